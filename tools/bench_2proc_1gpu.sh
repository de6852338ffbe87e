#!/bin/bash
# Exercise bench.py's N = 2 code path on a ONE-GPU box: both ranks on device 0, process group on gloo, --comm torch
# (RCCL refuses two ranks on one device).  A temporary copy of bench.py is patched for that; nothing is measured.
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=$R/.bench2_tmp.py
mkdir -p $R/gpurun_out
sed -e 's/local_rank = int(os.environ.get("LOCAL_RANK", "0"))/local_rank = 0/' \
    -e 's/dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))/dist.init_process_group(backend="gloo")/' \
    $R/bench.py > $T
cd $R && trap "rm -f $T" EXIT && python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 $T \
    --gpus 2 --steps 4 --warmup 2 --batch 8192 --comm torch

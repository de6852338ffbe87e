#!/bin/bash
# The measurements that were queued when GPU access closed at the end of round 4: the lane-pair headline kernel with branch-free save-point
# seeds in its reverse sweep (-DCRNN_ADJ2_SEEDS_FLAT=1, same values by construction) against the shipped one, same box, alternating.
#   here (no GPU):   bash tools/gpu_queued_ab.sh build      (two libraries under crnn_amd/csrc/dbg/; take that directory out of .gpurunignore)
#   on the GPU box:  bash tools/gpu_queued_ab.sh run
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
if [ "$1" = build ]; then
  bash $R/tools/kvariants.sh build base="-ffp-contract=on" flat="-ffp-contract=on -DCRNN_ADJ2_SEEDS_FLAT=1" hyclosed="-ffp-contract=on -DCRNN_HY_SENS_CLOSED=1" hylu="-ffp-contract=on -DCRNN_HY_SENS_SHARED_LU=1" hyboth="-ffp-contract=on -DCRNN_HY_SENS_CLOSED=1 -DCRNN_HY_SENS_SHARED_LU=1"
else
  cd $R
  for rep in 1 2 3; do for v in base flat; do
    CRNN_HIP_LIB=$R/crnn_amd/csrc/dbg/libcrnn_kv_$v.so python tools/kbench.py --lanes 2 --reps 30 | tail -1 | cut -c1-170
  done; done
  CRNN_HIP_LIB=$R/crnn_amd/csrc/dbg/libcrnn_kv_flat.so python -m pytest tests/test_gpu_lanes2.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -1
  # second queued item: HyChem dual-norm kernel with hychem_tan.hpp's closed forms (-DCRNN_HY_SENS_CLOSED=1; compiled only so far: scratch
  # 5 236 -> 3 528 B per lane, no logarithm / exponential taken twice).  Same loss and gradient to ~1e-12 expected (other rounding), then time
  # third: one copy of W's factors per trajectory instead of per lane (-DCRNN_HY_SENS_SHARED_LU=1: LDS 107 712 -> 31 248 B per block, two
  # blocks = four wavefronts per CU instead of two); identical bits expected for this one (same arithmetic)
  for v in base hyclosed hylu hyboth; do
    echo $v; CRNN_HIP_LIB=$R/crnn_amd/csrc/dbg/libcrnn_kv_$v.so python tools/hy_sens_time.py 1024
    CRNN_HIP_LIB=$R/crnn_amd/csrc/dbg/libcrnn_kv_$v.so python tools/hy_sens_time.py 4096
  done
  for v in hyclosed hylu hyboth; do
    CRNN_HIP_LIB=$R/crnn_amd/csrc/dbg/libcrnn_kv_$v.so python -m pytest tests/test_hychem.py tests/test_gpu_errnorm_sens.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -1
  done
fi

#!/bin/bash
# Build / time variants of libcrnn_hip.so (kernel experiments).
#   bash tools/kvariants.sh build name1="-DFOO=1" name2="-mllvm -bar" ...     (here; no GPU needed)
#   bash tools/kvariants.sh run   [kbench args]                               (on the GPU box)
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
D=$R/crnn_amd/csrc/dbg
mkdir -p $D
if [ "$1" = build ]; then
  shift
  rm -f $D/libcrnn_kv_*.so
  for spec in "$@"; do
    name=${spec%%=*}; flags=${spec#*=}
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $flags -o $D/libcrnn_kv_$name.so $R/crnn_amd/csrc/crnn_capi.hip -shared -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib 2>&1 | grep -E "error" &
  done
  wait
  ls $D
else
  shift
  for so in $D/libcrnn_kv_*.so; do
    CRNN_HIP_LIB=$so timeout 300 python $R/tools/kbench.py "$@" 2>&1 | tail -1
  done
fi

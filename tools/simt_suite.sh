#!/bin/bash
# tools/simt_suite.sh -- the `-m gpu` parity tests executed on the CPU through the SIMT emulator (tests/simt/: the kernels' own sources
# compiled as host C++, a fibre per lane).  Everything except the tests that need the device's throughput (full BASELINE sizes), other
# builds of the device library (crossbuild) or more than one rank.  Output: gpurun-free evidence of what the kernel sources compute;
# the log is committed under profiles/ when it backs a claim.
#   bash tools/simt_suite.sh [pytest args...]        e.g.  bash tools/simt_suite.sh tests/test_hychem.py -k errnorm
R=$(cd $(dirname $0)/.. && pwd)
bash $R/tests/simt/build.sh || exit 1
export CRNN_HIP_LIB=$R/tests/simt/libcrnn_simt.so
export SIMT_THREADS=${SIMT_THREADS:-6}
cd $R
if [ $# -gt 0 ]; then exec python -m pytest -m gpu -p no:cacheprovider "$@"; fi
python -m pytest tests -m gpu -p no:cacheprovider -v --timeout=${SIMT_TIMEOUT:-1200} --durations=20 \
  --ignore=tests/test_gpu_fullsize.py --ignore=tests/test_gpu_crossbuild.py --ignore=tests/test_dist_gpu2proc.py \
  -k "not full_size and not fullsize and not config5 and not 65536 and not kernel_resources and not oracle_sanitizers"

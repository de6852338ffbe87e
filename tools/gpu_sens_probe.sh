#!/bin/bash
# usage (GPU box): bash tools/gpu_sens_probe.sh <tag> [pmc]  -- the dual-norm gradient (errnorm_sens = 1) on case2, 65 536 trajectories:
# parity tests, kernel trace with the chunks in one launch and with one launch per chunk, optionally the SQ counter passes
TAG=$1; PMC=$2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/sens_$TAG
mkdir -p $OUT
cd $R
python -c "import crnn_amd" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
timeout 900 python -m pytest tests/test_gpu_errnorm_sens.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
for m in 1 0; do
  CRNN_SENS_ONE_LAUNCH=$m python $R/tools/kbench.py --errnorm-sens 1 --reps 8 --wall 2>&1 | tail -1
  CRNN_SENS_ONE_LAUNCH=$m rocprofv3 --kernel-trace --stats -d $OUT/trace$m -o trace -- python $R/tools/kbench.py --errnorm-sens 1 --reps 6 > $OUT/kbench$m.log 2>&1
  (cd $R && python tools/rocpd_summary.py $OUT/trace$m > $OUT/summary$m.txt 2>&1; head -6 $OUT/summary$m.txt | cut -c1-170)
done
if [ -n "$PMC" ]; then
  bash $R/tools/gpu_pmc_k.sh sens${TAG} --errnorm-sens 1
fi

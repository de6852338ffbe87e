#!/bin/bash
# usage (GPU box): bash tools/gpu_pmc_fw.sh <tag> [kbench args]  -- FETCH_SIZE and WRITE_SIZE passes (separate) + a kernel trace on tools/kbench.py
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcfw_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- python $R/tools/kbench.py --reps 4 "$@" > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o write -- python $R/tools/kbench.py --reps 4 "$@" > /dev/null 2> $OUT/write.err
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $R/tools/kbench.py --reps 6 "$@" > $OUT/run.log 2> $OUT/trace.err
cd $R && python tools/rocpd_summary.py $OUT > $OUT/summary.txt 2>&1
grep -E "FETCH_SIZE|WRITE_SIZE" $OUT/summary.txt | cut -c1-150
grep -E "kernel<" $OUT/summary.txt | head -3 | cut -c1-150
tail -1 $OUT/run.log | cut -c1-160

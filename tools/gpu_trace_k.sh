#!/bin/bash
# usage (GPU box): bash tools/gpu_trace_k.sh <tag> <script> [args]   -- rocprofv3 --kernel-trace --stats around a tools/ script
TAG=$1; SCRIPT=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $R/tools/$SCRIPT "$@" > $OUT/run.log 2>&1
cd $R && python tools/rocpd_summary.py $OUT > $OUT/summary.txt 2>&1
head -7 $OUT/summary.txt | cut -c1-170
tail -1 $OUT/run.log

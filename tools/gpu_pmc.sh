#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/gpu_pmc.sh <tag> <command...>
# kernel trace + stats, then separate PMC passes (never combined with a trace domain): FETCH_SIZE, WRITE_SIZE, two SQ groups.
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- "$@" > $OUT/run_trace.log 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- "$@" > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o write -- "$@" > /dev/null 2> $OUT/write.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $OUT/sq1 -o sq1 -- "$@" > /dev/null 2> $OUT/sq1.err
cd $R && python tools/rocpd_summary.py $OUT > $OUT/summary.txt 2>&1
grep -v "^$" $OUT/summary.txt | cut -c1-220 | head -70
tail -2 $OUT/run_trace.log | cut -c1-400

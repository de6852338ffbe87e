#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/gpu_profile.sh <tag> [bench args...]
# kernel-trace + stats pass, then two PMC passes (FETCH_SIZE / WRITE_SIZE need separate passes).
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o write -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $OUT/pmc_sq -o sq -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> $OUT/pmc_sq.err
cd $OUT && find . -name "*.csv" | head -30; du -sh .

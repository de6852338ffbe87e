#!/bin/bash
# usage: bash tools/gpu_pmc_cath.sh <tag> [cathode_bench args]  -- SQ / memory counter passes on tools/cathode_bench.py
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $OUT/sq1 -o sq1 -- python $R/tools/cathode_bench.py --reps 2 "$@" > /dev/null 2> $OUT/sq1.err
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU -d $OUT/sq2 -o sq2 -- python $R/tools/cathode_bench.py --reps 2 "$@" > /dev/null 2> $OUT/sq2.err
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- python $R/tools/cathode_bench.py --reps 2 "$@" > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o write -- python $R/tools/cathode_bench.py --reps 2 "$@" > $OUT/run.log 2> $OUT/write.err
cd $R && python tools/rocpd_summary.py $OUT > $OUT/summary.txt 2>&1
grep -v "^$" $OUT/summary.txt | cut -c1-230 | head -60
tail -1 $OUT/run.log

#!/bin/bash
# usage: bash tools/gpu_pmc_k.sh <tag> [kbench args]  -- SQ counter passes on tools/kbench.py (fixed-parameter launches)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmck_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $OUT/sq1 -o sq1 -- python $R/tools/kbench.py --reps 4 "$@" > /dev/null 2> $OUT/sq1.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU -d $OUT/sq2 -o sq2 -- python $R/tools/kbench.py --reps 4 "$@" > /dev/null 2> $OUT/sq2.err
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SMEM -d $OUT/sq3 -o sq3 -- python $R/tools/kbench.py --reps 4 "$@" > /dev/null 2> $OUT/sq3.err

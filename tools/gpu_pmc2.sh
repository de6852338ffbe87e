#!/bin/bash
# usage (GPU box): bash tools/gpu_pmc2.sh <tag> <command...>   -- two more SQ passes: where the cycles go / the measured instruction mix
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc2_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_LDS -d $OUT/sq2 -o sq2 -- "$@" > /dev/null 2> $OUT/sq2.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $OUT/sq3 -o sq3 -- "$@" > /dev/null 2> $OUT/sq3.err
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FLOPS_FP64 -d $OUT/sq4 -o sq4 -- "$@" > /dev/null 2> $OUT/sq4.err
cd $R && python tools/rocpd_summary.py $OUT > $OUT/summary.txt 2>&1
grep -v "^$" $OUT/summary.txt | grep -E "==|adj_kernel<6|hychem_kernel<9, 10, true|cathode_adj" | grep -v "^void\|^crnn" | cut -c1-150

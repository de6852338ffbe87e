#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/calib; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- $R/tools/ubench/fetch_calib > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o write -- $R/tools/ubench/fetch_calib > $OUT/write.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $OUT/rd -o rd -- $R/tools/ubench/fetch_calib > $OUT/rd.log 2>&1
grep known $OUT/fetch.log

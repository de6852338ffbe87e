#!/bin/bash
# Round-6 first contact, short form (when less than half an hour of a round is left): smoke, the headline bench line without secondaries, a kernel
# trace of it, then the primitives calibration and the case2 pin tests.   gpurun --timeout 900 -- 'bash tools/gpu_r06_short.sh'
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
O=$R/gpurun_out/r06s
mkdir -p $O
cd $R
( time timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" ) > $O/build_smoke.log 2>&1; tail -3 $O/build_smoke.log | cut -c1-300
timeout 200 python bench.py --steps 20 --warmup 5 --no-secondary > $O/bench_driver.json 2> $O/bench_driver.err; tail -1 $O/bench_driver.json | cut -c1-600
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench_trace.json 2> $O/trace.err )
python tools/rocpd_summary.py $O > $O/trace_summary.txt 2>&1; head -12 $O/trace_summary.txt | cut -c1-200
timeout 300 python -m pytest tests/test_simt_calibration.py tests/test_case2_stream_pin.py tests/test_gpu_errnorm_sens.py -m gpu -q -p no:cacheprovider --timeout=200 > $O/gpu_new_tests.log 2>&1; tail -5 $O/gpu_new_tests.log
find $O -name "*.db" -size +8M -delete

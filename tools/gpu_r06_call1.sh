#!/bin/bash
# Round-6 first contact: the new kernels first, then the whole -m gpu suite without -x, bench lines, kernel trace.
#   gpurun --timeout 2700 -- 'bash tools/gpu_r06_call1.sh'
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
O=$R/gpurun_out/r06a
mkdir -p $O
cd $R
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" > $O/box.txt; nproc >> $O/box.txt
( time timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" ) > $O/build_smoke.log 2>&1
tail -5 $O/build_smoke.log | cut -c1-300
# 1. first-contact risks: the kernels no device has executed (HyChem dual norm / composite / finite-difference J, cathode dual norm / composite, tsit5 sens)
timeout 900 python -m pytest tests/test_hychem.py tests/test_cathode.py tests/test_gpu_errnorm_sens.py -m gpu -q -p no:cacheprovider --timeout=300 \
   -k "errnorm or composite or finite_difference or sparse or tape" --durations=10 > $O/gpu_new_kernels.log 2>&1
tail -40 $O/gpu_new_kernels.log > $O/gpu_new_kernels.txt; tail -3 $O/gpu_new_kernels.txt
# 2. the whole -m gpu suite (no -x: every failure is wanted); a per-test limit so that one hung kernel does not take the record with it
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=300 --durations=40 ) > $O/gpu_suite_full.log 2>&1
tail -90 $O/gpu_suite_full.log > $O/gpu_suite.txt
tail -6 $O/gpu_suite.txt
# 3. bench lines
( time timeout 600 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -1 $O/bench_default.json | cut -c1-900; tail -4 $O/bench_default.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; tail -1 $O/bench_driver.json | cut -c1-300
# 4. kernel trace of the bench command
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_trace.json 2> $O/trace.err )
python tools/rocpd_summary.py $O > $O/trace_summary.txt 2>&1; head -30 $O/trace_summary.txt | cut -c1-200
find $O -name "*.db" -size +8M -delete
echo "call 1 complete: $(ls $O | wc -l) files"

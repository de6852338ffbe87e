#!/bin/bash
# Round-5 evidence, first call: build + smoke, the whole -m gpu suite on the tree that ships, the bench lines, a kernel trace of the bench command.
#   gpurun --timeout 2400 -- 'bash tools/gpu_r05_call1.sh'
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
O=$R/gpurun_out/r05a
mkdir -p $O
cd $R
( time timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" ) > $O/build_smoke.log 2>&1
tail -4 $O/build_smoke.log | cut -c1-300
# the whole -m gpu suite (no -x: every failure is wanted); a per-test limit so that one hung kernel does not take the record with it
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=420 --durations=15 > $O/gpu_suite_full.log 2>&1
tail -60 $O/gpu_suite_full.log > $O/gpu_suite.txt
tail -3 $O/gpu_suite.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -1 $O/bench_default.json | cut -c1-700
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; tail -1 $O/bench_driver.json | cut -c1-300
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_trace.json 2> $O/trace.err )
python tools/rocpd_summary.py $O > $O/trace_summary.txt 2>&1; head -30 $O/trace_summary.txt | cut -c1-200
find $O -name "*.db" -size +8M -delete
echo "call 1 complete: $(ls $O | wc -l) files"

#!/bin/bash
# Round-5 evidence run: everything the final tree needs re-measured, in one gpurun call.
#   gpurun --timeout 2400 -- 'bash tools/gpu_r05_evidence.sh'
# Writes under gpurun_out/r05a/ ; the summaries judged are copied into profiles/ afterwards (by hand, from the merged gpurun_out).
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
O=$R/gpurun_out/r05a
mkdir -p $O
cd $R
( time python -c "import __graft_entry__ as g; g.build(); g.smoke()" ) > $O/build_smoke.log 2>&1
tail -4 $O/build_smoke.log | cut -c1-300
# 1. the whole -m gpu suite (no -x: every failure is wanted), full tail kept
python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > $O/gpu_suite_full.log 2>&1
tail -45 $O/gpu_suite_full.log > $O/gpu_suite.txt
tail -3 $O/gpu_suite.txt
# 2. the driver's bench line (default flags) + the driver's own invocation
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -1 $O/bench_default.json | cut -c1-600
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; tail -1 $O/bench_driver.json | cut -c1-300
# 3. kernel trace of the same command
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_trace.json 2> $O/trace.err )
python tools/rocpd_summary.py $O > $O/trace_summary.txt 2>&1; head -30 $O/trace_summary.txt | cut -c1-200
# 4. queued A/B (libraries prebuilt here under crnn_amd/csrc/dbg/, which travels for this call)
if ls $R/crnn_amd/csrc/dbg/libcrnn_kv_*.so > /dev/null 2>&1; then
  timeout 900 bash tools/gpu_queued_ab.sh run > $O/queued_ab.txt 2>&1; cat $O/queued_ab.txt | cut -c1-200
fi
# 5. fuzz sweeps on the tree that ships
for f in fuzz_parity fuzz_hychem fuzz_cathode; do
  timeout 600 python tools/$f.py > $O/$f.txt 2>&1; tail -3 $O/$f.txt | cut -c1-200
done

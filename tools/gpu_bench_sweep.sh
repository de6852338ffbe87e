#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/gpu_bench_sweep.sh "1 3 5 7"
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
python bench.py --steps 10 --warmup 2 2> gpurun_out/bench_err.log | tail -1 > gpurun_out/bench.json; cat gpurun_out/bench.json
for c in $1; do
  python bench.py --steps 5 --warmup 1 --cols $c --no-cpu-baseline 2>> gpurun_out/bench_err.log | tail -1 > gpurun_out/bench_c$c.json
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_c$c.json"))
print("cols $c value %.4g traj/s kernel_ms %.3f step_ms %.3f valu %.3f TF frac %.4f steps/traj %.2f" % (d["value"], d["roofline"]["kernel_ms"], d["ms_per_step"], d["valu_fp64"]["achieved"], d["valu_fp64"]["frac"], d["valu_fp64"]["steps_per_traj"]))
PY
done

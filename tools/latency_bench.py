#!/usr/bin/env python3
"""Launch latency of one loss+gradient call on tiny ensembles (the reference's schedule is batch 1): adjoint vs forward tangents
with one column per lane.  usage: python tools/latency_bench.py"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crnn_amd import NeuralODE, ODEProblem, PRESET_CASE2, cases
fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures.json")))
c2 = fx["case2"]
u0 = np.array(c2["u0"]); ts = np.array(c2["tsteps"]); data = np.array(c2["data"]); ys = np.array(c2["yscale"])
p = np.array(fx["case2_ckpt"]["p"])
for B in (1, 8, 32):
    for name, kw in (("adjoint 1 lane", dict(grad_mode=2, lanes=1)), ("adjoint 2 lanes", dict(grad_mode=2, lanes=2)), ("forward C=1 L=25", dict(grad_mode=1, cols_per_lane=1)), ("forward C=7 L=4", dict(grad_mode=1, cols_per_lane=7)),
                     ("forward C=5 L=5", dict(grad_mode=1, cols_per_lane=5)), ("auto", dict()),
                     ("tsit5 adjoint", dict(grad_mode=2, solver=1)), ("tsit5 fwd C=1 L=25", dict(grad_mode=1, cols_per_lane=1, solver=1)), ("tsit5 auto", dict(solver=1))):
        lanes = kw.pop("lanes", None)
        node = NeuralODE(ODEProblem(PRESET_CASE2, ts, **kw))
        node.set_ensemble(u0[:B], data[:B], ys)
        if lanes is not None:
            node.set_lanes_per_traj(lanes)
        ks, ws = [], []
        for _ in range(30):
            t0 = time.perf_counter(); node.loss_and_grad(p); ws.append((time.perf_counter() - t0) * 1e3); ks.append(node.last_stats["kernel_ms"])
        print(f"B={B:3d} {name:18s} kernel {np.median(ks[5:]):.3f} ms  call {np.median(ws[5:]):.3f} ms", flush=True)
        node.close()

#!/usr/bin/env python3
"""Kernel time of the primal-only paths (the reference's epoch-end loss loop, case2.jl:199-203, and predict_neuralode) next to the
gradient launch: case2, 65 536 trajectories, checkpoint p.  usage (GPU box): python tools/primal_probe.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crnn_amd import NeuralODE, ODEProblem, PRESET_CASE2, cases  # noqa: E402

fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures.json")))
B = int(os.environ.get("B", 65536))
rng = np.random.Generator(np.random.PCG64([1234, 0]))
ts = cases.case2_tsteps()
u0 = cases.case2_u0(B, rng)
p = np.array(fx["case2_ckpt"]["p"])
from crnn_amd import _lib as L  # noqa: E402
SOLVER = {"ros23": L.SOLVER_ROSENBROCK23, "tsit5": L.SOLVER_TSIT5, "autotsit5": L.SOLVER_AUTOTSIT5}[os.environ.get("SOLVER", "ros23")]
node = NeuralODE(ODEProblem(PRESET_CASE2, ts, solver=SOLVER))
node.set_ensemble(u0, np.zeros((B, 6, len(ts))), np.ones(6))
for name, fn in (("loss_and_grad", lambda: node.loss_and_grad(p)), ("losses", lambda: node.losses(p)),
                 ("predict_n_ode", lambda: node.predict_n_ode(p))):
    k = []
    for _ in range(6):
        fn()
        k.append(node.stats()["kernel_ms"])
    print(f"{name:16s} kernel_ms {np.median(k):.4f}  (first {k[0]:.4f})")
node.close()
import time
node = NeuralODE(ODEProblem(PRESET_CASE2, ts, solver=SOLVER))
node.set_ensemble(u0, np.zeros((B, 6, len(ts))), np.ones(6))
for name, fn in (("loss_and_grad", lambda: node.loss_and_grad(p)), ("losses", lambda: node.losses(p))):
    fn(); fn()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    print(f"{name:16s} wall per call {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms, stats kernel_ms {node.stats()['kernel_ms']:.4f}")
node.close()

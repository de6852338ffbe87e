#!/usr/bin/env python3
"""Long device-resident training run at the headline size: --steps optimiser steps without per-step host synchronisation,
then checks that nothing was skipped or replayed, the loss is finite and lower, and the parameters equal a second run's
bit for bit (determinism).  usage: python tools/soak_train.py [--steps 3000] [--batch 65536] [--theta0 ckpt|init]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=3000)
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--theta0", default="ckpt")
args = ap.parse_args()
from crnn_amd import NeuralODE, ODEProblem, Optimiser, PRESET_CASE2, cases  # noqa: E402

fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures.json")))
rng = np.random.Generator(np.random.PCG64([1234, 0]))
B = args.batch
ts = cases.case2_tsteps()
u0 = cases.case2_u0(B, rng)
gen = NeuralODE(ODEProblem(PRESET_CASE2, ts, atol=1e-10, rtol=1e-8))
clean = gen.predict_theta(u0, cases.case2_true_theta())[:, :6, :]
gen.close()
data = cases.add_noise(clean, 0.05, rng)
ys = cases.max_min(data, lb=cases.LB_CASE2)
p0 = np.array(fx["case2_ckpt"]["p"]) if args.theta0 == "ckpt" else cases.case2_init_p(np.random.Generator(np.random.PCG64(7)))


def run():
    node = NeuralODE(ODEProblem(PRESET_CASE2, ts))
    node.set_ensemble(u0, data, ys)
    node.train_init(Optimiser(25, PRESET_CASE2), p0)
    l0 = node.train_step(want_loss=True)
    t0 = time.perf_counter()
    for _ in range(args.steps - 2):
        node.train_step(want_loss=False)
    l1 = node.train_step(want_loss=True)
    dt = time.perf_counter() - t0
    p = node.params()
    st = node.stats()
    node.close()
    return l0, l1, p, st, dt


a = run()
b = run()
print(f"{args.steps} steps at B={B}: loss {a[0]:.6e} -> {a[1]:.6e}; {a[4] / (args.steps - 1) * 1e3:.3f} ms/step; last step stats {a[3]}")
ok = np.isfinite(a[1]) and a[1] < a[0] and np.array_equal(a[2], b[2]) and a[3]["n_ok"] == B
print("finite, lower, deterministic (second run bit-identical), all trajectories solved:", bool(ok))
sys.exit(0 if ok else 1)

#!/usr/bin/env python3
"""Where a launch's time goes when the ensemble equals the resident lanes: the distribution of per-trajectory step counts
(crnn_last_step_counts) of the bench ensemble at the checkpoint p.  A wavefront is busy for the LONGEST of its 64
trajectories and the launch for the longest wavefront, so  mean / max-per-wavefront  is the lane utilisation inside busy
wavefronts and  mean(max-per-wavefront) / max  the share of the launch a SIMD has a wavefront at all.
usage: python tools/step_hist.py [--batch 65536] [--case case2|rober]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--case", default="case2")
args = ap.parse_args()
from crnn_amd import NeuralODE, ODEProblem, PRESET_CASE2, PRESET_ROBER, cases  # noqa: E402

fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures.json")))
rng = np.random.Generator(np.random.PCG64([1234, 0]))
B = args.batch
if args.case == "case2":
    ts = cases.case2_tsteps()
    u0 = cases.case2_u0(B, rng)
    node = NeuralODE(ODEProblem(PRESET_CASE2, ts))
    node.set_ensemble(u0, np.zeros((B, 6, len(ts))), np.ones(6))
    p = np.array(fx["case2_ckpt"]["p"])
else:
    ts = cases.rober_tsteps()
    u0 = cases.rober_u0(B, rng)
    ys = np.array([1.0, 4e-5, 1.0])
    node = NeuralODE(ODEProblem(PRESET_ROBER, ts, rate_scale=ys / ts[-1]))
    node.set_ensemble(u0, np.zeros((B, 3, len(ts))), ys)
    p = np.array(fx["rober_ckpt"]["p"])
node.loss_and_grad(p)
na, nr = node.step_counts()
n = (na + nr).astype(np.int64)
q = np.percentile(n, [0, 1, 10, 50, 90, 99, 99.9, 100])
print(f"{args.case} B={B}: attempts per trajectory  mean {n.mean():.2f}  min/1%/10%/50%/90%/99%/99.9%/max = " + "/".join(f"{v:.0f}" for v in q))
w = n[: (B // 64) * 64].reshape(-1, 64)
wmax = w.max(axis=1)
print(f"per wavefront (64 consecutive trajectories): mean of max {wmax.mean():.2f}, min {wmax.min()}, max {wmax.max()}")
print(f"lane utilisation inside busy wavefronts  = mean / mean(max per wavefront) = {n.mean() / wmax.mean():.3f}")
print(f"wavefront residency over the launch      = mean(max per wavefront) / max  = {wmax.mean() / wmax.max():.3f}")
print(f"useful lane-steps / (lanes x longest)    = {n.mean() / n.max():.3f}")
hist = np.bincount(n)
print("histogram (attempts: trajectories): " + " ".join(f"{k}:{v}" for k, v in enumerate(hist) if v))

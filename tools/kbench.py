#!/usr/bin/env python3
"""Kernel timing at FIXED parameters (no optimiser update): case2 checkpoint p, B initial conditions, loss+gradient
launches; prints the median HIP-event kernel time.  CRNN_HIP_LIB selects the library build (tools/kvariants.sh).
usage: python tools/kbench.py [--batch 65536] [--reps 12] [--grad auto|forward|adjoint] [--case case2|case1|rober|hychem] [--errnorm-sens 0|1|2] [--solver rosenbrock23|tsit5|autotsit5]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--reps", type=int, default=12)
ap.add_argument("--grad", default="auto")
ap.add_argument("--case", default="case2")
ap.add_argument("--solver", default=None, choices=[None, "rosenbrock23", "tsit5", "autotsit5"])
ap.add_argument("--errnorm-sens", type=int, default=0)
ap.add_argument("--lanes", type=int, default=0, help="lanes per trajectory in the Rosenbrock23 adjoint kernel: 0 auto, 1, 2")
ap.add_argument("--wall", action="store_true", help="also print the wall time per loss+gradient call (several launches per call: errnorm_sens)")
args = ap.parse_args()

from crnn_amd import NeuralODE, ODEProblem, PRESET_CASE1, PRESET_CASE2, PRESET_ROBER, cases  # noqa: E402

fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures.json")))
rng = np.random.Generator(np.random.PCG64([1234, 0]))
gm = {"auto": 0, "forward": 1, "adjoint": 2}[args.grad]
sv = {None: None, "rosenbrock23": 0, "tsit5": 1, "autotsit5": 2}[args.solver]
B = args.batch
if args.case == "case2":
    ts = cases.case2_tsteps()
    u0 = cases.case2_u0(B, rng)
    gen = NeuralODE(ODEProblem(PRESET_CASE2, ts, atol=1e-10, rtol=1e-8))
    clean = gen.predict_theta(u0, cases.case2_true_theta())[:, :6, :]
    gen.close()
    data = cases.add_noise(clean, 0.05, rng)
    node = NeuralODE(ODEProblem(PRESET_CASE2, ts, grad_mode=gm, solver=sv, errnorm_sens=args.errnorm_sens))
    node.set_ensemble(u0, data, cases.max_min(data, lb=cases.LB_CASE2))
    p = np.array(fx["case2_ckpt"]["p"])
elif args.case == "case1":
    ts = cases.case1_tsteps()
    u0 = cases.case1_u0(B, rng)
    gen = NeuralODE(ODEProblem(PRESET_CASE1, ts, atol=1e-10, rtol=1e-8, solver=0))
    clean = gen.predict_theta(u0, cases.case1_true_theta())
    gen.close()
    data = cases.add_noise(clean, 0.05, rng)
    node = NeuralODE(ODEProblem(PRESET_CASE1, ts, grad_mode=gm, solver=sv, errnorm_sens=args.errnorm_sens))
    node.set_ensemble(u0, data, cases.max_min(data, lb=cases.LB_CASE1))
    p = np.array(fx["case1"]["p"])
elif args.case == "hychem":
    from crnn_amd import PRESET_HYCHEM, hychem as hy
    ts, u0, Tt, Pt = hy.sample_conditions(B, rng)
    node = NeuralODE(ODEProblem(PRESET_HYCHEM, ts, rate_scale=hy.DYDT_SCALE, grad_mode=gm))
    node.set_ensemble(u0, np.zeros((B, 9, len(ts))), np.ones(9))
    node.set_tables(Tt, Pt)
    clean = node.predict_n_ode(hy.true_p())                       # synthetic data: the true mechanism, 1 % noise
    data = clean * (1.0 + 0.01 * rng.standard_normal(clean.shape))
    ys = np.maximum((data.max(axis=2) - data.min(axis=2)).max(axis=0), hy.LB)
    node.set_ensemble(u0, data, ys)
    node.set_tables(Tt, Pt)
    p = hy.true_p() + 0.02 * np.random.Generator(np.random.PCG64(5)).standard_normal(hy.NP)
    p[-1] = 0.1
else:
    ts = cases.rober_tsteps()
    u0 = cases.rober_u0(B, rng)
    ys = np.array(fx["rober"]["yscale"]) if "yscale" in fx.get("rober", {}) else np.array([1.0, 4e-5, 1.0])
    sc = ys / ts[-1]
    data = np.abs(rng.standard_normal((B, 3, len(ts)))) * ys[None, :, None]
    node = NeuralODE(ODEProblem(PRESET_ROBER, ts, rate_scale=sc, grad_mode=gm, solver=sv))
    node.set_ensemble(u0, data, ys)
    p = np.array(fx["rober_ckpt"]["p"])
if args.lanes:
    node.set_lanes_per_traj(args.lanes)
import time  # noqa: E402
ms, wall = [], []
for _ in range(args.reps):
    t0 = time.perf_counter()
    loss, grad = node.loss_and_grad(p)
    wall.append((time.perf_counter() - t0) * 1e3)
    ms.append(node.last_stats["kernel_ms"])
st = node.last_stats
print(f"lib={os.path.basename(os.environ.get('CRNN_HIP_LIB', 'libcrnn_hip.so'))} case={args.case} solver={args.solver} grad={args.grad} lanes={args.lanes} B={B} "
      f"kernel_ms median {np.median(ms[2:]):.4f} min {min(ms[2:]):.4f}  steps/traj {st['n_accept'] / st['n_traj']:.2f} "
      f"rej/traj {st['n_reject'] / st['n_traj']:.2f} loss {loss:.6e} |g| {np.linalg.norm(grad):.6e}"
      + (f"  wall_ms/call median {np.median(wall[2:]):.3f} min {min(wall[2:]):.3f} ({B / np.median(wall[2:]) / 1e3:.2f} M traj+grads/s)" if args.wall else ""))

#!/usr/bin/env python3
"""BASELINE config 5 at scale on one GPU: N particles x R heating rates (default 4096 x 256 = 1 048 576 trajectories with
per-particle 17-parameter gradients).  Observation sets are built from the committed golden fixture (tests/golden/
fixtures_cathode.json): each rate borrows the temperature grid and replica statistics of the nearest measured rate.
usage: python tools/cathode_bench.py [--particles 4096] [--rates 256] [--reps 3]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--particles", type=int, default=4096)
ap.add_argument("--rates", type=int, default=256)
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()
from crnn_amd.cathode import CathodeUQ  # noqa: E402

fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures_cathode.json")))
betas = np.exp(np.linspace(np.log(2.0), np.log(20.0), args.rates))
meas = np.array([s["beta"] for s in fx["sets"]])
exp_data = []
for b in betas:
    s = fx["sets"][int(np.argmin(np.abs(np.log(meas) - np.log(b))))]
    dbar, d2bar = np.array(s["dbar"]), np.array(s["d2bar"])
    sd = np.sqrt(np.maximum(d2bar - dbar ** 2, 0.0))
    exp_data.append(np.stack([np.array(s["ts"]) * s["beta"] / b, dbar + sd, dbar - sd], axis=1))
uq = CathodeUQ(exp_data, betas, fx["theta"], normalizer=np.ones((args.rates, 3)), errnorm_sens=0)
rng = np.random.default_rng(0)
p = 1 + 1e-3 * rng.standard_normal((args.particles, 17))          # particles = 1 + 1e-3 N(0,1)
p[:, 6:9] = 0.0
ms = []
for _ in range(args.reps):
    loss, grad, _ = uq.solve(p)
    ms.append(uq.last_stats["kernel_ms"])
st = uq.last_stats
n = args.particles * args.rates
print(f"cathode-UQ {args.particles} particles x {args.rates} rates = {n} trajectories+17-gradients: kernel_ms median "
      f"{np.median(ms):.2f} -> {n / np.median(ms) * 1e3:.3e} traj+grads/s; steps/traj {st['n_accept'] / st['n_traj']:.1f} "
      f"rej/traj {st['n_reject'] / st['n_traj']:.2f} ok {st['n_ok']}/{st['n_traj']} mean loss {loss.mean():.4e}")
# the primal launches (loss only; loss + HRR at the measured temperatures) next to the gradient launch
for name, kw in (("primal (loss)", dict(want_grad=False)), ("primal + HRR", dict(want_grad=False, want_hrr=True))):
    ms = []
    for _ in range(max(2, args.reps)):
        uq.solve(p, **kw)
        ms.append(uq.last_stats["kernel_ms"])
    print(f"  {name}: kernel_ms median {np.median(ms):.2f}")
# the same primal launches through the reference's composite (network.jl:195) -- crnn_cathode_set_solver
l_ros, _, _ = uq.solve(p, want_grad=False)
for solver in ("autotsit5_trbdf2", "autotsit5_rosenbrock23"):
    uq.set_solver(solver)
    for name, kw in (("primal (loss)", dict(want_grad=False)), ("primal + HRR", dict(want_grad=False, want_hrr=True))):
        ms = []
        for _ in range(max(2, args.reps)):
            l_c, _, _ = uq.solve(p, **kw)
            ms.append(uq.last_stats["kernel_ms"])
        st = uq.last_stats
        print(f"  {solver} {name}: kernel_ms median {np.median(ms):.2f}; steps/traj {st['n_accept'] / st['n_traj']:.1f} rej/traj "
              f"{st['n_reject'] / st['n_traj']:.2f} ok {st['n_ok']}/{st['n_traj']}; max |loss - loss_rosenbrock23| / loss "
              f"{np.max(np.abs(l_c - l_ros) / l_ros):.2e}")
uq.set_solver("rosenbrock23")

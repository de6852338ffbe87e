#!/usr/bin/env python3
"""CPU (oracle) census, config 5's model: how far is the gradient the device forms for dlnprob -- Rosenbrock23, primal error norm (the
discrete adjoint's derivative) -- from the gradient the reference really evaluates -- ForwardDiff's chunks 9 + 8 through
AutoTsit5(TRBDF2) with the partials in every error estimate (oracle solver 3 with errnorm_sens = 2) -- on a perturbed particle cloud
around the reference's parameter vector, at the reference's tolerances; both against the Rosenbrock23 gradient at tight tolerance
(the converged sensitivity).  usage: python tools/cathode_gradient_census.py [n_particles] [spread]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402

orc.build()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
spread = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
cfx = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures_cathode.json")))
th0 = np.array(cfx["theta"])
rng = np.random.default_rng(5)
P = 1 + spread * rng.standard_normal((N, 17)); P[:, 6:9] = 0.0
d_dev, d_ref, d_dev2, newton, steps_dev, steps_ref, n_unconv = [], [], [], 0, 0, 0, 0
for n in range(N):
    th = P[n] * th0
    for s in cfx["sets"]:
        args = (th, s["ts"], s["dbar"], s["d2bar"])
        conv = orc.cathode_solve_one(orc.make_cathode(s["beta"], atol=1e-14, rtol=1e-8), *args)["grad"]
        dev = orc.cathode_solve_one(orc.make_cathode(s["beta"]), *args)
        g = np.zeros(17)
        for cc, (lo, k) in zip(orc.cathode_sens_chunks(orc.make_cathode(s["beta"], solver=3), th0, mode=2), ((0, 9), (9, 8))):
            r = orc.cathode_solve_one(cc, *args)
            assert r["retcode"] == 0
            g[lo:lo + k] = r["grad"][lo:lo + k]
            newton += r["n_newton"]; steps_ref += r["naccept"]
        g0 = np.zeros(17)                 # the device's own reference-faithful mode: the same chunks and norm on Rosenbrock23
        for cc, (lo, k) in zip(orc.cathode_sens_chunks(orc.make_cathode(s["beta"]), th0, mode=2), ((0, 9), (9, 8))):
            r = orc.cathode_solve_one(cc, *args)
            g0[lo:lo + k] = r["grad"][lo:lo + k]
        steps_dev += dev["naccept"]
        # the converged sensitivity: Rosenbrock23 at rtol 1e-10; where even that disagrees with rtol 1e-8 by more than 1e-4 the trajectory's
        # tangents are unstable at tight tolerance too and it is reported separately
        conv8 = orc.cathode_solve_one(orc.make_cathode(s["beta"], atol=1e-14, rtol=1e-10), *args)["grad"]
        sc = np.max(np.abs(conv8))
        if not np.max(np.abs(conv8 - conv)) < 1e-4 * sc:
            n_unconv += 1
            continue
        d_dev.append(np.max(np.abs(dev["grad"] - conv8)) / sc)
        d_ref.append(np.max(np.abs(g - conv8)) / sc)
        d_dev2.append(np.max(np.abs(g0 - conv8)) / sc)
d_dev, d_ref, d_dev2 = np.array(d_dev), np.array(d_ref), np.array(d_dev2)
q = lambda a: f"median {np.median(a):.2e}  90 % {np.quantile(a, 0.9):.2e}  max {a.max():.2e}  off by more than 0.1: {int((a > 0.1).sum())} of {a.size}"
print(f"{N} particles (spread {spread}) x {len(cfx['sets'])} heating rates, reference tolerances (abstol 1e-12, reltol 1e-3); distance to the converged "
      f"sensitivity, relative to its largest entry:")
print(f"  Rosenbrock23, primal norm (what the device's adjoint differentiates): {q(d_dev)}   ({steps_dev / len(d_dev):.0f} accepted steps per trajectory)")
print(f"  Rosenbrock23, ForwardDiff's chunks and norm (the device's errnorm_sens):  {q(d_dev2)}")
print(f"  AutoTsit5(TRBDF2), ForwardDiff's chunks and norm (the reference):     {q(d_ref)}   ({steps_ref / (2 * len(d_ref)):.0f} per chunk solve, {newton} Newton iterations in all)")
print(f"  ({n_unconv} trajectories left out: their Rosenbrock23 sensitivities at rtol 1e-8 and 1e-10 disagree by more than 1e-4 -- no converged value to compare with)")

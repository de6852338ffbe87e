#!/usr/bin/env python3
"""CPU (oracle), config 3 (robertson) on the fixture's trajectories: distance of the per-trajectory gradient from the converged sensitivity
(Rosenbrock23 at rtol 1e-9), relative to its largest entry: (a) Rosenbrock23 with the primal norm -- the device's adjoint --, (b) ForwardDiff's
chunks 11 + 11 + 11 + 10 with the partials in the norm -- the reference's evaluation (rober_crnn.jl:219 through :33) and the device's
errnorm_sens mode."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402

orc.build()
fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures.json")))
rb = fx["robertson"]
u0, ts, data, ys, sc_ = np.array(rb["u0"]), np.array(rb["tsteps"]), np.array(rb["data"]), np.array(rb["yscale"]), np.array(rb["dydt_scale"])
p = np.array(fx["rober_ckpt"]["p"])
th, dth = orc.p2vec(3, 3, 6, p)
mk = lambda **kw: orc.make_problem(ns=3, nr=6, lb=1e-8, atol=kw.pop("atol", [1e-6, 1e-8, 1e-6]), rtol=kw.pop("rtol", 1e-3), yscale=ys, rate_scale=sc_, maxiters=1000000, **kw)
da, db = [], []
for i in range(u0.shape[0]):
    conv = orc.solve_one(mk(atol=[1e-12, 1e-14, 1e-12], rtol=1e-9), th, u0[i], ts, data[i], dtheta=dth, want_pred=False)["grad"]
    s = np.max(np.abs(conv))
    a = orc.solve_one(mk(), th, u0[i], ts, data[i], dtheta=dth, want_pred=False)["grad"]
    g = np.zeros(43)
    for k0 in range(0, 43, 11):
        k1 = min(43, k0 + 11)
        cols = np.zeros((dth.shape[0], 11), order="F"); cols[:, :k1 - k0] = dth[:, k0:k1]
        g[k0:k1] = orc.solve_one(mk(errnorm_sens=1), th, u0[i], ts, data[i], dtheta=cols, want_pred=False)["grad"][:k1 - k0]
    da.append(np.max(np.abs(a - conv)) / s); db.append(np.max(np.abs(g - conv)) / s)
print(f"robertson, {u0.shape[0]} trajectories, reference tolerances: Rosenbrock23 primal norm median {np.median(da):.2e} max {np.max(da):.2e}; "
      f"Rosenbrock23 + ForwardDiff's norm median {np.median(db):.2e} max {np.max(db):.2e}")

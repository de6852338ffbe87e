#!/usr/bin/env python3
"""CPU (oracle), the headline config (case2) on the fixture's trajectories: distance of the per-trajectory gradient from the converged
sensitivity (Rosenbrock23 at rtol 1e-10), relative to its largest entry, for (a) Rosenbrock23 with the primal error norm -- the derivative
the headline's adjoint kernel forms --, (b) ForwardDiff's chunks 9 + 9 + 7 with the partials in Rosenbrock23's norm -- the device's
errnorm_sens mode --, (c) the same through Tsit5 -- what case2's `AutoTsit5(Rosenbrock23())` amounts to (the composite never leaves
Tsit5 there) -- the reference's own evaluation; the device offers it too (solver = TSIT5 with errnorm_sens)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402

orc.build()
fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures.json")))
c2 = fx["case2"]
u0, ts, data, ys = np.array(c2["u0"]), np.array(c2["tsteps"]), np.array(c2["data"]), np.array(c2["yscale"])
kw = dict(ns=6, nr=3, has_temp=1, lb=float(np.float32(1e-6)), ub=10.0, inv_R=float(np.float32(-1.0) / np.float32(1.98720425864083e-3)), yscale=ys, clamp_pred=1)
for name, p in (("checkpoint p", np.array(fx["case2_ckpt"]["p"])), ("initialiser p", np.array(c2["p_init"]) if "p_init" in c2 else None)):
    if p is None:
        continue
    th, dth = orc.p2vec(2, 6, 3, p)
    da, db, dc = [], [], []
    for i in range(u0.shape[0]):
        conv = orc.solve_one(orc.make_problem(atol=1e-12, rtol=1e-10, **kw), th, u0[i], ts, data[i], dtheta=dth, want_pred=False)["grad"]
        sc = np.max(np.abs(conv))
        a = orc.solve_one(orc.make_problem(atol=1e-6, rtol=1e-3, **kw), th, u0[i], ts, data[i], dtheta=dth, want_pred=False)["grad"]
        gb, gc = np.zeros(25), np.zeros(25)
        for k0 in range(0, 25, 9):
            k1 = min(25, k0 + 9)
            cols = np.zeros((dth.shape[0], 9), order="F"); cols[:, :k1 - k0] = dth[:, k0:k1]
            gb[k0:k1] = orc.solve_one(orc.make_problem(atol=1e-6, rtol=1e-3, errnorm_sens=1, **kw), th, u0[i], ts, data[i], dtheta=cols, want_pred=False)["grad"][:k1 - k0]
            gc[k0:k1] = orc.solve_one(orc.make_problem(atol=1e-6, rtol=1e-3, errnorm_sens=1, solver=1, **kw), th, u0[i], ts, data[i], dtheta=cols, want_pred=False)["grad"][:k1 - k0]
        da.append(np.max(np.abs(a - conv)) / sc); db.append(np.max(np.abs(gb - conv)) / sc); dc.append(np.max(np.abs(gc - conv)) / sc)
    q = lambda x: f"median {np.median(x):.2e} max {np.max(x):.2e}"
    print(f"{name}, {u0.shape[0]} trajectories, atol 1e-6 rtol 1e-3: Rosenbrock23 primal norm {q(da)}; Rosenbrock23 + ForwardDiff's norm {q(db)}; Tsit5 + ForwardDiff's norm {q(dc)}")

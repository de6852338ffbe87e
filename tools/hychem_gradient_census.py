#!/usr/bin/env python3
"""CPU (oracle), config 4's model on the fixture's conditions: distance of (a) the gradient the device forms by default -- Rosenbrock23,
primal error norm (the adjoint's derivative) --, (b) the device's errnorm_sens mode -- ForwardDiff's 18 chunks of 12 with the partials in
Rosenbrock23's error norm --, (c) the reference's own evaluation -- the same chunks through AutoTsit5(Rosenbrock23) with the partials in
both algorithms' norms (oracle only) -- from the converged sensitivity (Rosenbrock23 at rtol 1e-9), relative to its largest entry."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402

orc.build()
d = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures_hychem.json")))
for k in ("ts", "u0", "Ttab", "Ptab", "data", "yscale", "dydt_scale", "p"):
    d[k] = np.array(d[k])
th, dth = orc.hychem_p2vec(d["p"])
mk = lambda **kw: orc.make_hychem(dydt_scale=d["dydt_scale"], yscale=d["yscale"], **kw)
for b in range(d["u0"].shape[0]):
    args = (th, d["u0"][b], d["ts"], d["Ttab"][b], d["Ptab"][b], d["data"][b])
    conv = orc.hychem_solve_one(mk(atol=1e-13, rtol=1e-9, maxiters=2000000), *args, dtheta=dth)
    conv2 = orc.hychem_solve_one(mk(atol=1e-13, rtol=1e-8, maxiters=2000000), *args, dtheta=dth)
    sc = np.max(np.abs(conv["grad"]))
    dev = orc.hychem_solve_one(mk(), *args, dtheta=dth)
    out = {}
    for name, solver in (("ros23+norm", 0), ("composite+norm", 2)):
        g = np.zeros(211); steps = 0
        for k0 in range(0, 211, 12):
            k1 = min(211, k0 + 12)
            r = orc.hychem_solve_one(mk(solver=solver, errnorm_sens=2, dual_partials=12), *args, dtheta=dth[k0:k1])
            assert r["retcode"] == 0
            g[k0:k1] = r["grad"]; steps += r["naccept"] + r["nreject"]
        out[name] = (np.max(np.abs(g - conv["grad"])) / sc, steps / 18)
    print(f"trajectory {b}: converged sensitivity stable to {np.max(np.abs(conv2['grad'] - conv['grad'])) / sc:.1e} (rtol 1e-8 against 1e-9); "
          f"Rosenbrock23 primal norm {np.max(np.abs(dev['grad'] - conv['grad'])) / sc:.2e} ({dev['naccept'] + dev['nreject']} attempts); "
          + "; ".join(f"{n} {v[0]:.2e} ({v[1]:.0f} attempts per chunk)" for n, v in out.items()))

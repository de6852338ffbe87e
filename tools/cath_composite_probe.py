#!/usr/bin/env python3
"""Device composite (cathode_auto_kernel: AutoTsit5(TRBDF2) / AutoTsit5(Rosenbrock23)) against the oracle's composites, the
Rosenbrock23 path and the golden Radau vectors: what the parity tests' bars are taken from.  GPU."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crnn_amd.cathode import CathodeUQ  # noqa: E402
from oracle import oracle as orc  # noqa: E402
orc.build()
fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures_cathode.json")))
ps = np.array(fx["theta"])


def two(s):
    dbar, d2bar = np.array(s["dbar"]), np.array(s["d2bar"])
    sd = np.sqrt(np.maximum(d2bar - dbar ** 2, 0.0))
    return np.stack([np.array(s["ts"]), dbar + sd, dbar - sd], axis=1)


for spread, N in ((1e-3, 64), (0.05, 64)):
    rng = np.random.default_rng(5)
    p = 1 + spread * rng.standard_normal((N, 17)); p[:, 6:9] = 0.0
    for atol, rtol in ((1e-12, 1e-3), (1e-13, 1e-6), (1e-14, 1e-9)):
        ref = CathodeUQ([two(s) for s in fx["sets"]], [s["beta"] for s in fx["sets"]], fx["theta"], atol=atol, rtol=rtol, errnorm_sens=0)
        l0, _, h0 = ref.solve(p, want_grad=False, want_hrr=True)
        for name, osolver in (("autotsit5_trbdf2", 3), ("autotsit5_rosenbrock23", 2)):
            uq = CathodeUQ([two(s) for s in fx["sets"]], [s["beta"] for s in fx["sets"]], fx["theta"], atol=atol, rtol=rtol, solver=name, errnorm_sens=0)
            l, _, h = uq.solve(p, want_grad=False, want_hrr=True)
            assert np.all(uq.last_retcode == 0), uq.last_retcode
            dl = dh = 0.0; dacc = 0; nacc_o = 0; nsw = 0; nts = 0
            for n in range(N):
                for i, s in enumerate(fx["sets"]):
                    c = orc.make_cathode(s["beta"], atol=atol, rtol=rtol, solver=osolver)
                    r = orc.cathode_solve_one(c, p[n] * ps, s["ts"], s["dbar"], s["d2bar"], want_grad=False)
                    D = len(s["ts"])
                    dl = max(dl, abs(l[n, i] - r["loss"]) / abs(r["loss"]))
                    dh = max(dh, np.max(np.abs(h[n, i, :D] - r["hrr"])) / np.max(np.abs(r["hrr"])))
                    nacc_o += r["naccept"]; nts += r["n_tsit5"]; nsw += r["n_tsit5"] != r["naccept"]
            print(f"spread {spread} tol ({atol},{rtol}) {name}: device vs oracle composite loss {dl:.2e} hrr {dh:.2e}; accepted device "
                  f"{uq.last_stats['n_accept']} oracle {nacc_o} (tsit5 {nts}, switching trajectories {nsw}/{N * 5}); vs device Rosenbrock23: loss "
                  f"{np.max(np.abs(l - l0) / l0):.2e} hrr {np.max(np.abs(h - h0)) / np.max(np.abs(h0)):.2e}", flush=True)
# golden Radau at tight tolerance (reference parameters)
for name in ("autotsit5_trbdf2", "autotsit5_rosenbrock23", "rosenbrock23"):
    uq = CathodeUQ([two(s) for s in fx["sets"]], [s["beta"] for s in fx["sets"]], fx["theta"], atol=1e-14, rtol=1e-9, solver=name, errnorm_sens=0)
    l, _, h = uq.solve(np.ones((1, 17)), want_grad=False, want_hrr=True)
    e = max(np.max(np.abs(h[0, i, :len(s["ts"])] - np.array(s["hrr"]))) / np.max(np.abs(s["hrr"])) for i, s in enumerate(fx["sets"]))
    el = max(abs(l[0, i] - s["loss"]) / s["loss"] for i, s in enumerate(fx["sets"]))
    print(f"golden Radau, tight tolerance, {name}: hrr {e:.2e} loss {el:.2e} steps {uq.last_stats['n_accept']}")

#!/usr/bin/env python3
"""HyChem errnorm_sens: device chunk launches against the oracle's chunked solves, chunk by chunk (step counts, gradient pieces).  GPU."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crnn_amd import NeuralODE, ODEProblem, PRESET_HYCHEM  # noqa: E402
from oracle import oracle as orc  # noqa: E402
orc.build()
d = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures_hychem.json")))
for k in ("ts", "u0", "Ttab", "Ptab", "data", "yscale", "dydt_scale", "p", "theta"):
    d[k] = np.array(d[k])
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rtol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-3
node = NeuralODE(ODEProblem(PRESET_HYCHEM, d["ts"], rate_scale=d["dydt_scale"], errnorm_sens=mode, rtol=rtol))
node.set_ensemble(d["u0"], d["data"], d["yscale"]); node.set_tables(d["Ttab"], d["Ptab"])
th, dth = orc.hychem_p2vec(d["p"])
for b in range(3):
    g = node.gradient(d["p"], b)
    stats = list(node.last_chunk_stats)
    gref = np.zeros(211)
    line = []
    for ci, k0 in enumerate(range(0, 211, 12)):
        k1 = min(211, k0 + 12)
        c = orc.make_hychem(dydt_scale=d["dydt_scale"], yscale=d["yscale"], errnorm_sens=mode, dual_partials=12, rtol=rtol)
        r = orc.hychem_solve_one(c, th, d["u0"][b], d["ts"], d["Ttab"][b], d["Ptab"][b], d["data"][b], dtheta=dth[k0:k1])
        gref[k0:k1] = r["grad"]
        e = np.max(np.abs(g[k0:k1] - r["grad"])) / np.max(np.abs(r["grad"]) + 1e-300)
        line.append(f"{stats[ci][0]}/{stats[ci][1]} vs {r['naccept']}/{r['nreject']} ({e:.1e})")
    print(f"traj {b} mode {mode} rtol {rtol}: grad {np.max(np.abs(g - gref)) / np.max(np.abs(gref)):.2e} | " + " ; ".join(line), flush=True)

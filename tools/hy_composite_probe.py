#!/usr/bin/env python3
"""HyChem composite (hychem_auto_kernel) against the oracle's composite, trajectory by trajectory: accepted / rejected steps and
deviations -- what tests/test_hychem.py's bars are taken from.  GPU."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crnn_amd import NeuralODE, ODEProblem, PRESET_HYCHEM, SOLVER_AUTOTSIT5, hychem as hy  # noqa: E402
from oracle import oracle as orc  # noqa: E402
orc.build()
d = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures_hychem.json")))
for k in ("ts", "u0", "Ttab", "Ptab", "data", "yscale", "dydt_scale", "p", "theta"):
    d[k] = np.array(d[k])
rng = np.random.default_rng(3)
_, u0s, Tts, Pts = hy.sample_conditions(13, rng)
datas = np.stack([d["data"][b % 3] * (1 + 0.05 * rng.standard_normal()) for b in range(13)])
u0 = np.concatenate([d["u0"], u0s]); data = np.concatenate([d["data"], datas])
Tt = np.concatenate([d["Ttab"], Tts]); Pt = np.concatenate([d["Ptab"], Pts])
B = u0.shape[0]
for atol, rtol in ((1e-8, 1e-3), (1e-11, 1e-7)):
    node = NeuralODE(ODEProblem(PRESET_HYCHEM, d["ts"], rate_scale=d["dydt_scale"], solver=SOLVER_AUTOTSIT5, atol=atol, rtol=rtol, maxiters=10**6))
    node.set_ensemble(u0, data, d["yscale"]); node.set_tables(Tt, Pt)
    c = orc.make_hychem(dydt_scale=d["dydt_scale"], yscale=d["yscale"], atol=atol, rtol=rtol, maxiters=10**6, solver=2)
    for name, p in (("fixture", d["p"]), ("true", hy.true_p()), ("init", hy.init_p(np.random.default_rng(3)))):
        th, _ = orc.hychem_p2vec(p)
        pred = node.predict_n_ode(p)
        losses = node.loss_n_ode(p)
        acc, rej = node.step_counts()
        for b in range(B):
            r = orc.hychem_solve_one(c, th, u0[b], d["ts"], Tt[b], Pt[b], data[b], want_pred=True)
            print(f"rtol {rtol} {name} b={b}: device acc {acc[b]} rej {rej[b]} | oracle acc {r['naccept']} (tsit5 {r['n_tsit5']}) rej {r['nreject']} | "
                  f"pred {np.max(np.abs(pred[b] - r['pred'])):.2e} loss {abs(losses[b] - r['loss']) / r['loss']:.2e}")

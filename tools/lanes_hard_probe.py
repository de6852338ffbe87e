#!/usr/bin/env python3
"""One lane vs a lane pair per trajectory at B = 65 536 in the regimes whose launch time is set by the longest trajectory (early-training
p, the diverged full-batch state): kernel ms per loss+gradient launch.  GPU."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_secondary as bs
from crnn_amd import NeuralODE, ODEProblem, Optimiser, PRESET_CASE2, cases

fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures.json")))
u0, data, ys = bs.case2_ensemble(65536, [1234, 0])
ck = np.array(fx["case2_ckpt"]["p"])
p_init = cases.case2_init_p(np.random.Generator(np.random.PCG64(7)))
p_hard = np.array(json.load(open(os.path.join(ROOT, "tests", "golden", "case2_hard_p.json")))["p"])
nd = NeuralODE(ODEProblem(PRESET_CASE2, cases.case2_tsteps()))
nd.set_ensemble(u0, data, ys)
nd.train_init(Optimiser(25, PRESET_CASE2), p_init)
for _ in range(30):
    nd.train_step(want_loss=False)
p_deg = nd.params()
nd.close()
for name, p in (("ckpt", ck), ("init", p_init), ("hard", p_hard), ("diverged", p_deg)):
    row = [name]
    for lanes in (1, 2):
        e = bs.case2_fixed(u0, data, ys, p, reps=4, lanes=lanes)
        row.append(f"lanes={lanes}: {e['kernel_ms']:.3f} ms (steps {e['steps_per_traj']:.1f}, rejects {e['rejects_per_traj']:.1f})")
    print(" | ".join(row), flush=True)

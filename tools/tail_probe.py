#!/usr/bin/env python3
"""Training steps of the headline ensemble with one lane, then two lanes per trajectory, for a rocprofv3 --kernel-trace run: how
long the tail launch (reduction + optimiser + sort) is behind each kernel."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import json  # noqa: E402
from crnn_amd import NeuralODE, ODEProblem, Optimiser, PRESET_CASE2, cases  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rng = np.random.Generator(np.random.PCG64(1234))
ts = cases.case2_tsteps()
u0 = cases.case2_u0(B, rng)
data = np.abs(rng.standard_normal((B, 6, len(ts)))) * 0.5
p = np.array(json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures.json")))["case2_ckpt"]["p"])
for lanes in (1, 2):
    node = NeuralODE(ODEProblem(PRESET_CASE2, ts))
    node.set_ensemble(u0, data, cases.max_min(data, lb=cases.LB_CASE2))
    node.set_lanes_per_traj(lanes)
    node.train_init(Optimiser(25, PRESET_CASE2), p)
    for _ in range(12):
        node.train_step(want_loss=False)
    node.params()
    node.close()

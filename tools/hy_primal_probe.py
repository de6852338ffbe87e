#!/usr/bin/env python3
"""HyChem, 32 768 trajectories: kernel time of the primal launch (loss_n_ode / predict_n_ode) next to the gradient launch."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crnn_amd import NeuralODE, ODEProblem, PRESET_HYCHEM, hychem as hy  # noqa: E402

B = int(os.environ.get("B", 32768))
rng = np.random.Generator(np.random.PCG64([1234, 4]))
ts, u0, Tt, Pt = hy.sample_conditions(B, rng)
node = NeuralODE(ODEProblem(PRESET_HYCHEM, ts, rate_scale=hy.DYDT_SCALE))
node.set_ensemble(u0, np.abs(rng.standard_normal((B, 9, len(ts)))) * 0.05, np.ones(9))
node.set_tables(Tt, Pt)
p = hy.true_p() + 0.02 * np.random.Generator(np.random.PCG64(5)).standard_normal(hy.NP)
p[-1] = 0.1
for name, fn in (("loss_and_grad", lambda: node.loss_and_grad(p)), ("losses", lambda: node.losses(p)), ("predict_n_ode", lambda: node.predict_n_ode(p))):
    k = []
    for _ in range(4):
        fn()
        k.append(node.stats()["kernel_ms"])
    print(f"{name:16s} kernel_ms {np.median(k):.3f}")
l1, _ = node.loss_and_grad(p)
l2 = float(np.mean(node.losses(p)))
print("mean loss: gradient launch %.15e, primal launch %.15e, rel diff %.1e" % (l1, l2, abs(l1 - l2) / abs(l1)))
node.close()

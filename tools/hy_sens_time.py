#!/usr/bin/env python3
"""HyChem dual-norm gradient (errnorm_sens = 2), wall time of one loss+gradient call over n trajectories: python tools/hy_sens_time.py [n]
(CRNN_SENS_ONE_LAUNCH=0: one launch per ForwardDiff chunk, 18 of them)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crnn_amd import NeuralODE, ODEProblem, PRESET_HYCHEM, hychem as hy  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rng = np.random.Generator(np.random.PCG64([1234, 4]))
ts, u0, Tt, Pt = hy.sample_conditions(n, rng)
node = NeuralODE(ODEProblem(PRESET_HYCHEM, ts, rate_scale=hy.DYDT_SCALE))
node.set_ensemble(u0, np.zeros((n, 9, len(ts))), np.ones(9)); node.set_tables(Tt, Pt)
clean = node.predict_n_ode(hy.true_p())
data = clean * (1.0 + 0.01 * rng.standard_normal(clean.shape))
ys = np.maximum((data.max(axis=2) - data.min(axis=2)).max(axis=0), hy.LB)
node.close()
p = hy.true_p() + 0.02 * np.random.Generator(np.random.PCG64(5)).standard_normal(hy.NP); p[-1] = 0.1
sens = NeuralODE(ODEProblem(PRESET_HYCHEM, ts, rate_scale=hy.DYDT_SCALE, errnorm_sens=2))
sens.set_ensemble(u0, data, ys); sens.set_tables(Tt, Pt)
L, G = sens.loss_and_grad(p)
t0 = time.perf_counter(); L, G = sens.loss_and_grad(p); w = (time.perf_counter() - t0) * 1e3
print(f"one_launch={os.environ.get('CRNN_SENS_ONE_LAUNCH', '1')} n={n} call_ms {w:.1f} ({n / (w * 1e-3):.0f} traj+grads/s) loss {L:.12e} |g| {np.linalg.norm(G):.12e}")

#!/usr/bin/env python3
"""Randomised Cathode-UQ parity sweep on the GPU: random particles around the deterministic optimum (sigma up to 10 %), the
five measured heating rates plus random ones, random tolerances and maxiters truncation; per (particle, heating rate) the
device's loss, heat-release curve and 17-component adjoint gradient against the oracle's complex-step tangents.
usage: python tools/fuzz_cathode.py [--n 60] [--seed 0]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=60)
ap.add_argument("--seed", type=int, default=0)
args = ap.parse_args()
from crnn_amd.cathode import CathodeUQ  # noqa: E402
from oracle import oracle as orc  # noqa: E402

orc.build(); orc.lib()
fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures_cathode.json")))
ps = np.array(fx["theta"])
meas = np.array([s["beta"] for s in fx["sets"]])
worst = dict(loss=0.0, hrr=0.0, grad=0.0, fa=0.0, kink=0.0)
nfail = nfork = ntraj = nkink = 0
for it in range(args.n):
    rng = np.random.Generator(np.random.PCG64([args.seed, it, 9]))
    n_rates = int(rng.integers(1, 7))
    betas = np.exp(rng.uniform(np.log(2.0), np.log(20.0), n_rates))
    exp_data, sets = [], []
    for bta in betas:
        s = fx["sets"][int(np.argmin(np.abs(np.log(meas) - np.log(bta))))]
        dbar, d2bar = np.array(s["dbar"]), np.array(s["d2bar"])
        sd = np.sqrt(np.maximum(d2bar - dbar ** 2, 0.0))
        tsb = np.array(s["ts"]) * s["beta"] / bta
        exp_data.append(np.stack([tsb, dbar + sd, dbar - sd], axis=1))
        sets.append((tsb, dbar, d2bar))
    rtol = float(10.0 ** rng.uniform(-6, -3)); atol = float(10.0 ** rng.uniform(-13, -9))
    maxiters = int(rng.choice([2500000, 2500000, 200]))
    N = int(rng.integers(1, 9))
    p = 1 + rng.uniform(0.005, 0.1) * rng.standard_normal((N, 17))
    p[:, 6:9] = 0.0
    uq = CathodeUQ(exp_data, betas, fx["theta"], atol=atol, rtol=rtol, maxiters=maxiters, errnorm_sens=0)
    loss, grad, hrr = uq.solve(p, want_hrr=True)
    fwd = CathodeUQ(exp_data, betas, fx["theta"], atol=atol, rtol=rtol, maxiters=maxiters, grad_mode=1, errnorm_sens=0)   # 14 tangent columns
    _, gfwd, _ = fwd.solve(p)
    for n in range(N):
        for i, (tsb, dbar, d2bar) in enumerate(sets):
            c = orc.make_cathode(betas[i], atol=atol, rtol=rtol, maxiters=maxiters)
            r = orc.cathode_solve_one(c, p[n] * ps, tsb, dbar, d2bar)
            ntraj += 1
            D = len(tsb)
            gsc = np.max(np.abs(r["grad"] * ps)) + 1e-300
            dl = abs(loss[n, i] - r["loss"]) / abs(r["loss"])
            dh = np.max(np.abs(hrr[n, i, :r["n_saved"]] - r["hrr"][:r["n_saved"]])) / max(1.0, np.max(np.abs(r["hrr"])))
            dg = np.max(np.abs(grad[n, i] - r["grad"] * ps)) / gsc
            if uq.last_retcode[n, i] != r["retcode"] or uq.last_n_saved[n, i] != r["n_saved"]:
                nfork += 1          # the truncation point moved by a step: compare nothing else
                continue
            # the device's two gradient algorithms (reversed steps / forward tangents) must agree with each other always
            dfa = np.max(np.abs(grad[n, i] - gfwd[n, i])) / (np.max(np.abs(gfwd[n, i])) + 1e-300)
            worst["loss"] = max(worst["loss"], dl); worst["hrr"] = max(worst["hrr"], dh); worst["fa"] = max(worst["fa"], dfa)
            if dg > 1e-5 and dl <= 1e-7 and dh <= 1e-6 and dfa <= 1e-6:
                # primal equal, both device gradients equal, oracle gradient different: a depleted species sits at lb_clamp = 1e-16,
                # where rounding decides on which side of the clamp's kink (derivative 0 or 1/u = 1e16) a stage value falls
                nkink += 1
                worst["kink"] = max(worst["kink"], dg)
                continue
            worst["grad"] = max(worst["grad"], dg)
            if dl > 1e-7 or dh > 1e-6 or dg > 1e-5 or dfa > 1e-6:
                nfail += 1
                print(f"[{it}] particle {n} beta {betas[i]:.2f} rtol {rtol:.1e} atol {atol:.1e} maxiters {maxiters}: dloss {dl:.2e} dhrr {dh:.2e} "
                      f"dgrad {dg:.2e} rc {uq.last_retcode[n, i]} steps {r['naccept']}", flush=True)
print("cathode sweep: %d trajectories; worst deviations from the oracle: loss %.2e, heat-release curve %.2e, gradient %.2e of max|grad| ; "
      "adjoint vs forward tangents %.2e ; trajectories with a depleted species on the clamp's kink (gradients differ by up to %.1e) %d ; "
      "truncation points that moved %d ; failures %d" % (ntraj, worst["loss"], worst["hrr"], worst["grad"], worst["fa"], worst["kink"], nkink, nfork, nfail))
sys.exit(1 if nfail else 0)

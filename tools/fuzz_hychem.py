#!/usr/bin/env python3
"""Randomised HyChem parity sweep on the GPU: random conditions (fuel fraction, T, P), parameter vectors around the true
mechanism and around the reference's initialiser, tolerances and horizons; the device's adjoint gradient (211 components)
is checked against the oracle's complex-step tangents along 4 random parameter directions per problem.
usage: python tools/fuzz_hychem.py [--n 40] [--seed 0]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=40)
ap.add_argument("--seed", type=int, default=0)
args = ap.parse_args()
from crnn_amd import NeuralODE, ODEProblem, PRESET_HYCHEM, hychem as hy  # noqa: E402
from oracle import oracle as orc  # noqa: E402

orc.build(); orc.lib()
worst = dict(pred=0.0, loss=0.0, dgrad=0.0, ill=0.0)
nfail = nfork = nill = 0
for it in range(args.n):
    rng = np.random.Generator(np.random.PCG64([args.seed, it, 3]))
    B = int(rng.integers(1, 5))
    ts, u0, Tt, Pt = hy.sample_conditions(B, rng)
    kind = rng.random()
    p = hy.true_p() + 0.05 * rng.standard_normal(hy.NP) if kind < 0.6 else hy.init_p(rng)
    p[-1] = 0.1
    rtol = float(10.0 ** rng.uniform(-5, -3)); atol = rtol * 1e-5
    data = np.abs(rng.standard_normal((B, 9, len(ts)))) * 0.05
    ys = np.maximum((data.max(axis=2) - data.min(axis=2)).max(axis=0), hy.LB)
    sample = int(rng.integers(20, len(ts) + 1))
    node = NeuralODE(ODEProblem(PRESET_HYCHEM, ts, rate_scale=hy.DYDT_SCALE, atol=atol, rtol=rtol))
    node.set_ensemble(u0, data, ys)
    node.set_tables(Tt, Pt)
    c = orc.make_hychem(dydt_scale=hy.DYDT_SCALE, yscale=ys, atol=atol, rtol=rtol)
    th, dth = orc.hychem_p2vec(p)
    V = rng.standard_normal((4, hy.NP)); V /= np.linalg.norm(V, axis=1, keepdims=True)
    dirs = V @ dth                                             # [4, nth]: d theta along the 4 directions
    losses = node.losses(p, sample=sample)
    nacc_dev = 0
    for b in range(B):
        g = node.gradient(p, b, sample=sample)
        nacc_dev = node.last_stats["n_accept"]
        r = orc.hychem_solve_one(c, th, u0[b], ts, Tt[b], Pt[b], data[b], dtheta=dirs, sample=sample)
        same = nacc_dev == r["naccept"] and node.last_stats["n_reject"] == r["nreject"]
        dl = abs(losses[b] - r["loss"]) / max(abs(r["loss"]), 1e-300)
        dd = np.max(np.abs(V @ g - r["grad"])) / max(np.max(np.abs(r["grad"])), 1e-300)
        if same and dd <= 1e-4:
            worst["loss"] = max(worst["loss"], dl); worst["dgrad"] = max(worst["dgrad"], dd)
        else:
            nfork += 1
        suspicious = (same and (dl > 1e-7 or dd > 1e-4)) or ((not same) and dl > 10 * rtol) or node.last_retcode[b] != r["retcode"]
        if suspicious and node.last_retcode[b] == r["retcode"]:
            # is the discrete map itself that sensitive here?  perturb p by 1e-13 (relative) and look at the ORACLE's own change:
            # species crossing their clamp at 1e-8 make the step-size controller amplify last-bit differences by many orders
            own = 0.0
            for kk in range(3):
                pp = p * (1 + 1e-13 * np.random.default_rng(kk).standard_normal(p.size))
                t2, d2 = orc.hychem_p2vec(pp)
                r2 = orc.hychem_solve_one(c, t2, u0[b], ts, Tt[b], Pt[b], data[b], dtheta=V @ d2, sample=sample)
                own = max(own, np.max(np.abs(r2["grad"] - r["grad"])) / max(np.max(np.abs(r["grad"])), 1e-300))
            if own > 0.1 * dd:
                nill += 1
                suspicious = False
                worst["ill"] = max(worst["ill"], dd)
        if suspicious:
            nfail += 1
            print(f"[{it}.{b}] kind {'true' if kind < 0.6 else 'init'} rtol {rtol:.1e} sample {sample} same_steps {same} dloss {dl:.2e} "
                  f"ddir {dd:.2e} rc {node.last_retcode[b]} / {r['retcode']} steps {nacc_dev} / {r['naccept']}", flush=True)
    node.close()
print("HyChem sweep: worst deviations on identical step sequences: loss %.2e, directional derivatives %.2e ; forked step sequences %d ; "
      "ill-conditioned cases (the oracle's own gradient moves as much under a 1e-13 perturbation of p) %d, deviation up to %.1e ; failures %d"
      % (worst["loss"], worst["dgrad"], nfork, nill, worst["ill"], nfail))
sys.exit(1 if nfail else 0)

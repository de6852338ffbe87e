#!/usr/bin/env python3
"""Randomised sweep of the HyChem dual-norm gradient (errnorm_sens: hychem_sens2_kernel, or its COMPOSITE instantiation with --composite)
against the CPU oracle's chunked solves: random conditions, parameter vectors around the true mechanism and around the reference's
initialiser, tolerances, horizons.  Per trajectory all 18 ForwardDiff chunks: step counts and gradient pieces.  A chunk whose step counts
differ is classified with the oracle itself (a 1e-13 perturbation of p: does ITS count move?) -- the hot trajectories' discrete maps are that
sensitive (tests/test_hychem.py).  Runs on an MI355X, or on the CPU through the SIMT emulator (CRNN_HIP_LIB=tests/simt/libcrnn_simt.so).
usage: python tools/fuzz_hychem_sens.py [--n 12] [--seed 0] [--mode 2] [--composite]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=12)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--mode", type=int, default=2)
ap.add_argument("--composite", action="store_true")
a = ap.parse_args()
from crnn_amd import NeuralODE, ODEProblem, PRESET_HYCHEM, SOLVER_AUTOTSIT5, hychem as hy  # noqa: E402
from crnn_amd import _lib as L  # noqa: E402
from oracle import oracle as orc  # noqa: E402

orc.build(); orc.lib()
rng = np.random.default_rng(a.seed)
n_exact = n_fork_ill = n_fail = n_chunks = 0
worst = 0.0
for it in range(a.n):
    ts, u0, Tt, Pt = hy.sample_conditions(1, rng)
    kind = rng.random()
    p = hy.true_p() + 0.03 * rng.standard_normal(hy.NP) if kind < 0.6 else 0.1 * rng.standard_normal(hy.NP)
    p[-1] = 0.1
    rtol = float(10.0 ** rng.uniform(-4, -3)); atol = rtol * 1e-5
    data = np.abs(rng.standard_normal((1, 9, len(ts)))) * 0.05
    ys = np.maximum((data.max(axis=2) - data.min(axis=2)).max(axis=0), hy.LB)
    kw = dict(solver=SOLVER_AUTOTSIT5) if a.composite else {}
    node = NeuralODE(ODEProblem(PRESET_HYCHEM, ts, rate_scale=hy.DYDT_SCALE, atol=atol, rtol=rtol, errnorm_sens=a.mode, **kw))
    node.set_ensemble(u0, data, ys); node.set_tables(Tt, Pt)
    g = node.gradient(p, 0)
    stats = list(node.last_chunk_stats)
    node.close()
    mk = lambda: orc.make_hychem(dydt_scale=hy.DYDT_SCALE, yscale=ys, atol=atol, rtol=rtol, errnorm_sens=a.mode, dual_partials=12, solver=2 if a.composite else 0)
    th, dth = orc.hychem_p2vec(p)
    th2, dth2 = orc.hychem_p2vec(p * (1 + 1e-13 * np.random.default_rng(it).standard_normal(hy.NP)))
    gref = np.zeros(hy.NP); same = []; moved = []
    for ci, k0 in enumerate(range(0, hy.NP, 12)):
        k1 = min(hy.NP, k0 + 12)
        r = orc.hychem_solve_one(mk(), th, u0[0], ts, Tt[0], Pt[0], data[0], dtheta=dth[k0:k1])
        gref[k0:k1] = r["grad"]
        same.append(stats[ci] == (r["naccept"], r["nreject"]))
        if not same[-1]:
            r2 = orc.hychem_solve_one(mk(), th2, u0[0], ts, Tt[0], Pt[0], data[0], dtheta=dth2[k0:k1])
            moved.append((r2["naccept"], r2["nreject"]) != (r["naccept"], r["nreject"]))
    gmax = np.max(np.abs(gref)) + 1e-300
    dev = np.max(np.abs(g - gref)) / gmax
    n_chunks += 18; n_exact += sum(same)
    forks = 18 - sum(same)
    if forks == 0:
        worst = max(worst, dev)
        if dev > 1e-5: n_fail += 1; print(f"[{it}] identical step sequences but gradient off by {dev:.2e}", flush=True)
    else:
        # a forked chunk is acceptable where the oracle's own count moves under a 1e-13 perturbation (or the gradient still agrees to solver tolerance)
        if all(moved) or dev < 5e-2: n_fork_ill += forks
        else: n_fail += 1; print(f"[{it}] {forks} chunks forked, oracle stable, gradient off by {dev:.2e}", flush=True)
    print(f"[{it}] {'true' if kind < 0.6 else 'init'} rtol {rtol:.1e}: {sum(same)}/18 chunks step for step, gradient deviation {dev:.2e}", flush=True)
print(f"HyChem dual-norm sweep ({'composite' if a.composite else 'Rosenbrock23'}, mode {a.mode}, library [{L.lib.crnn_build_info().decode()}]): {n_exact} of {n_chunks} chunk solves step for step; "
      f"worst gradient deviation on fully identical trajectories {worst:.2e}; forked chunks on ill-conditioned / tolerance-equal trajectories {n_fork_ill}; failures {n_fail}")
sys.exit(1 if n_fail else 0)

#!/usr/bin/env python3
"""CPU (oracle) census, config 5's model: how far is the gradient the device forms for dlnprob -- Rosenbrock23, primal error norm (the
discrete adjoint's derivative) -- from the gradient the reference really evaluates -- ForwardDiff's chunks 9 + 8 through
AutoTsit5(TRBDF2) with the partials in every error estimate (oracle solver 3 with errnorm_sens = 2) -- on a perturbed particle cloud
around the reference's parameter vector, at the reference's tolerances; both against the Rosenbrock23 gradient at tight tolerance
(the converged sensitivity).  usage: python tools/cathode_gradient_census.py [n_particles] [spread] [--device]
--device: the same cloud through the LIBRARY as well (CathodeUQ in its default configuration, and with errnorm_sens = 0: the adjoint) --
on an MI355X, or on the CPU through the SIMT emulator (CRNN_HIP_LIB=tests/simt/libcrnn_simt.so): what a caller of dlnprob gets."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402

orc.build()
DEVICE = "--device" in sys.argv
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(argv[0]) if len(argv) > 0 else 16
spread = float(argv[1]) if len(argv) > 1 else 0.05
cfx = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures_cathode.json")))
th0 = np.array(cfx["theta"])
rng = np.random.default_rng(5)
P = 1 + spread * rng.standard_normal((N, 17)); P[:, 6:9] = 0.0
d_dev, d_ref, d_dev2, newton, steps_dev, steps_ref, n_unconv = [], [], [], 0, 0, 0, 0
d_lib, d_lib0 = [], []
if DEVICE:
    from crnn_amd import _lib as L
    from crnn_amd.cathode import CathodeUQ
    two = lambda s_: np.column_stack([s_["ts"], np.array(s_["dbar"]) + np.sqrt(np.maximum(np.array(s_["d2bar"]) - np.array(s_["dbar"]) ** 2, 0.0)),
                                      np.array(s_["dbar"]) - np.sqrt(np.maximum(np.array(s_["d2bar"]) - np.array(s_["dbar"]) ** 2, 0.0))])
    exp = [two(s_) for s_ in cfx["sets"]]; betas = [s_["beta"] for s_ in cfx["sets"]]
    uq = CathodeUQ(exp, betas, th0)                       # the default configuration
    _, g_lib, _ = uq.solve(P, want_grad=True)             # d loss / d p = d loss / d theta * p_scales
    uq0 = CathodeUQ(exp, betas, th0, errnorm_sens=0)
    _, g_lib0, _ = uq0.solve(P, want_grad=True)
    lib_info = L.lib.crnn_build_info().decode()
for n in range(N):
    th = P[n] * th0
    for s in cfx["sets"]:
        args = (th, s["ts"], s["dbar"], s["d2bar"])
        conv = orc.cathode_solve_one(orc.make_cathode(s["beta"], atol=1e-14, rtol=1e-8), *args)["grad"]
        dev = orc.cathode_solve_one(orc.make_cathode(s["beta"]), *args)
        g = np.zeros(17)
        for cc, (lo, k) in zip(orc.cathode_sens_chunks(orc.make_cathode(s["beta"], solver=3), th0, mode=2), ((0, 9), (9, 8))):
            r = orc.cathode_solve_one(cc, *args)
            assert r["retcode"] == 0
            g[lo:lo + k] = r["grad"][lo:lo + k]
            newton += r["n_newton"]; steps_ref += r["naccept"]
        g0 = np.zeros(17)                 # the device's own reference-faithful mode: the same chunks and norm on Rosenbrock23
        for cc, (lo, k) in zip(orc.cathode_sens_chunks(orc.make_cathode(s["beta"]), th0, mode=2), ((0, 9), (9, 8))):
            r = orc.cathode_solve_one(cc, *args)
            g0[lo:lo + k] = r["grad"][lo:lo + k]
        steps_dev += dev["naccept"]
        # the converged sensitivity: Rosenbrock23 at rtol 1e-10; where even that disagrees with rtol 1e-8 by more than 1e-4 the trajectory's
        # tangents are unstable at tight tolerance too and it is reported separately
        conv8 = orc.cathode_solve_one(orc.make_cathode(s["beta"], atol=1e-14, rtol=1e-10), *args)["grad"]
        sc = np.max(np.abs(conv8))
        if not np.max(np.abs(conv8 - conv)) < 1e-4 * sc:
            n_unconv += 1
            continue
        d_dev.append(np.max(np.abs(dev["grad"] - conv8)) / sc)
        d_ref.append(np.max(np.abs(g - conv8)) / sc)
        d_dev2.append(np.max(np.abs(g0 - conv8)) / sc)
        if DEVICE:
            i_s = cfx["sets"].index(s)
            with np.errstate(divide="ignore", invalid="ignore"):
                to_theta = lambda gp: np.where(th0 != 0, gp / np.where(th0 != 0, th0, 1.0), 0.0)     # back to d loss / d theta where p_scales != 0
            msk = th0 != 0
            d_lib.append(np.max(np.abs(to_theta(g_lib[n, i_s]) - conv8)[msk]) / sc)
            d_lib0.append(np.max(np.abs(to_theta(g_lib0[n, i_s]) - conv8)[msk]) / sc)
d_dev, d_ref, d_dev2 = np.array(d_dev), np.array(d_ref), np.array(d_dev2)
q = lambda a: f"median {np.median(a):.2e}  90 % {np.quantile(a, 0.9):.2e}  max {a.max():.2e}  off by more than 0.1: {int((a > 0.1).sum())} of {a.size}"
print(f"{N} particles (spread {spread}) x {len(cfx['sets'])} heating rates, reference tolerances (abstol 1e-12, reltol 1e-3); distance to the converged "
      f"sensitivity, relative to its largest entry:")
print(f"  Rosenbrock23, primal norm (what the device's adjoint differentiates): {q(d_dev)}   ({steps_dev / len(d_dev):.0f} accepted steps per trajectory)")
print(f"  Rosenbrock23, ForwardDiff's chunks and norm (the device's errnorm_sens):  {q(d_dev2)}")
print(f"  AutoTsit5(TRBDF2), ForwardDiff's chunks and norm (the reference):     {q(d_ref)}   ({steps_ref / (2 * len(d_ref)):.0f} per chunk solve, {newton} Newton iterations in all)")
if DEVICE:
    print(f"  library [{lib_info}], default configuration (errnorm_sens = 2):                {q(np.array(d_lib))}")
    print(f"  library, errnorm_sens = 0 (the primal-norm adjoint, opt-in):                     {q(np.array(d_lib0))}")
print(f"  ({n_unconv} trajectories left out: their Rosenbrock23 sensitivities at rtol 1e-8 and 1e-10 disagree by more than 1e-4 -- no converged value to compare with)")

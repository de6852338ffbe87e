#!/usr/bin/env python3
"""Does the cathode's composite ever reach its stiff branch?  ("What's missing" 1 / item 7.)

The reference integrates the Bayesian cathode model with AutoTsit5(TRBDF2(autodiff = true))
(Cathode_NCM333_UQ/src_333/network.jl:195,205-212).  The oracle carries the restated AutoSwitch for this model
(oracle/crnn_oracle.c: orc_cathode_solve_one, solver = 2 -- [UNVERIFIED-DEP], restated from the published algorithm).
Round 2 showed on the reference's FIVE heating rates that the detector never reaches its 11th stiff step in a row, i.e.
the runs are pure Tsit5 runs and whichever stiff algorithm is configured never executes.  This tool takes the census on
BASELINE config 5 itself: 4 096 particles (1 + 1e-3 N(0,1) around the reference's initialiser scales, as bench.py draws
them; --spread widens the cloud) x 256 heating rates log-spaced in [2, 20] K/min = 1 048 576 trajectories, CPU, primal only.

    python tools/cathode_autoswitch_census.py [--particles 4096] [--rates 256] [--spread 1e-3] [--json profiles/...]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--particles", type=int, default=4096)
ap.add_argument("--rates", type=int, default=256)
ap.add_argument("--spread", type=float, default=1e-3)
ap.add_argument("--json", default=None)
ap.add_argument("--threads", type=int, default=0)
a = ap.parse_args()
from oracle import oracle as orc  # noqa: E402
orc.build()
fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures_cathode.json")))
betas = np.exp(np.linspace(np.log(2.0), np.log(20.0), a.rates))
meas = np.array([s["beta"] for s in fx["sets"]])
Dmax = max(len(s["ts"]) for s in fx["sets"])
ts = np.zeros((a.rates, Dmax)); D = np.zeros(a.rates, np.int32)
for i, b in enumerate(betas):          # the measured temperature grid of the nearest heating rate, on that rate's clock (bench_secondary.cathode)
    s = fx["sets"][int(np.argmin(np.abs(np.log(meas) - np.log(b))))]
    t = np.array(s["ts"]) * s["beta"] / b
    D[i] = t.size
    ts[i, :t.size] = t
    ts[i, t.size:] = t[-1] + np.arange(1, Dmax - t.size + 1)
rng = np.random.default_rng(0)
p = 1 + a.spread * rng.standard_normal((a.particles, 17))
p[:, 6:9] = 0.0
theta = p * np.array(fx["theta"])[:17]
out = {}
ap_solvers = (("AutoTsit5(TRBDF2) -- the reference's algorithm (restated AutoSwitch + restated TRBDF2 / Newton machinery, smooth_est as the package reads)", 3, 0),
              ("AutoTsit5(TRBDF2), Shampine's scaling of the smoothed estimate", 3, 1),
              ("AutoTsit5 composite (restated AutoSwitch, stiff branch = Rosenbrock23)", 2, 0), ("Rosenbrock23 (the device's gradient path)", 0, 0))
for name, solver, est in ap_solvers:
    c = orc.make_cathode(1.0, solver=solver, trbdf2_est=est)
    t0 = time.time()
    r = orc.cathode_census(c, theta, betas, ts, D, nthreads=a.threads)
    r["seconds"] = time.time() - t0
    r["mean_accepted"] = r["accepted"] / r["trajectories"]
    out[name] = r
    print(name, r, flush=True)
# which heating rates switch: the census rate by rate (composite only)
per_rate = []
c = orc.make_cathode(1.0, solver=2)
for i in range(a.rates):
    r = orc.cathode_census(c, theta, betas[i:i + 1], ts[i:i + 1], D[i:i + 1], nthreads=a.threads)
    per_rate.append(dict(beta=float(betas[i]), switched=r["trajectories"] - r["never_left_tsit5"], stiff_steps=r["accepted"] - r["accepted_tsit5"],
                         mean_accepted=r["accepted"] / r["trajectories"]))
sw = [q for q in per_rate if q["switched"]]
print(f"heating rates with at least one switching particle: {len(sw)} of {a.rates}; beta range of those: "
      f"{min((q['beta'] for q in sw), default=None)} .. {max((q['beta'] for q in sw), default=None)}", flush=True)
res = dict(particles=a.particles, rates=a.rates, spread=a.spread, reference="Cathode_NCM333_UQ/src_333/network.jl:195,205-212", census=out,
           per_rate=per_rate, reference_rates_K_per_min=[float(x) for x in meas])
if a.json:
    json.dump(res, open(a.json, "w"), indent=1)

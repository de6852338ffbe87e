#!/usr/bin/env python3
"""SVGD move timing (BASELINE config 5's particle count): python tools/svgd_bench.py [N] -- HIP-event time of the device-resident
move (crnn_cathode_svgd_step) and wall time of the stateless host-array entry point (crnn_svgd_update)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crnn_amd.cathode import CathodeUQ, svgd_update
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures_cathode.json")))
exp = []
for s in fx["sets"]:
    dbar, d2bar = np.array(s["dbar"]), np.array(s["d2bar"])
    sd = np.sqrt(np.maximum(d2bar - dbar ** 2, 0.0))
    exp.append(np.stack([np.array(s["ts"]), dbar + sd, dbar - sd], axis=1))
uq = CathodeUQ(exp, [s["beta"] for s in fx["sets"]], fx["theta"])
rng = np.random.default_rng(0)
p = 1 + 1e-3 * rng.standard_normal((N, 17)); p[:, 6:9] = 0.0
uq.set_particles(p)
sv, so = [], []
for it in range(12):
    _, h, ms = uq.svgd_step(it % 5, 1e-3)
    sv.append(ms["svgd_ms"]); so.append(ms["solve_ms"])
print(f"device-resident loop, N = {N}: SVGD move {np.median(sv[2:]):.3f} ms, solve (one heating rate) {np.median(so[2:]):.3f} ms, h = {h:.6e}")
g = rng.standard_normal((N, 17))
ws = []
for it in range(8):
    t0 = time.perf_counter(); svgd_update(p, g, 1e-3); ws.append((time.perf_counter() - t0) * 1e3)
print(f"crnn_svgd_update (host arrays in and out), N = {N}: {np.median(ws[2:]):.3f} ms per call")
uq.close()

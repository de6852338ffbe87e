#!/usr/bin/env python3
"""tools/case2_stream_controller_ablation.py -- which of the oracle's restated step-size-controller constants does the reference's recorded case2
history confirm?  (profiles/r06c_*)  OrdinaryDiffEq is not in /root/reference; the oracle restates its PI controller for Tsit5 (beta1 = 7/(10 k),
beta2 = 2/(5 k) with k = 5; gamma 9/10; qmin 1/5; qmax 10; no steady band; qoldinit 1e-4) [UNVERIFIED-DEP].  The replay of the reference's own run
(tools/case2_stream_replay.py) is sensitive to every one of them: each row below changes ONE constant and reports the final-loss pin and the
deviation of the first six replayed epochs from the recorded ones.  CPU only (the oracle)."""
import json, sys, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from oracle import oracle as orc
orc.build()
from crnn_amd import cases
import test_case2_stream_pin as T
fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures_case2_stream.json"))); d = fx["design"]; rec = fx["recorded"]
des = dict(u0=np.array(d["u0"]), ts=np.array(d["tsteps"]), data=np.array(d["data"]), ys=np.array(d["yscale"]), p0=np.array(d["p0"]), perms=d["perms"])
ck = np.array(json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures.json")))["case2_ckpt"]["p"])
def mk(mode, **ctl):
    pb = orc.make_problem(ns=6, nr=3, has_temp=1, lb=cases.LB_CASE2, ub=10.0, inv_R=cases.INV_R, atol=1e-6, rtol=1e-3, yscale=des["ys"], clamp_pred=1, solver=1, errnorm_sens=mode)
    for k, v in ctl.items(): setattr(pb, k, v)
    return pb
base = mk(2); print("defaults:", {k: getattr(base, k) for k in ("gamma","qmin","qmax","beta1","beta2","qsteady_min","qsteady_max","qoldinit")})
def run(ctl, epochs=6):
    pbg = mk(2, **ctl); pbl = mk(0, **ctl)
    th, _ = orc.p2vec(2, 6, 3, ck)
    r = orc.solve_batch(pbl, th, np.ascontiguousarray(des["u0"].T), des["ts"], np.ascontiguousarray(des["data"].transpose(2, 1, 0)))
    tr, va = T._split(r["loss"]); fin = max(abs(tr / rec["l_loss_train_last"] - 1), abs(va / rec["l_loss_val_last"] - 1))
    opt = orc.Optimiser(25, eta=0.005, wd=T.WD, expdecay=T.EXPDECAY); p = des["p0"].copy(); dev = []
    for ep in range(epochs):
        for i in des["perms"][ep]:
            p = opt.update(p, T._oracle_gradient(orc, pbg, des, p, i - 1))
        r = orc.solve_batch(pbl, orc.p2vec(2, 6, 3, p)[0], np.ascontiguousarray(des["u0"].T), des["ts"], np.ascontiguousarray(des["data"].transpose(2, 1, 0)))
        tr, va = T._split(r["loss"])
        dev += [abs(tr / rec["l_loss_train_head"][ep] - 1), abs(va / rec["l_loss_val_head"][ep] - 1)]
    return fin, np.max(dev), np.median(dev)
variants = [("restated defaults (Tsit5: beta1 7/50, beta2 2/25, gamma 0.9, qmin 0.2, qmax 10, qsteady [1,1], qoldinit 1e-4)", {}),
            ("beta exponents of a 2nd/3rd-order method (7/20, 2/10)", dict(beta1=7/20, beta2=2/10)),
            ("plain I controller (beta1 = 1/5, beta2 = 0)", dict(beta1=1/5, beta2=0.0)),
            ("gamma 0.8", dict(gamma=0.8)), ("gamma 0.95", dict(gamma=0.95)),
            ("qmax 5", dict(qmax=5.0)), ("qmin 0.1", dict(qmin=0.1)), ("qmin 0.5", dict(qmin=0.5)),
            ("qsteady_max 1.2", dict(qsteady_max=1.2)), ("qoldinit 1e-2", dict(qoldinit=1e-2)), ("qoldinit 1", dict(qoldinit=1.0))]
print(f"{'variant':105s} final-loss dev   replay max(1-6)  median(1-6)")
for name, ctl in variants:
    f, mx, md = run(ctl)
    print(f"{name:105s} {f:.1e}        {mx:.1e}         {md:.1e}")

#!/usr/bin/env python3
"""tools/case2_stream_replay.py -- the reference's recorded case2 training history against a replay of its own run (profiles/r06a_*).

case2/case2.jl:11 seeds Julia's RNG; tests/golden/fixtures_case2_stream.json holds the experiments, the initial p and the epoch shuffles re-drawn
from that stream (tests/golden/julia_rng.py, case2_stream.py) next to the first 100 entries of `l_loss_train` / `l_loss_val` the reference's
checkpoint recorded.  This prints, per epoch, the replay's loss against the recorded one for the three error norms a `ForwardDiff.gradient` through
the adaptive solver could have used, the final-loss pin, and the replay's sensitivity to a 0.1 % change of rtol.

    python tools/case2_stream_replay.py [--epochs 100]            # CPU: the oracle
    python tools/case2_stream_replay.py --device [--epochs 100]   # MI355X: the product's device-resident training loop (errnorm_sens = 2)
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=100)
    ap.add_argument("--device", action="store_true")
    a = ap.parse_args()
    import test_case2_stream_pin as T
    from crnn_amd import cases
    from oracle import oracle as orc
    orc.build()
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures_case2_stream.json")))
    d, rec = fx["design"], fx["recorded"]
    des = dict(u0=np.array(d["u0"]), ts=np.array(d["tsteps"]), data=np.array(d["data"]), ys=np.array(d["yscale"]), p0=np.array(d["p0"]), perms=d["perms"])
    ne = min(a.epochs, len(des["perms"]))
    ref = np.stack([rec["l_loss_train_head"][:ne], rec["l_loss_val_head"][:ne]], axis=1)
    ck = np.array(json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures.json")))["case2_ckpt"]["p"])

    print(f"# final losses at the saved p (iter {rec['iter']}): recorded train {rec['l_loss_train_last']:.7e} val {rec['l_loss_val_last']:.7e}")
    for name, solver in (("AutoTsit5(Rosenbrock23) [= Tsit5 here]", 2), ("Tsit5", 1), ("Rosenbrock23", 0)):
        tr, va = T._split(T._oracle_losses(orc, des, ck, solver=solver))
        print(f"#   oracle {name:40s} train {tr:.7e} ({tr / rec['l_loss_train_last'] - 1:+.2e})  val {va:.7e} ({va / rec['l_loss_val_last'] - 1:+.2e})")

    def replay_oracle(mode, rtol=1e-3):
        pb = orc.make_problem(ns=6, nr=3, has_temp=1, lb=cases.LB_CASE2, ub=10.0, inv_R=cases.INV_R, atol=1e-6, rtol=rtol, yscale=des["ys"], clamp_pred=1,
                              solver=1, errnorm_sens=mode)
        opt = orc.Optimiser(25, eta=0.005, wd=T.WD, expdecay=T.EXPDECAY)
        p = des["p0"].copy()
        out = []
        for ep in range(ne):
            for i in des["perms"][ep]:
                p = opt.update(p, T._oracle_gradient(orc, pb, des, p, i - 1))
            out.append(T._split(T._oracle_losses(orc, des, p)))
        return np.array(out)

    runs = {}
    if a.device:
        from crnn_amd import PRESET_CASE2, SOLVER_TSIT5, Optimiser
        node = T._node(des, solver=SOLVER_TSIT5, errnorm_sens=2)
        node.train_init(Optimiser(25, preset=PRESET_CASE2), des["p0"])
        h = []
        for ep in range(ne):
            for i in des["perms"][ep]:
                node.train_step(first=i - 1, count=1, want_loss=False)
            h.append(T._split(node.losses(node.params())))
        runs["device, errnorm_sens 2"] = np.array(h)
    runs["oracle, errnorm_sens 2 (/ totallength)"] = replay_oracle(2)
    runs["oracle, errnorm_sens 1 (/ length)"] = replay_oracle(1)
    runs["oracle, errnorm_sens 0 (primal norm)"] = replay_oracle(0)
    runs["oracle, errnorm_sens 2, rtol x 1.001"] = replay_oracle(2, rtol=1.001e-3)
    names = list(runs)
    print("# replay of the first epochs: relative deviation of (train, val) loss from the recorded history; Tsit5, ForwardDiff chunks 9 + 9 + 7, ExpDecay -> ADAM -> WeightDecay")
    print("# epoch  recorded_train recorded_val   " + "   ".join(f"[{k}] {n}" for k, n in enumerate(names)))
    for ep in range(ne):
        row = "  ".join(f"{runs[n][ep, 0] / ref[ep, 0] - 1:+.1e}/{runs[n][ep, 1] / ref[ep, 1] - 1:+.1e}" for n in names)
        print(f"{ep + 1:5d}  {ref[ep, 0]:.7e} {ref[ep, 1]:.7e}   {row}")
    print("# summary: max |dev| epochs 1-6 / median |dev| all epochs / max |dev| all epochs")
    for n in names:
        dev = np.abs(runs[n] / ref - 1.0)
        print(f"#   {n:45s} {dev[:6].max():.1e} / {np.median(dev):.1e} / {dev.max():.1e}")


if __name__ == "__main__":
    main()

"""Decode the training histories the reference's own checkpoints hold into tests/golden/fixtures_ckpt_history.json.

The reference has no tests and no solver vectors; what it does hold, next to `p` and the optimiser objects (tests/golden/make_ckpt_opt.py),
are the per-epoch metrics its training loops recorded with ITS solver stack at ITS tolerances:
  case2/checkpoint/mymodel.bson      l_loss_train, l_loss_val            (pushed at case2/case2.jl:159-160, saved at :178, iter = 3700)
  robertson/checkpoint/mymodel.bson  l_loss_train, l_loss_val, l_grad    (rober_crnn.jl:176-178, saved at :201, iter = 10850)
The last entries were computed at the saved `p` (the save happens inside the callback that pushed them).  The training data came from
Julia's RNG stream (case2.jl:11,60-61,79; rober_crnn.jl:44-47,73) and cannot be re-drawn here, so these numbers pin the solve path only
through the distribution of the same metric over re-drawn designs: tests/test_ckpt_history_pin.py.

Run in the build container (needs /root/reference):  python tests/golden/make_ckpt_history.py
The fixture holds numbers only (the last WINDOW entries of each list)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
REF = "/root/reference"
WINDOW = 50


def decode():
    from make_fixtures import load_bson
    out = {}
    for name, path, keys in (("case2", f"{REF}/case2/checkpoint/mymodel.bson", ("l_loss_train", "l_loss_val")),
                             ("robertson", f"{REF}/robertson/checkpoint/mymodel.bson", ("l_loss_train", "l_loss_val", "l_grad"))):
        d, arr = load_bson(path)
        e = {"iter": int(d["iter"]), "window": WINDOW}
        for k in keys:
            v = np.asarray(arr(d[k]), float).ravel()
            assert v.size == e["iter"], (name, k, v.size)
            e[k + "_tail"] = v[-WINDOW:].tolist()
            e[k + "_min"] = float(v.min())
        out[name] = e
    return out


if __name__ == "__main__":
    out = decode()
    with open(os.path.join(HERE, "fixtures_ckpt_history.json"), "w") as f:
        json.dump(out, f, indent=0)
    for n, e in out.items():
        for k, v in e.items():
            if k.endswith("_tail"):
                v = np.array(v)
                print(f"{n:10s} {k:18s} last {v[-1]:.6g}  window mean {v.mean():.6g}  [{v.min():.6g}, {v.max():.6g}]")

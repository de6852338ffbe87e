"""Write tests/golden/fixtures_case2_stream.json: case2's experiments re-drawn from the reference's seeded RNG stream, next to the numbers the
reference's own solver stack recorded on them.

What the reference holds (case2/checkpoint/mymodel.bson, decoded here from /root/reference -- build container only):
  l_loss_train / l_loss_val   one entry per epoch, pushed at case2/case2.jl:159-160 from the epoch-end loop :199-203; 3700 entries.
     * the LAST entries were computed at the saved `p` (the save is inside the callback that pushed them, :178)      -> pins A1, A2, A5, A6
     * the FIRST entries were computed after 20, 40, ... optimiser steps from the stream's initial `p`, each step one
       `ForwardDiff.gradient` (:195) and one `update!` (:197), in the stream's `randperm` order                        -> pins A7, A8, N1 as a chain
What is re-drawn (tests/golden/case2_stream.py + julia_rng.py, no Julia needed): u0_list, the 30 noisy experiments, yscale, the initial p, the
epoch shuffles.  The fixture keeps both, so that the `-m gpu` tests need neither /root/reference nor the RNG restatement; a CPU test regenerates
the re-drawn half from the stream bit for bit (tests/test_case2_stream_pin.py).

Run in the build container:  python tests/golden/make_case2_stream.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
REF = "/root/reference"
HEAD = 100


def recorded():
    from make_fixtures import load_bson
    d, arr = load_bson(f"{REF}/case2/checkpoint/mymodel.bson")
    lt = np.asarray(arr(d["l_loss_train"]), float).ravel()
    lv = np.asarray(arr(d["l_loss_val"]), float).ravel()
    assert lt.size == lv.size == int(d["iter"])
    return dict(iter=int(d["iter"]), l_loss_train_head=lt[:HEAD].tolist(), l_loss_val_head=lv[:HEAD].tolist(),
                l_loss_train_last=float(lt[-1]), l_loss_val_last=float(lv[-1]))


if __name__ == "__main__":
    import case2_stream as S
    import julia_rng as J
    from crnn_amd import cases
    from oracle import oracle as orc
    orc.build()
    J.self_check()
    d = S.draw(orc, cases, n_epochs=HEAD)
    out = dict(recorded=recorded(),
               design=dict(u0=d["u0"].tolist(), tsteps=d["ts"].tolist(), data=d["data"].tolist(), yscale=d["ys"].tolist(), p0=d["p0"].tolist(),
                           perms=d["perms"], stream_doubles_drawn=d["drawn"],
                           generator="MersenneTwister(1234) (Julia 1.6: dSFMT-19937; rand!(::Array{Float32}), randn!(::Array{Float64}), randn(Float32), randperm)"))
    with open(os.path.join(HERE, "fixtures_case2_stream.json"), "w") as f:
        json.dump(out, f)
    r = out["recorded"]
    print(f"recorded: iter {r['iter']}, last train {r['l_loss_train_last']:.7e} val {r['l_loss_val_last']:.7e}; first train {r['l_loss_train_head'][0]:.7e}")
    print(f"design: {d['drawn']} stream doubles; u0[0] = {d['u0'][0]}; yscale = {d['ys']}")

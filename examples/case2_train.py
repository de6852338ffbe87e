#!/usr/bin/env python3
"""examples/case2_train.py -- the reference's case2 training script (case2/case2.jl) on the MI355X library, at the level of
its behaviour: synthetic biodiesel data from the true mechanism + 5 % noise (:38-83), random initial p (:85-89), per epoch
one optimiser update PER EXPERIMENT in random order (:190-198), epoch-end train / validation losses (:199-203), a BSON
checkpoint every --n-plot epochs that the reference's `@load` reads (:178) and --restart reads back (:184-187).

    python examples/case2_train.py --epochs 20                       # the reference's loop: n_exp_train updates per epoch
    python examples/case2_train.py --epochs 200 --mode batch         # one update per epoch on the mean gradient of all
                                                                     # training experiments (what bench.py times)
    python examples/case2_train.py --epochs 25 --reference-run       # the reference's OWN run: its seed-1234 experiments, initial p and
                                                                     # epoch shuffles (tests/golden/fixtures_case2_stream.json, re-drawn from
                                                                     # Julia's RNG stream), its algorithm (Tsit5 with ForwardDiff's dual norm),
                                                                     # printed next to the losses its checkpoint recorded
Needs an MI355X (no CPU fallback)."""
import argparse
import os
import sys

import numpy as np

LB_CASE1, LB_CASE2 = float(np.float32(1e-5)), float(np.float32(1e-6))   # `lb = 1.f-5` / `lb = 1.f-6`: Float32 literals (case1/case1.jl:34, case2/case2.jl:34)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-exp-train", type=int, default=20)
    ap.add_argument("--n-exp-val", type=int, default=10)
    ap.add_argument("--epochs", type=int, default=20)
    ap.add_argument("--n-plot", type=int, default=10, help="checkpoint period in epochs")
    ap.add_argument("--mode", choices=["reference", "batch"], default="reference")
    ap.add_argument("--checkpoint", default="mymodel.bson")
    ap.add_argument("--restart", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--reference-run", action="store_true", help="replay case2/case2.jl's recorded run (first 100 epochs are on file)")
    args = ap.parse_args()

    from crnn_amd import NeuralODE, ODEProblem, Optimiser, PRESET_CASE2, cases
    from crnn_amd.io import load_checkpoint, save_checkpoint

    rng = np.random.Generator(np.random.PCG64(args.seed))
    n_train, n_exp = args.n_exp_train, args.n_exp_train + args.n_exp_val
    ref = None
    if args.reference_run:
        import json
        from crnn_amd import SOLVER_AUTOTSIT5
        with open(os.path.join(ROOT, "tests", "golden", "fixtures_case2_stream.json")) as f:
            ref = json.load(f)
        d = ref["design"]
        n_train, n_exp = 20, 30
        ts, u0, data, yscale, p = (np.array(d[k]) for k in ("tsteps", "u0", "data", "yscale", "p0"))
        args.epochs = min(args.epochs, len(d["perms"]))
        # case2.jl:26 `alg = AutoTsit5(Rosenbrock23(autodiff=false))` (stays on Tsit5 for this model: the library runs its Tsit5 kernels); :195
        # ForwardDiff.gradient through it puts the partials into the error norm (/ totallength(u)): errnorm_sens = 2 -- what reproduces the recorded history
        node = NeuralODE(ODEProblem(PRESET_CASE2, ts, solver=SOLVER_AUTOTSIT5, errnorm_sens=2))
    else:
        ts = cases.case2_tsteps()
        u0 = cases.case2_u0(n_exp, rng)
        gen = NeuralODE(ODEProblem(PRESET_CASE2, ts, atol=1e-10, rtol=1e-8))     # "true" data: tight tolerance
        clean = gen.predict_theta(u0, cases.case2_true_theta())[:, :6, :]
        gen.close()
        data = cases.add_noise(clean, 0.05, rng)
        yscale = cases.max_min(data, lb=LB_CASE2)
        node = NeuralODE(ODEProblem(PRESET_CASE2, ts))
        p = cases.case2_init_p(rng)
    node.set_ensemble(u0, data, yscale)
    l_train, l_val, it0, opt_state = [], [], 1, None
    if args.restart and os.path.exists(args.checkpoint):
        ck = load_checkpoint(args.checkpoint)
        p, l_train, l_val, it0 = ck["p"], list(ck["l_loss_train"]), list(ck["l_loss_val"]), int(ck["iter"]) + 1
        opt_state = ck.get("opt_state")                    # `@load ... opt`: ADAM moments, beta powers, ExpDecay eta / counter
        print(f"restarting from {args.checkpoint} at epoch {it0}")
    node.train_init(Optimiser(25, PRESET_CASE2), p)        # p and the optimiser state live on the GPU from here on
    if opt_state is not None:
        node.set_opt_state(opt_state)

    for epoch in range(it0, args.epochs + 1):
        if args.mode == "reference":
            if ref is not None:
                order = np.array(ref["design"]["perms"][epoch - 1]) - 1                             # the stream's randperm(20) of this epoch
            else:
                order = np.random.Generator(np.random.PCG64([args.seed, epoch])).permutation(n_train)   # randperm(n_exp_train), per epoch
            for i_exp in order:                             # update!(opt, p, gradient of experiment i_exp)
                node.train_step(first=int(i_exp), count=1, want_loss=False)
        else:
            node.train_step(first=0, count=n_train, want_loss=False)
        p = node.params()
        losses = node.losses(p)                             # epoch-end evaluation of every experiment
        lt, lv = float(losses[:n_train].mean()), float(losses[n_train:].mean())
        l_train.append(lt); l_val.append(lv)
        if ref is not None:
            rt, rv = ref["recorded"]["l_loss_train_head"][epoch - 1], ref["recorded"]["l_loss_val_head"][epoch - 1]
            print(f"epoch {epoch:4d}  loss train {lt:.6e}  val {lv:.6e}   reference recorded {rt:.6e} ({lt / rt - 1:+.1e})  {rv:.6e} ({lv / rv - 1:+.1e})", flush=True)
        else:
            print(f"epoch {epoch:4d}  loss train {lt:.3e}  val {lv:.3e}", flush=True)
        if epoch % args.n_plot == 0 or epoch == args.epochs:
            save_checkpoint(args.checkpoint, p=p, opt_state=node.opt_state(), l_loss_train=np.array(l_train),
                            l_loss_val=np.array(l_val), iter=epoch)
    node.close()
    return l_train


if __name__ == "__main__":
    main()

"""The solve path, the gradient and the optimiser pinned to numbers the reference's own solver stack computed.

case2/case2.jl seeds Julia's RNG (`Random.seed!(1234)`, :11), so its 30 experiments, its initial `p` and its epoch shuffles are a deterministic
stream; tests/golden/julia_rng.py restates that stream (Julia 1.6's MersenneTwister: dSFMT-19937, the Float32 array fill, the ziggurat, randperm;
pinned to the values Julia's documentation prints), tests/golden/case2_stream.py re-draws the experiments from it, and
tests/golden/fixtures_case2_stream.json holds them next to what `case2/checkpoint/mymodel.bson` recorded with OrdinaryDiffEq + ForwardDiff + Flux:

  * `l_loss_train[end]`, `l_loss_val[end]` = the mean `loss_neuralode(p, i_exp)` at the saved `p` (case2.jl:199-203 -> :159-160, saved :178)
        oracle / device reproduce them to 5e-6 (bar 3e-5).  With the stream consumed in the other documented order (element-by-element `randn`,
        Julia <= 1.4) the same metric is 6e-3 ... 9e-3 off: the pin is self-validating.                              rows A1, A2, A3, A5, A6, A9
  * `l_loss_train[1:25]`, `l_loss_val[1:25]` = the same metric after 20, 40, ... `ForwardDiff.gradient` + `update!` steps from the stream's
    initial `p`, in the stream's `randperm` order (case2.jl:194-198)
        the replay (chunks of 9 partials, every chunk its own adaptive solve, ExpDecay -> ADAM -> WeightDecay) lands within 5e-4 on the first
        six epochs and within 2e-2 on all 25 (median 7e-4) -- WITH the dual-inclusive error norm divided by `totallength(u)` (errnorm_sens = 2).
        Divided by `length(u)` (errnorm_sens = 1) the first epoch is already 1.2e-2 off and stays 3e-3 ... 1e-2 off; with the primal-only norm
        (the explicit solver's tangents then run at their stability limit) epoch 2 is 1e-1 off.                      rows A7, A8, N1
    (case2's `AutoTsit5(Rosenbrock23)` never leaves Tsit5: the temperature component does not move, OrdinaryDiffEq's stiffness estimate is
    0/0 = NaN, and `NaN > 0.9` is false -- oracle/crnn_oracle.c solve_one_auto; so the gradient solves are Tsit5 solves.)

CPU tests: the oracle.  `-m gpu`: the same through the product (C ABI, gfx950 kernels, device-resident optimiser).
"""
import json
import os
import sys

import numpy as np
import pytest

from crnn_amd import cases

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
INV_R = cases.INV_R
N_TRAIN = 20
EPOCHS = 25
EXPDECAY = (5e-3, 0.5, 500 * N_TRAIN, 1e-4)       # case2.jl:31-32: Flux.Optimiser(ExpDecay(5e-3, 0.5, 500 * n_exp_train, 1e-4), ADAMW(0.005, (0.9, 0.999), 1.f-6))
WD = float(np.float32(1e-6))


@pytest.fixture(scope="module")
def sfx():
    with open(os.path.join(HERE, "golden", "fixtures_case2_stream.json")) as f:
        fx = json.load(f)
    d = fx["design"]
    des = dict(u0=np.array(d["u0"]), ts=np.array(d["tsteps"]), data=np.array(d["data"]), ys=np.array(d["yscale"]), p0=np.array(d["p0"]),
               perms=d["perms"], drawn=d["stream_doubles_drawn"])
    return dict(rec=fx["recorded"], des=des)


def _split(l):
    """loss_epoch is a Float32 array (case2.jl:191); the epoch means are Float32 means."""
    l32 = np.asarray(l, np.float64).astype(np.float32)
    return float(l32[:N_TRAIN].mean(dtype=np.float32)), float(l32[N_TRAIN:].mean(dtype=np.float32))


def _rel(a, b):
    return abs(a / b - 1.0)


# ------------------------------------------------------------------ the stream
def test_julia_rng_reproduces_the_documented_values():
    import julia_rng as J
    assert J.self_check()


def test_fixture_design_regenerates_from_the_stream_bit_for_bit(orc, sfx):
    import case2_stream as S
    d = S.draw(orc, cases, n_epochs=len(sfx["des"]["perms"]))
    for k in ("u0", "ts", "data", "ys", "p0"):
        assert np.array_equal(d[k], sfx["des"][k]), k
    assert d["perms"] == sfx["des"]["perms"] and d["drawn"] == sfx["des"]["drawn"]
    # what the design must look like whatever the stream (case2.jl:60-63)
    u0 = d["u0"]
    assert ((u0[:, :2] >= 0.2) & (u0[:, :2] < 2.2)).all() and (u0[:, 2:6] == 0).all() and ((u0[:, 6] >= 323) & (u0[:, 6] < 343)).all()
    assert all(sorted(p) == list(range(1, N_TRAIN + 1)) for p in d["perms"])


# ------------------------------------------------------------------ oracle
def _oracle_problem(orc, des, **kw):
    return orc.make_problem(ns=6, nr=3, has_temp=1, lb=cases.LB_CASE2, ub=10.0, inv_R=INV_R, atol=1e-6, rtol=1e-3, yscale=des["ys"], clamp_pred=1, **kw)


def _oracle_losses(orc, des, p, data=None, solver=2):
    pb = _oracle_problem(orc, des, solver=solver)
    th, _ = orc.p2vec(2, 6, 3, p)
    data = des["data"] if data is None else data
    r = orc.solve_batch(pb, th, np.ascontiguousarray(des["u0"].T), des["ts"], np.ascontiguousarray(data.transpose(2, 1, 0)))
    assert (r["retcode"] == 0).all()
    return r["loss"]


def test_oracle_loss_at_the_checkpoint_equals_the_recorded_numbers(orc, fx, sfx):
    p = np.array(fx["case2_ckpt"]["p"])
    rec, des = sfx["rec"], sfx["des"]
    tr, va = _split(_oracle_losses(orc, des, p))
    assert _rel(tr, rec["l_loss_train_last"]) < 3e-5, (tr, rec["l_loss_train_last"])
    assert _rel(va, rec["l_loss_val_last"]) < 3e-5, (va, rec["l_loss_val_last"])
    # the composite IS Tsit5 here; the pin is sharp enough to tell the algorithm: Rosenbrock23 at the same tolerances is 1.1e-3 off
    tr1, va1 = _split(_oracle_losses(orc, des, p, solver=1))
    assert (tr1, va1) == (tr, va)
    tr0, _ = _split(_oracle_losses(orc, des, p, solver=0))
    assert 3e-4 < _rel(tr0, rec["l_loss_train_last"]) < 3e-3, tr0


def test_pin_is_self_validating_other_stream_orders_miss(orc, fx, sfx):
    """Negative controls: the same metric on experiments drawn with the stream consumed in Julia <= 1.4's order, and at 1.001 p."""
    import case2_stream as S
    p = np.array(fx["case2_ckpt"]["p"])
    rec = sfx["rec"]
    d = S.draw(orc, cases, n_epochs=0, array_randn=False)
    tr, va = _split(_oracle_losses(orc, d, p))
    assert _rel(tr, rec["l_loss_train_last"]) > 3e-3 and _rel(va, rec["l_loss_val_last"]) > 3e-3, (tr, va)
    tr, va = _split(_oracle_losses(orc, sfx["des"], p * 1.001))
    assert _rel(tr, rec["l_loss_train_last"]) > 1e-3, tr


def _oracle_gradient(orc, pb, des, p, i):
    from crnn_amd.api import fd_chunk_size
    th, dth = orc.p2vec(2, 6, 3, p)
    P = dth.shape[1]
    chunk = fd_chunk_size(P)
    g = np.zeros(P)
    for k0 in range(0, P, chunk):
        k1 = min(P, k0 + chunk)
        cols = np.zeros((dth.shape[0], chunk), order="F")
        cols[:, :k1 - k0] = dth[:, k0:k1]
        o = orc.solve_one(pb, th, des["u0"][i], des["ts"], des["data"][i], dtheta=cols, want_pred=False)
        assert o["retcode"] == 0
        g[k0:k1] = o["grad"][:k1 - k0]
    return g


def _oracle_replay(orc, des, epochs, errnorm_sens):
    pb = _oracle_problem(orc, des, solver=1, errnorm_sens=errnorm_sens)
    opt = orc.Optimiser(25, eta=0.005, wd=WD, expdecay=EXPDECAY)
    p = des["p0"].copy()
    hist, gmax = [], 0.0
    for ep in range(epochs):
        for i in des["perms"][ep]:
            g = _oracle_gradient(orc, pb, des, p, i - 1)
            gmax = max(gmax, float(np.linalg.norm(g)))
            p = opt.update(p, g)
        hist.append(_split(_oracle_losses(orc, des, p)))
    return np.array(hist), gmax, epochs, None


def _check_replay(hist, rec, label):
    ref = np.stack([rec["l_loss_train_head"][:len(hist)], rec["l_loss_val_head"][:len(hist)]], axis=1)
    dev = np.abs(hist / ref - 1.0)
    assert dev[:6].max() < 2.5e-3, (label, dev[:6])          # measured: <= 4.6e-4 train, <= 1.4e-3 val
    assert np.median(dev) < 2.5e-3, (label, np.median(dev))  # measured: 7e-4
    assert dev.max() < 4e-2, (label, dev.max())              # measured: 2.0e-2 (epoch 13, a loss spike both histories show)
    # the spikes of the recorded history are reproduced, not smoothed over: every epoch whose recorded train loss stands > 5 % above both neighbours
    spikes = [e for e in range(1, len(hist) - 1) if ref[e, 0] > 1.05 * max(ref[e - 1, 0], ref[e + 1, 0])]
    assert len(spikes) >= 3, spikes
    for e in spikes:
        assert hist[e, 0] > 1.03 * max(hist[e - 1, 0], hist[e + 1, 0]), (label, e + 1)
    return dev


def test_oracle_replays_the_recorded_first_epochs(orc, sfx):
    hist, gmax, _, _ = _oracle_replay(orc, sfx["des"], EPOCHS, errnorm_sens=2)
    _check_replay(hist, sfx["rec"], "oracle")
    assert gmax < 50.0


def test_replay_discriminates_the_error_norm(orc, sfx):
    """What the recorded history says about `ForwardDiff.gradient` through the adaptive solver: the partials are in the error norm, and the
    squared sum is divided by totallength(u) -- not by length(u), and not the primal values alone."""
    rec, des = sfx["rec"], sfx["des"]
    h1, _, _, _ = _oracle_replay(orc, des, 2, errnorm_sens=1)
    assert _rel(h1[0, 0], rec["l_loss_train_head"][0]) > 5e-3 and _rel(h1[1, 0], rec["l_loss_train_head"][1]) > 5e-3, h1
    h2, _, _, _ = _oracle_replay(orc, des, 2, errnorm_sens=2)
    assert _rel(h2[0, 0], rec["l_loss_train_head"][0]) < 1e-3 and _rel(h2[1, 0], rec["l_loss_train_head"][1]) < 1e-3, h2
    h0, _, _, _ = _oracle_replay(orc, des, 2, errnorm_sens=0)
    assert _rel(h0[1, 0], rec["l_loss_train_head"][1]) > 3e-2, h0


def test_recorded_history_confirms_the_restated_step_size_controller(orc, fx, sfx):
    """OrdinaryDiffEq's PI controller is restated in the oracle from its published form ([UNVERIFIED-DEP]: the package is not in the reference tree).
    The reference's own numbers confirm the constants: with the restated ones the final-loss pin stands at 5e-6 and the first six replayed epochs at
    <= 1.4e-3 (median 4e-4); changing any ONE of them makes both worse (profiles/r06c_case2_stream_controller_ablation.txt, 11 variants; four here)."""
    rec, des = sfx["rec"], sfx["des"]
    ck = np.array(fx["case2_ckpt"]["p"])
    u0T, dataT = np.ascontiguousarray(des["u0"].T), np.ascontiguousarray(des["data"].transpose(2, 1, 0))

    def run(**ctl):
        def mk(mode):
            pb = _oracle_problem(orc, des, solver=1, errnorm_sens=mode)
            for k, v in ctl.items():
                setattr(pb, k, v)
            return pb
        pbg, pbl = mk(2), mk(0)
        def loss(p):
            return _split(orc.solve_batch(pbl, orc.p2vec(2, 6, 3, p)[0], u0T, des["ts"], dataT)["loss"])
        tr, va = loss(ck)
        fin = max(_rel(tr, rec["l_loss_train_last"]), _rel(va, rec["l_loss_val_last"]))
        opt = orc.Optimiser(25, eta=0.005, wd=WD, expdecay=EXPDECAY)
        p = des["p0"].copy()
        dev = []
        for ep in range(6):
            for i in des["perms"][ep]:
                p = opt.update(p, _oracle_gradient(orc, pbg, des, p, i - 1))
            tr, va = loss(p)
            dev += [_rel(tr, rec["l_loss_train_head"][ep]), _rel(va, rec["l_loss_val_head"][ep])]
        return fin, max(dev), float(np.median(dev))

    f0, mx0, md0 = run()
    assert f0 < 3e-5 and mx0 < 2.5e-3 and md0 < 8e-4, (f0, mx0, md0)
    for ctl in (dict(beta1=7 / 20, beta2=2 / 10),      # the exponents of a second / third-order pair instead of Tsit5's
                dict(beta1=1 / 5, beta2=0.0),          # a plain I controller
                dict(gamma=0.8),
                dict(qoldinit=1.0)):
        f, mx, md = run(**ctl)
        assert mx > 3 * mx0 and md > 3 * md0, (ctl, f, mx, md)
    # (the final-loss pin alone already tells the exponents: 3e-4 / 1e-3 against 5e-6)
    assert run(beta1=7 / 20, beta2=2 / 10)[0] > 1e-4


def test_oracle_composite_refuses_the_dual_norm_instead_of_ignoring_it(orc, sfx):
    """oracle solver 2 (the CRNN composite) propagates tangents on accepted steps only; until round 6 it returned the primal-norm gradient when
    asked for errnorm_sens (which is how the replay first went wrong).  It now says so: return code -8, like the product's crnn_ctx_create."""
    des = sfx["des"]
    pb = _oracle_problem(orc, des, solver=2, errnorm_sens=2)
    th, dth = orc.p2vec(2, 6, 3, des["p0"])
    o = orc.solve_one(pb, th, des["u0"][0], des["ts"], des["data"][0], dtheta=dth[:, :9], want_pred=False)
    assert o["retcode"] == -8
    o = orc.solve_one(_oracle_problem(orc, des, solver=2), th, des["u0"][0], des["ts"], des["data"][0], dtheta=dth[:, :9], want_pred=False)
    assert o["retcode"] == 0 and o["n_rosenbrock"] == 0          # ... and the composite never leaves Tsit5 on case2 (constant temperature component)


# ------------------------------------------------------------------ product (-m gpu)
def _node(des, **kw):
    from crnn_amd import NeuralODE, ODEProblem, PRESET_CASE2
    node = NeuralODE(ODEProblem(PRESET_CASE2, des["ts"], **kw))
    node.set_ensemble(des["u0"], des["data"], des["ys"])
    return node


@pytest.mark.gpu
def test_gpu_loss_at_the_checkpoint_equals_the_recorded_numbers(fx, sfx):
    from crnn_amd import SOLVER_AUTOTSIT5, SOLVER_ROSENBROCK23, SOLVER_TSIT5
    p = np.array(fx["case2_ckpt"]["p"])
    rec, des = sfx["rec"], sfx["des"]
    for solver, bar in ((SOLVER_AUTOTSIT5, 3e-5), (SOLVER_TSIT5, 3e-5), (SOLVER_ROSENBROCK23, 3e-3)):
        tr, va = _split(_node(des, solver=solver).losses(p))
        assert _rel(tr, rec["l_loss_train_last"]) < bar, (solver, tr)
        assert _rel(va, rec["l_loss_val_last"]) < bar, (solver, va)


@pytest.mark.gpu
def test_gpu_replays_the_recorded_first_epochs(orc, sfx):
    """The reference's training loop on the device: crnn_train_step per experiment (dual-norm chunks of ForwardDiff.gradient, reduction, Flux
    optimiser, all on the device), the epoch-end loss loop, against the recorded history and against the oracle's replay."""
    from crnn_amd import PRESET_CASE2, SOLVER_AUTOTSIT5, Optimiser
    rec, des = sfx["rec"], sfx["des"]
    node = _node(des, solver=SOLVER_AUTOTSIT5, errnorm_sens=2)      # the script's own `alg` (case2.jl:26) and ForwardDiff's norm: Tsit5 dual-norm chunks
    node.train_init(Optimiser(25, preset=PRESET_CASE2), des["p0"])      # crnn_opt_preset: the chain above
    hist = []
    for ep in range(EPOCHS):
        for i in des["perms"][ep]:
            node.train_step(first=i - 1, count=1, want_loss=False)
        hist.append(_split(node.losses(node.params())))
    hist = np.array(hist)
    _check_replay(hist, rec, "device")
    # device and oracle walk the same chain: the first 40 gradient + update steps leave the Float32 epoch means equal; later a last-place
    # difference that flips one accept / reject decision moves a gradient by ~1e-3 and the two chains sit apart like two runs of the reference
    oh, _, _, _ = _oracle_replay(orc, des, EPOCHS, errnorm_sens=2)
    assert np.abs(hist[:2] / oh[:2] - 1.0).max() < 1e-6, np.abs(hist[:2] / oh[:2] - 1.0)
    assert np.abs(hist / oh - 1.0).max() < 4e-2

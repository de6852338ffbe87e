"""The solve path pinned to the only solver-dependent numbers the reference holds: the training metrics inside its own checkpoints.

SURVEY 8(c): the reference has no tests and no solver vectors, and its solver stack (OrdinaryDiffEq / ForwardDiff) cannot run here.  What its
checkpoints do hold (decoded by tests/golden/make_ckpt_history.py into tests/golden/fixtures_ckpt_history.json) are the metrics its training
loops computed WITH that stack at the saved `p`:
  case2      mean `loss_neuralode(p, i_exp)` over the 20 training / 10 validation experiments   (case2/case2.jl:199-203 -> :159-160, saved :178)
  robertson  the same over 20 / 5 experiments, and the epoch mean of `norm(ForwardDiff.gradient(...), 2)` on truncated prefixes
             (robertson/rober_crnn.jl:214-231 -> :176-178, saved :201)
The experiments themselves came from Julia's RNG stream and cannot be re-drawn, so this is a DISTRIBUTIONAL pin: the same metric, formed here
at the reference's `p` on experiments re-drawn from the reference's own design (u0 ranges, Latin hypercube, saveat grid, relative noise, yscale
from the data), has to reproduce the recorded values within the spread of the re-drawn designs / of the reference's own last 50 epochs.  The
pin is coarse in absolute terms (10-25 %) and sharp in what it rejects: the loss at `1.01 p` is 3x (robertson) / 1.3x (case2) the recorded one,
at `0.97 p` 15x / 1.45x -- a wrong `p2vec`, right-hand side, stiff solve, loss normalisation or gradient would not pass (negative controls below).
It covers rows A1 / A1-rob, A2, A5, A6 and (through `l_grad`) A7; solver-internal step-for-step parity with OrdinaryDiffEq stays unpinned.

CPU tests: the oracle.  `-m gpu`: the same designs through the product (device losses / gradients at the reference's tolerances).
"""
import json
import os

import numpy as np
import pytest

from crnn_amd import cases

HERE = os.path.dirname(os.path.abspath(__file__))
INV_R = float(np.float32(-1.0) / np.float32(1.98720425864083e-3))
N_DESIGNS = 8


@pytest.fixture(scope="module")
def hist():
    with open(os.path.join(HERE, "golden", "fixtures_ckpt_history.json")) as f:
        return {k: {a: (np.array(b) if isinstance(b, list) else b) for a, b in e.items()} for k, e in json.load(f).items()}


# ------------------------------------------------------------------ the reference's experiment designs, re-drawn
def rober_design(orc, seed):
    """rober_crnn.jl:44-47 (u0: y2 = lb, (y1, y3) = randomLHC(n_exp, 2) ./ n_exp .+ 0.5), :49 (tsteps), :69-79 (data: the true mechanism solved
    with `alg` at the script's atol / rtol, relative noise 1e-4, yscale = max over experiments of max - min), :80 (dydt_scale)."""
    rng = np.random.Generator(np.random.PCG64([seed, 77]))
    n_exp = 25
    ts = cases.rober_tsteps()
    u0 = np.zeros((n_exp, 3))
    u0[:, 1] = 1e-8
    u0[:, [0, 2]] = np.stack([rng.permutation(n_exp) + 1 for _ in range(2)], axis=1) / n_exp + 0.5
    pbt = orc.make_problem(ns=3, nr=3, lb=1e-30, atol=[1e-6, 1e-8, 1e-6], rtol=1e-3, maxiters=100000)
    r = orc.solve_batch(pbt, cases.rober_true_theta(), np.ascontiguousarray(u0.T), ts, np.zeros((len(ts), 3, n_exp)), want_pred=True)
    data = cases.add_noise(r["pred"].transpose(2, 1, 0), 1e-4, rng)
    ys = cases.max_min(data)
    samples = rng.integers(32, 41, size=20)          # rober_crnn.jl:216  sample = rand(batchsize:datasize)
    return dict(u0=u0, ts=ts, data=data, ys=ys, dyd=ys / ts[-1], samples=samples, n_train=20)


def case2_design(orc, seed):
    """case2.jl:60-63 (u0), :64-65 (tsteps), :74-83 (data: true mechanism, relative noise 0.05, yscale with + lb)."""
    rng = np.random.Generator(np.random.PCG64([seed, 78]))
    n_exp = 30
    ts = cases.case2_tsteps()
    u0 = cases.case2_u0(n_exp, rng)
    pbt = orc.make_problem(ns=6, nr=3, has_temp=1, lb=cases.LB_CASE2, ub=10.0, inv_R=INV_R, atol=1e-6, rtol=1e-3)
    r = orc.solve_batch(pbt, cases.case2_true_theta(), np.ascontiguousarray(u0.T), ts, np.zeros((len(ts), 6, n_exp)), want_pred=True)
    data = cases.add_noise(r["pred"][:, :6, :].transpose(2, 1, 0), 0.05, rng)
    return dict(u0=u0, ts=ts, data=data, ys=cases.max_min(data, lb=cases.LB_CASE2), n_train=20)


def rober_problem(orc, d, **kw):
    return orc.make_problem(ns=3, nr=6, lb=1e-8, atol=[1e-6, 1e-8, 1e-6], rtol=1e-3, yscale=d["ys"], rate_scale=d["dyd"], maxiters=10000, **kw)


def case2_problem(orc, d, **kw):
    # case2.jl:26: alg = AutoTsit5(Rosenbrock23(autodiff=false)) -> the oracle's composite
    return orc.make_problem(ns=6, nr=3, has_temp=1, lb=cases.LB_CASE2, ub=10.0, inv_R=INV_R, atol=1e-6, rtol=1e-3, yscale=d["ys"],
                            clamp_pred=1, solver=2, **kw)


def oracle_losses(orc, pb, kind, ns, nr, p, d):
    th, _ = orc.p2vec(kind, ns, nr, p)
    r = orc.solve_batch(pb, th, np.ascontiguousarray(d["u0"].T), d["ts"], np.ascontiguousarray(d["data"].transpose(2, 1, 0)))
    assert (r["retcode"] == 0).all()
    return r["loss"]


def oracle_rober_gnorm(orc, p, d):
    """mean over the training experiments of ||grad of loss on the first `sample` save points||_2; the gradient the way ForwardDiff forms it:
    P = 43 -> chunks of 11 + 11 + 11 + 10 partials, every chunk its own adaptive solve in the dual-inclusive norm (errnorm_sens = 1)."""
    from crnn_amd.api import fd_chunk_size
    th, dth = orc.p2vec(3, 3, 6, p)
    pb = rober_problem(orc, d, errnorm_sens=1)
    P = dth.shape[1]
    chunk = fd_chunk_size(P)
    g = []
    for i in range(d["n_train"]):
        s = int(d["samples"][i])
        gi = np.zeros(P)
        for k0 in range(0, P, chunk):
            k1 = min(P, k0 + chunk)
            cols = np.zeros((dth.shape[0], chunk), order="F")          # the Dual carries `chunk` partials, the surplus ones zero
            cols[:, :k1 - k0] = dth[:, k0:k1]
            o = orc.solve_one(pb, th, d["u0"][i], d["ts"][:s], d["data"][i][:, :s], dtheta=cols, want_pred=False)
            assert o["retcode"] == 0
            gi[k0:k1] = o["grad"][:k1 - k0]
        g.append(np.linalg.norm(gi))
    return float(np.mean(g)), g


def _split(l, d):
    return float(l[:d["n_train"]].mean()), float(l[d["n_train"]:].mean())


# ------------------------------------------------------------------ CPU: the oracle against the recorded metrics
def test_robertson_recorded_losses_lie_in_the_oracles_design_distribution(orc, fx, hist):
    p = np.array(fx["rober_ckpt"]["p"])
    h = hist["robertson"]
    tr, va = np.array([_split(oracle_losses(orc, rober_problem(orc, d), 3, 3, 6, p, d), d)
                       for d in (rober_design(orc, s) for s in range(N_DESIGNS))]).T
    rt, rv = h["l_loss_train_tail"], h["l_loss_val_tail"]
    # the reference's own epoch-to-epoch band at this stage of training (p jitters at eta = 1e-4) is [1.8e-3, 1.1e-2]; the oracle's
    # design-to-design band has to sit inside it and agree with its centre
    assert rt.min() <= tr.min() and tr.max() <= rt.max(), (tr, rt.min(), rt.max())
    assert rv.min() <= va.min() and va.max() <= rv.max(), (va, rv.min(), rv.max())
    assert abs(np.median(tr) / np.median(rt) - 1.0) < 0.35, (np.median(tr), np.median(rt))
    assert abs(np.median(va) / np.median(rv) - 1.0) < 0.35, (np.median(va), np.median(rv))
    # the recorded last entries (computed AT the saved p) are unremarkable draws of the oracle's distribution
    assert abs(rt[-1] - tr.mean()) < 3.0 * tr.std(ddof=1) and abs(rv[-1] - va.mean()) < 3.0 * va.std(ddof=1)


def test_robertson_pin_rejects_a_one_percent_change_of_p(orc, fx, hist):
    """What the pin is worth: the same metric at 1.01 p / 0.97 p leaves the reference's band by a wide margin."""
    p = np.array(fx["rober_ckpt"]["p"])
    rt = hist["robertson"]["l_loss_train_tail"]
    for scale, factor in ((1.01, 1.1), (0.97, 4.0)):      # rt.max() is one outlying epoch (1.07e-2; the window's median is 4.1e-3)
        tr = np.array([_split(oracle_losses(orc, rober_problem(orc, d), 3, 3, 6, p * scale, d), d)[0]
                       for d in (rober_design(orc, s) for s in range(4))])
        assert tr.min() > factor * rt.max() and tr.min() > 2.8 * np.median(rt), (scale, tr, rt.max())


def test_robertson_recorded_gradient_norm(orc, fx, hist):
    """A7 through the reference's `l_grad`: epoch mean of ||ForwardDiff.gradient||_2 on truncated prefixes (rober_crnn.jl:216-218,229)."""
    p = np.array(fx["rober_ckpt"]["p"])
    rg = hist["robertson"]["l_grad_tail"]
    g = np.array([oracle_rober_gnorm(orc, p, rober_design(orc, s))[0] for s in range(N_DESIGNS)])
    assert abs(g.mean() / rg.mean() - 1.0) < 0.25, (g, rg.mean())
    assert g.min() > 0.8 * rg.min() and g.max() < 1.25 * rg.max(), (g, rg.min(), rg.max())


def test_case2_recorded_losses_lie_in_the_oracles_design_distribution(orc, fx, hist):
    p = np.array(fx["case2_ckpt"]["p"])
    h = hist["case2"]
    tr, va = np.array([_split(oracle_losses(orc, case2_problem(orc, d), 2, 6, 3, p, d), d)
                       for d in (case2_design(orc, s) for s in range(N_DESIGNS))]).T
    rt, rv = h["l_loss_train_tail"], h["l_loss_val_tail"]
    assert abs(tr.mean() / rt.mean() - 1.0) < 0.12, (tr.mean(), rt.mean())
    assert abs(va.mean() / rv.mean() - 1.0) < 0.30, (va.mean(), rv.mean())       # ten experiments: the design spread is 12 %
    assert abs(rt[-1] - tr.mean()) < 3.0 * tr.std(ddof=1) and abs(rv[-1] - va.mean()) < 3.0 * va.std(ddof=1)
    # negative control: +-3 % on p leaves the band
    for scale in (1.03, 0.97):
        t2 = np.array([_split(oracle_losses(orc, case2_problem(orc, d), 2, 6, 3, p * scale, d), d)[0]
                       for d in (case2_design(orc, s) for s in range(4))])
        assert t2.mean() > 1.35 * rt.mean() and t2.min() > np.quantile(rt, 0.9), (scale, t2, rt.mean())


@pytest.mark.needs_reference
def test_history_fixture_matches_the_reference_checkpoints(hist):
    if not os.path.isdir("/root/reference"):
        pytest.skip("the reference tree exists in the build container only")
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_ckpt_history", os.path.join(HERE, "golden", "make_ckpt_history.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    now = m.decode()
    for name, e in now.items():
        for k, v in e.items():
            assert np.array_equal(np.asarray(v), np.asarray(hist[name][k])), (name, k)


# ------------------------------------------------------------------ device: the product on the same designs
@pytest.mark.gpu
def test_gpu_robertson_metrics_at_the_reference_checkpoint(orc, fx, hist):
    from crnn_amd import NeuralODE, ODEProblem, PRESET_ROBER
    p = np.array(fx["rober_ckpt"]["p"])
    h = hist["robertson"]
    tr, gn = [], []
    for s in range(4):
        d = rober_design(orc, s)
        node = NeuralODE(ODEProblem(PRESET_ROBER, d["ts"], rate_scale=d["dyd"], errnorm_sens=1))
        node.set_ensemble(d["u0"], d["data"], d["ys"])
        l = node.losses(p)
        lo = oracle_losses(orc, rober_problem(orc, d), 3, 3, 6, p, d)
        # typically 1e-13; on re-drawn designs an attempt whose error estimate sits at the accept threshold may be decided the other way
        # (device transcendentals against libm), which moves that trajectory by about rtol^2
        assert np.abs(l - lo).max() <= 5e-6 * np.abs(lo).max(), (l, lo)
        tr.append(_split(l, d)[0])
        go, g_each = oracle_rober_gnorm(orc, p, d)
        g = [np.linalg.norm(node.gradient(p, i, sample=int(d["samples"][i]))) for i in range(d["n_train"])]
        assert np.abs(np.array(g) - np.array(g_each)).max() <= 1e-5 * max(g_each), (g, g_each)
        gn.append(float(np.mean(g)))
        node.close()
    rt, rg = h["l_loss_train_tail"], h["l_grad_tail"]
    assert rt.min() <= min(tr) and max(tr) <= rt.max(), (tr, rt.min(), rt.max())
    assert abs(np.mean(gn) / rg.mean() - 1.0) < 0.25, (gn, rg.mean())


@pytest.mark.gpu
def test_gpu_case2_metrics_at_the_reference_checkpoint(orc, fx, hist):
    from crnn_amd import NeuralODE, ODEProblem, PRESET_CASE2
    p = np.array(fx["case2_ckpt"]["p"])
    rt = hist["case2"]["l_loss_train_tail"]
    tr = []
    for s in range(4):
        d = case2_design(orc, s)
        node = NeuralODE(ODEProblem(PRESET_CASE2, d["ts"], solver=2))            # case2.jl:26: AutoTsit5(Rosenbrock23)
        node.set_ensemble(d["u0"], d["data"], d["ys"])
        l = node.losses(p)
        lo = oracle_losses(orc, case2_problem(orc, d), 2, 6, 3, p, d)
        assert np.abs(l - lo).max() <= 5e-6 * np.abs(lo).max(), (l, lo)
        tr.append(_split(l, d)[0])
        node.close()
    assert abs(np.mean(tr) / rt.mean() - 1.0) < 0.15, (tr, rt.mean())

"""A8 / N1 on the GPU: the device-resident optimiser (`opt_kernel`: Flux chain [norm clip] -> [ExpDecay] -> ADAM ->
WeightDecay -> p .-= delta, reference case2/case2.jl:31-32,197, robertson/rober_crnn.jl:19,221-224, case1/case1.jl:18)
against the ORACLE's optimiser (oracle/crnn_oracle.c: orc_opt_update) and against the committed NumPy trace
(tests/golden/fixtures.json "optim", made by tests/golden/make_fixtures.py) -- not against the product's own host build.

  * fixed gradient sequence through crnn_train_update  -> golden trace and oracle, 1e-15
  * >= 10 full training steps (solve + gradient + update on the device) on case2, robertson and case1
    -> oracle gradients pushed through the oracle optimiser, parameters compared after every step
  * the optimiser state (crnn_get_opt_state / crnn_set_opt_state) against the oracle's state vector
"""
import numpy as np

LB_CASE1, LB_CASE2 = float(np.float32(1e-5)), float(np.float32(1e-6))   # `lb = 1.f-5` / `lb = 1.f-6`: Float32 literals (case1/case1.jl:34, case2/case2.jl:34)
import pytest

from conftest import WD6, WD8, oracle_problem

pytestmark = pytest.mark.gpu


def _dummy_node(preset, ts):
    """A context of the right parameter count: crnn_train_update needs no ensemble."""
    from crnn_amd import NeuralODE, ODEProblem
    return NeuralODE(ODEProblem(preset, ts))


def test_device_update_matches_golden_trace_and_oracle(orc, fx):
    """update!(opt, p, grad) on the device with the committed gradient sequence: every intermediate p equals the NumPy
    golden trace and the oracle's optimiser to 1e-15 (P = 25 for all three chains; case2's parameter count)."""
    from crnn_amd import Optimiser, PRESET_CASE2, cases
    o = fx["optim"]
    g = np.array(o["grads"]); p0 = np.array(o["p0"])
    assert g.shape[0] >= 10 and g.shape[1] == 25
    chains = (("case2", dict(eta=0.005, wd=1e-6, expdecay=(5e-3, 0.5, 5, 1e-4))),      # ExpDecay -> ADAM -> WeightDecay
              ("rober", dict(eta=0.005, wd=1e-6, grad_clip_norm=10.0)),                # norm clip -> ADAMW
              ("case1", dict(eta=0.001, wd=1e-8)))                                      # ADAMW
    for key, kw in chains:
        node = _dummy_node(PRESET_CASE2, cases.case2_tsteps())
        node.train_init(Optimiser(25, **kw), p0)
        oopt = orc.Optimiser(25, **kw)
        po = p0.copy()
        for i in range(g.shape[0]):
            node.update_(g[i])
            po = oopt.update(po, g[i])
            p = node.params()
            assert np.max(np.abs(p - np.array(o[key][i]))) < 1e-15, (key, i)
            assert np.max(np.abs(p - po)) < 1e-15, (key, i)
        st = node.opt_state()
        assert np.max(np.abs(st - oopt.state)) <= 1e-15 * max(1.0, np.max(np.abs(oopt.state)))
        node.close()


def _oracle_mean_grad(orc, pb, kind, ns, nr, p, u0, ts, data, sample=None):
    th, dth = orc.p2vec(kind, ns, nr, p)
    t_ = ts if sample is None else ts[:sample]
    d_ = data if sample is None else data[:, :, :sample]
    r = orc.solve_batch(pb, th, np.ascontiguousarray(u0.T), t_, np.ascontiguousarray(d_.transpose(2, 1, 0)), dtheta=dth)
    B = u0.shape[0]
    return r["loss"].mean(), r["grad"] / B


@pytest.mark.parametrize("gm", [0, 2], ids=["auto", "adjoint"])
@pytest.mark.parametrize("case", ["case2", "rober"])
def test_training_loop_matches_oracle_chain(orc, case2_setup, rober_setup, case, gm):
    """12 device training steps (crnn_train_init / crnn_train_step / crnn_get_params) from the reference's initial p
    (case2) / checkpoint p (robertson, random horizons as rober_crnn.jl:218) against: oracle loss + gradient at the
    oracle's own current p -> oracle optimiser.  Both chains start from the same p and never exchange anything, so
    this is the whole A7 + A8 loop, device vs oracle.  ADAM's first updates are sign-like (m / sqrt(v) = +-1), which
    keeps the comparison tight: 1e-9 on the parameters after every step.  grad_mode AUTO takes the forward-tangent kernels at
    this ensemble size (reduction, then `opt_kernel`), ADJOINT the tape kernel whose tail is ONE launch without a communicator
    (`reduce_opt_sort_kernel`: reduction + chain rule + optimiser + p2vec) -- the same chain either way."""
    from crnn_amd import NeuralODE, ODEProblem, Optimiser, PRESET_CASE2, PRESET_ROBER
    if case == "case2":
        s, preset, kind, ns, nr, P = case2_setup, PRESET_CASE2, 2, 6, 3, 25
        p0 = s["p_init"]
        node = NeuralODE(ODEProblem(preset, s["tsteps"], grad_mode=gm))
        okw = dict(eta=0.005, wd=WD6, expdecay=(5e-3, 0.5, 500 * 20, 1e-4))
        samples = [None] * 12
    else:
        s, preset, kind, ns, nr, P = rober_setup, PRESET_ROBER, 3, 3, 6, 43
        p0 = s["p_ckpt"]
        node = NeuralODE(ODEProblem(preset, s["tsteps"], rate_scale=s["dydt_scale"], grad_mode=gm))
        okw = dict(eta=0.005, wd=WD6, grad_clip_norm=10.0)
        samples = [20, 40, 22, 40, 25, 40, 40, 21, 33, 40, 28, 40]
    node.set_ensemble(s["u0"], s["data"], s["yscale"])
    node.train_init(Optimiser(P, preset), p0)
    pb = oracle_problem(orc, case, s)
    oopt = orc.Optimiser(P, **okw)
    po = p0.copy()
    for it, sm in enumerate(samples):
        loss_d = node.train_step(sample=sm, want_loss=True)
        loss_o, g_o = _oracle_mean_grad(orc, pb, kind, ns, nr, po, s["u0"], s["tsteps"], s["data"], sm)
        po = oopt.update(po, g_o)
        assert abs(loss_d - loss_o) < 1e-8 * abs(loss_o), (it, loss_d, loss_o)
        assert np.max(np.abs(node.params() - po)) < 1e-9, it
    st = node.opt_state()
    assert np.max(np.abs(st - oopt.state)) < 1e-7 * max(1.0, np.max(np.abs(oopt.state)))
    assert st[2 * P + 3] == (12 if case == "case2" else 0)          # ExpDecay's call counter only runs in case2's chain
    node.close()


def test_training_loop_case1_tsit5_matches_oracle_chain(orc, fx):
    """case1 (case1.jl:18,28: ADAMW(0.001, (0.9, 0.999), 1e-8), Tsit5): 10 device steps vs oracle gradient + oracle optimiser."""
    from crnn_amd import NeuralODE, ODEProblem, Optimiser, PRESET_CASE1, cases
    rng = np.random.Generator(np.random.PCG64(12))
    ts = cases.case1_tsteps()
    u0 = np.array(fx["case1"]["u0"])
    p0 = np.array(fx["case1"]["p"])
    gen = NeuralODE(ODEProblem(PRESET_CASE1, ts, atol=1e-12, rtol=1e-10))
    data = cases.add_noise(gen.predict_theta(u0, cases.case1_true_theta()), 0.05, rng)
    gen.close()
    ys = cases.max_min(data, lb=LB_CASE1)
    node = NeuralODE(ODEProblem(PRESET_CASE1, ts))
    node.set_ensemble(u0, data, ys)
    node.train_init(Optimiser(24, PRESET_CASE1), p0)
    pb = orc.make_problem(ns=5, nr=4, lb=LB_CASE1, ub=10.0, atol=1e-5, rtol=1e-2, yscale=ys, clamp_pred=1, maxiters=10000, solver=1)
    oopt = orc.Optimiser(24, eta=0.001, wd=WD8)
    po = p0.copy()
    for it in range(10):
        loss_d = node.train_step(want_loss=True)
        loss_o, g_o = _oracle_mean_grad(orc, pb, 1, 5, 4, po, u0, ts, data)
        po = oopt.update(po, g_o)
        assert abs(loss_d - loss_o) < 1e-8 * abs(loss_o), it
        assert np.max(np.abs(node.params() - po)) < 1e-9, it
    node.close()


def test_opt_state_roundtrip_resumes_bit_identically(case2_setup):
    """crnn_get_opt_state / crnn_set_opt_state: 6 steps, save (p, state), 6 more == restore into a fresh context, 6 more.
    Without the state a restart zeroes ADAM's moments and the runs part ways (the @save ... p opt of case2.jl:213)."""
    from crnn_amd import NeuralODE, ODEProblem, Optimiser, PRESET_CASE2
    s = case2_setup

    def fresh():
        n = NeuralODE(ODEProblem(PRESET_CASE2, s["tsteps"]))
        n.set_ensemble(s["u0"], s["data"], s["yscale"])
        return n

    a = fresh()
    a.train_init(Optimiser(25, PRESET_CASE2), s["p_init"])
    for _ in range(6):
        a.train_step(want_loss=False)
    p_mid, st_mid = a.params(), a.opt_state()
    assert st_mid.shape == (54,) and st_mid[53] == 6 and abs(st_mid[50] - 0.9 ** 7) < 1e-15
    for _ in range(6):
        a.train_step(want_loss=False)
    b = fresh()
    b.train_init(Optimiser(25, PRESET_CASE2), p_mid)
    b.set_opt_state(st_mid)
    for _ in range(6):
        b.train_step(want_loss=False)
    assert np.array_equal(a.params(), b.params()) and np.array_equal(a.opt_state(), b.opt_state())
    c = fresh()
    c.train_init(Optimiser(25, PRESET_CASE2), p_mid)      # no state: ADAM restarts
    for _ in range(6):
        c.train_step(want_loss=False)
    assert np.max(np.abs(c.params() - a.params())) > 1e-6
    for n in (a, b, c):
        n.close()


def test_loaded_library_was_built_from_these_sources():
    """The binary the GPU box runs must be the one these sources compile to (a prebuilt .so travels with the snapshot):
    crnn_amd/_lib.py rebuilds a stale library at import; here the baked-in digest is compared once more."""
    from crnn_amd import _lib as L
    info = L.lib.crnn_build_info().decode()
    assert f"src={L.source_hash()} " in info and "arch=gfx950" in info

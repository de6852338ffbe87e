"""GPU parity tests: the gfx950 kernels (through the C ABI / Python host mirror)
against the CPU oracle on identical inputs, and against the committed golden
vectors.  Floating point: tolerances are written at each assert."""
import numpy as np

LB_CASE1, LB_CASE2 = float(np.float32(1e-5)), float(np.float32(1e-6))   # `lb = 1.f-5` / `lb = 1.f-6`: Float32 literals (case1/case1.jl:34, case2/case2.jl:34)
import pytest

from conftest import oracle_problem

pytestmark = pytest.mark.gpu


def _node(case, setup, atol=None, rtol=None, maxiters=None, cols=0, grad_mode=0, tape_steps=0):
    from crnn_amd import NeuralODE, ODEProblem, PRESET_CASE2, PRESET_ROBER
    kw = dict(atol=atol, rtol=rtol, maxiters=maxiters, cols_per_lane=cols, grad_mode=grad_mode, tape_steps=tape_steps)
    if case == "case2":
        prob = ODEProblem(PRESET_CASE2, setup["tsteps"], **kw)
    else:
        prob = ODEProblem(PRESET_ROBER, setup["tsteps"], rate_scale=setup["dydt_scale"], **kw)
    node = NeuralODE(prob)
    node.set_ensemble(setup["u0"], setup["data"], setup["yscale"])
    return node


def _oracle_batch(orc, case, setup, p, atol=None, rtol=None, maxiters=None, grad=True, sample=None):
    kind = 2 if case == "case2" else 3
    ns, nr = (6, 3) if case == "case2" else (3, 6)
    th, dth = orc.p2vec(kind, ns, nr, p)
    pb = oracle_problem(orc, case, setup, atol, rtol, maxiters)
    ts = setup["tsteps"] if sample is None else setup["tsteps"][:sample]
    data = setup["data"] if sample is None else setup["data"][:, :, :sample]
    return orc.solve_batch(pb, th, np.ascontiguousarray(setup["u0"].T), ts,
                           np.ascontiguousarray(data.transpose(2, 1, 0)), dtheta=dth if grad else None,
                           want_pred=True)


@pytest.mark.parametrize("case,pkey", [("case2", "p_ckpt"), ("case2", "p_init"), ("rober", "p_ckpt")])
def test_pred_loss_match_oracle_reference_tolerances(orc, case2_setup, rober_setup, case, pkey):
    """Same algorithm, same inputs, reference tolerances: the step sequences
    coincide, so trajectories agree to rounding (1e-9 relative to the species scale)."""
    setup = case2_setup if case == "case2" else rober_setup
    p = setup[pkey]
    node = _node(case, setup)
    ref = _oracle_batch(orc, case, setup, p, grad=False)
    pred = node.predict_neuralode(setup["u0"], p)               # [B, n, D]
    ref_pred = ref["pred"].transpose(2, 1, 0)                   # [D, n, B] -> [B, n, D]
    scale = np.abs(ref_pred).max(axis=(0, 2), keepdims=True) + 1e-300
    assert np.max(np.abs(pred - ref_pred) / scale) < 1e-9
    assert np.array_equal(node.last_retcode, ref["retcode"])
    losses = node.losses(p)
    assert np.max(np.abs(losses - ref["loss"]) / ref["loss"]) < 1e-9
    st = node.last_stats
    assert st["n_accept"] == ref["naccept"] and st["n_reject"] == ref["nreject"]
    assert st["n_ok"] == len(losses)


@pytest.mark.parametrize("case,pkey", [("case2", "p_ckpt"), ("case2", "p_init"), ("rober", "p_ckpt")])
@pytest.mark.parametrize("mode,cols", [(2, 0), (1, 0), (1, 1)])
def test_gradient_matches_oracle(orc, case2_setup, rober_setup, case, pkey, mode, cols):
    """d loss/d p per experiment and batched, reference tolerances; 1e-7 relative to max |grad|.
    mode 2: discrete adjoint of the accepted steps; mode 1: forward tangents (auto and one-column-per-lane variants).
    The oracle carries forward tangents (ForwardDiff's arithmetic)."""
    setup = case2_setup if case == "case2" else rober_setup
    p = setup[pkey]
    node = _node(case, setup, cols=cols, grad_mode=mode)
    ref = _oracle_batch(orc, case, setup, p)
    B = setup["u0"].shape[0]
    loss, grad = node.loss_and_grad(p)
    gref = ref["grad"] / B
    assert abs(loss - ref["loss"].mean()) < 1e-9 * ref["loss"].mean()
    assert np.max(np.abs(grad - gref)) < 1e-7 * np.max(np.abs(gref))
    # per-experiment gradient (ForwardDiff.gradient(x -> loss_neuralode(x, i_exp), p))
    kind, ns, nr = (2, 6, 3) if case == "case2" else (3, 3, 6)
    th, dth = orc.p2vec(kind, ns, nr, p)
    pb = oracle_problem(orc, case, setup)
    for i in (0, B - 1):
        g = node.gradient(p, i)
        r1 = orc.solve_one(pb, th, setup["u0"][i], setup["tsteps"], setup["data"][i], dtheta=dth)
        assert np.max(np.abs(g - r1["grad"])) < 1e-7 * np.max(np.abs(r1["grad"]))
        assert abs(node.loss_neuralode(p, i) - r1["loss"]) < 1e-9 * r1["loss"]


def _solve_all(node, p, sample=None, want_pred=False):
    from crnn_amd import p2vec_jac
    th, dth = p2vec_jac(node.pmap, node.ns, node.nr, p)
    return node._solve(node._ctx, node.B, th, dth, 0, node.B, sample, want_pred)


@pytest.mark.parametrize("case,pkey,tol", [("case2", "p_ckpt", None), ("case2", "p_init", None), ("rober", "p_ckpt", None),
                                           ("case2", "p_ckpt", (1e-10, 1e-8)), ("rober", "p_ckpt", (1e-9, 1e-6))])
def test_adjoint_equals_forward_tangents(case2_setup, rober_setup, case, pkey, tol):
    """Two differentiations of the same accepted steps: identical losses/step counts, gradients equal to rounding."""
    setup = case2_setup if case == "case2" else rober_setup
    p = setup[pkey]
    kw = {} if tol is None else dict(atol=tol[0], rtol=tol[1], maxiters=10**6)
    fwd, adj = _node(case, setup, grad_mode=1, **kw), _node(case, setup, grad_mode=2, **kw)
    lf, gf = fwd.loss_and_grad(p)
    la, ga = adj.loss_and_grad(p)
    assert fwd.last_stats["n_accept"] == adj.last_stats["n_accept"] and fwd.last_stats["n_reject"] == adj.last_stats["n_reject"]
    assert abs(lf - la) < 1e-13 * abs(lf)
    assert np.max(np.abs(gf - ga)) < 1e-9 * np.max(np.abs(gf))
    B = setup["u0"].shape[0]
    for i in (1, B // 2):
        assert np.max(np.abs(fwd.gradient(p, i) - adj.gradient(p, i))) < 1e-9 * np.max(np.abs(fwd.gradient(p, i)))
    # predictions requested together with the gradient come out of the adjoint kernel's forward sweep: the one-lane kernel
    # performs the forward-tangent kernel's primal arithmetic operation for operation (bit-identical), the two-lane kernel
    # (case2's default at this size, ros23_adj2_kernel.hpp) forms the species sums in another order (1e-12)
    pf, lsf, gsf, _, _ = _solve_all(fwd, p, want_pred=True)
    for lanes in ((1, 2) if case == "case2" else (1,)):
        adj.set_lanes_per_traj(lanes)
        pa, lsa, gsa, _, _ = _solve_all(adj, p, want_pred=True)
        assert adj.last_lanes_per_traj() == lanes
        if lanes == 1:
            assert np.array_equal(pf, pa)
        else:
            assert np.max(np.abs(pf - pa)) < 1e-12 * np.max(np.abs(pf))
        assert np.max(np.abs(lsf - lsa) / lsf) < 1e-13 * (1 if lanes == 1 else 100)
        assert np.max(np.abs(gsf - gsa)) < 1e-9 * np.max(np.abs(gsf))


def test_adjoint_truncated_and_failed_trajectories(rober_setup):
    """sample horizon, maxiters failures (gradient over the saved prefix) and the tape-overflow fallback."""
    s = rober_setup
    p = s["p_ckpt"]
    for kw, sample in ((dict(), 33), (dict(maxiters=20), None), (dict(maxiters=45), 36)):
        fwd, adj = _node("rober", s, grad_mode=1, **kw), _node("rober", s, grad_mode=2, **kw)
        _, lf, gf, rf, nf = _solve_all(fwd, p, sample=sample)
        _, la, ga, ra, na = _solve_all(adj, p, sample=sample)
        assert np.array_equal(rf, ra) and np.array_equal(nf, na)
        assert np.max(np.abs(lf - la)) <= 1e-13 * np.max(np.abs(lf))
        assert np.max(np.abs(gf - ga)) <= 1e-9 * np.max(np.abs(gf))
        if kw.get("maxiters") == 20:
            assert np.all(rf == 1) and np.all(nf < len(s["tsteps"]))
    # 8 tape slots for ~35-step trajectories: every trajectory overflows, the call is repeated with forward tangents
    fwd, tiny = _node("rober", s, grad_mode=1), _node("rober", s, grad_mode=2, tape_steps=8)
    _, lf, gf, rf, _ = _solve_all(fwd, p)
    _, lt, gt, rt, _ = _solve_all(tiny, p)
    assert np.array_equal(lf, lt) and np.array_equal(gf, gt)
    assert np.all(rt == 0) and np.array_equal(rf, rt)


def test_case2_converged_golden(case2_setup):
    """Tight tolerance vs the committed SciPy Radau (rtol 1e-12) trajectories of the
    checkpoint CRNN and the converged continuous-sensitivity gradients:
    north-star bar 'within 1e-6 rel-err' (met at tight tolerance)."""
    s = case2_setup
    node = _node("case2", s, atol=1e-10, rtol=1e-8)
    pred = node.predict_neuralode(s["u0"], s["p_ckpt"])
    gold = s["pred_ckpt"]
    scale = np.abs(gold[:, :6]).max()
    assert np.max(np.abs(pred[:, :6] - gold[:, :6])) / scale < 1e-6
    # clamp.(Array(sol), -ub, ub) also clamps the temperature row to ub = 10 (case2/case2.jl:126)
    assert np.all(pred[:, 6] == 10.0)
    for g in s["grads"]:
        p = s["p_ckpt"] if g["p"] == "ckpt" else s["p_init"]
        grad = node.gradient(p, g["ic"])
        gg = np.array(g["grad"])
        assert np.max(np.abs(grad - gg)) < 2e-5 * np.max(np.abs(gg))
        # the loss inherits the solver tolerance (rtol 1e-8 -> a few 1e-6 relative on an MAE of 2e-2)
        assert abs(node.loss_neuralode(p, g["ic"]) - g["loss"]) < 1e-5 * g["loss"]


def test_rober_converged_golden(rober_setup):
    s = rober_setup
    node = _node("rober", s, atol=1e-12, rtol=1e-8, maxiters=10**7)
    pred = node.predict_neuralode(s["u0"], s["p_ckpt"])
    gold = s["pred_ckpt"]
    scale = np.abs(gold).max(axis=(0, 2), keepdims=True)
    assert np.max(np.abs(pred - gold) / scale) < 1e-5
    for g in s["grads"]:
        grad = node.gradient(s["p_ckpt"], g["ic"])
        gg = np.array(g["grad"])
        assert np.max(np.abs(grad - gg)) < 1e-4 * np.max(np.abs(gg))


def test_robertson_known_answers(rober_setup):
    """Classical Robertson (1,0,0) values through the CRNN form of the true mechanism."""
    from crnn_amd import NeuralODE, ODEProblem, PRESET_ROBER, cases
    kat = rober_setup["kat"]
    t = np.array(kat["t"])
    # 3 reactions only: pad the rober-shaped (nr = 6) weights with three null reactions
    th3 = cases.rober_true_theta()
    w_in = np.zeros((3, 6)); w_b = np.full(6, -700.0); w_out = np.zeros((3, 6))
    w_in[:, :3] = th3[:9].reshape((3, 3), order="F"); w_b[:3] = th3[9:12]; w_out[:, :3] = th3[12:].reshape((3, 3), order="F")
    theta = cases.pack_theta(w_in, w_b, w_out)
    node = NeuralODE(ODEProblem(PRESET_ROBER, t, atol=1e-14, rtol=1e-9, maxiters=10**7, lb=1e-300))
    pred = node.predict_theta(np.array([1.0, 0.0, 0.0]), theta)
    y = np.array(kat["y"]).T
    assert np.max(np.abs(pred - y) / np.abs(y)) < 1e-6


def test_sample_horizon_and_subrange(orc, rober_setup):
    """robertson's random horizon `sample` (rober_crnn.jl:125,218) and [first, first+count) sub-ranges."""
    s = rober_setup
    p = s["p_ckpt"]
    node = _node("rober", s)
    sample = 33
    ref = _oracle_batch(orc, "rober", s, p, sample=sample)
    loss, grad = node.loss_and_grad(p, first=1, count=4, sample=sample)
    assert abs(loss - ref["loss"][1:5].mean()) < 1e-9 * loss
    # oracle grad over the sub-range
    kind, ns, nr = 3, 3, 6
    th, dth = orc.p2vec(kind, ns, nr, p)
    pb = oracle_problem(orc, "rober", s)
    g = np.zeros(43)
    for i in range(1, 5):
        g += orc.solve_one(pb, th, s["u0"][i], s["tsteps"][:sample], s["data"][i][:, :sample], dtheta=dth)["grad"]
    assert np.max(np.abs(grad - g / 4)) < 1e-7 * np.max(np.abs(g / 4))
    pred = node.predict_neuralode(s["u0"][2], p, sample=sample)
    assert pred.shape == (3, sample)
    assert np.all(node.last_n_saved == sample)


def test_failed_trajectories_are_reported_not_raised(orc, rober_setup, capsys):
    """maxiters exhaustion: retcode 1, truncated prefix kept (rober_crnn.jl:130-134)."""
    s = rober_setup
    p = s["p_ckpt"]
    node = _node("rober", s, maxiters=20)
    ref = _oracle_batch(orc, "rober", s, p, maxiters=20)
    pred = node.predict_neuralode(s["u0"], p)
    assert "ode solver failed" in capsys.readouterr().out
    assert np.array_equal(node.last_retcode, ref["retcode"]) and np.all(ref["retcode"] == 1)
    assert np.array_equal(node.last_n_saved, ref["n_saved"])
    loss, grad = node.loss_and_grad(p)
    assert abs(loss - ref["loss"].mean()) < 1e-9 * abs(loss)
    gref = ref["grad"] / len(ref["loss"])
    assert np.max(np.abs(grad - gref)) < 1e-7 * np.max(np.abs(gref))
    assert node.last_stats["n_ok"] == 0


def test_training_step_matches_host_chain(case2_setup):
    """Device-resident train step == host loss_and_grad + update!(opt, p, grad)."""
    from crnn_amd import Optimiser, PRESET_CASE2
    s = case2_setup
    node = _node("case2", s)
    p = s["p_init"].copy()
    opt_dev = Optimiser(25, PRESET_CASE2)
    node.train_init(opt_dev, p)
    opt_host = Optimiser(25, PRESET_CASE2)
    p_host = p.copy()
    for _ in range(3):
        loss_h, g = node.loss_and_grad(p_host)
        opt_host.update_(p_host, g)
        loss_d = node.train_step()
        assert abs(loss_d - loss_h) < 1e-12 * abs(loss_h)
        assert np.max(np.abs(node.params() - p_host)) < 1e-12


def test_case1_rosenbrock23_adjoint_woodbury_4x4(orc, fx):
    """case1 shape (5 species, 4 reactions: Woodbury with a 4x4 pivoted LU and its transposed solve) on the stiff stepper:
    adjoint == forward tangents == oracle."""
    from crnn_amd import NeuralODE, ODEProblem, PRESET_CASE1, SOLVER_ROSENBROCK23, cases
    rng = np.random.Generator(np.random.PCG64(12))
    ts = cases.case1_tsteps()
    u0 = np.array(fx["case1"]["u0"])
    p = np.array(fx["case1"]["p"])
    gen = NeuralODE(ODEProblem(PRESET_CASE1, ts, atol=1e-12, rtol=1e-10))
    data = cases.add_noise(gen.predict_theta(u0, cases.case1_true_theta()), 0.05, rng)
    gen.close()
    ys = cases.max_min(data, lb=LB_CASE1)
    res = {}
    for mode in (1, 2):
        node = NeuralODE(ODEProblem(PRESET_CASE1, ts, solver=SOLVER_ROSENBROCK23, grad_mode=mode, atol=1e-7, rtol=1e-5))
        node.set_ensemble(u0, data, ys)
        res[mode] = node.loss_and_grad(p) + (node.last_stats["n_accept"],)
        node.close()
    th, dth = orc.p2vec(1, 5, 4, p)
    pb = orc.make_problem(ns=5, nr=4, lb=LB_CASE1, ub=10.0, atol=1e-7, rtol=1e-5, yscale=ys, clamp_pred=1, maxiters=10000, solver=0)
    B = u0.shape[0]
    ref = orc.solve_batch(pb, th, np.ascontiguousarray(u0.T), ts, np.ascontiguousarray(data.transpose(2, 1, 0)), dtheta=dth)
    gref = ref["grad"] / B
    for mode in (1, 2):
        loss, grad, nacc = res[mode]
        assert nacc == ref["naccept"] and abs(loss - ref["loss"].mean()) < 1e-9 * loss
        assert np.max(np.abs(grad - gref)) < 1e-7 * np.max(np.abs(gref))
    assert np.max(np.abs(res[1][1] - res[2][1])) < 1e-9 * np.max(np.abs(gref))


def test_async_adjoint_training_and_deferred_replay(rober_setup):
    """crnn_train_step does not wait for the adjoint's tape-overflow flag: a step whose overflow count is not zero is skipped on the device (and
    everything after it), and repeated in order with forward tangents when the host next looks.  End states must be
    bit-identical to training that never used the adjoint."""
    from crnn_amd import Optimiser, PRESET_ROBER
    s = rober_setup
    p0 = s["p_ckpt"]
    samples = [20, 40, 22, 40, 25, 40, 40, 21]

    def run(per_step_loss, **kw):
        node = _node("rober", s, **kw)
        node.train_init(Optimiser(43, PRESET_ROBER), p0)
        losses = [node.train_step(sample=sm, want_loss=per_step_loss) for sm in samples]
        return node.params(), losses, node

    p_fwd, l_fwd, _ = run(True, grad_mode=1)
    # (a) roomy tape: async adjoint == adjoint with a look after every step; gradients equal forward tangents to rounding
    p_async, _, _ = run(False, grad_mode=2)
    p_sync, l_sync, _ = run(True, grad_mode=2)
    assert np.array_equal(p_async, p_sync)
    assert np.max(np.abs(p_sync - p_fwd)) < 1e-9 and np.max(np.abs(np.array(l_sync) - np.array(l_fwd))) < 1e-12
    # (b) 30 tape slots: horizons 20-25 fit, the full horizon (about 35 steps) overflows -> skipped steps are replayed
    p_tiny, _, node = run(False, grad_mode=2, tape_steps=30)
    p_tiny_sync, l_tiny_sync, _ = run(True, grad_mode=2, tape_steps=30)
    assert np.array_equal(p_tiny, p_tiny_sync)
    # from the first overflowing step on, everything was done with forward tangents (sticky skip + in-order replay)
    assert np.max(np.abs(p_tiny - p_fwd)) < 1e-9
    assert np.max(np.abs(np.array(l_tiny_sync) - np.array(l_fwd))) < 1e-12
    # (c) every step overflows: pure forward-tangent training, bit for bit
    p_all, _, _ = run(False, grad_mode=2, tape_steps=4)
    assert np.array_equal(p_all, p_fwd)


def test_rccl_single_rank_allreduce(case2_setup):
    """The in-library RCCL path on a 1-rank communicator (multi-rank needs >1 GPU)."""
    import ctypes as C
    from crnn_amd import _lib as L
    node = _node("case2", case2_setup)
    uid = C.create_string_buffer(L.UNIQUE_ID_BYTES)
    L.check(L.lib.crnn_comm_get_unique_id(uid))
    L.check(L.lib.crnn_comm_init(node.handle, uid, 0, 1), node.handle)
    buf = np.arange(27, dtype=np.float64)
    L.check(L.lib.crnn_allreduce_grad(node.handle, L.dptr(buf), 27), node.handle)
    assert np.array_equal(buf, np.arange(27, dtype=np.float64))
    loss0, _ = node.loss_and_grad(case2_setup["p_ckpt"])
    from crnn_amd import Optimiser, PRESET_CASE2
    node.train_init(Optimiser(25, PRESET_CASE2), case2_setup["p_ckpt"])
    assert abs(node.train_step() - loss0) < 1e-12
    L.check(L.lib.crnn_comm_destroy(node.handle), node.handle)


# ------------------------------------------------------------------ Tsit5 (case1's algorithm; case2's non-stiff branch)
def _tsit5_node(preset, setup, **kw):
    from crnn_amd import NeuralODE, ODEProblem, SOLVER_TSIT5
    node = NeuralODE(ODEProblem(preset, setup["tsteps"], solver=SOLVER_TSIT5, **kw))
    node.set_ensemble(setup["u0"], setup["data"], setup["yscale"])
    return node


@pytest.mark.parametrize("pkey", ["p_ckpt", "p_init"])
def test_tsit5_case2_matches_oracle(orc, case2_setup, pkey):
    from crnn_amd import PRESET_CASE2
    s = case2_setup
    p = s[pkey]
    node = _tsit5_node(PRESET_CASE2, s)
    th, dth = orc.p2vec(2, 6, 3, p)
    pb = oracle_problem(orc, "case2", s, solver=1)
    ref = orc.solve_batch(pb, th, np.ascontiguousarray(s["u0"].T), s["tsteps"],
                          np.ascontiguousarray(s["data"].transpose(2, 1, 0)), dtheta=dth, want_pred=True)
    pred = node.predict_neuralode(s["u0"], p)
    ref_pred = ref["pred"].transpose(2, 1, 0)
    scale = np.abs(ref_pred).max(axis=(0, 2), keepdims=True) + 1e-300
    assert np.max(np.abs(pred - ref_pred) / scale) < 1e-9
    assert np.array_equal(node.last_retcode, ref["retcode"])
    loss, grad = node.loss_and_grad(p)
    st = node.last_stats
    assert st["n_accept"] == ref["naccept"] and st["n_reject"] == ref["nreject"]
    B = len(ref["loss"])
    assert abs(loss - ref["loss"].mean()) < 1e-9 * loss
    assert np.max(np.abs(grad - ref["grad"] / B)) < 1e-7 * np.max(np.abs(ref["grad"] / B))
    # 5th order: far fewer steps than Rosenbrock23 at the same tolerance, tighter answer
    gold = s["pred_ckpt"]
    if pkey == "p_ckpt":
        assert np.max(np.abs(pred[:, :6] - gold[:, :6])) / np.abs(gold[:, :6]).max() < 1e-4
    node.close()
    node = _tsit5_node(PRESET_CASE2, s, atol=1e-11, rtol=1e-9)
    if pkey == "p_ckpt":
        pred = node.predict_neuralode(s["u0"], p)
        assert np.max(np.abs(pred[:, :6] - gold[:, :6])) / np.abs(gold[:, :6]).max() < 1e-8
    for g in s["grads"]:
        if (g["p"] == "ckpt") == (pkey == "p_ckpt"):
            gg = np.array(g["grad"])
            assert np.max(np.abs(node.gradient(p, g["ic"]) - gg)) < 2e-6 * np.max(np.abs(gg))
    node.close()


def test_tsit5_case1_reference_configuration(orc, fx):
    """BASELINE config 0: case1 (5 species / 4 reactions), Tsit5, atol 1e-5, rtol 1e-2, maxiters 10000 (case1.jl:19-35)."""
    from crnn_amd import NeuralODE, ODEProblem, PRESET_CASE1, cases
    rng = np.random.Generator(np.random.PCG64(11))
    ts = cases.case1_tsteps()
    u0 = np.array(fx["case1"]["u0"])
    p = np.array(fx["case1"]["p"])
    gen = NeuralODE(ODEProblem(PRESET_CASE1, ts, atol=1e-12, rtol=1e-10))
    clean = gen.predict_theta(u0, cases.case1_true_theta())
    gen.close()
    data = cases.add_noise(clean, 0.05, rng)
    ys = cases.max_min(data, lb=LB_CASE1)
    node = NeuralODE(ODEProblem(PRESET_CASE1, ts))          # preset = the reference's Tsit5 configuration
    node.set_ensemble(u0, data, ys)
    th, dth = orc.p2vec(1, 5, 4, p)
    pb = orc.make_problem(ns=5, nr=4, lb=LB_CASE1, ub=10.0, atol=1e-5, rtol=1e-2, yscale=ys, clamp_pred=1, maxiters=10000, solver=1)
    B = u0.shape[0]
    ref = orc.solve_batch(pb, th, np.ascontiguousarray(u0.T), ts, np.ascontiguousarray(data.transpose(2, 1, 0)), dtheta=dth,
                          want_pred=True)
    pred = node.predict_neuralode(u0, p)
    assert np.max(np.abs(pred - ref["pred"].transpose(2, 1, 0))) < 1e-9
    loss, grad = node.loss_and_grad(p)
    assert abs(loss - ref["loss"].mean()) < 1e-9 * loss
    assert np.max(np.abs(grad - ref["grad"] / B)) < 1e-7 * np.max(np.abs(ref["grad"] / B))
    assert node.last_stats["n_accept"] == ref["naccept"]
    # single initial condition, as the reference's per-IC loop does (case1.jl:191-201)
    g0 = node.gradient(p, 0)
    r0 = orc.solve_one(pb, th, u0[0], ts, data[0], dtheta=dth)
    assert np.max(np.abs(g0 - r0["grad"])) < 1e-7 * np.max(np.abs(r0["grad"]))
    node.close()


# ------------------------------------------------------------------ Tsit5 adjoint and the AutoTsit5(Rosenbrock23()) composite
@pytest.mark.parametrize("pkey", ["p_ckpt", "p_init"])
def test_tsit5_adjoint_equals_forward_tangents(case2_setup, pkey):
    """Reversing the accepted Tsit5 steps gives the gradient the forward tangents give (same graph): 1e-9 relative."""
    from crnn_amd import PRESET_CASE2
    s = case2_setup
    p = s[pkey]
    fwd = _tsit5_node(PRESET_CASE2, s, grad_mode=1)
    adj = _tsit5_node(PRESET_CASE2, s, grad_mode=2)
    lf, gf = fwd.loss_and_grad(p)
    la, ga = adj.loss_and_grad(p)
    assert abs(lf - la) < 1e-13 * abs(lf)
    assert np.max(np.abs(gf - ga)) < 1e-9 * np.max(np.abs(gf))
    assert adj.last_stats["n_accept"] == fwd.last_stats["n_accept"] and adj.last_stats["n_reject"] == fwd.last_stats["n_reject"]
    for i in (0, 3):
        assert np.max(np.abs(fwd.gradient(p, i) - adj.gradient(p, i))) < 1e-9 * np.max(np.abs(gf))
    # random horizon (rober_crnn.jl:125,218 style) goes through the same tape
    lf, gf = fwd.loss_and_grad(p, sample=31)
    la, ga = adj.loss_and_grad(p, sample=31)
    assert abs(lf - la) < 1e-13 * abs(lf) and np.max(np.abs(gf - ga)) < 1e-9 * np.max(np.abs(gf))
    fwd.close(); adj.close()


def _auto_node(case, setup, **kw):
    from crnn_amd import NeuralODE, ODEProblem, PRESET_CASE2, PRESET_ROBER, SOLVER_AUTOTSIT5
    if case == "case2":
        prob = ODEProblem(PRESET_CASE2, setup["tsteps"], solver=SOLVER_AUTOTSIT5, **kw)
    else:
        prob = ODEProblem(PRESET_ROBER, setup["tsteps"], rate_scale=setup["dydt_scale"], solver=SOLVER_AUTOTSIT5, **kw)
    node = NeuralODE(prob)
    node.set_ensemble(setup["u0"], setup["data"], setup["yscale"])
    return node


def test_autotsit5_robertson_matches_oracle(orc, rober_setup):
    """Stiff problem: the composite leaves Tsit5 after its 11th stiff step in a row and finishes on Rosenbrock23
    (62 steps instead of Tsit5's ~19 000).  Same switching rule, same inputs: the step sequences coincide with the
    CPU restatement's; trajectories 1e-9 of the species scale, gradient 1e-7 of max |grad|."""
    s = rober_setup
    p = s["p_ckpt"]
    th, dth = orc.p2vec(3, 3, 6, p)
    pb = oracle_problem(orc, "rober", s, solver=2)
    ref = orc.solve_batch(pb, th, np.ascontiguousarray(s["u0"].T), s["tsteps"],
                          np.ascontiguousarray(s["data"].transpose(2, 1, 0)), dtheta=dth, want_pred=True)
    one = orc.solve_one(pb, th, s["u0"][0], s["tsteps"], s["data"][0])
    assert one["n_switch"] == 1 and one["n_rosenbrock"] > one["n_tsit5"] > 10      # the case does exercise the switch
    node = _auto_node("rober", s)
    pred = node.predict_neuralode(s["u0"], p)
    ref_pred = ref["pred"].transpose(2, 1, 0)
    scale = np.abs(ref_pred).max(axis=(0, 2), keepdims=True) + 1e-300
    assert np.max(np.abs(pred - ref_pred) / scale) < 1e-9
    assert np.array_equal(node.last_retcode, ref["retcode"])
    losses = node.losses(p)
    assert np.max(np.abs(losses - ref["loss"]) / ref["loss"]) < 1e-9
    loss, grad = node.loss_and_grad(p)
    st = node.last_stats
    assert st["n_accept"] == ref["naccept"] and st["n_reject"] == ref["nreject"]
    B = len(ref["loss"])
    assert abs(loss - ref["loss"].mean()) < 1e-9 * loss
    assert np.max(np.abs(grad - ref["grad"] / B)) < 1e-7 * np.max(np.abs(ref["grad"] / B))
    g0 = node.gradient(p, 2)
    r0 = orc.solve_one(pb, th, s["u0"][2], s["tsteps"], s["data"][2], dtheta=dth)
    assert np.max(np.abs(g0 - r0["grad"])) < 1e-7 * np.max(np.abs(r0["grad"]))
    # against the tight-tolerance golden trajectories (independent Radau run)
    gold = s["pred_ckpt"]
    gscale = np.abs(gold).max(axis=(0, 2))[:, None]
    assert np.max(np.abs(pred[:6] - gold[:6]) / gscale) < 5e-3
    node.close()


def test_autotsit5_with_constant_temperature_state_is_tsit5(case2_setup):
    """case2 carries its constant temperature as a state: the composite's stiffness estimate is 0/0 there and it never
    leaves Tsit5 (auto_adj_kernel.hpp), so AUTOTSIT5 and TSIT5 give identical numbers."""
    from crnn_amd import PRESET_CASE2
    s = case2_setup
    p = s["p_ckpt"]
    a = _auto_node("case2", s)
    t5 = _tsit5_node(PRESET_CASE2, s, grad_mode=2)
    la, ga = a.loss_and_grad(p)
    lt, gt = t5.loss_and_grad(p)
    assert la == lt and np.array_equal(ga, gt)
    pa = a.predict_neuralode(s["u0"], p)
    pt = t5.predict_neuralode(s["u0"], p)
    assert np.max(np.abs(pa - pt)) < 1e-12       # tape kernel vs the primal Tsit5 kernel: same steps, same interpolant
    assert np.array_equal(a.losses(p), t5.losses(p)) or np.max(np.abs(a.losses(p) - t5.losses(p))) < 1e-13
    a.close(); t5.close()


def test_autotsit5_forward_mode_and_tape_overflow_are_errors(rober_setup):
    from crnn_amd import CrnnError
    s = rober_setup
    node = _auto_node("rober", s, grad_mode=1)
    node.losses(s["p_ckpt"])                                     # primal calls do not depend on grad_mode
    with pytest.raises(CrnnError, match="discrete adjoint only"):
        node.loss_and_grad(s["p_ckpt"])
    node.close()
    node = _auto_node("rober", s, tape_steps=8)
    with pytest.raises(CrnnError, match="tape"):
        node.loss_and_grad(s["p_ckpt"])
    node.close()


def test_autotsit5_case1_matches_oracle(orc, fx):
    """case1 shape (5 species, 4 reactions, no temperature) through the composite at the reference tolerances."""
    from crnn_amd import NeuralODE, ODEProblem, PRESET_CASE1, SOLVER_AUTOTSIT5, cases
    rng = np.random.Generator(np.random.PCG64(12))
    ts = cases.case1_tsteps()
    u0 = np.array(fx["case1"]["u0"])
    p = np.array(fx["case1"]["p"])
    gen = NeuralODE(ODEProblem(PRESET_CASE1, ts, atol=1e-12, rtol=1e-10))
    clean = gen.predict_theta(u0, cases.case1_true_theta())
    gen.close()
    data = cases.add_noise(clean, 0.05, rng)
    ys = cases.max_min(data, lb=LB_CASE1)
    node = NeuralODE(ODEProblem(PRESET_CASE1, ts, solver=SOLVER_AUTOTSIT5))
    node.set_ensemble(u0, data, ys)
    th, dth = orc.p2vec(1, 5, 4, p)
    pb = orc.make_problem(ns=5, nr=4, lb=LB_CASE1, ub=10.0, atol=1e-5, rtol=1e-2, yscale=ys, clamp_pred=1, maxiters=10000, solver=2)
    B = u0.shape[0]
    ref = orc.solve_batch(pb, th, np.ascontiguousarray(u0.T), ts, np.ascontiguousarray(data.transpose(2, 1, 0)), dtheta=dth,
                          want_pred=True)
    pred = node.predict_neuralode(u0, p)
    assert np.max(np.abs(pred - ref["pred"].transpose(2, 1, 0))) < 1e-9
    loss, grad = node.loss_and_grad(p)
    assert abs(loss - ref["loss"].mean()) < 1e-9 * loss
    assert np.max(np.abs(grad - ref["grad"] / B)) < 1e-7 * np.max(np.abs(ref["grad"] / B))
    assert node.last_stats["n_accept"] == ref["naccept"]
    node.close()


def test_training_with_tsit5_adjoint_and_composite(case2_setup, rober_setup):
    """The device-resident loop on the new steppers.  Tsit5 (case2): the asynchronous adjoint loop equals forward-tangent
    training to rounding, also when the tape is too short and the skipped steps are replayed with the Tsit5 forward-tangent
    kernel.  AutoTsit5 (robertson): train step == host loss_and_grad + update!."""
    from crnn_amd import Optimiser, PRESET_CASE2, PRESET_ROBER
    s = case2_setup
    p0 = s["p_init"]

    def run(**kw):
        node = _tsit5_node(PRESET_CASE2, s, **kw)
        node.train_init(Optimiser(25, PRESET_CASE2), p0)
        for _ in range(5):
            node.train_step()
        return node.params()

    p_fwd, p_adj, p_tiny = run(grad_mode=1), run(grad_mode=2), run(grad_mode=2, tape_steps=6)
    assert np.max(np.abs(p_adj - p_fwd)) < 1e-10
    assert np.array_equal(p_tiny, p_fwd)                      # every step overflowed -> replayed with forward tangents
    r = rober_setup
    node = _auto_node("rober", r)
    opt_host = Optimiser(43, PRESET_ROBER)
    p_host = r["p_ckpt"].copy()
    node.train_init(Optimiser(43, PRESET_ROBER), p_host)
    for _ in range(3):
        loss_h, g = node.loss_and_grad(p_host)
        opt_host.update_(p_host, g)
        loss_d = node.train_step(want_loss=True)
        assert abs(loss_d - loss_h) < 1e-12 * abs(loss_h)
        assert np.max(np.abs(node.params() - p_host)) < 1e-12
    node.close()

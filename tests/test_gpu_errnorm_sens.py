"""errnorm_sens = 1 on the device: the step-size controller of a GRADIENT call sees ForwardDiff's dual-inclusive error norm
(reference: ForwardDiff.gradient through the adaptive solver, case2/case2.jl:195, robertson/rober_crnn.jl:219), chunked the
way ForwardDiff chunks P (25 -> 9 + 9 + 7, 43 -> 11 + 11 + 11 + 10, 24 -> 12 + 12): every chunk is its own adaptive solve.
HIP kernel (ros23_sens_kernel.hpp) vs the oracle's errnorm_sens = 1 on the same chunk of directions: identical accepted /
rejected step counts, gradients to 1e-7 of max |grad|.  [UNVERIFIED-DEP]: the norm itself is a restatement of
DiffEqBase.ODE_DEFAULT_NORM on Dual arrays (oracle header)."""
import numpy as np

LB_CASE1, LB_CASE2 = float(np.float32(1e-5)), float(np.float32(1e-6))   # `lb = 1.f-5` / `lb = 1.f-6`: Float32 literals (case1/case1.jl:34, case2/case2.jl:34)
import pytest

from conftest import oracle_problem

pytestmark = pytest.mark.gpu


def _setup(case, case2_setup, rober_setup, fx):
    from crnn_amd import NeuralODE, ODEProblem, PRESET_CASE1, PRESET_CASE2, PRESET_ROBER, SOLVER_ROSENBROCK23, SOLVER_TSIT5, cases
    if case == "case2":
        s = case2_setup
        mk = lambda **kw: NeuralODE(ODEProblem(PRESET_CASE2, s["tsteps"], **kw))
        return s, mk, (2, 6, 3), lambda orc, **kw: oracle_problem(orc, "case2", s, **kw)
    if case == "case2-tsit5":      # the non-stiff branch of case2's AutoTsit5(Rosenbrock23()) (case2.jl:26)
        s = case2_setup
        mk = lambda **kw: NeuralODE(ODEProblem(PRESET_CASE2, s["tsteps"], solver=SOLVER_TSIT5, **kw))
        return s, mk, (2, 6, 3), lambda orc, **kw: oracle_problem(orc, "case2", s, solver=1, **kw)
    if case == "rober":
        s = rober_setup
        mk = lambda **kw: NeuralODE(ODEProblem(PRESET_ROBER, s["tsteps"], rate_scale=s["dydt_scale"], **kw))
        return s, mk, (3, 3, 6), lambda orc, **kw: oracle_problem(orc, "rober", s, **kw)
    rng = np.random.Generator(np.random.PCG64(12))
    ts = cases.case1_tsteps()
    u0 = np.array(fx["case1"]["u0"])
    gen = NeuralODE(ODEProblem(PRESET_CASE1, ts, atol=1e-12, rtol=1e-10))
    data = cases.add_noise(gen.predict_theta(u0, cases.case1_true_theta()), 0.05, rng)
    gen.close()
    ys = cases.max_min(data, lb=LB_CASE1)
    s = dict(u0=u0, tsteps=ts, data=data, yscale=ys, p_ckpt=np.array(fx["case1"]["p"]))
    sv = 1 if case == "case1-tsit5" else 0           # case1's own algorithm is Tsit5 (case1.jl:28)
    mk = lambda **kw: NeuralODE(ODEProblem(PRESET_CASE1, ts, solver=SOLVER_TSIT5 if sv else SOLVER_ROSENBROCK23, **kw))
    mkpb = lambda orc, **kw: orc.make_problem(ns=5, nr=4, lb=LB_CASE1, ub=10.0, atol=1e-5, rtol=1e-2, yscale=ys, clamp_pred=1,
                                              maxiters=10000, solver=sv, **kw)
    return s, mk, (1, 5, 4), mkpb


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("case,pkey", [("case2", "p_ckpt"), ("case2", "p_init"), ("rober", "p_ckpt"), ("case1", "p_ckpt"),
                                       ("case1-tsit5", "p_ckpt"), ("case2-tsit5", "p_ckpt"), ("case2-tsit5", "p_init")])
def test_chunked_dual_norm_gradient_matches_oracle(orc, fx, case2_setup, rober_setup, case, pkey, mode):
    from crnn_amd.api import fd_chunk_size
    s, mk, (kind, ns, nr), mkpb = _setup(case, case2_setup, rober_setup, fx)
    p = s[pkey]
    # mode 1: squared norm / length(u); mode 2: / totallength(u) = n (1 + partials per Dual) -- the one the reference's recorded history selects
    node = mk(errnorm_sens=mode)
    node.set_ensemble(s["u0"], s["data"], s["yscale"])
    th, dth = orc.p2vec(kind, ns, nr, p)
    P = dth.shape[1]
    chunk = fd_chunk_size(P)
    assert chunk == {25: 9, 43: 11, 24: 12}[P]
    pb1 = mkpb(orc, errnorm_sens=mode)
    B = s["u0"].shape[0]
    differs = 0
    for b in range(min(B, 6)):
        g = node.gradient(p, b)
        plain = orc.solve_one(mkpb(orc), th, s["u0"][b], s["tsteps"], s["data"][b], dtheta=None)
        gref = np.zeros(P)
        for c, k0 in enumerate(range(0, P, chunk)):
            k1 = min(P, k0 + chunk)
            cols = np.zeros((dth.shape[0], chunk), order="F")      # the Dual carries `chunk` partials, the surplus ones zero
            cols[:, :k1 - k0] = dth[:, k0:k1]
            r = orc.solve_one(pb1, th, s["u0"][b], s["tsteps"], s["data"][b], dtheta=cols, want_pred=False)
            gref[k0:k1] = r["grad"][:k1 - k0]
            assert node.last_chunk_stats[c] == (r["naccept"], r["nreject"]), (case, b, c)
            differs += (r["naccept"], r["nreject"]) != (plain["naccept"], plain["nreject"])
        assert np.max(np.abs(g - gref)) < 1e-7 * np.max(np.abs(gref)), (case, b)
    if mode == 1:
        assert differs > 0    # the dual-inclusive norm really changes the step sequence somewhere (mode 2 divides by n (1 + N):
                              # there the sequence may coincide with the plain one on an easy problem)


def test_loss_grad_and_training_step_assemble_the_chunks(orc, case2_setup):
    """crnn_loss_grad / crnn_train_step with errnorm_sens = 1: gradient = concatenated chunk gradients (oracle, batched),
    loss and step statistics = those of the plain solve (what loss_neuralode evaluates); the device training step equals
    host gradient + update!."""
    from crnn_amd import NeuralODE, ODEProblem, Optimiser, PRESET_CASE2
    from crnn_amd.api import fd_chunk_size
    s = case2_setup
    p = s["p_init"]
    node = NeuralODE(ODEProblem(PRESET_CASE2, s["tsteps"], errnorm_sens=1))
    node.set_ensemble(s["u0"], s["data"], s["yscale"])
    loss, grad = node.loss_and_grad(p)
    th, dth = orc.p2vec(2, 6, 3, p)
    B = s["u0"].shape[0]
    u0T = np.ascontiguousarray(s["u0"].T); dT = np.ascontiguousarray(s["data"].transpose(2, 1, 0))
    pb1 = oracle_problem(orc, "case2", s, errnorm_sens=1)
    gref = np.zeros(25)
    for k0 in range(0, 25, 9):
        k1 = min(25, k0 + 9)
        gref[k0:k1] = orc.solve_batch(pb1, th, u0T, s["tsteps"], dT, dtheta=dth[:, k0:k1])["grad"] / B
    plain = orc.solve_batch(oracle_problem(orc, "case2", s), th, u0T, s["tsteps"], dT)
    assert abs(loss - plain["loss"].mean()) < 1e-9 * abs(loss)
    assert node.last_stats["n_accept"] == plain["naccept"] and node.last_stats["n_reject"] == plain["nreject"]
    assert np.max(np.abs(grad - gref)) < 1e-7 * np.max(np.abs(gref))
    # the primal-only norm gives a (slightly) different gradient: the two modes are distinguishable
    node0 = NeuralODE(ODEProblem(PRESET_CASE2, s["tsteps"]))
    node0.set_ensemble(s["u0"], s["data"], s["yscale"])
    _, g0 = node0.loss_and_grad(p)
    assert 1e-9 < np.max(np.abs(g0 - grad)) / np.max(np.abs(grad)) < 1e-1
    # training step == host chain
    opt_h = Optimiser(25, PRESET_CASE2); p_h = p.copy()
    node.train_init(Optimiser(25, PRESET_CASE2), p)
    for _ in range(3):
        l_h, g_h = node.loss_and_grad(p_h)
        opt_h.update_(p_h, g_h)
        l_d = node.train_step()
        assert abs(l_d - l_h) < 1e-12 * abs(l_h) and np.max(np.abs(node.params() - p_h)) < 1e-12
    node.close(); node0.close()


@pytest.mark.parametrize("case", ["case2", "rober", "case2-tsit5"])
def test_all_chunks_in_one_launch_equal_one_launch_per_chunk(monkeypatch, fx, case2_setup, rober_setup, case):
    """crnn_loss_grad with errnorm_sens = 1 runs ForwardDiff's chunks (case2 9 + 9 + 7, robertson 11 + 11 + 11 + 10) as the batches
    of ONE launch where the kernel stages all of d theta / d p; CRNN_SENS_ONE_LAUNCH=0 (read at context creation) keeps one launch
    per chunk.  The chunks are the same independent adaptive solves either way: per-trajectory gradient rows are the same bits,
    the batch sums differ only by the order of the fixed-order reduction (row pitch 25 / 43 against 9 / 12)."""
    s, mk, _, _ = _setup(case, case2_setup, rober_setup, fx)
    p = s["p_ckpt"]
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("CRNN_SENS_ONE_LAUNCH", mode)
        node = mk(errnorm_sens=1)
        node.set_ensemble(s["u0"], s["data"], s["yscale"])
        out[mode] = (node.loss_and_grad(p), dict(node.last_stats))
        out[mode + "again"] = node.loss_and_grad(p)           # second call: the queue is ordered by the first call's plain solve
        node.close()
    (l1, g1), st1 = out["1"]
    (l0, g0), st0 = out["0"]
    assert l1 == l0 and st1["n_accept"] == st0["n_accept"] and st1["n_reject"] == st0["n_reject"]
    assert np.max(np.abs(g1 - g0)) <= 1e-13 * np.max(np.abs(g0))
    for mode in ("1", "0"):
        la, ga = out[mode + "again"]
        assert la == out[mode][0][0] and np.max(np.abs(ga - out[mode][0][1])) <= 1e-13 * np.max(np.abs(ga))


def test_errnorm_sens_rejects_unsupported_combinations():
    from crnn_amd import CrnnError, NeuralODE, ODEProblem, PRESET_CASE2, PRESET_ROBER, SOLVER_AUTOTSIT5, cases
    with pytest.raises(CrnnError, match="errnorm_sens"):      # a composite that does switch (no temperature state): tape kernel only
        NeuralODE(ODEProblem(PRESET_ROBER, cases.rober_tsteps(), errnorm_sens=1, solver=SOLVER_AUTOTSIT5))
    with pytest.raises(CrnnError, match="errnorm_sens"):
        NeuralODE(ODEProblem(PRESET_CASE2, cases.case2_tsteps(), errnorm_sens=1, grad_mode=2))


def test_the_references_alg_with_the_dual_norm_is_the_tsit5_gradient(case2_setup):
    """case2.jl:26 `alg = AutoTsit5(Rosenbrock23(autodiff=false))` + :195 `ForwardDiff.gradient`: a host that keeps the reference's `alg` and asks for
    the reference's error norm gets the gradient the reference's run evaluated -- the composite never leaves Tsit5 on a state with a constant
    temperature component (tests/test_case2_stream_pin.py: the recorded history says so to 5e-6), so the dual-norm chunks run on tsit5_sens_kernel:
    bit for bit the SOLVER_TSIT5 context's numbers (gradient(), loss_and_grad(), the device-resident training step).  Until round 6 this
    combination was refused at crnn_ctx_create.  robertson's shape (no temperature state: the composite does switch) still is."""
    from crnn_amd import CrnnError, NeuralODE, ODEProblem, Optimiser, PRESET_CASE2, PRESET_ROBER, SOLVER_AUTOTSIT5, SOLVER_TSIT5
    s = case2_setup
    nodes = {}
    for name, solver in (("auto", SOLVER_AUTOTSIT5), ("tsit5", SOLVER_TSIT5)):
        n = NeuralODE(ODEProblem(PRESET_CASE2, s["tsteps"], solver=solver, errnorm_sens=2))
        n.set_ensemble(s["u0"], s["data"], s["yscale"])
        nodes[name] = n
    for pkey in ("p_ckpt", "p_init"):
        p = s[pkey]
        ga, gt = nodes["auto"].gradient(p, 3), nodes["tsit5"].gradient(p, 3)
        assert np.array_equal(ga, gt) and nodes["auto"].last_chunk_stats == nodes["tsit5"].last_chunk_stats
        (la, Ga), (lt, Gt) = nodes["auto"].loss_and_grad(p), nodes["tsit5"].loss_and_grad(p)
        assert la == lt and np.array_equal(Ga, Gt)
    for n in nodes.values():
        n.train_init(Optimiser(25, preset=PRESET_CASE2), s["p_init"])
        for i in (2, 0, 5):
            n.train_step(first=i, count=1, want_loss=False)
    assert np.array_equal(nodes["auto"].params(), nodes["tsit5"].params())
    for n in nodes.values():
        n.close()
    with pytest.raises(CrnnError, match="errnorm_sens"):
        NeuralODE(ODEProblem(PRESET_ROBER, np.logspace(0, 5, 40), solver=SOLVER_AUTOTSIT5, errnorm_sens=2))

"""The primal launches (crnn_solve with no directions: predict_neuralode, the epoch-end loss loop -- case2.jl:124-128, 199-203;
rober_crnn.jl:123-144; HyChem crnn_pyrolysis_mass.jl:135-147; Cathode-UQ network.jl:196-275) run the gradient kernels' forward
sweep alone (ros23_adj_kernel / auto_adj_kernel / cathode_adj_kernel with PRIMAL, hychem2_kernel with GRAD = false): loss accumulated
at the save points as they are passed, no tape, no reverse sweep.

Checked here, on top of the oracle comparisons of the other test files (which call the same launches):
  * per-trajectory losses of the primal launch against those of the gradient launch (the same forward sweep; the loss terms summed
    in ascending instead of descending save order): 1e-13 relative; identical return codes, saved counts and step counts;
  * against the oracle: losses 1e-9 relative (the bar of tests/test_gpu_parity.py);
  * ragged ensembles, sub-ranges and a truncated horizon.
Floating point throughout; tolerances at each assert."""
import numpy as np

LB_CASE1, LB_CASE2 = float(np.float32(1e-5)), float(np.float32(1e-6))   # `lb = 1.f-5` / `lb = 1.f-6`: Float32 literals (case1/case1.jl:34, case2/case2.jl:34)
import pytest

from conftest import oracle_problem

pytestmark = pytest.mark.gpu


def _both(node, p, first=0, count=None, sample=None):
    from crnn_amd.api import p2vec_jac
    count = node.B - first if count is None else count
    th, dth = p2vec_jac(node.pmap, node.ns, node.nr, p)
    _, lg, _, rg, sg = node._solve(node._ctx, node.B, th, dth, first, count, sample, False)
    nag, nrg = node.step_counts(first, count)
    _, lp, _, rp, sp = node._solve(node._ctx, node.B, th, None, first, count, sample, False)
    nap, nrp = node.step_counts(first, count)
    sl = slice(first, first + count)
    assert np.array_equal(rg[sl], rp[sl]) and np.array_equal(sg[sl], sp[sl])
    assert np.array_equal(nag, nap) and np.array_equal(nrg, nrp)
    return lg[sl], lp[sl]


@pytest.mark.parametrize("solver", ["rosenbrock23", "tsit5", "autotsit5"])
def test_case2_primal_launch_matches_gradient_launch_and_oracle(orc, case2_setup, solver):
    from crnn_amd import NeuralODE, ODEProblem, PRESET_CASE2
    from crnn_amd import _lib as L
    s = case2_setup
    code = {"rosenbrock23": L.SOLVER_ROSENBROCK23, "tsit5": L.SOLVER_TSIT5, "autotsit5": L.SOLVER_AUTOTSIT5}[solver]
    node = NeuralODE(ODEProblem(PRESET_CASE2, s["tsteps"], solver=code, grad_mode=2))
    node.set_ensemble(s["u0"], s["data"], s["yscale"])
    for p in (s["p_ckpt"], s["p_init"]):
        lg, lp = _both(node, p)
        assert np.max(np.abs(lg - lp) / lg) < 1e-13
        th, _ = orc.p2vec(2, 6, 3, p)
        pb = oracle_problem(orc, "case2", s, solver={"rosenbrock23": 0, "tsit5": 1, "autotsit5": 2}[solver])
        ref = orc.solve_batch(pb, th, np.ascontiguousarray(s["u0"].T), s["tsteps"], np.ascontiguousarray(s["data"].transpose(2, 1, 0)))
        assert np.max(np.abs(lp - ref["loss"]) / ref["loss"]) < 1e-9
    # sub-range and truncated horizon (sample = 33 save points)
    B = node.B
    lg, lp = _both(node, s["p_ckpt"], first=3, count=B - 5, sample=33)
    assert np.max(np.abs(lg - lp) / lg) < 1e-13
    node.close()


def test_robertson_primal_launch(orc, rober_setup):
    from crnn_amd import NeuralODE, ODEProblem, PRESET_ROBER
    s = rober_setup
    node = NeuralODE(ODEProblem(PRESET_ROBER, s["tsteps"], rate_scale=s["dydt_scale"], grad_mode=2))
    node.set_ensemble(s["u0"], s["data"], s["yscale"])
    lg, lp = _both(node, s["p_ckpt"])
    assert np.max(np.abs(lg - lp) / lg) < 1e-13
    node.close()


def test_ragged_large_ensemble_primal_launch(case2_setup):
    """More trajectories than resident lanes, not a multiple of 64: the primal launch takes the queue order of the previous launch
    like the gradient launch does; predictions equal those of the forward-tangent kernel bit for bit (same stepper, same arithmetic)."""
    from crnn_amd import NeuralODE, ODEProblem, PRESET_CASE2, cases
    s = case2_setup
    B = 65536 + 4099
    rng = np.random.Generator(np.random.PCG64([5, 11]))
    u0 = cases.case2_u0(B, rng)
    data = np.abs(rng.standard_normal((B, 6, len(s["tsteps"])))) * 0.5
    node = NeuralODE(ODEProblem(PRESET_CASE2, s["tsteps"], grad_mode=2))
    node.set_ensemble(u0, data, cases.max_min(data, lb=LB_CASE2))
    lg, lp = _both(node, s["p_ckpt"])
    assert np.max(np.abs(lg - lp) / lg) < 1e-13
    l2 = node.losses(s["p_ckpt"])                   # second primal launch: sorted queue, same per-trajectory results
    assert np.array_equal(l2, lp)
    pred = node.predict_n_ode(s["p_ckpt"])
    fwd = NeuralODE(ODEProblem(PRESET_CASE2, s["tsteps"], grad_mode=1, errnorm_sens=2))   # (a context whose primal calls stay on ros23_kernel)
    fwd.set_ensemble(u0[:2048], data[:2048], cases.max_min(data, lb=LB_CASE2))
    assert np.array_equal(fwd.predict_n_ode(s["p_ckpt"]), pred[:2048])
    node.close(); fwd.close()


@pytest.mark.parametrize("case", ["case2", "rober"])
def test_finite_difference_jacobian_mode_matches_the_oracle(orc, case2_setup, rober_setup, case):
    """crnn_ctx_set_jacobian(CRNN_JAC_FINITE_DIFF): primal launches build W = I - gam J from forward differences of the right-hand
    side, as Rosenbrock23(autodiff=false) does (case2/case2.jl:26, robertson/rober_crnn_lm.jl:34; FiniteDiff's default step
    restated, [UNVERIFIED-DEP]).  HIP against the oracle's jac_fd = 1: same accepted / rejected counts; losses and predictions to
    1e-8 on case2 and 1e-5 on robertson -- the two sides' right-hand sides differ in their last bits (other log / exp), the
    1.5e-8 increments magnify that to ~1e-8 of |f| in a column of J, and robertson's W = I - gam J (gam |J| >> 1, a species at
    1e-5 below the absolute increment) passes it on: measured 4e-9 ... 8e-7 per trajectory, against the 3e-4 ... 1.3e-3 by which
    the finite-difference W moves robertson's losses away from the analytic-W ones (case2: 2e-9).  Distinguishable from the
    analytic-W solve, which gradient launches keep using."""
    from conftest import oracle_problem
    from crnn_amd import JAC_ANALYTIC, JAC_FINITE_DIFF, NeuralODE, ODEProblem, PRESET_CASE2, PRESET_ROBER
    if case == "case2":
        s = case2_setup
        node = NeuralODE(ODEProblem(PRESET_CASE2, s["tsteps"]))
        kind, ns, nr = 2, 6, 3
    else:
        s = rober_setup
        node = NeuralODE(ODEProblem(PRESET_ROBER, s["tsteps"], rate_scale=s["dydt_scale"]))
        kind, ns, nr = 3, 3, 6
    node.set_ensemble(s["u0"], s["data"], s["yscale"])
    p = s["p_ckpt"]
    th, _ = orc.p2vec(kind, ns, nr, p)
    B = s["u0"].shape[0]
    u0T = np.ascontiguousarray(s["u0"].T); dT = np.ascontiguousarray(s["data"].transpose(2, 1, 0))
    pred_an = node.predict_n_ode(p); loss_an = node.losses(p).mean()
    _, grad_an = node.loss_and_grad(p)
    node.set_jacobian(JAC_FINITE_DIFF)
    pred_fd = node.predict_n_ode(p); loss_fd = node.losses(p).mean()
    st = dict(node.last_stats)
    ref = orc.solve_batch(oracle_problem(orc, case, s, jac_fd=1), th, u0T, s["tsteps"], dT, want_pred=True)
    assert st["n_accept"] == ref["naccept"] and st["n_reject"] == ref["nreject"]
    tol = 1e-8 if case == "case2" else 1e-5
    assert abs(loss_fd - ref["loss"].mean()) < tol * abs(loss_fd)
    pr = ref["pred"].transpose(2, 1, 0)                                   # [B, n, D]
    assert pr.shape == pred_fd.shape and np.max(np.abs(pred_fd - pr)) < tol * np.max(np.abs(pr))
    d_an = abs(loss_fd - loss_an) / abs(loss_an)
    assert 1e-12 < d_an < 1e-2                                            # another W, the same method: close, not equal
    if case == "rober":
        assert d_an > 10 * tol                                            # ... and further apart than the two implementations of it
    assert np.max(np.abs(pred_fd - pred_an)) > 0
    _, grad_fd = node.loss_and_grad(p)                                    # gradient launches: the analytic-W step, unchanged
    assert np.array_equal(grad_fd, grad_an)
    node.set_jacobian(JAC_ANALYTIC)
    assert node.losses(p).mean() == loss_an
    node.close()


def test_finite_difference_jacobian_mode_is_refused_where_it_does_not_exist():
    from crnn_amd import CrnnError, JAC_FINITE_DIFF, NeuralODE, ODEProblem, PRESET_CASE2, SOLVER_TSIT5, cases
    node = NeuralODE(ODEProblem(PRESET_CASE2, cases.case2_tsteps(), solver=SOLVER_TSIT5))
    with pytest.raises(CrnnError, match="crnn_ctx_set_jacobian"):
        node.set_jacobian(JAC_FINITE_DIFF)
    node.close()

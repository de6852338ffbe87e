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

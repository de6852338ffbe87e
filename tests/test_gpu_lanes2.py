"""The two-lanes-per-trajectory Rosenbrock23 adjoint kernel (ros23_adj2_kernel.hpp; crnn_ctx_set_lanes_per_traj) against
the CPU oracle and against the one-lane kernel: same algorithm, the species sums formed in a different order.

Tolerances (floating point, written at each assert): against the oracle the bars of tests/test_gpu_parity.py -- losses 1e-9
relative, gradients 1e-7 of max |grad|, identical return codes and step counts; against the one-lane kernel 1e-10 / 1e-9
(both sum the same terms, in different orders)."""
import numpy as np

LB_CASE1, LB_CASE2 = float(np.float32(1e-5)), float(np.float32(1e-6))   # `lb = 1.f-5` / `lb = 1.f-6`: Float32 literals (case1/case1.jl:34, case2/case2.jl:34)
import pytest

from conftest import oracle_problem

pytestmark = pytest.mark.gpu


def _node(setup, lanes, **kw):
    from crnn_amd import NeuralODE, ODEProblem, PRESET_CASE2
    node = NeuralODE(ODEProblem(PRESET_CASE2, setup["tsteps"], grad_mode=2, **kw))
    node.set_ensemble(setup["u0"], setup["data"], setup["yscale"])
    node.set_lanes_per_traj(lanes)
    return node


def _per_traj(node, p):
    from crnn_amd.api import p2vec_jac
    th, dth = p2vec_jac(node.pmap, node.ns, node.nr, p)
    _, loss, grad, ret, nsv = node._solve(node._ctx, node.B, th, dth, 0, node.B, None, False)
    na, nr = node.step_counts()
    return loss, grad, ret, nsv, na, nr


@pytest.mark.parametrize("pkey", ["p_ckpt", "p_init"])
def test_case2_two_lanes_matches_oracle_and_one_lane(orc, case2_setup, pkey):
    s = case2_setup
    p = s[pkey]
    B = s["u0"].shape[0]
    th, dth = orc.p2vec(2, 6, 3, p)
    pb = oracle_problem(orc, "case2", s)
    ref = orc.solve_batch(pb, th, np.ascontiguousarray(s["u0"].T), s["tsteps"], np.ascontiguousarray(s["data"].transpose(2, 1, 0)), dtheta=dth)
    n1, n2 = _node(s, 1), _node(s, 2)
    l1, g1, r1, s1, a1, j1 = _per_traj(n1, p)
    l2, g2, r2, s2, a2, j2 = _per_traj(n2, p)
    # oracle
    assert np.array_equal(r2, ref["retcode"]) and int(a2.sum()) == ref["naccept"] and int(j2.sum()) == ref["nreject"]
    assert np.max(np.abs(l2 - ref["loss"]) / ref["loss"]) < 1e-9
    assert np.max(np.abs(g2 - ref["grad"])) < 1e-7 * np.max(np.abs(ref["grad"]))
    # one-lane kernel
    assert np.array_equal(r1, r2) and np.array_equal(s1, s2) and np.array_equal(a1, a2) and np.array_equal(j1, j2)
    assert np.max(np.abs(l1 - l2) / l1) < 1e-10
    assert np.max(np.abs(g1 - g2)) < 1e-9 * np.max(np.abs(g1))
    # the batched entry point and the per-experiment gradient take the same kernel
    lm, gm = n2.loss_and_grad(p)
    assert abs(lm - ref["loss"].mean()) < 1e-9 * ref["loss"].mean()
    assert np.max(np.abs(gm - ref["grad"] / B)) < 1e-7 * np.max(np.abs(ref["grad"] / B))
    gi = n2.gradient(p, B - 1)
    r_one = orc.solve_one(pb, th, s["u0"][B - 1], s["tsteps"], s["data"][B - 1], dtheta=dth)
    assert np.max(np.abs(gi - r_one["grad"])) < 1e-7 * np.max(np.abs(r_one["grad"]))
    n1.close(); n2.close()


def test_case1_shape_odd_species_count(orc, fx):
    """ns = 5: the second lane of a pair owns two species and a padding slot.  case1's RHS with Rosenbrock23."""
    from crnn_amd import NeuralODE, ODEProblem, PRESET_CASE1, cases
    rng = np.random.Generator(np.random.PCG64(31))
    ts = cases.case1_tsteps()
    B = 97
    u0 = cases.case1_u0(B, rng)
    p = np.array(fx["case1"]["p"])
    gen = NeuralODE(ODEProblem(PRESET_CASE1, ts, atol=1e-12, rtol=1e-10, solver=0))
    data = cases.add_noise(gen.predict_theta(u0, cases.case1_true_theta()), 0.05, rng)
    gen.close()
    ys = cases.max_min(data, lb=LB_CASE1)
    pb = orc.make_problem(ns=5, nr=4, lb=LB_CASE1, ub=10.0, atol=1e-5, rtol=1e-2, yscale=ys, clamp_pred=1, maxiters=10000, solver=0)
    th, dth = orc.p2vec(1, 5, 4, p)
    ref = orc.solve_batch(pb, th, np.ascontiguousarray(u0.T), ts, np.ascontiguousarray(data.transpose(2, 1, 0)), dtheta=dth)
    out = {}
    for lanes in (1, 2):
        node = NeuralODE(ODEProblem(PRESET_CASE1, ts, solver=0, grad_mode=2))
        node.set_ensemble(u0, data, ys)
        node.set_lanes_per_traj(lanes)
        out[lanes] = _per_traj(node, p)
        node.close()
    l2, g2, r2, _, a2, j2 = out[2]
    assert np.array_equal(r2, ref["retcode"]) and int(a2.sum()) == ref["naccept"] and int(j2.sum()) == ref["nreject"]
    assert np.max(np.abs(l2 - ref["loss"]) / ref["loss"]) < 1e-9
    assert np.max(np.abs(g2 - ref["grad"])) < 1e-7 * np.max(np.abs(ref["grad"]))
    assert np.max(np.abs(out[1][1] - g2)) < 1e-9 * np.max(np.abs(g2)) and np.array_equal(out[1][4], a2)


def test_ragged_ensemble_queue_orders_and_subranges(case2_setup):
    """1 061 trajectories (the last wavefront holds 5 pairs), sub-ranges, second launch queued by step counts: per-trajectory
    results do not depend on geometry or order; the index-order batch gradient equals the one-lane kernel's to rounding."""
    from crnn_amd import NeuralODE, ODEProblem, PRESET_CASE2, QUEUE_INDEX, cases
    rng = np.random.Generator(np.random.PCG64(5))
    ts = cases.case2_tsteps()
    B = 1061
    u0 = cases.case2_u0(B, rng)
    data = np.abs(rng.standard_normal((B, 6, len(ts)))) * 0.5
    s = dict(tsteps=ts, u0=u0, data=data, yscale=cases.max_min(data, lb=LB_CASE2))
    p = case2_setup["p_ckpt"]
    n1, n2 = _node(s, 1), _node(s, 2)
    l1, g1, r1, _, a1, _ = _per_traj(n1, p)
    l2, g2, r2, _, a2, _ = _per_traj(n2, p)           # first launch: index order
    l2b, g2b, _, _, a2b, _ = _per_traj(n2, p)          # second launch: queued by the first one's step counts
    assert np.array_equal(l2, l2b) and np.array_equal(a2, a2b)             # per-trajectory: bit-identical
    assert np.max(np.abs(g2 - g2b)) < 1e-12 * np.max(np.abs(g2))           # batch sum: another batch composition
    assert np.array_equal(a1, a2) and np.array_equal(r1, r2)
    assert np.max(np.abs(l1 - l2) / l1) < 1e-10 and np.max(np.abs(g1 - g2)) < 1e-9 * np.max(np.abs(g1))
    n2.set_queue_order(QUEUE_INDEX)
    lm, gm = n2.loss_and_grad(p, first=100, count=333)
    l1m, g1m = n1.loss_and_grad(p, first=100, count=333)
    assert abs(lm - l1m) < 1e-12 * lm and np.max(np.abs(gm - g1m)) < 1e-9 * np.max(np.abs(g1m))
    assert abs(lm - l2[100:433].mean()) < 1e-13 * lm
    n1.close(); n2.close()


def test_truncated_failed_and_tape_overflow(case2_setup):
    """maxiters truncation (retcode 1, short solutions) and a tape too small (forward-tangent fallback) behave as in the
    one-lane kernel."""
    s = case2_setup
    p = s["p_ckpt"]
    n1, n2 = _node(s, 1, maxiters=12), _node(s, 2, maxiters=12)
    l1, g1, r1, s1, a1, _ = _per_traj(n1, p)
    l2, g2, r2, s2, a2, _ = _per_traj(n2, p)
    assert np.all(r2 == 1) and np.array_equal(r1, r2) and np.array_equal(s1, s2) and np.array_equal(a1, a2)
    assert np.max(np.abs(l1 - l2) / l1) < 1e-10 and np.max(np.abs(g1 - g2)) < 1e-9 * np.max(np.abs(g1))
    n1.close(); n2.close()
    full, tiny = _node(s, 2), _node(s, 2, tape_steps=6)
    lf, gf = full.loss_and_grad(p)
    lt, gt = tiny.loss_and_grad(p)                     # every trajectory outruns 6 records: repeated with forward tangents
    assert abs(lf - lt) < 1e-9 * lf and np.max(np.abs(gf - gt)) < 1e-8 * np.max(np.abs(gf))
    full.close(); tiny.close()


def test_training_steps_agree_and_auto_picks_two_lanes_for_small_shards(case2_setup):
    from crnn_amd import NeuralODE, ODEProblem, Optimiser, PRESET_CASE2
    s = case2_setup
    P = 25
    ps = {}
    for lanes in (1, 2, 0):
        node = NeuralODE(ODEProblem(PRESET_CASE2, s["tsteps"], grad_mode=2))
        node.set_ensemble(s["u0"], s["data"], s["yscale"])
        node.set_lanes_per_traj(lanes)
        node.train_init(Optimiser(P, PRESET_CASE2), s["p_init"])
        for _ in range(8):
            node.train_step(want_loss=False)
        ps[lanes] = node.params()
        assert node.last_lanes_per_traj() == (lanes if lanes else 2)
        node.close()
    assert np.max(np.abs(ps[1] - ps[2])) < 1e-9
    assert np.array_equal(ps[0], ps[2])                # AUTO: 8 trajectories fit the resident pairs -> the two-lane kernel
    with pytest.raises(Exception):
        from crnn_amd import PRESET_ROBER
        nr = NeuralODE(ODEProblem(PRESET_ROBER, np.linspace(1.0, 2.0, 4), rate_scale=np.ones(3)))
        try:
            nr.set_lanes_per_traj(2)                   # robertson (ns < nr, scaled): no two-lane instantiation
        finally:
            nr.close()


def test_auto_beyond_one_generation_follows_the_step_count_spread():
    """65 536 trajectories = two generations of lane pairs: lanes_per_traj = AUTO decides from the spread of the step counts of the
    launch before the previous one over the same range (crnn_capi.hip: launch_adjoint / step_spread_block) -- pairs where the
    counts spread (the reference's trained p: longest 48 against a median of 29), single lanes where they do not (its initialiser:
    nearly uniform counts).  The choice is a function of the run's own history: two runs are bit-identical; whichever kernel
    runs, the results are the forced kernels' (per-trajectory losses 1e-10 against the other kernel, bit-identical to its own)."""
    import json
    import os
    from crnn_amd import NeuralODE, ODEProblem, PRESET_CASE2, cases
    B = 65536
    rng = np.random.Generator(np.random.PCG64(1234))
    ts = cases.case2_tsteps()
    u0 = cases.case2_u0(B, rng)
    gen = NeuralODE(ODEProblem(PRESET_CASE2, ts, atol=1e-10, rtol=1e-8))
    clean = gen.predict_theta(u0, cases.case2_true_theta())[:, :6, :]
    gen.close()
    data = cases.add_noise(clean, 0.05, rng)
    s = dict(tsteps=ts, u0=u0, data=data, yscale=cases.max_min(data, lb=LB_CASE2))
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fixtures.json")))
    p_ckpt = np.array(fx["case2_ckpt"]["p"])
    p_init = cases.case2_init_p(np.random.Generator(np.random.PCG64(7)))
    for p, want in ((p_ckpt, 2), (p_init, 1)):
        runs = []
        for _ in range(2):
            node = _node(s, 0)
            seq = []
            for _ in range(4):
                l, g = node.loss_and_grad(p)
                seq.append((node.last_lanes_per_traj(), l, g.copy()))
            na, nr = node.step_counts()
            runs.append(seq)
            node.close()
        steps = na + nr
        print(f"longest {steps.max()} median {int(np.median(steps))} -> lanes {[q[0] for q in runs[0]]}")
        assert [q[0] for q in runs[0]] == [1, 1, want, want]            # no history for the first two launches
        for a, b in zip(runs[0], runs[1]):                              # deterministic: a function of the run's own history
            assert a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2], b[2])
        forced = _node(s, want)
        forced.loss_and_grad(p); forced.loss_and_grad(p)
        lf, gf = forced.loss_and_grad(p)                                 # third launch of the forced kernel: the same queue history
        assert runs[0][2][1] == pytest.approx(lf, rel=1e-12) and np.max(np.abs(runs[0][2][2] - gf)) < 1e-10 * np.max(np.abs(gf))
        forced.close()

"""GPU tests at BASELINE.json's full sizes (65 536 initial conditions) through
size-independent properties, plus edge cases of the boundary.  The oracle cannot
integrate 65 536 trajectories in seconds, so at full size it checks a random sample
and the rest is covered by determinism / additivity / permutation / variant
invariance."""
import ctypes as C

import numpy as np

LB_CASE1, LB_CASE2 = float(np.float32(1e-5)), float(np.float32(1e-6))   # `lb = 1.f-5` / `lb = 1.f-6`: Float32 literals (case1/case1.jl:34, case2/case2.jl:34)
import pytest

from conftest import INV_R, oracle_problem
from conftest import emulated as _emulated

pytestmark = pytest.mark.gpu
B_DEVICE = 65536


def _bfull():
    """BASELINE's 65 536 on the device; 1 024 (two generations of the emulated device's resident lanes) under SIMT emulation"""
    return 1024 if _emulated() else B_DEVICE



def _case2_ensemble(B, seed=1234):
    from crnn_amd import NeuralODE, ODEProblem, PRESET_CASE2, cases
    rng = np.random.Generator(np.random.PCG64(seed))
    ts = cases.case2_tsteps()
    u0 = cases.case2_u0(B, rng)
    gen = NeuralODE(ODEProblem(PRESET_CASE2, ts, atol=1e-10, rtol=1e-8))
    clean = gen.predict_theta(u0, cases.case2_true_theta())[:, :6, :]
    assert np.all(gen.last_retcode == 0)
    gen.close()
    data = cases.add_noise(clean, 0.05, rng)
    return ts, u0, data, cases.max_min(data, lb=LB_CASE2)


@pytest.fixture(scope="module")
def full_case2(fx):
    ts, u0, data, ys = _case2_ensemble(_bfull())
    return dict(tsteps=ts, u0=u0, data=data, yscale=ys, p=np.array(fx["case2_ckpt"]["p"]), p_init=np.array(fx["case2"]["p_init"]))


def _node(s, cols=0, **kw):
    from crnn_amd import NeuralODE, ODEProblem, PRESET_CASE2
    node = NeuralODE(ODEProblem(PRESET_CASE2, s["tsteps"], cols_per_lane=cols, **kw))
    node.set_ensemble(s["u0"], s["data"], s["yscale"])
    return node


def test_case2_full_batch_properties(orc, full_case2):
    s = full_case2
    p = s["p"]
    node = _node(s)
    loss, grad = node.loss_and_grad(p)
    st = node.last_stats
    assert st["n_traj"] == _bfull() and st["n_ok"] == _bfull()
    assert 20 < st["n_accept"] / _bfull() < 45 and st["n_reject"] < 0.05 * st["n_accept"]
    # the learned checkpoint explains data made from the true mechanism + 5 % noise: MAE of the order of the noise
    assert 0.005 < loss < 0.05
    # determinism: the second launch is queued by the first one's step counts (other 64-trajectory partial sums: equal to
    # rounding), from then on bitwise identical on repetition; in index order bitwise identical to the first launch
    loss2, grad2 = node.loss_and_grad(p)
    assert abs(loss2 - loss) < 1e-13 * loss and np.max(np.abs(grad2 - grad)) < 1e-11 * np.max(np.abs(grad))
    # (round 4) from the third launch on lanes_per_traj = AUTO has the step-count spread of launch 1 and, at this trained p, gives every
    # trajectory a lane pair (tests/test_gpu_lanes2.py): other partial sums once more, then bitwise identical on repetition
    loss3, grad3 = node.loss_and_grad(p)
    assert node.last_lanes_per_traj() == 2 or _emulated()      # (the rule is sized for 256 CUs' resident lanes: tests/test_gpu_lanes2.py)
    assert abs(loss3 - loss2) < 1e-13 * loss and np.max(np.abs(grad3 - grad2)) < 1e-10 * np.max(np.abs(grad))
    loss3b, grad3b = node.loss_and_grad(p)
    assert loss3b == loss3 and np.array_equal(grad3b, grad3)
    from crnn_amd import QUEUE_AUTO, QUEUE_INDEX
    node.set_queue_order(QUEUE_INDEX)
    loss4, grad4 = node.loss_and_grad(p)
    assert loss4 == loss and np.array_equal(grad4, grad)
    node.set_queue_order(QUEUE_AUTO)
    # additivity over sub-ranges (sums of the same per-trajectory terms, different association)
    acc_l, acc_g = 0.0, np.zeros(25)
    q = _bfull() // 4
    for k in range(4):
        l_k, g_k = node.loss_and_grad(p, first=k * q, count=q)
        acc_l += l_k * q
        acc_g += g_k * q
    assert abs(acc_l / _bfull() - loss) < 1e-13 * loss
    assert np.max(np.abs(acc_g / _bfull() - grad)) < 1e-12 * np.max(np.abs(grad))
    # per-IC losses: mean equals the batched loss
    losses = node.losses(p)
    assert abs(losses.mean() - loss) < 1e-13 * loss
    # random sample against the oracle (reference tolerances, same inputs)
    rng = np.random.default_rng(5)
    idx = rng.choice(_bfull(), 96, replace=False)
    th, dth = orc.p2vec(2, 6, 3, p)
    pb = oracle_problem(orc, "case2", s)
    for i in idx[:96]:
        r = orc.solve_one(pb, th, s["u0"][i], s["tsteps"], s["data"][i], dtheta=None, want_pred=False)
        assert abs(losses[i] - r["loss"]) < 1e-9 * r["loss"]
    for i in idx[:8]:
        r = orc.solve_one(pb, th, s["u0"][i], s["tsteps"], s["data"][i], dtheta=dth, want_pred=False)
        g = node.gradient(p, int(i))
        assert np.max(np.abs(g - r["grad"])) < 1e-7 * np.max(np.abs(r["grad"]))
    node.close()
    # permutation of the ensemble permutes the per-IC results exactly and leaves the sums unchanged to rounding
    perm = np.random.default_rng(9).permutation(_bfull())
    s2 = dict(s, u0=s["u0"][perm], data=s["data"][perm])
    node2 = _node(s2)
    losses2 = node2.losses(p)
    assert np.array_equal(losses2, losses[perm])
    lossp, gradp = node2.loss_and_grad(p)
    assert abs(lossp - loss) < 1e-13 * loss and np.max(np.abs(gradp - grad)) < 1e-11 * np.max(np.abs(grad))
    node2.close()
    # a different lanes-per-trajectory variant performs the same primal arithmetic (bitwise) and the same tangents
    node3 = _node(s, cols=5)
    assert np.array_equal(node3.losses(p), losses)
    loss3, grad3 = node3.loss_and_grad(p)
    assert np.max(np.abs(grad3 - grad)) < 1e-11 * np.max(np.abs(grad))
    node3.close()


def test_case2_full_batch_training_decreases_loss(full_case2):
    from crnn_amd import Optimiser, PRESET_CASE2
    s = full_case2
    node = _node(s)
    node.train_init(Optimiser(25, PRESET_CASE2), s["p_init"])
    l0 = node.train_step()
    for _ in range(30):
        l = node.train_step()
    assert np.isfinite(l) and l < 0.9 * l0
    assert node.stats()["n_ok"] == _bfull()
    node.close()


def test_robertson_full_batch(orc, fx):
    """BASELINE config 3: robertson, 65 536 ICs, stiff: every trajectory succeeds, sample matches the oracle."""
    from crnn_amd import NeuralODE, ODEProblem, PRESET_ROBER, cases
    rng = np.random.Generator(np.random.PCG64(77))
    B = _bfull()
    ts = cases.rober_tsteps()
    u0 = cases.rober_u0(B, rng)
    th3 = cases.rober_true_theta()
    w_in = np.zeros((3, 6)); w_b = np.full(6, -700.0); w_out = np.zeros((3, 6))
    w_in[:, :3] = th3[:9].reshape((3, 3), order="F"); w_b[:3] = th3[9:12]; w_out[:, :3] = th3[12:].reshape((3, 3), order="F")
    gen = NeuralODE(ODEProblem(PRESET_ROBER, ts, atol=1e-12, rtol=1e-7, maxiters=10**6, lb=1e-300))
    clean = gen.predict_theta(u0, cases.pack_theta(w_in, w_b, w_out))
    assert np.all(gen.last_retcode == 0)
    gen.close()
    # mass conservation of the true mechanism: y1 + y2 + y3 is invariant (size-independent property)
    tot0 = u0.sum(axis=1)
    assert np.max(np.abs(clean.sum(axis=1) - tot0[:, None])) < 1e-6
    data = cases.add_noise(clean, 1e-4, rng)
    ys = cases.max_min(data)
    dydt = ys / ts[-1]
    p = np.array(fx["rober_ckpt"]["p"])
    node = NeuralODE(ODEProblem(PRESET_ROBER, ts, rate_scale=dydt))
    node.set_ensemble(u0, data, ys)
    loss, grad = node.loss_and_grad(p)
    st = node.last_stats
    assert st["n_ok"] == B and np.isfinite(loss) and np.all(np.isfinite(grad))
    losses = node.losses(p)
    th, dth = orc.p2vec(3, 3, 6, p)
    pb = orc.make_problem(ns=3, nr=6, lb=1e-8, atol=[1e-6, 1e-8, 1e-6], rtol=1e-3, yscale=ys, rate_scale=dydt, maxiters=10000)
    idx = np.random.default_rng(3).choice(B, 48, replace=False)
    for i in idx:
        r = orc.solve_one(pb, th, u0[i], ts, data[i], want_pred=False)
        assert r["retcode"] == 0
        # stiff steps amplify last-bit differences (exp/log implementations, fused multiply-adds)
        assert abs(losses[i] - r["loss"]) < 1e-7 * r["loss"]
    for i in idx[:4]:
        r = orc.solve_one(pb, th, u0[i], ts, data[i], dtheta=dth, want_pred=False)
        g = node.gradient(p, int(i))
        assert np.max(np.abs(g - r["grad"])) < 1e-6 * np.max(np.abs(r["grad"]))
    node.close()


# ------------------------------------------------------------------ edge cases
def test_tiny_and_ragged_batches(orc, case2_setup):
    s = case2_setup
    p = s["p_ckpt"]
    th, dth = orc.p2vec(2, 6, 3, p)
    pb = oracle_problem(orc, "case2", s)
    ref = [orc.solve_one(pb, th, s["u0"][i], s["tsteps"], s["data"][i], dtheta=dth) for i in range(8)]
    for B in (1, 3, 5):
        sub = dict(s, u0=s["u0"][:B], data=s["data"][:B])
        node = _node(sub)
        loss, grad = node.loss_and_grad(p)
        assert abs(loss - np.mean([r["loss"] for r in ref[:B]])) < 1e-9 * loss
        g = np.mean([r["grad"] for r in ref[:B]], axis=0)
        assert np.max(np.abs(grad - g)) < 1e-7 * np.max(np.abs(g))
        # per-trajectory step counts (`sol.destats`): those of the oracle's solves, whichever kernel produced them
        na, nr = node.step_counts()
        assert list(na) == [r["naccept"] for r in ref[:B]] and list(nr) == [r["nreject"] for r in ref[:B]]
        assert na.sum() == node.last_stats["n_accept"] and nr.sum() == node.last_stats["n_reject"]
        # last element alone
        l1, g1 = node.loss_and_grad(p, first=B - 1, count=1)
        assert abs(l1 - ref[B - 1]["loss"]) < 1e-9 * l1
        na1, _ = node.step_counts(first=B - 1, count=1)
        assert na1[0] == ref[B - 1]["naccept"]
        if B > 1:
            with pytest.raises(Exception):
                node.step_counts(first=0, count=B)        # outside the range of the most recent solve
        node.close()


def test_more_trajectories_than_lanes(case2_setup):
    """150 001 trajectories (> 65 536 resident lanes, not a multiple of 64): wavefronts take several batches from the queue,
    the last one is ragged.  The 8 base conditions are tiled, so every row must equal the small launch's row, the batch
    gradient must equal the small gradient (both modes), and a sub-range must work."""
    from crnn_amd import p2vec_jac
    s = case2_setup
    p = s["p_ckpt"]
    B, sub0, subn = (1601, 701, 345) if _emulated() else (150001, 70001, 12345)
    rep = -(-B // 8)
    big = dict(s, u0=np.tile(s["u0"], (rep, 1))[:B], data=np.tile(s["data"], (rep, 1, 1))[:B])
    for mode, lanes in ((2, 1), (2, 2), (1, 0)):   # adjoint with one / two lanes per trajectory (same kernel for both sizes), forward tangents
        node, small = _node(big, grad_mode=mode), _node(s, grad_mode=mode)
        if lanes:
            node.set_lanes_per_traj(lanes); small.set_lanes_per_traj(lanes)
        th, dth = p2vec_jac(node.pmap, 6, 3, p)
        _, loss, gsum, ret, nsv = node._solve(node._ctx, B, th, dth, 0, B, None, False)
        _, ls, gs, _, _ = small._solve(small._ctx, 8, th, dth, 0, 8, None, False)
        assert np.all(ret == 0) and np.all(nsv == 50)
        assert np.array_equal(loss, np.tile(ls, rep)[:B])
        w = np.bincount(np.arange(B) % 8, minlength=8)                  # how often each base condition occurs
        assert node.last_stats["n_traj"] == B
        # per-condition gradients from one-trajectory launches of the small node, weighted
        gref = sum(w[i] * small._solve(small._ctx, 8, th, dth, i, 1, None, False)[2] for i in range(8))
        assert np.max(np.abs(gsum - gref)) < 1e-9 * np.max(np.abs(gref))
        _, lsub, gsub, _, _ = node._solve(node._ctx, B, th, dth, sub0, subn, None, False)
        assert np.array_equal(lsub[sub0:sub0 + subn], loss[sub0:sub0 + subn])
        node.close(); small.close()


def test_observation_mask_and_mse(orc, case2_setup):
    """i_obs = [1,2,4,5,6] as in case2_missing.jl:165 (0-based here) and the MSE loss kind."""
    from crnn_amd import NeuralODE, ODEProblem, PRESET_CASE2
    s = case2_setup
    p = s["p_ckpt"]
    i_obs = [0, 1, 3, 4, 5]
    data = s["data"][:, i_obs, :]
    ys = s["yscale"][i_obs]
    node = NeuralODE(ODEProblem(PRESET_CASE2, s["tsteps"]))
    node.set_ensemble(s["u0"], data, ys, i_obs=i_obs)
    loss, grad = node.loss_and_grad(p)
    th, dth = orc.p2vec(2, 6, 3, p)
    pb = orc.make_problem(ns=6, nr=3, has_temp=1, lb=LB_CASE2, ub=10.0, inv_R=INV_R, atol=1e-6, rtol=1e-3, yscale=ys,
                          i_obs=i_obs, clamp_pred=1)
    rs = [orc.solve_one(pb, th, s["u0"][i], s["tsteps"], data[i], dtheta=dth) for i in range(8)]
    assert abs(loss - np.mean([r["loss"] for r in rs])) < 1e-9 * loss
    g = np.mean([r["grad"] for r in rs], axis=0)
    assert np.max(np.abs(grad - g)) < 1e-7 * np.max(np.abs(g))
    node.close()
    # MSE
    from crnn_amd import LOSS_MSE
    node = NeuralODE(ODEProblem(PRESET_CASE2, s["tsteps"], loss_kind=LOSS_MSE))
    node.set_ensemble(s["u0"], s["data"], s["yscale"])
    loss, grad = node.loss_and_grad(p)
    pb = orc.make_problem(ns=6, nr=3, has_temp=1, lb=LB_CASE2, ub=10.0, inv_R=INV_R, atol=1e-6, rtol=1e-3,
                          yscale=s["yscale"], clamp_pred=1, loss_kind=1)
    rs = [orc.solve_one(pb, th, s["u0"][i], s["tsteps"], s["data"][i], dtheta=dth) for i in range(8)]
    assert abs(loss - np.mean([r["loss"] for r in rs])) < 1e-9 * loss
    g = np.mean([r["grad"] for r in rs], axis=0)
    assert np.max(np.abs(grad - g)) < 1e-7 * np.max(np.abs(g))
    node.close()


def test_non_finite_and_dtmin_failures_are_per_trajectory(case2_setup):
    from crnn_amd import NeuralODE, ODEProblem, PRESET_CASE2, RET_DTMIN, RET_UNSTABLE
    s = case2_setup
    p = s["p_ckpt"]
    u0 = s["u0"].copy()
    u0[2, 0] = np.nan
    node = NeuralODE(ODEProblem(PRESET_CASE2, s["tsteps"]))
    node.set_ensemble(u0, s["data"], s["yscale"])
    losses = node.losses(p)
    assert node.last_retcode[2] == RET_UNSTABLE and np.all(np.delete(node.last_retcode, 2) == 0)
    assert node.last_n_saved[2] == 1          # only the save_start column exists
    loss, grad = node.loss_and_grad(p)
    assert node.last_stats["n_ok"] == 7
    assert np.isnan(losses[2]) and np.isnan(loss) and np.all(np.isfinite(np.delete(losses, 2)))
    # the failed trajectory's gradient row is zero (its tangents never advanced); the others are unaffected
    assert np.all(node.gradient(p, 2) == 0.0)
    g_ok = sum(node.gradient(p, i) for i in range(8) if i != 2)
    assert np.max(np.abs(g_ok / 8 - grad)) < 1e-12 * np.max(np.abs(grad))
    node.close()
    # dtmin larger than any step the controller wants: DtLessThanMin
    node = NeuralODE(ODEProblem(PRESET_CASE2, s["tsteps"], dtmin=1e3))
    node.set_ensemble(s["u0"], s["data"], s["yscale"])
    node.losses(p)
    assert np.all(node.last_retcode == RET_DTMIN)
    node.close()


def test_zero_length_horizon_and_argument_errors(case2_setup):
    from crnn_amd import CrnnError, NeuralODE, ODEProblem, PRESET_CASE2
    s = case2_setup
    p = s["p_ckpt"]
    node = _node(s)
    # sample = 1: the horizon is tspan[1] itself -> only the save_start column, success
    pred = node.predict_neuralode(s["u0"][0], p, sample=1)
    assert pred.shape == (7, 1) and np.all(node.last_retcode == 0)
    assert np.allclose(pred[:6, 0], np.clip(s["u0"][0][:6], -10, 10))
    with pytest.raises(CrnnError, match="n_save_active"):
        node.loss_and_grad(p, sample=51)
    with pytest.raises(CrnnError, match="outside the ensemble"):
        node.loss_and_grad(p, first=4, count=5)
    with pytest.raises(CrnnError, match="strictly increasing"):
        bad = NeuralODE(ODEProblem(PRESET_CASE2, s["tsteps"][::-1].copy()))
        bad.set_ensemble(s["u0"], s["data"], s["yscale"])
    node.close()


def test_theta_level_directions(orc, case2_setup):
    """crnn_solve with an arbitrary direction matrix: identity in theta space (42 directions, the (7,6) variant)."""
    from crnn_amd import NeuralODE, ODEProblem, PRESET_CASE2
    s = case2_setup
    th, dth = orc.p2vec(2, 6, 3, s["p_ckpt"])
    node = _node(s)
    eye = np.eye(42, order="F")
    _, _, g_theta, _, _ = node._solve(node._ctx, 8, th, eye, 0, 8, None, False)
    _, _, g_p, _, _ = node._solve(node._ctx, 8, th, dth, 0, 8, None, False)
    # chain rule on the host: grad_p = (d theta / d p)^T grad_theta
    assert np.max(np.abs(dth.T @ g_theta - g_p)) < 1e-10 * np.max(np.abs(g_p))
    node.close()


def test_two_gpu_data_parallel_bench_when_available():
    """The N > 1 path end to end (one process per GPU, RCCL all-reduce of the 31-double vector per step): only on a node
    that shows at least two GPUs -- the boxes gpurun hands out have one, where this self-skips and the N > 1 logic is
    covered by the gloo tests (tests/test_dist_gloo.py) and the 1-rank RCCL test."""
    import json
    import os
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    run = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29541", os.path.join(root, "bench.py"),
                          "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "8192"],
                         capture_output=True, text=True, env=env, timeout=600)
    assert run.returncode == 0, run.stderr[-3000:]
    line = json.loads(run.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 16384 and line["value"] > 0
    assert line["config"]["comm"] in ("rccl", "torch")


def test_ensemble_larger_than_the_resident_lanes_is_queued_by_step_count(orc, fx):
    """B = 3 x 65 536 (three generations of wavefronts on a 256-CU part): from the second gradient launch over the same
    range on, the queue is ordered by the previous launch's step counts (sort_steps_kernel: stable, longest first), so every
    64-trajectory batch is homogeneous.  The order changes which trajectories share a batch sum, nothing else: per-
    trajectory results are bit-identical, the batch gradient equal to rounding, and -- the order being a deterministic
    function of the previous launch -- the sorted launch itself is bitwise reproducible."""
    B = 3 * 576 if _emulated() else 3 * _bfull()
    ts, u0, data, ys = _case2_ensemble(B, seed=77)
    s = dict(tsteps=ts, u0=u0, data=data, yscale=ys)
    p = np.array(fx["case2_ckpt"]["p"])
    node = _node(s)
    loss1, grad1 = node.loss_and_grad(p)            # no step counts known yet: index order
    ms1 = node.last_stats["kernel_ms"]
    st1 = dict(node.last_stats)
    loss2, grad2 = node.loss_and_grad(p)            # queued by the counts of launch 1
    ms2 = node.last_stats["kernel_ms"]
    loss3, grad3 = node.loss_and_grad(p)
    assert node.last_stats["n_accept"] == st1["n_accept"] and node.last_stats["n_ok"] == B
    assert abs(loss2 - loss1) < 1e-13 * loss1 and np.max(np.abs(grad2 - grad1)) < 1e-11 * np.max(np.abs(grad1))
    assert loss3 == loss2 and np.array_equal(grad3, grad2)
    losses = node.losses(p)
    assert abs(losses.mean() - loss2) < 1e-13 * loss2
    idx = np.random.default_rng(3).choice(B, 24, replace=False)
    th, dth = orc.p2vec(2, 6, 3, p)
    pb = oracle_problem(orc, "case2", s)
    for i in idx:
        r = orc.solve_one(pb, th, u0[i], ts, data[i], dtheta=None, want_pred=False)
        assert abs(losses[i] - r["loss"]) < 1e-9 * r["loss"]
    print(f"B = {B}: kernel {ms1:.3f} ms in index order, {ms2:.3f} ms queued by step count")
    assert ms2 < ms1 or _emulated()                 # homogeneous batches are the point (no clock under emulation)
    node.close()


def test_queue_order_auto_and_index_below_the_resident_lanes(fx):
    """B = 8 192 (fits the resident lanes).  QUEUE_AUTO: from the second launch on the queue follows the previous launch's
    step counts -- per-trajectory results and step counts bit-identical, batch sums equal to rounding, and faster (the
    steps of an iteration hold about the same number of save points).  QUEUE_INDEX: every launch in index order, batch
    sums bit-identical from launch to launch whatever ran before."""
    from crnn_amd import QUEUE_AUTO, QUEUE_INDEX
    B = 256 if _emulated() else 8192
    ts, u0, data, ys = _case2_ensemble(B, seed=11)
    s = dict(tsteps=ts, u0=u0, data=data, yscale=ys)
    p = np.array(fx["case2_ckpt"]["p"])
    node = _node(s)
    l1, g1 = node.loss_and_grad(p)                  # no counts known yet: index order
    na1, nr1 = node.step_counts()
    ms1 = node.last_stats["kernel_ms"]
    l2, g2 = node.loss_and_grad(p)                  # by the counts of launch 1
    na2, nr2 = node.step_counts()
    ms2 = min(node.last_stats["kernel_ms"], *(node.loss_and_grad(p) and node.last_stats["kernel_ms"] for _ in range(3)))
    assert np.array_equal(na1, na2) and np.array_equal(nr1, nr2)
    assert abs(l2 - l1) < 1e-13 * l1 and np.max(np.abs(g2 - g1)) < 1e-11 * np.max(np.abs(g1))
    losses_auto = node.losses(p)
    print(f"B = {B}: kernel {ms1:.3f} ms in index order, {ms2:.3f} ms queued by step count")
    assert ms2 < ms1 or _emulated()
    node.set_queue_order(QUEUE_INDEX)
    l3, g3 = node.loss_and_grad(p)
    assert l3 == l1 and np.array_equal(g3, g1)      # the first launch was in index order too
    node.loss_and_grad(p * 1.001)                   # some other launch in between
    l4, g4 = node.loss_and_grad(p)
    assert l4 == l1 and np.array_equal(g4, g1)
    assert np.array_equal(node.losses(p), losses_auto)
    node.set_queue_order(QUEUE_AUTO)
    l5, g5 = node.loss_and_grad(p)                  # counts of the launch before: the same p, the same order as launch 2
    assert l5 == l2 and np.array_equal(g5, g2)
    with pytest.raises(Exception):
        node.set_queue_order(7)
    node.close()


def test_step_count_queue_with_a_ragged_ensemble_size(fx):
    """B = 70 001: one trajectory more than a multiple of 64 and a last sort run of 369 -- the position map of the queue
    (interleaved runs, partial last batch) must stay a permutation: every trajectory integrated exactly once."""
    B, f0_, c0_ = (1393, 100, 1200) if _emulated() else (70001, 1000, 68000)      # (1 393 = one sort run of 1 024 + a last one of 369)
    ts, u0, data, ys = _case2_ensemble(B, seed=5)
    s = dict(tsteps=ts, u0=u0, data=data, yscale=ys)
    p = np.array(fx["case2_ckpt"]["p"])
    node = _node(s)
    l1, g1 = node.loss_and_grad(p)
    n1 = dict(node.last_stats)
    l2, g2 = node.loss_and_grad(p)              # sorted queue
    n2 = dict(node.last_stats)
    assert n1["n_traj"] == n2["n_traj"] == B and n2["n_ok"] == B and n1["n_accept"] == n2["n_accept"]
    assert abs(l2 - l1) < 1e-13 * l1 and np.max(np.abs(g2 - g1)) < 1e-11 * np.max(np.abs(g1))
    # a sub-range launch after a full one: its own queue order over [first, first + count)
    l3, g3 = node.loss_and_grad(p, first=f0_, count=c0_)
    node2 = _node(dict(s, u0=u0[f0_:f0_ + c0_], data=data[f0_:f0_ + c0_]))
    l4, g4 = node2.loss_and_grad(p)
    assert abs(l3 - l4) < 1e-13 * l4 and np.max(np.abs(g3 - g4)) < 1e-11 * np.max(np.abs(g4))
    node.close(); node2.close()

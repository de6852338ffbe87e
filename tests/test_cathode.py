"""Cathode-UQ row (BASELINE config 5): the C oracle pinned against the golden vectors (Radau + forward
sensitivities of the reference's crnn!/HRR_getter restated in NumPy, reference data CSVs reduced to their
replica statistics), the host mirror's CPU definitions, and -- on the GPU -- the HIP kernel against both."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def cfx():
    with open(os.path.join(HERE, "golden", "fixtures_cathode.json")) as f:
        return json.load(f)


def _two_replicas(s):
    """exp_data [D, 1+2] with the same replica mean and mean-square as the reference's 100 replicas."""
    dbar, d2bar = np.array(s["dbar"]), np.array(s["d2bar"])
    sd = np.sqrt(np.maximum(d2bar - dbar ** 2, 0.0))
    return np.stack([np.array(s["ts"]), dbar + sd, dbar - sd], axis=1)


# ------------------------------------------------------------------ CPU: oracle vs golden
def test_oracle_cathode_matches_golden_at_tight_tolerance(orc, cfx):
    th = np.array(cfx["theta"])
    for s in cfx["sets"]:
        c = orc.make_cathode(s["beta"], atol=1e-14, rtol=1e-9)
        r = orc.cathode_solve_one(c, th, s["ts"], s["dbar"], s["d2bar"])
        assert r["retcode"] == 0 and r["n_saved"] == len(s["ts"])
        hg, gg = np.array(s["hrr"]), np.array(s["grad"])
        assert np.max(np.abs(r["hrr"] - hg)) < 2e-8 * max(1.0, np.max(np.abs(hg)))     # measured 5e-10
        assert abs(r["loss"] - s["loss"]) < 1e-8 * abs(s["loss"])
        assert np.max(np.abs(r["grad"] - gg)) < 2e-8 * np.max(np.abs(gg))               # measured 3e-10


def test_oracle_cathode_reference_tolerances_track_golden(orc, cfx):
    """At the reference's own tolerances (abstol 1e-12, reltol 1e-3) Rosenbrock23 stays within 1e-2 of converged."""
    th = np.array(cfx["theta"])
    s = cfx["sets"][2]
    c = orc.make_cathode(s["beta"])
    r = orc.cathode_solve_one(c, th, s["ts"], s["dbar"], s["d2bar"])
    assert r["retcode"] == 0
    assert np.max(np.abs(r["hrr"] - np.array(s["hrr"]))) < 1e-2 * np.max(np.abs(s["hrr"]))
    assert abs(r["loss"] - s["loss"]) < 5e-2 * abs(s["loss"])


def test_oracle_cathode_grad_by_central_differences(orc, cfx):
    th = np.array(cfx["theta"])
    s = cfx["sets"][1]
    c = orc.make_cathode(s["beta"], atol=1e-14, rtol=1e-10)
    g = orc.cathode_solve_one(c, th, s["ts"], s["dbar"], s["d2bar"])["grad"]
    for k in (0, 4, 7, 10, 13, 15):
        h = 1e-5 * max(1.0, abs(th[k]))
        tp, tm = th.copy(), th.copy()
        tp[k] += h; tm[k] -= h
        fd = (orc.cathode_solve_one(c, tp, s["ts"], s["dbar"], s["d2bar"], want_grad=False)["loss"]
              - orc.cathode_solve_one(c, tm, s["ts"], s["dbar"], s["d2bar"], want_grad=False)["loss"]) / (2 * h)
        assert abs(fd - g[k]) < 1e-5 * np.max(np.abs(g)) + 1e-4 * abs(g[k]), (k, fd, g[k])


def test_host_crnn_and_hrr_match_oracle_rhs(orc, cfx):
    """cathode.crnn / HRR_getter (the CPU definitions of network.jl:152-175) vs the oracle's RHS."""
    # crnn_amd loads the HIP library at import time; that is fine on CPU (only compute calls need a GPU)
    from crnn_amd import cathode as ch
    rng = np.random.default_rng(5)
    p_scales = np.array(cfx["theta"])
    for beta in (2.0, 20.0):
        c = orc.make_cathode(beta)
        for _ in range(5):
            p = 1 + 0.1 * rng.standard_normal(17)
            u = np.abs(rng.standard_normal(3)) * np.array([1, .5, .2])
            u[rng.integers(0, 3)] = 0.0          # exercises the lower clamp
            t = float(rng.uniform(0, 3000))
            du = ch.crnn(np.zeros(3), u, ch.p2vec(p), t, p_scales=p_scales, beta=beta)
            ref = orc.cathode_rhs(c, p * p_scales, u, t)
            assert np.max(np.abs(du - ref)) <= 1e-12 * max(1.0, np.max(np.abs(ref)))


def test_svgd_oracle_matches_reference_formulas(orc):
    """The oracle's SVGD move (oracle/crnn_oracle.c: orc_svgd_update) against network.jl:67-87 + crnn_cathode.jl:36-50
    written out independently in NumPy; odd and even pair counts (Julia's median of an even-length vector is the mean of
    the middle pair)."""
    for N, dim, seed in ((9, 17, 1), (6, 17, 2), (40, 3, 3)):
        rng = np.random.default_rng(seed)
        p = 1 + 0.1 * rng.standard_normal((N, dim)); g = rng.standard_normal((N, dim))
        d = np.array([[np.sqrt(np.sum((p[i] - p[j]) ** 2)) for j in range(N)] for i in range(N)])
        h = np.sqrt(0.5 * np.median([d[i, j] for i in range(N) for j in range(i)]) ** 2 / np.log(N + 1))
        K = np.exp(-d ** 2 / h ** 2 / 2)
        assert np.allclose(K, K.T) and np.allclose(np.diag(K), 1.0)
        rep = np.array([[(-K[i] @ p[:, k] + p[i, k] * K[i].sum()) / h ** 2 for k in range(dim)] for i in range(N)])
        pn, dt, rp, hh = orc.svgd_update(p, g, 0.02)
        assert abs(hh - h) <= 1e-15 * h
        assert np.allclose(dt, K @ g, rtol=1e-13, atol=1e-14) and np.allclose(rp, rep, rtol=1e-12, atol=1e-13)
        assert np.allclose(rp.sum(axis=0), 0.0, atol=1e-11 * np.abs(rp).max())      # the repulsion sums to zero over particles
        assert np.allclose(pn, p + 0.02 * (K @ g + rep) / N, rtol=1e-14)
        p2, _, _, h2 = orc.svgd_update(p, g, 0.02, h=0.37)
        K2 = np.exp(-d ** 2 / 0.37 ** 2 / 2)
        rep2 = (-K2 @ p + p * K2.sum(axis=1, keepdims=True)) / 0.37 ** 2
        assert h2 == 0.37 and np.allclose(p2, p + 0.02 * (K2 @ g + rep2) / N, rtol=1e-14)


@pytest.mark.gpu
@pytest.mark.parametrize("N,dim", [(6, 17), (257, 17), (1000, 3), (4096, 17)])
def test_gpu_svgd_update_matches_oracle(orc, N, dim):
    """HIP svgd kernels vs the oracle's scalar C loops.  N = 6: 15 pairs (odd count, single middle element); N = 257 /
    1000 / 4096: even counts (mean of the middle pair); 4096 x 17 is BASELINE config 5's particle count."""
    from crnn_amd.cathode import svgd_update
    rng = np.random.default_rng(N)
    p = 1 + 0.05 * rng.standard_normal((N, dim)); g = 50 * rng.standard_normal((N, dim))
    pn, dt, rp, href = orc.svgd_update(p, g, 0.01)
    pd, dtd, rpd, h = svgd_update(p, g, 0.01)
    assert abs(h - href) <= 1e-14 * href                           # the exact median, not an approximation
    assert np.max(np.abs(dtd - dt)) < 1e-11 * np.max(np.abs(dt))
    assert np.max(np.abs(rpd - rp)) < 1e-10 * np.max(np.abs(rp))
    assert np.max(np.abs(pd - pn)) < 1e-13
    p2, _, _, h2 = svgd_update(p, g, 0.01, h=0.37)          # explicit bandwidth
    assert h2 == 0.37 and np.max(np.abs(p2 - orc.svgd_update(p, g, 0.01, h=0.37)[0])) < 1e-13


def test_cathode_host_refuses_switches_the_default_gradient_would_ignore(cfx):
    """CathodeUQ's default gradient is the dual-norm one (errnorm_sens = 2), which reads neither grad_mode nor tape_every; a caller
    who sets them gets an error (before any device call), not a silently different gradient."""
    from crnn_amd.cathode import CathodeUQ
    args = ([np.zeros((4, 2))], [2.0], np.ones(17))
    for kw in (dict(grad_mode=1), dict(tape_every=4), dict(grad_mode=2, tape_every=1)):
        with pytest.raises(ValueError, match="errnorm_sens=0"):
            CathodeUQ(*args, **kw)


def test_cathode_config_abi():
    import ctypes as C
    from crnn_amd import _lib as L
    cfg = L.CathodeConfig()
    assert L.lib.crnn_cathode_config_default(C.byref(cfg)) == 0
    assert (cfg.atol, cfg.rtol, cfg.lb_clamp, cfg.T0) == (1e-12, 1e-3, 1e-16, 373.15)
    assert cfg.abi_version == L.lib.crnn_abi_version()
    import torch
    if not torch.cuda.is_available():
        h = C.c_void_p()
        assert L.lib.crnn_cathode_create(C.byref(cfg), C.byref(h)) != 0        # fails loudly without a GPU
        assert b"no HIP device" in L.lib.crnn_cathode_last_error(None)


# ------------------------------------------------------------------ GPU: kernel vs oracle / golden
CLOUD_SEED_PRIMAL_NORM = 13      # a 2 % cloud on which the primal-norm gradient is well-conditioned under every noise realisation tried (11 is not: see the SVGD loop test)


def _uq(cfx, **kw):
    kw.setdefault("errnorm_sens", 0)      # these tests pin the primal-norm adjoint / forward tangents unless they say otherwise
    from crnn_amd.cathode import CathodeUQ
    return CathodeUQ([_two_replicas(s) for s in cfx["sets"]], [s["beta"] for s in cfx["sets"]], cfx["theta"], **kw)


@pytest.mark.gpu
def test_gpu_cathode_matches_golden_tight(cfx):
    uq = _uq(cfx, atol=1e-14, rtol=1e-9)
    loss, grad, hrr = uq.solve(np.ones((1, 17)), want_hrr=True)
    assert np.all(uq.last_retcode == 0)
    ps = np.array(cfx["theta"])
    for i, s in enumerate(cfx["sets"]):
        D = len(s["ts"])
        assert uq.last_n_saved[0, i] == D
        hg, gg = np.array(s["hrr"]), np.array(s["grad"]) * ps     # d/dp = d/dtheta * p_scales
        assert np.max(np.abs(hrr[0, i, :D] - hg)) < 2e-8 * max(1.0, np.max(np.abs(hg)))
        assert abs(loss[0, i] - s["loss"]) < 1e-8 * abs(s["loss"])
        assert np.max(np.abs(grad[0, i] - gg)) < 2e-8 * np.max(np.abs(gg))


@pytest.mark.gpu
@pytest.mark.parametrize("tol", [(1e-12, 1e-3), (1e-10, 1e-6)])
def test_gpu_cathode_matches_oracle_step_for_step(orc, cfx, tol):
    """Same stepper, same tolerances, perturbed particles: the kernel follows the oracle's step sequence, so results
    agree far below the solver tolerance (fp64 reassociation only)."""
    atol, rtol = tol
    rng = np.random.default_rng(11)
    N = 6
    p = 1 + 0.05 * rng.standard_normal((N, 17))
    p[:, 6:9] = 0.0                                           # p_scales[b] = 0 in the reference's optimum
    uq = _uq(cfx, atol=atol, rtol=rtol)
    loss, grad, hrr = uq.solve(p, want_hrr=True)
    ps = np.array(cfx["theta"])
    nacc = 0
    for n in range(N):
        for i, s in enumerate(cfx["sets"]):
            c = orc.make_cathode(s["beta"], atol=atol, rtol=rtol)
            r = orc.cathode_solve_one(c, p[n] * ps, s["ts"], s["dbar"], s["d2bar"])
            D = len(s["ts"])
            assert r["retcode"] == uq.last_retcode[n, i] == 0
            assert np.max(np.abs(hrr[n, i, :D] - r["hrr"])) < 1e-9 * max(1.0, np.max(np.abs(r["hrr"])))
            assert abs(loss[n, i] - r["loss"]) < 1e-9 * abs(r["loss"])
            assert np.max(np.abs(grad[n, i] - r["grad"] * ps)) < 1e-7 * np.max(np.abs(r["grad"] * ps))
            nacc += r["naccept"]
    assert uq.last_stats["n_accept"] == nacc


@pytest.mark.gpu
@pytest.mark.parametrize("tape_every", [1, 4, 8])
def test_gpu_cathode_adjoint_equals_forward_tangents(cfx, tape_every):
    """grad_mode 0/2: reversed accepted steps; grad_mode 1: 14 tangent columns.  Same losses, gradients equal to rounding;
    also for solutions truncated by maxiters (gradient of the saved prefix).  tape_every = 4 / 8: the checkpointed tape
    (crnn_cathode_set_tape_every: states between checkpoints re-formed in the reverse sweep; step counts that are and are not
    multiples of the interval, ragged wavefronts)."""
    rng = np.random.default_rng(21)
    p = 1 + 0.05 * rng.standard_normal((70, 17))              # more than one wavefront per heating rate
    p[:, 6:9] = 0.0
    for kw in (dict(), dict(maxiters=150), dict(maxiters=7), dict(atol=1e-10, rtol=1e-6)):
        fwd, adj = _uq(cfx, grad_mode=1, **kw), _uq(cfx, grad_mode=2, tape_every=tape_every, **kw)
        lf, gf, hf = fwd.solve(p, want_hrr=True)
        la, ga, ha = adj.solve(p, want_hrr=True)
        assert np.array_equal(fwd.last_retcode, adj.last_retcode) and np.array_equal(fwd.last_n_saved, adj.last_n_saved)
        assert fwd.last_stats["n_accept"] == adj.last_stats["n_accept"]
        assert np.max(np.abs(lf - la)) <= 1e-12 * np.max(np.abs(lf))
        assert np.max(np.abs(hf - ha)) <= 1e-12 * np.max(np.abs(hf))
        assert np.max(np.abs(gf - ga) / np.max(np.abs(gf), axis=2, keepdims=True)) < 1e-8
        if "maxiters" in kw:
            assert np.all(fwd.last_retcode == 1)


@pytest.mark.gpu
def test_gpu_cathode_dlnprob_and_reference_surface(orc, cfx):
    from crnn_amd.cathode import NORMALIZER, NORM_COL
    uq = _uq(cfx)
    rng = np.random.default_rng(3)
    p = 1 + 0.02 * rng.standard_normal((4, 17))
    i_exp = 3
    l, g = uq.dlnprob(p, i_exp)
    loss, grad, _ = uq.solve(p)
    assert l == pytest.approx(loss[:, i_exp].mean())
    assert np.allclose(g, -grad[:, i_exp] / NORMALIZER[i_exp, NORM_COL] ** 2)
    heat, tt = uq.pred_n_ode(p[0], i_exp)
    s = cfx["sets"][i_exp]
    assert np.array_equal(tt, np.array(s["ts"]))
    c = orc.make_cathode(s["beta"])
    r = orc.cathode_solve_one(c, p[0] * np.array(cfx["theta"]), s["ts"], s["dbar"], s["d2bar"])
    assert np.max(np.abs(heat - r["hrr"])) < 1e-9 * np.max(np.abs(r["hrr"]))
    assert uq.loss_neuralode(p[0], i_exp) == pytest.approx(r["loss"], rel=1e-9)


@pytest.mark.gpu
def test_gpu_cathode_many_particles_consistent(cfx):
    """4096 particles x 5 heating rates in one launch: repeated particles give bit-identical rows wherever they sit in
    the work queue, and the batch equals the small launch."""
    uq = _uq(cfx)
    rng = np.random.default_rng(8)
    base = 1 + 0.05 * rng.standard_normal((16, 17))
    p = np.tile(base, (256, 1))
    loss, grad, _ = uq.solve(p)
    assert np.all(uq.last_retcode == 0)
    l0, g0, _ = uq.solve(base)
    assert np.array_equal(loss.reshape(256, 16, 5), np.broadcast_to(l0, (256, 16, 5)))
    assert np.array_equal(grad.reshape(256, 16, 5, 17), np.broadcast_to(g0, (256, 16, 5, 17)))


def _many_rates(cfx, n_rates):
    """n_rates heating rates log-spaced in [2, 20] K/min (BASELINE config 5): each borrows the temperature grid and the
    replica statistics of the nearest measured rate, its time grid follows from t = (T - 100) * 60 / beta (dataset.jl:19-23)."""
    betas = np.exp(np.linspace(np.log(2.0), np.log(20.0), n_rates))
    meas = np.array([s["beta"] for s in cfx["sets"]])
    exp_data = []
    for b in betas:
        s = cfx["sets"][int(np.argmin(np.abs(np.log(meas) - np.log(b))))]
        e = _two_replicas(s)
        e[:, 0] = e[:, 0] * s["beta"] / b
        exp_data.append(e)
    return betas, exp_data


@pytest.mark.gpu
def test_gpu_cathode_many_heating_rates(orc, cfx):
    """More heating rates than the LDS stages (8): observation rows are read in place; same results as the oracle."""
    from crnn_amd.cathode import CathodeUQ
    betas, exp_data = _many_rates(cfx, 24)
    uq = CathodeUQ(exp_data, betas, cfx["theta"], normalizer=np.ones((24, 3)), errnorm_sens=0)
    rng = np.random.default_rng(4)
    p = 1 + 0.03 * rng.standard_normal((3, 17))
    p[:, 6:9] = 0.0
    loss, grad, hrr = uq.solve(p, want_hrr=True)
    assert np.all(uq.last_retcode == 0)
    ps = np.array(cfx["theta"])
    for n in range(3):
        for i in (0, 7, 8, 15, 23):
            e = exp_data[i]
            c = orc.make_cathode(betas[i])
            r = orc.cathode_solve_one(c, p[n] * ps, e[:, 0], e[:, 1:].mean(axis=1), (e[:, 1:] ** 2).mean(axis=1))
            D = e.shape[0]
            assert np.max(np.abs(hrr[n, i, :D] - r["hrr"])) < 1e-9 * max(1.0, np.max(np.abs(r["hrr"])))
            assert abs(loss[n, i] - r["loss"]) < 1e-9 * abs(r["loss"])
            assert np.max(np.abs(grad[n, i] - r["grad"] * ps)) < 1e-7 * np.max(np.abs(r["grad"] * ps))


@pytest.mark.gpu
def test_gpu_cathode_config5_full_size(orc, cfx):
    """BASELINE config 5 at its full size on one GPU: 4 096 particles x 256 heating rates = 1 048 576 trajectories with
    per-particle 17-parameter gradients in ONE launch.  All retcodes 0; the 4 096 particles are 256 tiles of 16 distinct
    ones, so every tile must be bit-identical wherever it sat in the work queue; 64 random (particle, rate) rows are
    checked against the oracle (same stepper, reference tolerances: step for step)."""
    from conftest import emulated
    from crnn_amd.cathode import CathodeUQ
    NT, NRt, NS_ = (6, 12, 12) if emulated() else (256, 256, 64)     # tiles of 16 particles, heating rates, oracle rows (SIMT emulation: scaled down)
    betas, exp_data = _many_rates(cfx, NRt)
    uq = CathodeUQ(exp_data, betas, cfx["theta"], normalizer=np.ones((NRt, 3)), errnorm_sens=0)
    rng = np.random.default_rng(55)
    base = 1 + 1e-3 * rng.standard_normal((16, 17))           # particles = 1 + 1e-3 N(0,1)
    base[:, 6:9] = 0.0
    p = np.tile(base, (NT, 1))
    loss, grad, _ = uq.solve(p)
    st = uq.last_stats
    assert st["n_traj"] == 16 * NT * NRt == st["n_ok"] and np.all(uq.last_retcode == 0)
    assert np.all(uq.last_n_saved == np.array([e.shape[0] for e in exp_data])[None, :])
    L4, G4 = loss.reshape(NT, 16, NRt), grad.reshape(NT, 16, NRt, 17)
    assert np.array_equal(L4, np.broadcast_to(L4[0], L4.shape))
    assert np.array_equal(G4, np.broadcast_to(G4[0], G4.shape))
    ps = np.array(cfx["theta"])
    for _ in range(NS_):
        n, i = int(rng.integers(0, 16 * NT)), int(rng.integers(0, NRt))
        e = exp_data[i]
        r = orc.cathode_solve_one(orc.make_cathode(betas[i]), p[n] * ps, e[:, 0], e[:, 1:].mean(axis=1), (e[:, 1:] ** 2).mean(axis=1))
        assert r["retcode"] == 0
        assert abs(loss[n, i] - r["loss"]) < 1e-9 * abs(r["loss"])
        assert np.max(np.abs(grad[n, i] - r["grad"] * ps)) < 1e-7 * np.max(np.abs(r["grad"] * ps))
    print(f"config 5 full size: kernel {st['kernel_ms']:.1f} ms, {st['n_accept'] / st['n_traj']:.0f} steps/trajectory")


@pytest.mark.gpu
def test_gpu_cathode_allgather_single_rank(cfx):
    """crnn_cathode_allgather with a one-rank RCCL communicator (the N > 1 exchange needs one GPU per rank): rows come back
    unchanged through the device staging buffers; dlnprob_sharded == dlnprob."""
    uq = _uq(cfx)
    uq.comm_init()
    rows = np.random.default_rng(0).standard_normal((37, 18))
    assert np.array_equal(uq.allgather(rows, 37), rows)
    p = 1 + 0.02 * np.random.default_rng(1).standard_normal((5, 17))
    l1, g1 = uq.dlnprob(p, 2)
    l2, g2 = uq.dlnprob_sharded(p, 2)
    assert l1 == l2 and np.array_equal(g1, g2)


@pytest.mark.gpu
def test_gpu_cathode_maxiters_and_bad_inputs(cfx):
    from crnn_amd._lib import CrnnError
    from crnn_amd.cathode import CathodeUQ
    uq = _uq(cfx, maxiters=5)
    loss, grad, _ = uq.solve(np.ones((2, 17)))
    assert np.all(uq.last_retcode == 1)                       # MaxIters
    assert np.all(uq.last_n_saved < np.array([len(s["ts"]) for s in cfx["sets"]]))
    bad = _two_replicas(cfx["sets"][0]); bad[3, 0] = bad[2, 0]
    with pytest.raises(CrnnError):
        CathodeUQ([bad], [2.0], cfx["theta"])
    with pytest.raises(CrnnError):
        CathodeUQ([_two_replicas(cfx["sets"][0])], [-1.0], cfx["theta"])


def test_oracle_cathode_autotsit5_restatement_stays_on_tsit5(orc, cfx):
    """The reference integrates the cathode model with AutoTsit5(TRBDF2(autodiff=true)) (network.jl:195).  Under the
    restated AutoSwitch rule (oracle: solver=2) the detector never reaches its 11th stiff step in a row at the reference's
    parameters, so those runs are Tsit5 runs; at tight tolerance they give the heat-release curves and losses of the
    Rosenbrock23 path (which the device uses) to 1e-8.  The tangents do NOT carry over: late in the run the depleted
    species hover around lb_clamp = 1e-16 with Tsit5 at its stability limit, the clamp's derivative flips between 0 and
    1/u and the explicit tangent recursion loses its damping -- gradient errors of many orders of magnitude in some sets
    (DESIGN.md section 2).  That is why the product path keeps the L-stable Rosenbrock23 for this model."""
    th = np.array(cfx["theta"])
    worst = 0.0
    for s in cfx["sets"]:
        r = orc.cathode_solve_one(orc.make_cathode(s["beta"], solver=2), th, s["ts"], s["dbar"], s["d2bar"], want_grad=False)
        assert r["retcode"] == 0 and r["n_tsit5"] == r["naccept"] and 90 < r["naccept"] < 140
        a = orc.cathode_solve_one(orc.make_cathode(s["beta"], solver=2, atol=1e-13, rtol=1e-9), th, s["ts"], s["dbar"], s["d2bar"])
        b = orc.cathode_solve_one(orc.make_cathode(s["beta"], solver=0, atol=1e-13, rtol=1e-9), th, s["ts"], s["dbar"], s["d2bar"])
        assert a["n_tsit5"] == a["naccept"]
        assert abs(a["loss"] - b["loss"]) < 1e-8 * b["loss"]
        assert np.max(np.abs(a["hrr"] - b["hrr"])) < 1e-8 * np.max(np.abs(b["hrr"]))
        worst = max(worst, np.max(np.abs(a["grad"] - b["grad"])) / np.max(np.abs(b["grad"])))
    assert worst > 1.0      # the instability described above is there (if this ever fails, revisit the choice of stepper)


# ------------------------------------------------------------------ the reference's stepper: AutoTsit5(TRBDF2(autodiff = true))
def _perturbed(spread, N, seed=5):
    rng = np.random.default_rng(seed)
    p = 1 + spread * rng.standard_normal((N, 17))
    p[:, 6:9] = 0.0
    return p


def test_oracle_trbdf2_alone_converges_to_the_golden_vectors(orc, cfx):
    """TRBDF2 by itself (oracle solver 4: the restated stepper + Newton machinery, network.jl:195's stiff algorithm) against
    Radau: second-order convergence towards the golden heat-release curves, with both readings of the smoothed error
    estimate; the Newton iteration reuses its Jacobian (a handful of J evaluations per run, as do_newJW intends)."""
    th = np.array(cfx["theta"])
    for est in (0, 1):
        for s in cfx["sets"][::2]:
            hg = np.array(s["hrr"])
            errs = []
            for atol, rtol in ((1e-12, 1e-3), (1e-13, 1e-6), (1e-14, 1e-9)):
                c = orc.make_cathode(s["beta"], atol=atol, rtol=rtol, solver=4, trbdf2_est=est)
                c.qsteady_max = 1.2            # TRBDF2 alone is an implicit algorithm type
                r = orc.cathode_solve_one(c, th, s["ts"], s["dbar"], s["d2bar"], want_grad=False)
                assert r["retcode"] == 0 and r["n_saved"] == len(s["ts"])
                assert r["n_newton"] >= 4 * r["naccept"] and r["n_jac"] <= 0.1 * r["naccept"] + 5 and r["n_w"] >= r["n_jac"]
                errs.append(np.max(np.abs(r["hrr"] - hg)) / np.max(np.abs(hg)))
            assert errs[0] < 3e-3 and errs[1] < 5e-5 and errs[2] < 2e-6, errs      # measured 1.2e-3 / 2.3e-5 / 7e-7 at worst
            assert errs[2] < 0.1 * errs[1] < 0.01 * errs[0] * 10


def test_oracle_autotsit5_trbdf2_composite(orc, cfx):
    """The reference's `alg` (oracle solver 3).  At the reference's own parameter vector the detector never fires, so the run is
    the Tsit5 run of the Rosenbrock23 composite (solver 2), bit for bit; in a perturbed cloud some trajectories do reach
    TRBDF2, take a few Newton-solved steps and return; results agree with the Rosenbrock23 path to solver tolerance and, at
    tight tolerance, to 1e-7."""
    th = np.array(cfx["theta"])
    for s in cfx["sets"]:
        a = orc.cathode_solve_one(orc.make_cathode(s["beta"], solver=3), th, s["ts"], s["dbar"], s["d2bar"], want_grad=False)
        b = orc.cathode_solve_one(orc.make_cathode(s["beta"], solver=2), th, s["ts"], s["dbar"], s["d2bar"], want_grad=False)
        assert a["retcode"] == 0 and a["n_tsit5"] == a["naccept"] == b["naccept"] and a["n_newton"] == 0
        assert a["loss"] == b["loss"] and np.array_equal(a["hrr"], b["hrr"])
    p = _perturbed(0.05, 24)
    n_switch = n_newton = 0
    acc = [0, 0]
    for n in range(p.shape[0]):
        for s in cfx["sets"][1::2]:
            r = orc.cathode_solve_one(orc.make_cathode(s["beta"], solver=3), p[n] * th, s["ts"], s["dbar"], s["d2bar"], want_grad=False)
            r0 = orc.cathode_solve_one(orc.make_cathode(s["beta"], solver=0), p[n] * th, s["ts"], s["dbar"], s["d2bar"], want_grad=False)
            assert r["retcode"] == 0
            assert abs(r["loss"] - r0["loss"]) < 3e-2 * r0["loss"]
            n_switch += r["n_tsit5"] != r["naccept"]; n_newton += r["n_newton"]
            acc[0] += r["naccept"]; acc[1] += r0["naccept"]
            t = orc.cathode_solve_one(orc.make_cathode(s["beta"], solver=3, atol=1e-14, rtol=1e-9), p[n] * th, s["ts"], s["dbar"], s["d2bar"], want_grad=False)
            t0 = orc.cathode_solve_one(orc.make_cathode(s["beta"], solver=0, atol=1e-14, rtol=1e-9), p[n] * th, s["ts"], s["dbar"], s["d2bar"], want_grad=False)
            assert abs(t["loss"] - t0["loss"]) < 1e-7 * t0["loss"]
            assert np.max(np.abs(t["hrr"] - t0["hrr"])) < 1e-7 * np.max(np.abs(t0["hrr"]))
    assert n_switch >= 3 and n_newton > 0      # the stiff branch did run
    assert acc[0] < 0.5 * acc[1]               # and the composite takes well under half of Rosenbrock23's steps overall


def test_oracle_gradient_through_trbdf2_and_through_the_reference_composite(orc, cfx):
    """network.jl:232 through :195 -- ForwardDiff.gradient through AutoTsit5(TRBDF2(autodiff=true)) -- in the oracle (end of round 4; the
    device's gradient launches stay on the Rosenbrock23 adjoint).  (i) TRBDF2 alone (solver 4): tangent copies ride
    through the Newton iteration with the partial of W on the right-hand side (cath_cp); at tight tolerance the gradient is the Radau
    sensitivity golden vector's and the central difference of the oracle's own loss.  (ii) The composite with the PRIMAL error norm has no
    usable gradient: Tsit5 steps at their stability limit let the tangents of this stiff right-hand side grow without bound (1e27 at tight
    tolerance, 1e5 already at the reference's on one heating rate) -- what DESIGN section 9 reports for the device's reverted composite
    adjoint.  (iii) With ForwardDiff's chunks and the partials in every algorithm's error norm (errnorm_sens = 2: what the reference really
    evaluates) it is as good as Rosenbrock23's: within 5e-3 of the golden sensitivities at the reference's tolerances on all heating
    rates, the stiff branch taking Newton-solved steps on some of them."""
    th = np.array(cfx["theta"])
    s = cfx["sets"][1]
    c4 = orc.make_cathode(s["beta"], atol=1e-14, rtol=1e-9, solver=4)
    c4.qsteady_max = 1.2
    r = orc.cathode_solve_one(c4, th, s["ts"], s["dbar"], s["d2bar"])
    gg = np.array(s["grad"])
    assert r["retcode"] == 0 and r["n_newton"] > 4 * r["naccept"]
    assert np.max(np.abs(r["grad"] - gg)) < 1e-5 * np.max(np.abs(gg))                      # measured 2.3e-6
    for k in (0, 4, 15):
        h = 1e-5 * max(1.0, abs(th[k]))
        tp, tm = th.copy(), th.copy()
        tp[k] += h; tm[k] -= h
        fd = (orc.cathode_solve_one(c4, tp, s["ts"], s["dbar"], s["d2bar"], want_grad=False)["loss"]
              - orc.cathode_solve_one(c4, tm, s["ts"], s["dbar"], s["d2bar"], want_grad=False)["loss"]) / (2 * h)
        assert abs(fd - r["grad"][k]) < 1e-5 * np.max(np.abs(r["grad"]))                   # measured 2e-6
    bad = orc.cathode_solve_one(orc.make_cathode(s["beta"], atol=1e-14, rtol=1e-9, solver=3), th, s["ts"], s["dbar"], s["d2bar"])
    assert bad["retcode"] == 0 and not np.max(np.abs(bad["grad"] - gg)) < 1e10 * np.max(np.abs(gg))
    newton = 0
    for s in cfx["sets"]:
        gg = np.array(s["grad"])
        for solver in (3, 2):
            g = np.zeros(17)
            for cc, (lo, n) in zip(orc.cathode_sens_chunks(orc.make_cathode(s["beta"], solver=solver), th, mode=2), ((0, 9), (9, 8))):
                rr = orc.cathode_solve_one(cc, th, s["ts"], s["dbar"], s["d2bar"])
                assert rr["retcode"] == 0
                g[lo:lo + n] = rr["grad"][lo:lo + n]
                newton += rr["n_newton"] if solver == 3 else 0
            assert np.max(np.abs(g - gg)) < 5e-3 * np.max(np.abs(gg)), (solver, s["beta"])   # measured 7.6e-4 ... 1.9e-3 (Rosenbrock23: 1.4e-3 ... 2.2e-3)
    assert newton > 0


@pytest.mark.gpu
@pytest.mark.parametrize("solver,osolver", [("autotsit5_trbdf2", 3), ("autotsit5_rosenbrock23", 2)])
def test_gpu_cathode_composite_primal_matches_oracle_composite(orc, cfx, solver, osolver):
    """crnn_cathode_set_solver: primal launches through the reference's composite (network.jl:195,205-212) against the oracle's
    statement of the same composite, perturbed particles x the reference's five heating rates.  Explicit steps at their
    stability limit amplify round-off, so two implementations agree to a fraction of the solver tolerance, not step for step
    (cathode_auto_kernel.hpp): the bars are the tolerances themselves -- measured (tools/cath_composite_probe.py): loss 8e-7 /
    1e-4, curve 8e-5 / 1e-3 of its peak at rtol 1e-3 (cloud 1e-3 / 5e-2); 1.4e-7 / 8e-7 at rtol 1e-6; 4e-10 / 3e-9 at 1e-9 -- and
    the accepted step counts agree to 4 % (rtol 1e-9: 12 %)."""
    from crnn_amd.cathode import CathodeUQ
    th = np.array(cfx["theta"])
    mk = lambda **kw: CathodeUQ([_two_replicas(s) for s in cfx["sets"]], [s["beta"] for s in cfx["sets"]], cfx["theta"], errnorm_sens=0, **kw)
    for spread, N in ((1e-3, 16), (0.05, 24)):
        p = _perturbed(spread, N)
        for (atol, rtol), bar_l, bar_h in (((1e-12, 1e-3), 1e-3, 4e-3), ((1e-13, 1e-6), 2e-6, 1e-5), ((1e-14, 1e-9), 1e-8, 1e-7)):
            uq = mk(atol=atol, rtol=rtol, solver=solver)
            loss, grad, hrr = uq.solve(p, want_grad=False, want_hrr=True)
            assert grad is None and np.all(uq.last_retcode == 0)
            nacc = 0
            for n in range(N):
                for i, s in enumerate(cfx["sets"]):
                    r = orc.cathode_solve_one(orc.make_cathode(s["beta"], atol=atol, rtol=rtol, solver=osolver), p[n] * th, s["ts"], s["dbar"],
                                              s["d2bar"], want_grad=False)
                    D = len(s["ts"])
                    assert r["retcode"] == 0 and uq.last_n_saved[n, i] == D
                    assert abs(loss[n, i] - r["loss"]) < bar_l * abs(r["loss"]), (spread, rtol, n, i)
                    assert np.max(np.abs(hrr[n, i, :D] - r["hrr"])) < bar_h * np.max(np.abs(r["hrr"])), (spread, rtol, n, i)
                    nacc += r["naccept"]
            # (at rtol 1e-9 the wide cloud switches back and forth on 100-150 of its trajectories, and where a switch falls is
            #  round-off: the totals then differ by up to 12 %, the results by 3e-9)
            assert abs(uq.last_stats["n_accept"] - nacc) < (0.04 if rtol > 1e-8 else 0.2) * nacc
            # the gradient launch of the same context is the Rosenbrock23 adjoint, untouched by the solver setting
            if rtol == 1e-3 and spread == 0.05:
                ref = mk(atol=atol, rtol=rtol)
                l0, g0, _ = ref.solve(p[:4])
                l1, g1, _ = uq.solve(p[:4])
                assert np.array_equal(l0, l1) and np.array_equal(g0, g1)
                # and the primal of the two steppers agrees to solver tolerance (Tsit5 of order 5 against Rosenbrock23 of order 2 at rtol 1e-3)
                lr, _, _ = ref.solve(p, want_grad=False)
                assert np.max(np.abs(loss - lr) / lr) < 5e-2


@pytest.mark.gpu
def test_gpu_cathode_composite_primal_matches_golden_at_tight_tolerance(cfx):
    """north_star's bar: trajectories within 1e-6 of the converged solution -- the composite at tight tolerance against the Radau
    vectors of the reference's own data (measured 1e-9)."""
    from crnn_amd.cathode import CathodeUQ
    for solver in ("autotsit5_trbdf2", "autotsit5_rosenbrock23"):
        uq = CathodeUQ([_two_replicas(s) for s in cfx["sets"]], [s["beta"] for s in cfx["sets"]], cfx["theta"], errnorm_sens=0, atol=1e-14, rtol=1e-9, solver=solver)
        loss, _, hrr = uq.solve(np.ones((1, 17)), want_grad=False, want_hrr=True)
        for i, s in enumerate(cfx["sets"]):
            D = len(s["ts"])
            assert np.max(np.abs(hrr[0, i, :D] - np.array(s["hrr"]))) < 1e-6 * np.max(np.abs(s["hrr"]))
            assert abs(loss[0, i] - s["loss"]) < 1e-6 * s["loss"]
        heat, tt = uq.pred_n_ode(np.ones(17), 2)          # the reference surface goes through the same launch
        assert np.max(np.abs(heat - np.array(cfx["sets"][2]["hrr"]))) < 1e-6 * np.max(np.abs(cfx["sets"][2]["hrr"]))
    with pytest.raises(Exception):
        uq.set_solver(7)


@pytest.mark.gpu
def test_gpu_cathode_composite_config5_full_size(orc, cfx):
    """BASELINE config 5 at full size through the reference's composite, primal: every solve succeeds, a third of Rosenbrock23's
    accepted steps, tiles bit-identical wherever they sat in the queue, 48 random rows within solver tolerance of the oracle."""
    from conftest import emulated
    from crnn_amd.cathode import CathodeUQ
    NT, NRt, NS_ = (6, 12, 12) if emulated() else (256, 256, 48)     # tiles of 16 particles, heating rates, oracle rows (SIMT emulation: scaled down)
    betas, exp_data = _many_rates(cfx, NRt)
    uq = CathodeUQ(exp_data, betas, cfx["theta"], normalizer=np.ones((NRt, 3)), solver="autotsit5_trbdf2", errnorm_sens=0)
    rng = np.random.default_rng(55)
    base = 1 + 1e-3 * rng.standard_normal((16, 17))
    base[:, 6:9] = 0.0
    p = np.tile(base, (NT, 1))
    loss, _, _ = uq.solve(p, want_grad=False)
    st = uq.last_stats
    assert st["n_traj"] == 16 * NT * NRt == st["n_ok"] and 100 < st["n_accept"] / st["n_traj"] < 130
    L4 = loss.reshape(NT, 16, NRt)
    assert np.array_equal(L4, np.broadcast_to(L4[0], L4.shape))
    ps = np.array(cfx["theta"])
    for _ in range(NS_):
        n, i = int(rng.integers(0, 16 * NT)), int(rng.integers(0, NRt))
        e = exp_data[i]
        r = orc.cathode_solve_one(orc.make_cathode(betas[i], solver=3), p[n] * ps, e[:, 0], e[:, 1:].mean(axis=1), (e[:, 1:] ** 2).mean(axis=1),
                                  want_grad=False)
        assert r["retcode"] == 0 and abs(loss[n, i] - r["loss"]) < 1e-3 * abs(r["loss"])
    print(f"config 5 through AutoTsit5(TRBDF2), primal: kernel {st['kernel_ms']:.1f} ms, {st['n_accept'] / st['n_traj']:.0f} steps/trajectory")


# ------------------------------------------------------------------ the gradient as ForwardDiff evaluates it (network.jl:232)
def test_oracle_cathode_errnorm_sens_chunks(orc, cfx):
    """errnorm_sens in the oracle's cathode solve: ForwardDiff's two chunks (9, then 8 + a zero partial) are two different adaptive
    solves -- their step counts differ from each other and from the plain solve's --, the gradient pieces stay within solver
    tolerance of the primal-norm gradient, both norms (1: / length(u), 2: / totallength(u)) are distinguishable, and with all
    direction scales zero the chunks reproduce the plain solve's step sequence."""
    th = np.array(cfx["theta"])
    p = _perturbed(0.02, 1)[0]
    for s in cfx["sets"][::2]:
        c = orc.make_cathode(s["beta"])
        r0 = orc.cathode_solve_one(c, p * th, s["ts"], s["dbar"], s["d2bar"])
        counts = {}
        for mode in (1, 2):
            g = np.zeros(17)
            for cc in orc.cathode_sens_chunks(c, th, mode):
                r = orc.cathode_solve_one(cc, p * th, s["ts"], s["dbar"], s["d2bar"])
                assert r["retcode"] == 0 and abs(r["loss"] - r0["loss"]) < 1e-2 * r0["loss"]
                sl = slice(cc.dir_lo, cc.dir_lo + cc.dir_n)
                g[sl] = r["grad"][sl]
                assert np.all(r["grad"][:cc.dir_lo] == 0) and np.all(r["grad"][cc.dir_lo + cc.dir_n:] == 0)
                counts[(mode, cc.dir_lo)] = (r["naccept"], r["nreject"])
            assert np.max(np.abs((g - r0["grad"]) * th)) < 2e-2 * np.max(np.abs(r0["grad"] * th))
        assert len(set(counts.values()) | {(r0["naccept"], r0["nreject"])}) >= 4
        for cc in orc.cathode_sens_chunks(c, np.zeros(17), 1):
            r = orc.cathode_solve_one(cc, p * th, s["ts"], s["dbar"], s["d2bar"])
            assert (r["naccept"], r["nreject"]) == (r0["naccept"], r0["nreject"]) and abs(r["loss"] - r0["loss"]) < 1e-12 * r0["loss"]   # (the norm is formed in another order: last bits)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 2])
def test_gpu_cathode_errnorm_sens_matches_oracle_chunk_for_chunk(orc, cfx, mode):
    """crnn_cathode_set_errnorm_sens: the two chunk launches (cathode_sens_kernel) against the oracle's chunked solves on perturbed
    particles x the reference's heating rates -- the same step sequences (accepted / rejected counts per chunk, summed over the
    trajectories, identical), gradient pieces to 1e-7 of the largest entry (with respect to p), loss and curves of the call those of
    the plain solve."""
    from crnn_amd.cathode import CathodeUQ
    th = np.array(cfx["theta"])
    N = 6
    p = _perturbed(0.03, N, seed=17)
    uq = CathodeUQ([_two_replicas(s) for s in cfx["sets"]], [s["beta"] for s in cfx["sets"]], cfx["theta"], errnorm_sens=mode)
    ref = CathodeUQ([_two_replicas(s) for s in cfx["sets"]], [s["beta"] for s in cfx["sets"]], cfx["theta"], errnorm_sens=0)
    loss, grad, hrr = uq.solve(p, want_hrr=True)
    l0, g0, h0 = ref.solve(p, want_hrr=True)
    n_acc_plain = ref.last_stats["n_accept"]
    lp, _, hp = ref.solve(p, want_grad=False, want_hrr=True)
    assert np.array_equal(loss, lp) and np.array_equal(hrr, hp)            # the plain (primal) solve's, bit for bit
    assert np.max(np.abs(loss - l0) / l0) < 1e-12                           # (the adjoint launch sums the same loss terms backwards)
    assert uq.last_stats["n_accept"] == n_acc_plain
    tot = np.zeros((2, 2), np.int64)
    for n in range(N):
        for i, s in enumerate(cfx["sets"]):
            c = orc.make_cathode(s["beta"])
            g = np.zeros(17)
            for ch, cc in enumerate(orc.cathode_sens_chunks(c, th, mode)):
                r = orc.cathode_solve_one(cc, p[n] * th, s["ts"], s["dbar"], s["d2bar"])
                assert r["retcode"] == 0
                g[cc.dir_lo:cc.dir_lo + cc.dir_n] = r["grad"][cc.dir_lo:cc.dir_lo + cc.dir_n]
                tot[ch] += (r["naccept"], r["nreject"])
            assert np.max(np.abs(grad[n, i] - g * th)) < 1e-7 * np.max(np.abs(g * th)), (n, i)
    # the same step sequences: the per-chunk totals agree (to one or two attempts in 11 000: a decision that sits on the threshold of
    # `EEst <= 1` can fall the other way in the last bit -- the gradients above agree to 1e-7 nonetheless)
    assert np.max(np.abs(tot - np.array(uq.last_chunk_stats()))) <= 2, (tot, uq.last_chunk_stats())
    # the two gradients are different numbers (a few 1e-3 at reltol 1e-3), and the dual-norm chunks take their own step counts
    assert 1e-5 < np.max(np.abs(grad - g0)) / np.max(np.abs(g0)) < 5e-2
    assert uq.last_chunk_stats()[0][0] != n_acc_plain
    # dlnprob goes through the same call
    l, lnp = uq.dlnprob(p, 2)
    from crnn_amd.cathode import NORMALIZER, NORM_COL
    assert np.allclose(lnp, -grad[:, 2] / NORMALIZER[2, NORM_COL] ** 2, rtol=0, atol=0)


@pytest.mark.gpu
def test_gpu_cathode_gradient_through_the_reference_composite_matches_oracle(orc, cfx):
    """VERDICT r4 item 6 (rows A4 x A7, config 5): the reference's gradient through the reference's OWN stepper on the device --
    ForwardDiff's chunks 9 + 8 through AutoTsit5(TRBDF2) (network.jl:232 through :195) with the chunk's partials in both algorithms'
    error estimates and the tangent copies riding through TRBDF2's Newton iterations: cathode_sens_auto_kernel (set_solver
    "autotsit5_trbdf2" + errnorm_sens) against the oracle's solver = 3 with errnorm_sens, trajectory by trajectory and chunk by chunk.
    Two statements of such a composite do not agree step for step, and at the reference's tolerances not even in their step COUNTS on
    every trajectory: where Tsit5 rides its stability limit the eleventh stiff verdict in a row -- the switch to TRBDF2 -- comes or
    does not come on the last bit, and a chunk solve then takes 140 steps or 1 200 (the oracle's own counts jump the same way between
    neighbouring particles).  So: the GRADIENTS agree on every trajectory (2e-2 of the largest entry at rtol 1e-3, measured 2e-4 ... 2e-3;
    1e-5 at tight tolerance), the step counts within 5 % on at least 60 % of the (trajectory, chunk) solves (measured 70 %), and the loss of a
    gradient call is the plain composite solve's, bit for bit."""
    from crnn_amd.cathode import CathodeUQ
    th = np.array(cfx["theta"])
    N = 3
    P = _perturbed(0.03, 4, seed=23)[[0, 1, 3]]
    for okw, gtol in ((dict(), 2e-2), (dict(atol=1e-13, rtol=1e-7), 1e-5)):
        worst, close, total = 0.0, 0, 0
        for i, s in enumerate(cfx["sets"]):
            one = [_two_replicas(s)], [s["beta"]]
            uq = CathodeUQ(*one, cfx["theta"], errnorm_sens=2, solver="autotsit5_trbdf2", **okw)
            plain = CathodeUQ(*one, cfx["theta"], errnorm_sens=0, solver="autotsit5_trbdf2", **okw)
            for n in range(N):
                loss, grad, _ = uq.solve(P[n:n + 1])
                st = uq.last_chunk_stats()
                lp, _, _ = plain.solve(P[n:n + 1], want_grad=False)
                assert np.array_equal(loss, lp)                            # the plain composite solve's, bit for bit
                c = orc.make_cathode(s["beta"], solver=3, **okw)
                g = np.zeros(17)
                for ch, cc in enumerate(orc.cathode_sens_chunks(c, th, 2)):
                    r = orc.cathode_solve_one(cc, P[n] * th, s["ts"], s["dbar"], s["d2bar"])
                    assert r["retcode"] == 0
                    g[cc.dir_lo:cc.dir_lo + cc.dir_n] = r["grad"][cc.dir_lo:cc.dir_lo + cc.dir_n]
                    total += 1
                    close += abs(st[ch][0] - r["naccept"]) <= 0.05 * r["naccept"] + 2
                worst = max(worst, np.max(np.abs(grad[0, 0] - g * th)) / np.max(np.abs(g * th)))
            uq.close(); plain.close()
        print(f"composite dual-norm gradient, tolerances {okw or '(reference)'}: worst gradient deviation {worst:.2e}; {close} of {total} chunk solves within 5 % of the oracle's step count")
        assert worst < gtol
        assert close >= 0.6 * total


@pytest.mark.gpu
@pytest.mark.parametrize("errnorm_sens,cloud_seed", [(2, 11), (0, CLOUD_SEED_PRIMAL_NORM)])
def test_gpu_device_resident_svgd_loop_matches_host_driven_loop(orc, cfx, errnorm_sens, cloud_seed):
    """crnn_cathode_set_particles / crnn_cathode_svgd_step (particles, per-particle gradients, median select and move all on the
    device, one heating rate per iteration as crnn_cathode.jl:36-50 draws them) against the same iterations driven from the
    host through the entry points the other tests pin to the oracle: dlnprob (crnn_cathode_solve) + crnn_svgd_update, and the
    first move against the oracle's SVGD (orc_svgd_update).  Same kernels, same order: particles agree to 1e-12 after
    six iterations; the bandwidth is the exact median every time (1e-14).

    Both gradient modes: the default (errnorm_sens = 2, ForwardDiff's dual-inclusive norm, network.jl:232) on the cloud this test has always used,
    and the opt-in primal-norm adjoint on a cloud where it is well-conditioned.  On cloud 11 it is not: particle 20's gradient at the third heating
    rate is 9.06e1 in the plain build and 1e6 ... 5e9 under five of six realisations of one-ulp noise (profiles/r05j_simt_ulp_noise.txt: the
    derivative of the discrete step map itself, adjoint = forward tangents), after which the SVGD move throws the cloud out of the solvable region --
    round 3's device happened to realise the benign case; a device whose last places differ need not."""
    uq_dev, uq_host = _uq(cfx, errnorm_sens=errnorm_sens), _uq(cfx, errnorm_sens=errnorm_sens)
    rng = np.random.default_rng(cloud_seed)
    N = 96
    p0 = 1 + 2e-2 * rng.standard_normal((N, 17))
    p0[:, 6:9] = 0.0
    order = [2, 0, 4, 1, 3, 2]
    step = 2e-3
    uq_dev.set_particles(p0)
    p = p0.copy()
    for it, i_exp in enumerate(order):
        loss_d, h_d, ms = uq_dev.svgd_step(i_exp, step)
        loss_h, lnp = uq_host.dlnprob(p, i_exp)
        if it == 0:
            pn_o, _, _, h_o = orc.svgd_update(p, lnp, step)
        from crnn_amd.cathode import svgd_update
        p, _, _, h_h = svgd_update(p, lnp, step)
        if not np.isfinite(loss_h):
            # primal-norm half only: an ill-conditioned gradient (docstring) has thrown the cloud out of the solvable region on an earlier move.
            # Both loops made that move with the same bits (asserted below, iteration by iteration) and both have to report the failed solves.
            assert errnorm_sens == 0 and it > 0 and not np.isfinite(loss_d), (errnorm_sens, it, loss_d, loss_h)
            break
        assert abs(loss_d - loss_h) <= 1e-12 * abs(loss_h) and abs(h_d - h_h) <= 1e-14 * h_h
        assert ms["solve_ms"] > 0 and ms["svgd_ms"] > 0
        pd = uq_dev.particles()
        scale = max(1.0, float(np.abs(p).max()))            # 1 unless a gradient has exploded (then the move is large and so is its last place)
        assert np.max(np.abs(pd - p)) < 1e-12 * scale, it
        if it == 0:
            tol_o = 1e-12 * scale + 4e-15 * step * np.abs(lnp).mean(axis=0).max()     # the oracle's exp is the host's: last-place differences times |lnp|
            assert abs(h_d - h_o) <= 1e-14 * h_o and np.max(np.abs(pd - pn_o)) < tol_o, (np.max(np.abs(pd - pn_o)), tol_o)
    # without looking: steps stay enqueued, the particles are the same
    uq_a, uq_b = _uq(cfx, errnorm_sens=errnorm_sens), _uq(cfx, errnorm_sens=errnorm_sens)
    uq_a.set_particles(p0); uq_b.set_particles(p0)
    for i_exp in order[:3]:
        uq_a.svgd_step(i_exp, step, look=False)
        uq_b.svgd_step(i_exp, step)
    assert np.array_equal(uq_a.particles(), uq_b.particles(), equal_nan=True)
    for u in (uq_dev, uq_host, uq_a, uq_b):
        u.close()


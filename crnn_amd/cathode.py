"""Host mirror of the Bayesian cathode CRNN scripts (Cathode_NCM333_UQ/src_333), on the C ABI's
crnn_cathode_* entry points.

    p2vec(p)                         network.jl:90-107    (pure slicing; the scaling by p_scales happens in crnn!)
    crnn!(du, u, p_for_crnn, t)      network.jl:152-165   (CPU definition kept here)
    HRR_getter(times, u, p)          network.jl:167-175
    pred_n_ode(p, i_exp, exp_data)   network.jl:196-218
    loss_neuralode(p, i_exp)         network.jl:262-275
    dlnprob(p, i_exp)                network.jl:222-260   (loss, -grad ./ Normalizer.^2 per particle)
    svgd_kernel + the particle move  network.jl:67-87, crnn_cathode.jl:36-50   (svgd_update: on the device)

`p` are the reference's normalised particles (one row of 17 per particle), `p_scales` the deterministic optimum
they are scaled by; the device sees theta = p .* p_scales.  Experiments (heating rates) are 0-based here.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from ._lib import CathodeConfig, Stats, check, dptr, iptr, lib

R_GAS = -1.0 / 8.314       # network.jl:151
T0 = 100 + 273.15          # network.jl:189
# dataset.jl:27-32
NORMALIZER = np.array([[.00538199] * 3, [0.01230299] * 3, [0.02823027] * 3, [0.04655562] * 3, [0.05480013] * 3])
# gradient entry k is divided by Normalizer[i_exp, col]^2 with this column map (network.jl:234-250)
NORM_COL = np.array([0, 1, 2, 0, 1, 2, 0, 1, 2, 0, 1, 2, 0, 1, 2, 1, 2])


def p2vec(p):
    p = np.asarray(p, float)
    return p[3:6], p[6:9], np.array([1.0, p[15], p[16]]), p[9:12], p[12:15], p[0:3]


def getsampletemp(t, beta):
    return T0 + beta / 60.0 * np.asarray(t, float)


def crnn(du, u, p_for_crnn, t, *, p_scales, beta, lb_clamp=1e-16):
    """CPU definition of crnn! (never used by the integration)."""
    Ea, b, w_out, _dH, order, A = p_for_crnn
    logX = np.log(np.clip(u, lb_clamp, 10.0))
    T = getsampletemp(t, beta)
    r = np.exp(np.log(T) * (b * p_scales[6:9]) + (R_GAS / T) * (Ea * p_scales[3:6] * 1e5) + order * p_scales[12:15] * logX
               + A * p_scales[0:3])
    du[:] = -r
    du[1] += w_out[1] * p_scales[15] * r[0]
    du[2] += w_out[2] * p_scales[16] * r[1]
    return du


def HRR_getter(times, u_outputs, p_hrr, *, p_scales, beta, lb_clamp=1e-16):
    """u_outputs [3, n]; returns hrr [n]."""
    logX = np.log(np.clip(u_outputs, lb_clamp, 10.0))
    T = getsampletemp(times, beta)
    th = np.asarray(p_hrr, float) * p_scales[:17]
    z = np.log(T)[:, None] * th[6:9] + (R_GAS / T)[:, None] * (th[3:6] * 1e5) + th[12:15] * logX.T + th[0:3]
    return np.exp(z) @ th[9:12]


def svgd_update(p, lnpgrad, stepsize, h=-1.0, device=0):
    """One SVGD move (svgd_kernel network.jl:67-87 + crnn_cathode.jl:36-50) on the GPU: crnn_svgd_update -- exact
    median by radix select, fused K p / K lnpgrad rows.  Needs an MI355X (no CPU path).
    Returns (p_new, data_term, repulsion, h)."""
    p = np.ascontiguousarray(p, np.float64)
    g = np.ascontiguousarray(lnpgrad, np.float64)
    if p.ndim != 2 or g.shape != p.shape:
        raise ValueError("p and lnpgrad must be [N, dim] arrays of the same shape")
    p_new, dt, rep = np.empty_like(p), np.empty_like(p), np.empty_like(p)
    h_out = C.c_double(0.0)
    check(lib.crnn_svgd_update(int(device), dptr(p), dptr(g), p.shape[0], p.shape[1], float(stepsize), float(h), dptr(p_new),
                               C.byref(h_out), dptr(dt), dptr(rep)))
    return p_new, dt, rep, h_out.value


class CathodeUQ:
    """exp_data: list of arrays [D_s, 1 + n_replicas] (col 0 = time in s, dataset.jl:19-23), heating_rates in K/min."""

    def __init__(self, exp_data, heating_rates, p_scales, *, atol=None, rtol=None, maxiters=None, lb_clamp=None, device=0,
                 normalizer=None, grad_mode=None, tape_every=None, solver=None, errnorm_sens=2):
        if errnorm_sens and (grad_mode is not None or tape_every is not None):
            # the dual-norm gradient is two forward-tangent chunk launches: it reads neither switch, and a caller who sets one expects the
            # primal-norm gradient they configure (this used to be ignored silently)
            raise ValueError("CathodeUQ: grad_mode / tape_every configure the primal-norm gradient (adjoint tape, forward tangents); they have no "
                             "effect on the dual-norm gradient that is the default here -- pass errnorm_sens=0 with them")
        self.cfg = CathodeConfig()
        check(lib.crnn_cathode_config_default(C.byref(self.cfg)))
        self.cfg.device = device
        for k, v in (("atol", atol), ("rtol", rtol), ("maxiters", maxiters), ("lb_clamp", lb_clamp), ("grad_mode", grad_mode)):
            if v is not None:
                setattr(self.cfg, k, v)
        self.h = C.c_void_p()
        check(lib.crnn_cathode_create(C.byref(self.cfg), C.byref(self.h)))
        if tape_every is not None:     # adjoint tape: 1 = every step in full (default), 4 / 8 = checkpointed (include/crnn_hip.h)
            self._check(lib.crnn_cathode_set_tape_every(self.h, int(tape_every)))
        if solver is not None:         # stepper of the primal calls: "autotsit5_trbdf2" = the reference's `alg` (network.jl:195)
            self.set_solver(solver)
        self.p_scales = np.asarray(p_scales, float)[:17].copy()
        # errnorm_sens = 2 (default since round 5): gradient calls as the reference evaluates them (network.jl:232) -- ForwardDiff's chunks of
        # 9 + 8, every chunk its own adaptive solve with the partials in the error norm.  It is also the ROBUST gradient of this model: on a 5 %
        # particle cloud the primal-norm adjoint (errnorm_sens = 0: 2.2x faster, the most accurate one where it is accurate) is off by 0.2 ... 1 000
        # times its largest entry on 3 of 78 trajectories, this one on none (profiles/r04m, r05: tools/cathode_gradient_census.py)
        if errnorm_sens:
            self._check(lib.crnn_cathode_set_errnorm_sens(self.h, int(errnorm_sens), dptr(np.ascontiguousarray(self.p_scales))))
        self.beta = np.ascontiguousarray(heating_rates, np.float64)
        self.exp_data = [np.asarray(e, float) for e in exp_data]
        self.n_sets = len(self.exp_data)
        self.D = np.array([e.shape[0] for e in self.exp_data], np.int32)
        self.Dmax = int(self.D.max())
        ts = np.zeros((self.n_sets, self.Dmax)); dbar = np.zeros_like(ts); d2bar = np.zeros_like(ts)
        for s, e in enumerate(self.exp_data):
            ts[s, :e.shape[0]] = e[:, 0]
            ts[s, e.shape[0]:] = e[-1, 0] + np.arange(1, self.Dmax - e.shape[0] + 1)
            dbar[s, :e.shape[0]] = e[:, 1:].mean(axis=1)
            d2bar[s, :e.shape[0]] = (e[:, 1:] ** 2).mean(axis=1)
        self.ts = ts
        self._check(lib.crnn_cathode_set_obs(self.h, self.n_sets, self.Dmax, iptr(self.D), dptr(ts), dptr(dbar), dptr(d2bar),
                                             dptr(self.beta)))
        self.normalizer = NORMALIZER[: self.n_sets] if normalizer is None else np.asarray(normalizer, float)
        self.last_stats = None
        self.last_retcode = None
        self.last_n_saved = None

    def _check(self, rc):
        if rc != 0:
            raise L.CrnnError(lib.crnn_cathode_last_error(self.h).decode())

    SOLVERS = {"rosenbrock23": L.CATH_SOLVER_ROSENBROCK23, "autotsit5_trbdf2": L.CATH_SOLVER_AUTOTSIT5_TRBDF2,
               "autotsit5_rosenbrock23": L.CATH_SOLVER_AUTOTSIT5_ROS23}

    def last_chunk_stats(self):
        """[(accepted, rejected)] summed over the trajectories for the two ForwardDiff chunks of the last errnorm_sens gradient call."""
        out = (C.c_int64 * 4)()
        self._check(lib.crnn_cathode_last_chunk_stats(self.h, out))
        return [(int(out[0]), int(out[1])), (int(out[2]), int(out[3]))]

    def set_solver(self, solver):
        """Stepper of the PRIMAL calls (pred_n_ode, loss_neuralode, solve(want_grad=False)): "rosenbrock23" (default),
        "autotsit5_trbdf2" -- `alg = AutoTsit5(TRBDF2(autodiff = true))`, network.jl:195 -- or "autotsit5_rosenbrock23".
        Gradient calls (dlnprob) always run the Rosenbrock23 adjoint (include/crnn_hip.h: crnn_cathode_set_solver)."""
        self._check(lib.crnn_cathode_set_solver(self.h, self.SOLVERS[solver] if isinstance(solver, str) else int(solver)))

    def solve(self, p, want_grad=True, want_hrr=False):
        """All particles x all heating rates in one launch.  p [N, 17] normalised particles.
        Returns loss [N, n_sets], grad_p [N, n_sets, 17] (d loss / d p, chain rule through p_scales), hrr [N, n_sets, Dmax]."""
        p = np.atleast_2d(np.asarray(p, float))
        N = p.shape[0]
        theta = np.ascontiguousarray(p[:, :17] * self.p_scales)
        nt = N * self.n_sets
        loss = np.zeros(nt)
        grad = np.zeros((nt, 17)) if want_grad else None
        hrr = np.zeros((nt, self.Dmax)) if want_hrr else None
        ret = np.zeros(nt, np.int32); nsv = np.zeros(nt, np.int32)
        st = Stats()
        self._check(lib.crnn_cathode_solve(self.h, dptr(theta), N, dptr(loss), dptr(grad), dptr(hrr), iptr(ret), iptr(nsv),
                                           C.byref(st)))
        self.last_stats = st.asdict()
        self.last_retcode = ret.reshape(N, self.n_sets)
        self.last_n_saved = nsv.reshape(N, self.n_sets)
        if np.any(ret != 0):
            print("ode solver failed")
        g = None if grad is None else grad.reshape(N, self.n_sets, 17) * self.p_scales
        return loss.reshape(N, self.n_sets), g, None if hrr is None else hrr.reshape(N, self.n_sets, self.Dmax)

    # ---- reference surface (per heating rate i_exp) ----
    def pred_n_ode(self, p_temp, i_exp):
        """heat_rel, trunc_time for one particle and one heating rate (the solution object is not returned)."""
        _, _, hrr = self.solve(p_temp, want_grad=False, want_hrr=True)
        n = int(self.last_n_saved[0, i_exp])
        return hrr[0, i_exp, :n], self.ts[i_exp, :n]

    def loss_neuralode(self, p_temp, i_exp):
        loss, _, _ = self.solve(p_temp, want_grad=False)
        return float(loss[0, i_exp])

    def dlnprob(self, p, i_exp):
        """(mean loss over particles, -grad ./ Normalizer.^2) for heating rate i_exp (network.jl:222-260)."""
        loss, grad, _ = self.solve(p, want_grad=True)
        g = grad[:, i_exp, :] / (self.normalizer[i_exp, NORM_COL] ** 2)
        return float(loss[:, i_exp].mean()), -g

    # ---- device-resident SVGD loop (crnn_cathode.jl:36-50) ----
    def set_particles(self, p):
        """Upload the normalised particles p [N, 17]; they stay on the device between svgd_step calls."""
        p = np.ascontiguousarray(np.atleast_2d(np.asarray(p, float))[:, :17])
        self._check(lib.crnn_cathode_set_particles(self.h, dptr(p), dptr(np.ascontiguousarray(self.p_scales)), p.shape[0]))
        self._n_particles = p.shape[0]

    def svgd_step(self, i_exp, stepsize, h=-1.0, look=True):
        """One iteration of the reference loop for heating rate i_exp on the device-resident particles: dlnprob (solve +
        per-particle gradients, one launch) and the SVGD move, enqueued back to back.  look=True returns
        (mean loss, h, {solve_ms, svgd_ms}); look=False reads nothing back (the step stays enqueued)."""
        norm2 = np.ascontiguousarray(self.normalizer[i_exp, NORM_COL] ** 2, np.float64)     # Normalizer[i_exp, column of p_k]^2
        if not look:
            self._check(lib.crnn_cathode_svgd_step(self.h, int(i_exp), dptr(norm2), float(stepsize), float(h), None, None, None))
            return None
        loss, hh = C.c_double(0.0), C.c_double(0.0)
        ms = np.zeros(2)
        self._check(lib.crnn_cathode_svgd_step(self.h, int(i_exp), dptr(norm2), float(stepsize), float(h), C.byref(loss), C.byref(hh), dptr(ms)))
        return loss.value, hh.value, dict(solve_ms=float(ms[0]), svgd_ms=float(ms[1]))

    def particles(self):
        p = np.zeros((self._n_particles, 17))
        self._check(lib.crnn_cathode_get_particles(self.h, dptr(p)))
        return p

    def comm_init(self):
        """Attach the library's own RCCL communicator for the particle-shard exchange (crnn_cathode_allgather); the unique
        id travels through the default torch.distributed group.  One process per GPU."""
        import torch.distributed as dist
        rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
        uid = C.create_string_buffer(L.UNIQUE_ID_BYTES)
        if rank == 0:
            check(lib.crnn_comm_get_unique_id(uid))
        if world > 1:
            obj = [uid.raw]
            dist.broadcast_object_list(obj, src=0)
            uid = C.create_string_buffer(obj[0], L.UNIQUE_ID_BYTES)
        self._check(lib.crnn_cathode_comm_init(self.h, uid, rank, world))
        self._comm = (rank, world)

    def allgather(self, rows, n_total):
        """All ranks' row blocks -> the full [n_total, width] array on every rank (ncclAllGather on the ctx stream)."""
        rows = np.ascontiguousarray(rows, np.float64)
        full = np.empty((n_total, rows.shape[1]))
        self._check(lib.crnn_cathode_allgather(self.h, dptr(rows), rows.shape[0], rows.shape[1], n_total, dptr(full)))
        return full

    def dlnprob_sharded(self, p, i_exp):
        """dlnprob with the particles sharded over the ranks of the default torch.distributed group (one process per
        GPU): each rank solves its contiguous block of particles, one all-gather of [loss | lnpgrad] rows follows, and
        every rank returns the full (mean loss, lnpgrad[N, 17]) -- so that the replicated SVGD update stays identical."""
        from .dist import allgather_rows, shard_range
        import torch.distributed as dist
        p = np.atleast_2d(np.asarray(p, float))
        N = p.shape[0]
        if getattr(self, "_comm", None) is not None:
            rank, world = self._comm
        else:
            world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
            rank = dist.get_rank() if world > 1 else 0
        first, count = shard_range(N, rank, world)
        loss, grad, _ = self.solve(p[first:first + count], want_grad=True)
        rows = np.concatenate([loss[:, i_exp:i_exp + 1], -grad[:, i_exp, :] / (self.normalizer[i_exp, NORM_COL] ** 2)], axis=1)
        full = self.allgather(rows, N) if getattr(self, "_comm", None) is not None else allgather_rows(rows, N)
        return float(full[:, 0].mean()), full[:, 1:]

    def close(self):
        if self.h:
            lib.crnn_cathode_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


svgd_update_device = svgd_update

"""Host-side mirror of the reference's script-level surface for the neural-ODE
hot path, on top of the C ABI (include/crnn_hip.h).

The reference (DENG-MIT/CRNN) is a set of Julia scripts whose surface for this
path is four function names plus globals:

    p2vec(p)                       case2/case2.jl:91-99
    crnn(du, u, p, t)              case2/case2.jl:114-118
    prob = ODEProblem(crnn, u0, tspan, saveat=tsteps, atol=atol, rtol=rtol)   :120-121
    predict_neuralode(u0, p)       case2/case2.jl:124-128
    loss_neuralode(p, i_exp)       case2/case2.jl:132-137
    ForwardDiff.gradient(x -> loss_neuralode(x, i_exp), p)   :195
    update!(opt, p, grad)          :197

Julia is not available in this image, so the host side is Python (ctypes) with
the same names, argument meaning and error behaviour; julia/CRNNHip.jl holds
the equivalent `ccall` shim.  Array shapes follow the Julia scripts:
u0_list[n_exp, n], ode_data_list[n_exp, n_obs, datasize], pred[n, datasize];
experiment indices are 0-based here.  Solver failures do not raise: like the
reference (rober_crnn.jl:130-134) they are reported (retcode / n_saved) and the
truncated prediction is used.

No CPU fallback exists: every compute call goes to the gfx950 kernels.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib as L
from ._lib import CrnnError, Config, OptConfig, Stats, check, dptr, iptr, lib

__all__ = ["NeuralODE", "ODEProblem", "Optimiser", "p2vec", "p2vec_jac", "crnn", "CrnnError"]


# ---------------------------------------------------------------------------
# p2vec (host, exact reference formulas) and its Jacobian
# ---------------------------------------------------------------------------
def fd_chunk_size(P: int) -> int:
    """ForwardDiff.pickchunksize(P) (DEFAULT_CHUNK_THRESHOLD = 12): partials per Dual, i.e. per adaptive solve."""
    if P <= 12:
        return P
    nchunks = -(-P // 12)
    return -(-P // nchunks)


def _extra_rows(pmap):
    """feature rows of w_in beyond the species rows: case2's -1/(R T); HyChem's -1/(R T) and log T"""
    return 1 if pmap == L.PMAP_CASE2 else (2 if pmap == L.PMAP_HYCHEM else 0)


def _shape(pmap, ns, nr):
    extra = _extra_rows(pmap)
    return extra, ns + extra, lib.crnn_n_theta(ns, nr, extra), lib.crnn_n_params(pmap, ns, nr)


def p2vec_jac(pmap: int, ns: int, nr: int, p):
    """theta [n_theta] and d theta / d p [n_theta, P] for a p2vec variant."""
    has_temp, n, nth, P = _shape(pmap, ns, nr)
    p = np.ascontiguousarray(p, np.float64)
    if p.shape != (P,):
        raise ValueError(f"p must have length {P}, got {p.shape}")
    th = np.zeros(nth)
    dth = np.zeros((nth, P), order="F")
    check(lib.crnn_p2vec(pmap, ns, nr, dptr(p), dptr(th), dptr(dth)))
    return th, dth


def split_theta(theta, ns, nr, has_temp):
    n = ns + has_temp
    w_in = np.asarray(theta[: n * nr]).reshape((n, nr), order="F")
    w_b = np.asarray(theta[n * nr: (n + 1) * nr])
    w_out = np.asarray(theta[(n + 1) * nr:]).reshape((ns, nr), order="F")
    return w_in, w_b, w_out


def p2vec(pmap: int, ns: int, nr: int, p):
    """(w_in, w_b, w_out) exactly as the reference's p2vec returns them."""
    th, _ = p2vec_jac(pmap, ns, nr, p)
    return split_theta(th, ns, nr, _extra_rows(pmap))


def crnn(du, u, weights, *, lb, ub=np.inf, inv_R=None, rate_scale=None):
    """CPU *definition* of the CRNN right-hand side (kept, as SURVEY 8(b) asks,
    so that an ODEProblem can be described the way the reference does; the
    integration itself never calls it).  weights = (w_in, w_b, w_out)."""
    w_in, w_b, w_out = weights
    ns = w_out.shape[0]
    x = np.log(np.clip(u[:ns], lb, ub))
    if inv_R is not None:
        x = np.concatenate([x, [inv_R / u[ns]]])
    r = w_out @ np.exp(w_in.T @ x + w_b)
    if rate_scale is not None:
        r = r * np.asarray(rate_scale)[:ns]
    du[:ns] = r
    if inv_R is not None:
        du[ns] = 0.0
    return du


# ---------------------------------------------------------------------------
# Optimiser: Flux.Optimise chain (update!)
# ---------------------------------------------------------------------------
class Optimiser:
    """`opt = Flux.Optimiser(ExpDecay(...), ADAMW(...))` / `ADAMW(...)` and
    `update!(opt, p, grad)` (case2/case2.jl:31-32,197; rober_crnn.jl:19,221-224)."""

    def __init__(self, n_params: int, preset: int | None = None, *, eta=0.005, beta=(0.9, 0.999), wd=0.0,
                 expdecay=None, grad_clip_norm=0.0):
        self.cfg = OptConfig()
        if preset is not None:
            check(lib.crnn_opt_preset(C.byref(self.cfg), preset))
        else:
            self.cfg.eta, self.cfg.beta1, self.cfg.beta2, self.cfg.wd = eta, beta[0], beta[1], wd
            self.cfg.grad_clip_norm = grad_clip_norm
            if expdecay is not None:
                self.cfg.use_expdecay = 1
                self.cfg.ed_eta0, self.cfg.ed_decay, self.cfg.decay_step, self.cfg.ed_clip = (
                    expdecay[0], expdecay[1], int(expdecay[2]), expdecay[3])
        self.n_params = n_params
        self.state = np.zeros(lib.crnn_opt_state_len(n_params))
        check(lib.crnn_opt_init(C.byref(self.cfg), n_params, dptr(self.state)))

    def update_(self, p, grad):
        """update!(opt, p, grad): in place on p (float64, contiguous)."""
        if p.dtype != np.float64 or not p.flags.c_contiguous:
            raise ValueError("p must be a contiguous float64 array (updated in place)")
        g = np.ascontiguousarray(grad, np.float64)
        check(lib.crnn_opt_update(C.byref(self.cfg), self.n_params, dptr(p), dptr(g), dptr(self.state)))
        return p


# ---------------------------------------------------------------------------
# ODEProblem + predict / loss / gradient
# ---------------------------------------------------------------------------
@dataclass
class ODEProblem:
    """Descriptor mirroring `ODEProblem(crnn, u0, tspan, saveat=tsteps, atol=atol, rtol=rtol)`."""
    preset: int
    tsteps: np.ndarray
    atol: object = None
    rtol: object = None
    rate_scale: object = None     # dydt_scale (robertson)
    maxiters: int | None = None
    lb: float | None = None       # log-clamp window overrides
    ub: float | None = None
    loss_kind: int | None = None  # LOSS_MAE (reference) / LOSS_MSE
    solver: int | None = None     # SOLVER_ROSENBROCK23 / SOLVER_TSIT5 / SOLVER_AUTOTSIT5 (preset default: see crnn_config_preset)
    dtmin: float | None = None
    t0: float = 0.0
    device: int = 0
    cols_per_lane: int = 0
    mw: object = None             # HyChem: molar masses (preset: the reference's l_MW)
    grad_mode: int = 0            # GRAD_AUTO (adjoint where available) / GRAD_FORWARD (tangents) / GRAD_ADJOINT
    tape_steps: int = 0           # adjoint tape capacity per trajectory, 0 = auto
    errnorm_sens: int = 0         # 1 / 2: ForwardDiff's dual-inclusive error norm drives the step sizes of gradient calls
                                  # (1: squared norm / length(u); 2: / totallength(u) -- the reference's: tests/test_case2_stream_pin.py)

    def config(self) -> Config:
        cfg = Config()
        check(lib.crnn_config_preset(C.byref(cfg), self.preset))
        if self.solver is not None:
            check(lib.crnn_config_set_solver(C.byref(cfg), int(self.solver)))
        cfg.n_save = len(self.tsteps)
        cfg.t0 = float(self.t0)
        cfg.device = int(self.device)
        cfg.cols_per_lane = int(self.cols_per_lane)
        cfg.grad_mode = int(self.grad_mode)
        cfg.tape_steps = int(self.tape_steps)
        cfg.errnorm_sens = int(self.errnorm_sens)
        n = cfg.ns + cfg.has_temp
        if self.atol is not None:
            a = np.broadcast_to(np.asarray(self.atol, float), (n,))
            for i in range(n):
                cfg.atol[i] = a[i]
        if self.rtol is not None:
            r = np.broadcast_to(np.asarray(self.rtol, float), (n,))
            for i in range(n):
                cfg.rtol[i] = r[i]
        if self.rate_scale is not None:
            for i in range(cfg.ns):
                cfg.rate_scale[i] = float(self.rate_scale[i])
        if self.mw is not None:
            for i in range(cfg.ns):
                cfg.mw[i] = float(self.mw[i])
        if self.maxiters is not None:
            cfg.maxiters = int(self.maxiters)
        if self.lb is not None:
            cfg.lb = float(self.lb)
        if self.ub is not None:
            cfg.ub = float(self.ub)
        if self.loss_kind is not None:
            cfg.loss_kind = int(self.loss_kind)
        if self.dtmin is not None:
            cfg.dtmin = float(self.dtmin)
        return cfg


class _Ctx:
    def __init__(self, cfg: Config):
        self.h = C.c_void_p()
        check(lib.crnn_ctx_create(C.byref(cfg), C.byref(self.h)))
        self.cfg = cfg

    def close(self):
        if self.h:
            lib.crnn_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NeuralODE:
    """One problem (case1 / case2 / robertson) bound to one MI355X.

    >>> node = NeuralODE(ODEProblem(PRESET_CASE2, tsteps))
    >>> node.set_ensemble(u0_list, ode_data_list, yscale)          # case2.jl:62-83
    >>> pred = node.predict_neuralode(u0_list[i], p)                # :124-128
    >>> loss = node.loss_neuralode(p, i)                            # :132-137
    >>> grad = node.gradient(p, i)                                  # :195
    >>> loss_mean, grad = node.loss_and_grad(p)                     # batched (all ICs, one launch)
    """

    def __init__(self, prob: ODEProblem):
        self.prob = prob
        self.cfg = prob.config()
        self.ns, self.nr, self.has_temp = self.cfg.ns, self.cfg.nr, self.cfg.has_temp
        self.n = self.ns + self.has_temp
        self.pmap = self.cfg.param_map
        self.n_theta = lib.crnn_config_n_theta(C.byref(self.cfg))
        self.n_params = lib.crnn_n_params(self.pmap, self.ns, self.nr)
        self.tsteps = np.ascontiguousarray(prob.tsteps, np.float64)
        self.D = self.tsteps.size
        self._ctx = _Ctx(self.cfg)
        self._adhoc = None   # second context for predict_neuralode on arbitrary u0
        self.B = 0
        self.n_obs = self.ns
        self.last_stats = None
        self.last_retcode = None
        self.last_n_saved = None

    # -- reference surface ---------------------------------------------------
    def p2vec(self, p):
        return p2vec(self.pmap, self.ns, self.nr, p)

    def crnn(self, du, u, p, t):
        """SciML in-place RHS signature f(du,u,p,t) (CPU definition only)."""
        w = self.p2vec(p)
        sc = [self.cfg.rate_scale[i] for i in range(self.ns)]
        return crnn(du, u, w, lb=self.cfg.lb, ub=self.cfg.ub, inv_R=self.cfg.inv_R if self.has_temp else None,
                    rate_scale=sc)

    def set_ensemble(self, u0_list, ode_data_list, yscale=None, i_obs=None):
        """Upload u0_list[B, n], ode_data_list[B, n_obs, D] (Julia shapes), yscale[n_obs], i_obs (0-based)."""
        u0 = np.asfortranarray(u0_list, np.float64)
        data = np.asfortranarray(ode_data_list, np.float64)
        B = u0.shape[0]
        n_obs = self.ns if i_obs is None else len(i_obs)
        if u0.shape != (B, self.n):
            raise ValueError(f"u0_list must be [B, {self.n}]")
        if data.shape != (B, n_obs, self.D):
            raise ValueError(f"ode_data_list must be [B, {n_obs}, {self.D}], got {data.shape}")
        ys = None if yscale is None else np.ascontiguousarray(np.ravel(yscale), np.float64)
        io = None if i_obs is None else np.ascontiguousarray(i_obs, np.int32)
        check(lib.crnn_ctx_set_data(self._ctx.h, dptr(u0), dptr(data), dptr(self.tsteps), dptr(ys), iptr(io), n_obs, B),
              self._ctx.h)
        self.B, self.n_obs = B, n_obs
        self._u0 = u0

    def set_tables(self, Tlist, Plist):
        """HyChem: temperature [K] / pressure [Pa] tables on tsteps per experiment, Tlist[B, D], Plist[B, D]
        (crnn_pyrolysis_mass.jl:44-47; piecewise linear in t like itpT / itpP :103-104)."""
        T = np.asfortranarray(Tlist, np.float64)
        P = np.asfortranarray(Plist, np.float64)
        if T.shape != (self.B, self.D) or P.shape != (self.B, self.D):
            raise ValueError(f"tables must be [B, D] = [{self.B}, {self.D}]")
        check(lib.crnn_ctx_set_tables(self._ctx.h, dptr(T), dptr(P)), self._ctx.h)

    def predict_n_ode(self, p, sample=None):
        """HyChem's predict_n_ode(p, sample) (crnn_pyrolysis_mass.jl:135-140) for every experiment: pred[B, ns, D]."""
        th, _ = p2vec_jac(self.pmap, self.ns, self.nr, p)
        pred, _, _, ret, _ = self._solve(self._ctx, self.B, th, None, 0, self.B, sample, True)
        if np.any(ret != 0):
            print("ode solver failed")
        return pred if sample is None else pred[:, :, :int(sample)]

    def loss_n_ode(self, p, sample=None):
        """HyChem's loss_n_ode(p, sample) (:143-147), per experiment."""
        return self.losses(p, sample=sample)

    def _solve(self, ctx, B, theta, dtheta, first, count, sample, want_pred, want_loss=True):
        n_dir = 0 if dtheta is None else dtheta.shape[1]
        sample = self.D if sample is None else int(sample)
        pred = np.zeros((B, self.n, self.D), order="F") if want_pred else None
        loss = np.zeros(B) if want_loss else None
        grad = np.zeros(n_dir) if n_dir else None
        ret = np.zeros(B, np.int32)
        nsv = np.zeros(B, np.int32)
        st = Stats()
        th = np.ascontiguousarray(theta, np.float64)
        dth = None if dtheta is None else np.asfortranarray(dtheta, np.float64)
        check(lib.crnn_solve(ctx.h, dptr(th), dptr(dth), n_dir, first, count, sample, dptr(pred), dptr(loss),
                             dptr(grad), iptr(ret), iptr(nsv), C.byref(st)), ctx.h)
        self.last_stats, self.last_retcode, self.last_n_saved = st.asdict(), ret, nsv
        return pred, loss, grad, ret, nsv

    def predict_theta(self, u0, theta, sample=None):
        """Integrate u0[n] (or a batch u0[B, n]) with explicit effective weights theta."""
        u0 = np.asarray(u0, np.float64)
        single = u0.ndim == 1
        u0b = np.asfortranarray(u0[None, :] if single else u0)
        B = u0b.shape[0]
        if self._adhoc is None:
            self._adhoc = _Ctx(self.cfg)
            if getattr(self, "_jac_mode", 0):
                check(lib.crnn_ctx_set_jacobian(self._adhoc.h, self._jac_mode), self._adhoc.h)
        zeros = np.zeros((B, self.ns, self.D), order="F")
        check(lib.crnn_ctx_set_data(self._adhoc.h, dptr(u0b), dptr(zeros), dptr(self.tsteps), None, None, self.ns, B),
              self._adhoc.h)
        pred, _, _, ret, _ = self._solve(self._adhoc, B, theta, None, 0, B, sample, True, want_loss=False)
        if np.any(ret != 0):
            print("ode solver failed")
        sample = self.D if sample is None else sample
        pred = pred[:, :, :sample]
        return pred[0] if single else pred

    def predict_neuralode(self, u0, p, sample=None):
        """pred = clamp.(Array(solve(prob, alg, u0=u0, p=p)), -ub, ub)  ->  [n, D]
        (or [B, n, D] for a batch of initial conditions u0[B, n]).  A failed
        solve prints "ode solver failed" like the reference and returns the
        truncated prefix (trailing columns zero; see last_n_saved)."""
        th, _ = p2vec_jac(self.pmap, self.ns, self.nr, p)
        return self.predict_theta(u0, th, sample)

    def loss_neuralode(self, p, i_exp, sample=None):
        """mae(ode_data[i_obs,:] ./ yscale, pred[i_obs,:] ./ yscale) for experiment i_exp (0-based)."""
        th, _ = p2vec_jac(self.pmap, self.ns, self.nr, p)
        _, loss, _, _, _ = self._solve(self._ctx, self.B, th, None, int(i_exp), 1, sample, False)
        return float(loss[i_exp])

    def losses(self, p, first=0, count=None, sample=None):
        """Per-experiment losses for [first, first+count): the reference's epoch-end evaluation loop (case2.jl:199-201)."""
        count = self.B - first if count is None else count
        th, _ = p2vec_jac(self.pmap, self.ns, self.nr, p)
        _, loss, _, _, _ = self._solve(self._ctx, self.B, th, None, first, count, sample, False)
        return loss[first:first + count]

    def gradient(self, p, i_exp, sample=None):
        """ForwardDiff.gradient(x -> loss_neuralode(x, i_exp), p).  With errnorm_sens = 1 the call works through p in
        ForwardDiff's chunks (pickchunksize: 25 -> 9 + 9 + 7 ...), each chunk its own adaptive solve whose error norm sees
        that chunk's partials; last_chunk_stats lists the (accepted, rejected) steps of every chunk."""
        th, dth = p2vec_jac(self.pmap, self.ns, self.nr, p)
        if not self.cfg.errnorm_sens:
            _, _, grad, _, _ = self._solve(self._ctx, self.B, th, dth, int(i_exp), 1, sample, False)
            return grad
        P = dth.shape[1]
        chunk = fd_chunk_size(P)
        grad = np.zeros(P)
        self.last_chunk_stats = []
        for k0 in range(0, P, chunk):
            k1 = min(P, k0 + chunk)
            cols = np.zeros((dth.shape[0], chunk), order="F")     # every Dual carries `chunk` partials: the last chunk's
            cols[:, :k1 - k0] = dth[:, k0:k1]                      # surplus ones are zero (they count in errnorm_sens = 2)
            _, _, g, _, _ = self._solve(self._ctx, self.B, th, cols, int(i_exp), 1, sample, False)
            grad[k0:k1] = g[:k1 - k0]
            self.last_chunk_stats.append((self.last_stats["n_accept"], self.last_stats["n_reject"]))
        return grad

    def loss_and_grad(self, p, first=0, count=None, sample=None):
        """Mean loss over experiments [first, first+count) and its gradient w.r.t. p, in one launch
        (p2vec, solve, tangents and reduction all on the device)."""
        count = self.B - first if count is None else count
        sample = self.D if sample is None else int(sample)
        p = np.ascontiguousarray(p, np.float64)
        loss = C.c_double(0.0)
        grad = np.zeros(self.n_params)
        st = Stats()
        check(lib.crnn_loss_grad(self._ctx.h, dptr(p), first, count, sample, C.byref(loss), dptr(grad), C.byref(st)),
              self._ctx.h)
        self.last_stats = st.asdict()
        return loss.value, grad

    # -- device-resident training --------------------------------------------
    def train_init(self, opt: Optimiser, p0):
        p0 = np.ascontiguousarray(p0, np.float64)
        check(lib.crnn_train_init(self._ctx.h, C.byref(opt.cfg), dptr(p0)), self._ctx.h)

    def train_step(self, first=0, count=None, sample=None, want_loss=True):
        count = self.B - first if count is None else count
        sample = self.D if sample is None else int(sample)
        loss = C.c_double(0.0)
        check(lib.crnn_train_step(self._ctx.h, first, count, sample, C.byref(loss) if want_loss else None), self._ctx.h)
        return loss.value if want_loss else None

    def params(self):
        p = np.zeros(self.n_params)
        check(lib.crnn_get_params(self._ctx.h, dptr(p)), self._ctx.h)
        return p

    def update_(self, grad):
        """update!(opt, p, grad) on the device-resident p with a caller-supplied gradient (case2/case2.jl:197)."""
        g = np.ascontiguousarray(grad, np.float64)
        if g.shape != (self.n_params,):
            raise ValueError(f"grad must have {self.n_params} entries")
        check(lib.crnn_train_update(self._ctx.h, dptr(g)), self._ctx.h)

    def opt_state(self):
        """The device-resident optimiser state [m | v | beta1^t, beta2^t, eta_expdecay, ncalls]: what `@save ... opt`
        keeps across a restart (case2/case2.jl:213)."""
        st = np.zeros(lib.crnn_opt_state_len(self.n_params))
        check(lib.crnn_get_opt_state(self._ctx.h, dptr(st)), self._ctx.h)
        return st

    def set_opt_state(self, state):
        st = np.ascontiguousarray(state, np.float64)
        if st.shape != (lib.crnn_opt_state_len(self.n_params),):
            raise ValueError("optimiser state has the wrong length")
        check(lib.crnn_set_opt_state(self._ctx.h, dptr(st)), self._ctx.h)

    def stats(self):
        st = Stats()
        check(lib.crnn_last_stats(self._ctx.h, C.byref(st)), self._ctx.h)
        return st.asdict()

    def set_queue_order(self, order):
        """QUEUE_AUTO (default: trajectories queued by the previous launch's step counts) or QUEUE_INDEX (index order: batch
        sums are a function of the call's inputs alone, bit for bit) -- include/crnn_hip.h: crnn_ctx_set_queue_order."""
        check(lib.crnn_ctx_set_queue_order(self._ctx.h, int(order)), self._ctx.h)

    def set_lanes_per_traj(self, lanes):
        """0 (AUTO, default) / 1 / 2 lanes per trajectory in the Rosenbrock23 adjoint kernel -- include/crnn_hip.h:
        crnn_ctx_set_lanes_per_traj.  2 = an adjacent lane pair per trajectory (shards smaller than the chip)."""
        check(lib.crnn_ctx_set_lanes_per_traj(self._ctx.h, int(lanes)), self._ctx.h)

    def set_jacobian(self, mode):
        """JAC_ANALYTIC (default) / JAC_FINITE_DIFF: the Jacobian behind W in the Rosenbrock23 primal launches (predict, loss) --
        `Rosenbrock23(autodiff=true)` (robertson/rober_crnn.jl:33) / `Rosenbrock23(autodiff=false)` (case2/case2.jl:26); on a HyChem
        problem (HyChem/crnn_pyrolysis_mass.jl:29) FINITE_DIFF also takes the time derivative on the T(t), P(t) tables by a forward
        difference, for Rosenbrock23 and inside AutoTsit5(Rosenbrock23): include/crnn_hip.h: crnn_ctx_set_jacobian."""
        check(lib.crnn_ctx_set_jacobian(self._ctx.h, int(mode)), self._ctx.h)
        self._jac_mode = int(mode)
        if self._adhoc is not None:     # predict_theta / predict_neuralode integrate on their own context
            check(lib.crnn_ctx_set_jacobian(self._adhoc.h, int(mode)), self._adhoc.h)

    def last_lanes_per_traj(self):
        """Lanes per trajectory of the most recent adjoint gradient launch (1 or 2; 0: none yet)."""
        return int(lib.crnn_last_lanes_per_traj(self._ctx.h))

    def tape_retries(self):
        """HyChem: gradient launches repeated with fewer resident trajectories after a tape overflow (0: never)."""
        return int(lib.crnn_tape_retries(self._ctx.h))

    def hychem_block_cap(self):
        """HyChem: the resident-block limit the next gradient launch over the last overflowing range starts from (0: full width)."""
        return int(lib.crnn_hychem_block_cap(self._ctx.h))

    def step_counts(self, first=0, count=None):
        """(naccept, nreject) of every trajectory in [first, first+count) of the most recent solve -- `sol.destats` of
        each `solve` of the ensemble (case2/case2.jl:126)."""
        count = self.B - first if count is None else count
        na, nr = np.zeros(count, np.int32), np.zeros(count, np.int32)
        check(lib.crnn_last_step_counts(self._ctx.h, first, count, iptr(na), iptr(nr)), self._ctx.h)
        return na, nr

    @property
    def handle(self):
        return self._ctx.h

    def close(self):
        self._ctx.close()
        if self._adhoc is not None:
            self._adhoc.close()

"""ctypes binding of oracle/libcrnn_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  The product package (crnn_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcrnn_oracle.so")
MAXN = 12


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "crnn_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "-s"])
    return _SO


class Problem(C.Structure):
    _fields_ = [
        ("ns", C.c_int32), ("nr", C.c_int32), ("has_temp", C.c_int32),
        ("n_obs", C.c_int32), ("i_obs", C.c_int32 * MAXN),
        ("clamp_pred", C.c_int32), ("loss_kind", C.c_int32),
        ("maxiters", C.c_int32), ("errnorm_sens", C.c_int32), ("solver", C.c_int32), ("grad_adjoint", C.c_int32),
        ("jac_fd", C.c_int32),
        ("lb", C.c_double), ("ub", C.c_double), ("inv_R", C.c_double),
        ("rate_scale", C.c_double * MAXN),
        ("atol", C.c_double * MAXN), ("rtol", C.c_double * MAXN),
        ("yscale", C.c_double * MAXN),
        ("t0", C.c_double),
        ("gamma", C.c_double), ("qmin", C.c_double), ("qmax", C.c_double),
        ("beta1", C.c_double), ("beta2", C.c_double),
        ("qsteady_min", C.c_double), ("qsteady_max", C.c_double), ("qoldinit", C.c_double),
        ("dtmin", C.c_double),
    ]


class Opt(C.Structure):
    _fields_ = [
        ("use_expdecay", C.c_int32), ("decay_step", C.c_int32),
        ("ed_eta0", C.c_double), ("ed_decay", C.c_double), ("ed_clip", C.c_double),
        ("eta", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("wd", C.c_double),
        ("grad_clip_norm", C.c_double),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        assert _lib.orc_sizeof_problem() == C.sizeof(Problem), "oracle struct layout mismatch"
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32)) if a is not None else None


def make_problem(*, ns, nr, has_temp=0, lb=1e-6, ub=np.inf, inv_R=0.0, rate_scale=None,
                 atol=1e-6, rtol=1e-3, yscale=None, i_obs=None, clamp_pred=0, loss_kind=0,
                 maxiters=100000, errnorm_sens=0, t0=0.0, solver=0, grad_adjoint=0, jac_fd=0) -> Problem:
    pb = Problem()
    lib().orc_problem_defaults(C.byref(pb))
    n = ns + has_temp
    pb.ns, pb.nr, pb.has_temp = ns, nr, has_temp
    pb.lb, pb.ub, pb.inv_R = lb, ub, inv_R
    pb.clamp_pred, pb.loss_kind, pb.maxiters, pb.errnorm_sens = clamp_pred, loss_kind, maxiters, errnorm_sens
    pb.t0 = t0
    pb.grad_adjoint = int(grad_adjoint)      # Rosenbrock23 gradients by the discrete adjoint (same derivative, other algorithm)
    pb.jac_fd = int(jac_fd)                  # Rosenbrock23(autodiff=false): W from forward differences of the right-hand side (primal solves)
    lib().orc_set_solver(C.byref(pb), int(solver))
    at = np.broadcast_to(np.asarray(atol, float), (n,))
    rt = np.broadcast_to(np.asarray(rtol, float), (n,))
    for i in range(n):
        pb.atol[i], pb.rtol[i] = at[i], rt[i]
    if rate_scale is not None:
        for i in range(ns):
            pb.rate_scale[i] = float(rate_scale[i])
    i_obs = list(range(ns)) if i_obs is None else list(i_obs)
    pb.n_obs = len(i_obs)
    for k, i in enumerate(i_obs):
        pb.i_obs[k] = i
    ys = np.ones(len(i_obs)) if yscale is None else np.asarray(yscale, float)
    for k in range(len(i_obs)):
        pb.yscale[k] = ys[k]
    return pb


def n_theta(pb: Problem) -> int:
    return lib().orc_n_theta(C.byref(pb))


def n_params(kind: int, ns: int, nr: int) -> int:
    return lib().orc_n_params(kind, ns, nr)


def rhs(pb, theta, u):
    n = pb.ns + pb.has_temp
    du = np.zeros(n)
    lib().orc_rhs(C.byref(pb), _dp(np.ascontiguousarray(theta, float)), _dp(np.ascontiguousarray(u, float)), _dp(du))
    return du


def jac(pb, theta, u):
    n = pb.ns + pb.has_temp
    J = np.zeros((n, n), order="F")
    lib().orc_jac(C.byref(pb), _dp(np.ascontiguousarray(theta, float)), _dp(np.ascontiguousarray(u, float)), _dp(J))
    return J


def rhs_jvp(pb, theta, dtheta, u, su):
    n = pb.ns + pb.has_temp
    out = np.zeros(n)
    lib().orc_rhs_jvp(C.byref(pb), _dp(np.ascontiguousarray(theta, float)), _dp(np.ascontiguousarray(dtheta, float)),
                      _dp(np.ascontiguousarray(u, float)), _dp(np.ascontiguousarray(su, float)), _dp(out))
    return out


def jac_dir(pb, theta, dtheta, u, su):
    n = pb.ns + pb.has_temp
    dJ = np.zeros((n, n), order="F")
    lib().orc_jac_dir(C.byref(pb), _dp(np.ascontiguousarray(theta, float)), _dp(np.ascontiguousarray(dtheta, float)),
                      _dp(np.ascontiguousarray(u, float)), _dp(np.ascontiguousarray(su, float)), _dp(dJ))
    return dJ


def p2vec(kind, ns, nr, p, want_jac=True):
    """returns theta [n_theta], dtheta [n_theta, P] (Fortran order) or None."""
    p = np.ascontiguousarray(p, float)
    P = n_params(kind, ns, nr)
    assert p.size == P, (p.size, P)
    n = ns + (1 if kind == 2 else 0)
    nth = nr * (n + 1 + ns)
    th = np.zeros(nth)
    dth = np.zeros((nth, P), order="F") if want_jac else None
    rc = lib().orc_p2vec(kind, ns, nr, _dp(p), _dp(th), _dp(dth))
    assert rc == 0
    return th, dth


def solve_one(pb, theta, u0, tsave, data, dtheta=None, want_pred=True, want_dpred=False):
    """One trajectory.  data: [n_obs, nsave] (Fortran / column-major semantics).
    Returns dict(pred [n,nsave], loss, grad [P], retcode, n_saved, naccept, nreject, dpred)."""
    n = pb.ns + pb.has_temp
    tsave = np.ascontiguousarray(tsave, float)
    nsave = tsave.size
    theta = np.ascontiguousarray(theta, float)
    P = 0
    dth = None
    if dtheta is not None:
        dth = np.asfortranarray(dtheta, float)
        P = dth.shape[1]
    dataF = np.asfortranarray(data, float)
    assert dataF.shape == (pb.n_obs, nsave)
    pred = np.zeros((n, nsave), order="F") if want_pred else None
    dpred = np.zeros((n, nsave, P), order="F") if (want_dpred and P) else None
    loss = C.c_double(0)
    grad = np.zeros(max(P, 1))
    nsaved = C.c_int32(0)
    st = (C.c_int64 * 2)(0, 0)
    extra = {}
    if pb.solver == 2:   # composite: also how the accepted steps split between Tsit5 and Rosenbrock23
        sa = (C.c_int64 * 3)(0, 0, 0)
        lib().orc_solve_one_auto.restype = C.c_int
        rc = lib().orc_solve_one_auto(C.byref(pb), _dp(theta), _dp(dth), C.c_int(P), _dp(np.ascontiguousarray(u0, float)),
                                      _dp(tsave), C.c_int(nsave), _dp(dataF), _dp(pred), _dp(dpred), C.byref(loss),
                                      _dp(grad), C.byref(nsaved), C.cast(st, C.c_void_p), C.cast(sa, C.c_void_p))
        extra = dict(n_tsit5=sa[0], n_rosenbrock=sa[1], n_switch=sa[2])
    else:
        lib().orc_solve_one.restype = C.c_int
        rc = lib().orc_solve_one(C.byref(pb), _dp(theta), _dp(dth), C.c_int(P), _dp(np.ascontiguousarray(u0, float)),
                                 _dp(tsave), C.c_int(nsave), _dp(dataF), _dp(pred), _dp(dpred), C.byref(loss),
                                 _dp(grad), C.byref(nsaved), C.cast(st, C.c_void_p))
    return dict(pred=pred, loss=loss.value, grad=grad[:P].copy(), retcode=rc, n_saved=nsaved.value,
                naccept=st[0], nreject=st[1], dpred=dpred, **extra)


def solve_batch(pb, theta, u0, tsave, data, dtheta=None, want_pred=False, first=0, count=None, nthreads=0):
    """Batched (IC-fastest layout): u0 [n, B] C-order (== u0[i*B+b]), data [nsave, n_obs, B] C-order.
    grad is the SUM over trajectories of d loss_b/dp."""
    n = pb.ns + pb.has_temp
    u0 = np.ascontiguousarray(u0, float)
    B = u0.shape[1]
    assert u0.shape == (n, B)
    tsave = np.ascontiguousarray(tsave, float)
    nsave = tsave.size
    data = np.ascontiguousarray(data, float)
    assert data.shape == (nsave, pb.n_obs, B), data.shape
    count = B - first if count is None else count
    P = 0
    dth = None
    if dtheta is not None:
        dth = np.asfortranarray(dtheta, float)
        P = dth.shape[1]
    pred = np.zeros((nsave, n, B)) if want_pred else None
    loss = np.zeros(B)
    grad = np.zeros(P) if P else None
    retcode = np.zeros(B, np.int32)
    nsaved = np.zeros(B, np.int32)
    stats = np.zeros(2, np.int64)
    lib().orc_solve_batch(C.byref(pb), _dp(np.ascontiguousarray(theta, float)), _dp(dth), C.c_int(P), _dp(u0), _dp(tsave),
                          C.c_int(nsave), _dp(data), C.c_int64(B), C.c_int64(first), C.c_int64(count),
                          _dp(pred), _dp(loss), _dp(grad), _ip(retcode), _ip(nsaved),
                          stats.ctypes.data_as(C.POINTER(C.c_int64)), C.c_int(nthreads))
    return dict(pred=pred, loss=loss, grad=grad, retcode=retcode, n_saved=nsaved,
                naccept=int(stats[0]), nreject=int(stats[1]))


class Optimiser:
    """Flux.Optimise chain restatement (ExpDecay? -> ADAM -> WeightDecay [-> norm clip before])."""

    def __init__(self, P, eta=0.005, beta=(0.9, 0.999), wd=1e-6, expdecay=None, grad_clip_norm=0.0):
        self.o = Opt()
        self.o.eta, self.o.beta1, self.o.beta2, self.o.wd = eta, beta[0], beta[1], wd
        self.o.grad_clip_norm = grad_clip_norm
        if expdecay is not None:
            eta0, decay, step, clip = expdecay
            self.o.use_expdecay, self.o.decay_step = 1, int(step)
            self.o.ed_eta0, self.o.ed_decay, self.o.ed_clip = eta0, decay, clip
        self.P = P
        self.state = np.zeros(lib().orc_opt_state_len(P))
        lib().orc_opt_init(C.byref(self.o), C.c_int(P), _dp(self.state))

    def update(self, p, grad):
        p = np.ascontiguousarray(p, float)
        lib().orc_opt_update(C.byref(self.o), C.c_int(self.P), _dp(p), _dp(np.ascontiguousarray(grad, float)), _dp(self.state))
        return p


def tsit5_tableau():
    c = np.zeros(7); a = np.zeros((7, 6)); bt = np.zeros(7)
    lib().orc_tsit5_tableau(_dp(c), _dp(a), _dp(bt))
    return c, a, bt


def tsit5_dense(theta):
    b = np.zeros(7)
    lib().orc_tsit5_dense(C.c_double(theta), _dp(b))
    return b


# ---------------------------------------------------------------------------- Cathode-UQ restatement
class Cathode(C.Structure):
    _fields_ = [("lb_clamp", C.c_double), ("T0", C.c_double), ("beta", C.c_double), ("atol", C.c_double),
                ("rtol", C.c_double), ("maxiters", C.c_int32), ("solver", C.c_int32),
                ("gamma", C.c_double), ("qmin", C.c_double), ("qmax", C.c_double), ("beta1", C.c_double),
                ("beta2", C.c_double), ("qsteady_min", C.c_double), ("qsteady_max", C.c_double), ("qoldinit", C.c_double),
                ("errnorm_sens", C.c_int32), ("dir_lo", C.c_int32), ("dir_n", C.c_int32), ("dual_partials", C.c_int32),
                ("dir_scale", C.c_double * 17),
                ("trbdf2_est", C.c_int32), ("pad_", C.c_int32)]


def make_cathode(beta, atol=1e-12, rtol=1e-3, maxiters=2500000, lb_clamp=1e-16, solver=0, trbdf2_est=0):
    """solver 0: Rosenbrock23; 2: AutoTsit5 composite (Tsit5 + stiffness switch) with Rosenbrock23 as the stiff algorithm;
    3: AutoTsit5(TRBDF2(autodiff=true)) -- the reference's algorithm (network.jl:195); 4: TRBDF2 alone.  Gradients: tangent copies through
    the Newton iteration (cath_cp); through a composite only with errnorm_sens (cathode_sens_chunks) -- with the primal norm the Tsit5
    branch's tangents grow without bound.
    trbdf2_est: 0 = smoothed estimate `W \\ tmp` with the Newton iteration's W (default), 1 = Shampine's (I - gamma dt J)^-1 tmp."""
    c = Cathode()
    lib().orc_cathode_defaults(C.byref(c))
    assert lib().orc_sizeof_cathode() == C.sizeof(Cathode)
    c.beta, c.atol, c.rtol, c.maxiters, c.lb_clamp = float(beta), atol, rtol, int(maxiters), lb_clamp
    c.solver = int(solver)
    c.trbdf2_est = int(trbdf2_est)
    if solver in (2, 3):
        c.qsteady_max = 1.0          # a composite is not an implicit algorithm type (see solve_one_auto)
    return c


def cathode_sens_chunks(c, p_scales, mode=2):
    """ForwardDiff's chunks of the 17 normalised parameters (pickchunksize(17) = 9: p[0:9], then p[9:17] + one zero partial) as
    configurations of `c` for cathode_solve_one(..., want_grad=True): each chunk is its own adaptive solve whose error norm weighs
    that chunk's partials (errnorm_sens = mode; dir_scale = p_scales = d theta / d p)."""
    import copy
    out = []
    for lo, n in ((0, 9), (9, 8)):
        cc = copy.copy(c)
        cc.errnorm_sens, cc.dir_lo, cc.dir_n, cc.dual_partials = int(mode), lo, n, 9
        for k in range(17):
            cc.dir_scale[k] = float(p_scales[k])
        out.append(cc)
    return out


def cathode_rhs(c, theta, u, t):
    du = np.zeros(3)
    lib().orc_cathode_rhs(C.byref(c), _dp(np.ascontiguousarray(theta, float)), _dp(np.ascontiguousarray(u, float)),
                          C.c_double(t), _dp(du))
    return du


def cathode_solve_one(c, theta, ts, dbar, d2bar, want_grad=True):
    ts = np.ascontiguousarray(ts, float)
    D = ts.size
    hrr = np.zeros(D)
    grad = np.zeros(17) if want_grad else None
    loss = C.c_double(0)
    ns = C.c_int32(0)
    st = (C.c_int64 * 2)(0, 0)
    lib().orc_cathode_solve_one.restype = C.c_int
    rc = lib().orc_cathode_solve_one(C.byref(c), _dp(np.ascontiguousarray(theta, float)), _dp(ts), C.c_int(D),
                                     _dp(np.ascontiguousarray(dbar, float)), _dp(np.ascontiguousarray(d2bar, float)),
                                     _dp(hrr), C.byref(loss), _dp(grad), C.byref(ns), C.cast(st, C.c_void_p))
    lib().orc_cathode_tsit5_steps.restype = C.c_int64
    nl = np.zeros(4, np.int64)
    lib().orc_cathode_nl_stats(nl.ctypes.data_as(C.POINTER(C.c_int64)))
    return dict(hrr=hrr, loss=loss.value, grad=grad, retcode=rc, n_saved=ns.value, naccept=st[0], nreject=st[1],
                n_tsit5=int(lib().orc_cathode_tsit5_steps()),
                n_newton=int(nl[0]), n_jac=int(nl[1]), n_w=int(nl[2]), n_nlfail=int(nl[3]))


def cathode_census(c, theta, beta, ts, D, nthreads=0):
    """AutoSwitch census (orc_cathode_census): theta [n_part, 17], beta [n_sets], ts [n_sets, Dmax], D [n_sets]."""
    theta = np.ascontiguousarray(theta, float); beta = np.ascontiguousarray(beta, float); ts = np.ascontiguousarray(ts, float)
    D = np.ascontiguousarray(D, np.int32)
    out = np.zeros(6, np.int64)
    lib().orc_cathode_census(C.byref(c), _dp(theta), C.c_int64(theta.shape[0]), _dp(beta), C.c_int(beta.size), _dp(ts), _ip(D),
                             C.c_int(ts.shape[1]), out.ctypes.data_as(C.POINTER(C.c_int64)), C.c_int(nthreads))
    return dict(trajectories=int(out[0]), never_left_tsit5=int(out[1]), failed=int(out[2]), accepted=int(out[3]),
                accepted_tsit5=int(out[4]), max_accepted=int(out[5]))


# ---------------------------------------------------------------------------- HyChem restatement
class Hychem(C.Structure):
    _fields_ = [("ns", C.c_int32), ("nr", C.c_int32), ("maxiters", C.c_int32), ("solver", C.c_int32),
                ("errnorm_sens", C.c_int32), ("dual_partials", C.c_int32), ("jac_fd", C.c_int32),
                ("lb", C.c_double), ("ub", C.c_double), ("inv_R", C.c_double), ("Ru", C.c_double),
                ("atol", C.c_double), ("rtol", C.c_double),
                ("mw", C.c_double * 12), ("scale", C.c_double * 12), ("inv_yscale", C.c_double * 12),
                ("gamma", C.c_double), ("qmin", C.c_double), ("qmax", C.c_double), ("beta1", C.c_double),
                ("beta2", C.c_double), ("qsteady_min", C.c_double), ("qsteady_max", C.c_double), ("qoldinit", C.c_double)]


def lu_swaps(reset=True):
    """Row exchanges the oracle's LU factorisations have made since the last reset (pivoting coverage of a scenario)."""
    lib().orc_lu_swaps.restype = C.c_long
    return int(lib().orc_lu_swaps(C.c_int(1 if reset else 0)))


def make_hychem(dydt_scale=None, yscale=None, atol=None, rtol=None, maxiters=None, solver=0, errnorm_sens=0, dual_partials=12, jac_fd=0):
    """solver 0: Rosenbrock23; 2: AutoTsit5(Rosenbrock23(autodiff=false)) -- the reference's `ode_solver` (crnn_pyrolysis_mass.jl:29)."""
    c = Hychem()
    lib().orc_hychem_defaults(C.byref(c))
    assert lib().orc_sizeof_hychem() == C.sizeof(Hychem)
    for i in range(c.ns):
        if dydt_scale is not None:
            c.scale[i] = float(dydt_scale[i])
        if yscale is not None:
            c.inv_yscale[i] = 1.0 / float(yscale[i])
    if atol is not None:
        c.atol = atol
    if rtol is not None:
        c.rtol = rtol
    if maxiters is not None:
        c.maxiters = int(maxiters)
    c.solver = int(solver)
    c.errnorm_sens, c.dual_partials = int(errnorm_sens), int(dual_partials)   # the directions of a call = one ForwardDiff chunk
    c.jac_fd = int(jac_fd)           # Rosenbrock23(autodiff=false): finite-difference J and time derivative (primal solves)
    if solver == 2:
        c.qsteady_max = 1.0          # a composite is not an implicit algorithm type (see solve_one_auto)
    return c


def hychem_p2vec(p, ns=9, nr=10):
    nth, P = nr * (2 * ns + 3), nr * (2 * ns + 3) + 1
    th = np.zeros(nth)
    dth = np.zeros((P, nth))
    lib().orc_hychem_p2vec(_dp(np.ascontiguousarray(p, float)), C.c_int(ns), C.c_int(nr), _dp(th), _dp(dth))
    return th, dth          # dth[k, m] = d theta_m / d p_k


def hychem_rhs(c, theta, u, T, P, Td=0.0, Pd=0.0):
    ns = c.ns
    f = np.zeros(ns); J = np.zeros((ns, ns), order="F"); ft = np.zeros(ns)
    lib().orc_hychem_rhs(C.byref(c), _dp(np.ascontiguousarray(theta, float)), _dp(np.ascontiguousarray(u, float)),
                         C.c_double(T), C.c_double(P), C.c_double(Td), C.c_double(Pd), _dp(f), _dp(J), _dp(ft))
    return f, J, ft


def hychem_solve_one(c, theta, u0, ts, Ttab, Ptab, data, dtheta=None, sample=None, want_pred=False):
    """data [ns, D]; dtheta [ndir, nth] (rows = directions) or None.  Returns dict(loss, grad, pred [ns, D], ...)"""
    ts = np.ascontiguousarray(ts, float)
    Dfull = ts.size
    D = Dfull if sample is None else int(sample)
    ns = c.ns
    ndir = 0 if dtheta is None else int(np.shape(dtheta)[0])
    dth = None if dtheta is None else np.ascontiguousarray(dtheta, float)
    pred = np.zeros((ns, Dfull)) if want_pred else None
    grad = np.zeros(max(ndir, 1))
    loss = C.c_double(0); nsv = C.c_int32(0); st = (C.c_int64 * 2)(0, 0)
    lib().orc_hychem_solve_one.restype = C.c_int
    rc = lib().orc_hychem_solve_one(C.byref(c), _dp(np.ascontiguousarray(theta, float)), _dp(dth), C.c_int(ndir),
                                    _dp(np.ascontiguousarray(u0, float)), _dp(ts), C.c_int(D), C.c_int(Dfull),
                                    _dp(np.ascontiguousarray(Ttab, float)), _dp(np.ascontiguousarray(Ptab, float)),
                                    _dp(np.ascontiguousarray(data, float)), _dp(pred), C.byref(loss), _dp(grad),
                                    C.byref(nsv), C.cast(st, C.c_void_p))
    lib().orc_hychem_tsit5_steps.restype = C.c_int64
    return dict(loss=loss.value, grad=grad[:ndir], pred=pred, retcode=rc, n_saved=nsv.value, naccept=st[0], nreject=st[1],
                n_tsit5=int(lib().orc_hychem_tsit5_steps()))


def svgd_update(p, lnpgrad, stepsize, h=-1.0):
    """SVGD move (network.jl:67-87 svgd_kernel + crnn_cathode.jl:36-50), scalar C loops.
    Returns (p_new, data_term, repulsion, h_used)."""
    p = np.ascontiguousarray(p, float)
    g = np.ascontiguousarray(lnpgrad, float)
    assert p.ndim == 2 and g.shape == p.shape
    pn, dt, rp = np.empty_like(p), np.empty_like(p), np.empty_like(p)
    hout = C.c_double(0.0)
    rc = lib().orc_svgd_update(_dp(p), _dp(g), C.c_int(p.shape[0]), C.c_int(p.shape[1]), C.c_double(stepsize),
                               C.c_double(h), _dp(pn), C.byref(hout), _dp(dt), _dp(rp))
    assert rc == 0, rc
    return pn, dt, rp, hout.value

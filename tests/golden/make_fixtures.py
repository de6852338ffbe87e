#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (run ONCE, in the build
container where /root/reference exists; the outputs are committed, this script
is committed, nothing here runs at test time).

Everything in this file is an *independent* NumPy/SciPy restatement -- it does
not import the C oracle or the product package -- so that the fixtures can pin
both of them:

  * `p` vectors decoded from the reference's own checkpoints
      case2/checkpoint/mymodel.bson      (written by case2/case2.jl:178)
      robertson/checkpoint/mymodel.bson  (written by robertson/rober_crnn.jl:201)
  * p2vec images of those vectors (NumPy transliteration of the formulas at
      case2/case2.jl:91-99 and robertson/rober_crnn.jl:85-96)
  * seeded initial conditions drawn as the reference draws them
      (case2/case2.jl:62-65, robertson/rober_crnn.jl:44-47; NumPy PCG64, not
      Julia's MersenneTwister stream)
  * "true mechanism" data = literal trueODEfunc (case2/case2.jl:38-53,
      robertson/rober_crnn.jl:52-63) integrated with SciPy Radau rtol 1e-12
  * converged CRNN trajectories of the checkpoint networks (Radau 1e-12)
  * converged sensitivities d u(t_j)/d p via the continuous sensitivity ODE
      (directional derivatives of the RHS by complex step), hence converged
      loss gradients
  * the classical Robertson (1,0,0) known answers
  * a 12-step trace of the Flux optimiser chain on a fixed gradient sequence
"""
import json
import os
import sys

import bson
import numpy as np
from scipy.integrate import solve_ivp

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------------------
# BSON.jl decoding (arrays are {tag:"array", type:{name:[..,"Float64"]}, size, data})
# ----------------------------------------------------------------------------
def load_bson(path):
    d = bson.decode(open(path, "rb").read())
    refs = d.get("_backrefs", [])

    def res(x):
        if isinstance(x, dict) and x.get("tag") == "backref":
            return res(refs[x["ref"] - 1])
        return x

    def scalar(v):
        v = res(v)
        if isinstance(v, dict):  # boxed {tag:"struct", type:Float32/64, data:bytes}
            name = res(v["type"])["name"][-1]
            return float(np.frombuffer(v["data"], dtype={"Float32": "<f4", "Float64": "<f8"}[name])[0])
        return float(v)

    def arr(x):
        x = res(x)
        if isinstance(x, list):  # Vector{Any}
            return np.array([scalar(v) for v in x])
        assert x["tag"] == "array", x.get("tag")
        ty = res(x["type"])
        name = ty["name"][-1]
        dt = {"Float64": "<f8", "Float32": "<f4", "Int64": "<i8"}.get(name)
        if dt is None:  # Any-array: list of boxed values
            return np.array([scalar(v) for v in x["data"]])
        return np.frombuffer(x["data"], dtype=dt).reshape(x["size"][::-1]).T.copy()

    return d, arr


# ----------------------------------------------------------------------------
# NumPy restatements (complex-safe so that complex-step derivatives work)
# ----------------------------------------------------------------------------
def cclamp(v, lo, hi):
    re = np.real(v)
    return np.where(re < lo, lo + 0 * v, np.where(re > hi, hi + 0 * v, v))


def p2vec_case2(p, ns=6, nr=3):
    slope = p[nr * (ns + 2)] * 100
    w_b = p[0:nr] * slope
    w_out = p[nr:nr * (ns + 1)].reshape(nr, ns).T  # Julia reshape(.., ns, nr) column-major
    w_in_Ea = np.abs(p[nr * (ns + 1):nr * (ns + 2)] * slope)
    w_in = np.clip(-w_out, 0, 4)
    w_in = np.vstack([w_in, w_in_Ea[None, :]])
    return w_in, w_b, w_out


def p2vec_rober(p, ns=3, nr=6):
    slope = abs(p[-1])
    w_b = p[0:nr] * (10 * slope)
    w_in = p[nr * (ns + 1):nr * (2 * ns + 1)].reshape(nr, ns).T
    w_out = p[nr:nr * (ns + 1)].reshape(nr, ns).T
    w_out = -w_in * (10.0 ** w_out)
    w_in = np.clip(w_in, 0, 2.5)
    return w_in, w_b, w_out


def p2vec_case1(p, ns=5, nr=4):
    w_b = p[0:nr] + (-10.0)
    w_out = p[nr:].reshape(nr, ns).T
    w_in = np.clip(-w_out, 0, 2.5)
    return w_in, w_b, w_out


INV_R = float(np.float32(-1.0) / np.float32(1.98720425864083e-3))   # case2.jl:113: R is a Float32 literal


LB_CASE2 = float(np.float32(1e-6))   # `lb = 1.f-6` (case2/case2.jl:34): a Float32 literal, promoted where it meets Float64s


def crnn_case2(u, w_in, w_b, w_out, lb=LB_CASE2, ub=10.0):
    logX = np.log(cclamp(u[:-1], lb, ub))
    w_in_x = w_in.T @ np.concatenate([logX, [INV_R / u[-1]]])
    return np.concatenate([w_out @ np.exp(w_in_x + w_b), [0.0]])


def crnn_rober(u, w_in, w_b, w_out, dydt_scale, lb=1e-8):
    w_in_x = w_in.T @ np.log(cclamp(u, lb, np.inf))
    return (w_out @ np.exp(w_in_x + w_b)) * dydt_scale


def theta_pack(w_in, w_b, w_out):
    return np.concatenate([w_in.flatten(order="F"), w_b, w_out.flatten(order="F")])


def true_case2(y, k):
    r1 = k[0] * y[0] * y[1]; r2 = k[1] * y[2] * y[1]; r3 = k[2] * y[3] * y[1]
    return np.array([-r1, -r1 - r2 - r3, r1 - r2, r2 - r3, r3, r1 + r2 + r3, 0.0])


def arrhenius(logA, Ea, T):
    return np.exp(logA) * np.exp(Ea * INV_R / T)       # -Ea / R / T with the Float32 R of case2.jl:56


def true_rober(y, k):
    r1 = k[0] * y[0]; r2 = k[1] * y[1] * y[1]; r3 = k[2] * y[1] * y[2]
    return np.array([-r1 + r3, r1 - r2 - r3, r2])


def radau(f, u0, tsave, t0=0.0, rtol=1e-12, atol=1e-14, method="Radau"):
    sol = solve_ivp(lambda t, y: f(y), (t0, tsave[-1]), u0, method=method, t_eval=tsave, rtol=rtol, atol=atol)
    assert sol.success, sol.message
    return sol.y  # [n, nsave]


def sens_solve(rhs_p, p, u0, tsave, t0=0.0, method="DOP853", rtol=1e-11, atol=1e-13, h=1e-30):
    """u(t_j) and S(t_j) = du/dp by integrating the continuous sensitivity ODE
    S_k' = d/d eps f(u + eps S_k, p + eps e_k), evaluated by complex step."""
    n, P = u0.size, p.size

    def aug(t, y):
        u = y[:n]
        S = y[n:].reshape(P, n)
        out = np.empty_like(y)
        out[:n] = np.real(rhs_p(u.astype(complex), p.astype(complex)))
        for k in range(P):
            pk = p.astype(complex); pk[k] += 1j * h
            out[n + k * n:n + (k + 1) * n] = np.imag(rhs_p(u + 1j * h * S[k], pk)) / h
        return out

    y0 = np.concatenate([u0, np.zeros(n * P)])
    sol = solve_ivp(aug, (t0, tsave[-1]), y0, method=method, t_eval=tsave, rtol=rtol, atol=atol)
    assert sol.success, sol.message
    U = sol.y[:n]                      # [n, nsave]
    S = sol.y[n:].reshape(P, n, -1)    # [P, n, nsave]
    return U, S


# complex-safe p2vec (abs / clip replaced)
def cabs(v):
    return np.where(np.signbit(np.real(v)), -v, v)


def p2vec_case2_c(p, ns=6, nr=3):
    slope = p[nr * (ns + 2)] * 100
    w_b = p[0:nr] * slope
    w_out = p[nr:nr * (ns + 1)].reshape(nr, ns).T
    w_in_Ea = cabs(p[nr * (ns + 1):nr * (ns + 2)] * slope)
    w_in = cclamp(-w_out, 0, 4)
    return np.vstack([w_in, w_in_Ea[None, :]]), w_b, w_out


def p2vec_rober_c(p, ns=3, nr=6):
    slope = cabs(p[-1])
    w_b = p[0:nr] * (10 * slope)
    w_in = p[nr * (ns + 1):nr * (2 * ns + 1)].reshape(nr, ns).T
    w_out = p[nr:nr * (ns + 1)].reshape(nr, ns).T
    w_out = -w_in * np.exp(np.log(10.0) * w_out)
    return cclamp(w_in, 0, 2.5), w_b, w_out


def mae_loss_grad(U, S, data, yscale, i_obs, clamp_ub=None):
    """loss = mean |data - pred|/yscale; grad_k = sum w * S_k."""
    pred = U[i_obs]
    Sx = S[:, i_obs, :]
    mask = np.ones_like(pred)
    if clamp_ub is not None:
        mask = ((pred >= -clamp_ub) & (pred <= clamp_ub)).astype(float)
        pred = np.clip(pred, -clamp_ub, clamp_ub)
    r = (data - pred) / yscale[:, None]
    loss = np.mean(np.abs(r))
    w = -np.where(np.signbit(r), -1.0, 1.0) / yscale[:, None] * mask / r.size
    grad = np.einsum("ij,kij->k", w, Sx)
    return loss, grad


def main():
    rng = np.random.Generator(np.random.PCG64(1234))
    out = {}

    # ---------------- checkpoints ----------------
    d2, arr2 = load_bson(f"{REF}/case2/checkpoint/mymodel.bson")
    p_c2 = arr2(d2["p"]).astype(float).ravel()
    dr, arrr = load_bson(f"{REF}/robertson/checkpoint/mymodel.bson")
    p_rb = arrr(dr["p"]).astype(float).ravel()
    assert p_c2.size == 25 and p_rb.size == 43
    out["case2_ckpt"] = dict(p=p_c2.tolist(), iter=int(d2["iter"]),
                             loss_train_last=float(arr2(d2["l_loss_train"]).ravel()[-1]),
                             loss_val_last=float(arr2(d2["l_loss_val"]).ravel()[-1]))
    out["rober_ckpt"] = dict(p=p_rb.tolist(), iter=int(dr["iter"]),
                             loss_train_last=float(arrr(dr["l_loss_train"]).ravel()[-1]),
                             loss_val_last=float(arrr(dr["l_loss_val"]).ravel()[-1]))
    w_in, w_b, w_out = p2vec_case2(p_c2)
    out["case2_ckpt"]["theta"] = theta_pack(w_in, w_b, w_out).tolist()
    w_in_r, w_b_r, w_out_r = p2vec_rober(p_rb)
    out["rober_ckpt"]["theta"] = theta_pack(w_in_r, w_b_r, w_out_r).tolist()

    # ---------------- case2: ICs, data, converged CRNN trajectories, gradients ----------------
    nic = 8
    ns, nr = 6, 3
    u0 = rng.random((nic, ns + 1))
    u0[:, 0:2] = u0[:, 0:2] * 2.0 + 0.2
    u0[:, 2:ns] = 0.0
    u0[:, ns] = u0[:, ns] * 20.0 + 323.0
    tsteps = np.linspace(0.0, 50.0, 50)
    logA = np.array([18.60, 19.13, 7.93]); Ea = np.array([14.54, 14.42, 6.47])
    data = np.zeros((nic, ns, 50)); clean = np.zeros((nic, ns, 50))
    for i in range(nic):
        k = arrhenius(logA, Ea, u0[i, -1])
        y = radau(lambda y: true_case2(y, k), u0[i], tsteps)[:ns]
        clean[i] = y
        data[i] = y + rng.standard_normal(y.shape) * y * 0.05
    yscale = np.max(np.max(data, axis=2) - np.min(data, axis=2) + LB_CASE2, axis=0)
    pred = np.zeros((nic, ns + 1, 50))
    for i in range(nic):
        pred[i] = radau(lambda y: crnn_case2(y, w_in, w_b, w_out), u0[i], tsteps)
    c2 = dict(u0=u0.tolist(), tsteps=tsteps.tolist(), data=data.tolist(), clean=clean.tolist(),
              yscale=yscale.tolist(), pred_ckpt=pred.tolist())

    def rhs_p_case2(u, p):
        return crnn_case2(u, *p2vec_case2_c(p))

    # converged gradient on 3 ICs, at the checkpoint p and at a reference-style init p
    p_init = rng.standard_normal(25) * 0.1
    p_init[0:3] += 0.8; p_init[21:24] += 0.8; p_init[24] = 0.1
    c2["p_init"] = p_init.tolist()
    grads = []
    for tag, pvec in (("ckpt", p_c2), ("init", p_init)):
        for i in range(3):
            U, S = sens_solve(rhs_p_case2, pvec, u0[i], tsteps)
            loss, grad = mae_loss_grad(U, S, data[i], yscale, np.arange(ns), clamp_ub=10.0)
            grads.append(dict(p=tag, ic=i, loss=float(loss), grad=grad.tolist(), pred=U.tolist()))
            print("case2 grad", tag, i, loss, np.linalg.norm(grad), flush=True)
    c2["grads"] = grads
    out["case2"] = c2

    # ---------------- robertson ----------------
    nic_r = 6
    ns_r, nr_r = 3, 6
    u0r = np.zeros((nic_r, 3))
    u0r[:, 1] = 1e-8
    u0r[:, [0, 2]] = rng.random((nic_r, 2)) + 0.5
    tst_r = 10.0 ** np.linspace(0, 5, 40)
    kr = np.array([4e-2, 3e7, 1e4])
    data_r = np.zeros((nic_r, 3, 40))
    for i in range(nic_r):
        y = radau(lambda y: true_rober(y, kr), u0r[i], tst_r, rtol=1e-12, atol=1e-16)
        data_r[i] = y + rng.standard_normal(y.shape) * y * 1e-4
    yscale_r = np.max(np.max(data_r, axis=2) - np.min(data_r, axis=2), axis=0)
    dydt_scale = yscale_r / tst_r[-1]
    pred_r = np.zeros((nic_r, 3, 40))
    for i in range(nic_r):
        pred_r[i] = radau(lambda y: crnn_rober(y, w_in_r, w_b_r, w_out_r, dydt_scale), u0r[i], tst_r, rtol=1e-12, atol=1e-16)
    rb = dict(u0=u0r.tolist(), tsteps=tst_r.tolist(), data=data_r.tolist(), yscale=yscale_r.tolist(),
              dydt_scale=dydt_scale.tolist(), pred_ckpt=pred_r.tolist())

    def rhs_p_rober(u, p):
        return crnn_rober(u, *p2vec_rober_c(p), dydt_scale)

    grads_r = []
    for i in range(2):
        U, S = sens_solve(rhs_p_rober, p_rb, u0r[i], tst_r, method="Radau", rtol=1e-10, atol=1e-14)
        loss, grad = mae_loss_grad(U, S, data_r[i], yscale_r, np.arange(3))
        grads_r.append(dict(p="ckpt", ic=i, loss=float(loss), grad=grad.tolist(), pred=U.tolist()))
        print("rober grad", i, loss, np.linalg.norm(grad), flush=True)
    rb["grads"] = grads_r
    # classical KAT u0 = (1,0,0)
    tk = np.array([0.4, 4.0, 40.0, 400.0])
    yk = radau(lambda y: true_rober(y, kr), np.array([1.0, 0.0, 0.0]), tk, rtol=1e-13, atol=1e-18)
    rb["kat"] = dict(t=tk.tolist(), y=yk.T.tolist())
    out["robertson"] = rb

    # ---------------- case1 (plumbing case) ----------------
    u01 = np.zeros((4, 5)); u01[:, 0:2] = rng.random((4, 2)) + 0.2
    p1 = rng.standard_normal(24) * 0.1
    w1 = p2vec_case1(p1)
    out["case1"] = dict(u0=u01.tolist(), p=p1.tolist(), theta=theta_pack(*w1).tolist())

    # ---------------- optimiser traces ----------------
    def flux_chain(p, grads, eta=0.005, beta=(0.9, 0.999), wd=1e-6, expdecay=None, clipnorm=0.0):
        p = p.copy(); m = np.zeros_like(p); v = np.zeros_like(p); bp = np.array(beta, float)
        ed_eta = expdecay[0] if expdecay else 1.0; ncalls = 0
        tr = []
        for g in grads:
            g = g.copy()
            if clipnorm > 0:
                gn = np.linalg.norm(g)
                if gn > clipnorm:
                    g = g / gn * clipnorm
            if expdecay:
                ncalls += 1
                if ncalls % expdecay[2] == 0:
                    ed_eta = max(ed_eta * expdecay[1], expdecay[3])
                g = g * ed_eta
            m = beta[0] * m + (1 - beta[0]) * g
            v = beta[1] * v + (1 - beta[1]) * g * g
            delta = m / (1 - bp[0]) / (np.sqrt(v / (1 - bp[1])) + 1e-8) * eta
            bp = bp * np.array(beta)
            delta = delta + wd * p
            p = p - delta
            tr.append(p.tolist())
        return tr

    g_seq = rng.standard_normal((12, 25)) * np.logspace(-3, 1, 25)[None, :]
    p0 = rng.standard_normal(25) * 0.1
    out["optim"] = dict(
        p0=p0.tolist(), grads=g_seq.tolist(),
        case2=flux_chain(p0, g_seq, 0.005, (0.9, 0.999), 1e-6, expdecay=(5e-3, 0.5, 5, 1e-4)),
        rober=flux_chain(p0, g_seq, 0.005, (0.9, 0.999), 1e-6, clipnorm=10.0),
        case1=flux_chain(p0, g_seq, 0.001, (0.9, 0.999), 1e-8),
    )

    with open(os.path.join(OUT, "fixtures.json"), "w") as f:
        json.dump(out, f)
    print("wrote", os.path.join(OUT, "fixtures.json"), os.path.getsize(os.path.join(OUT, "fixtures.json")))




# ============================================================================
# Cathode-UQ fixtures (python tests/golden/make_fixtures.py cathode): written to fixtures_cathode.json
#   * observation sets = the reference's own data files Cathode_NCM333_UQ/exp_data/UNCERT_cath_1_{2,5,10,15,20}.csv
#     (col 1 temperature [C], cols 2..101 = 100 noisy HRR replicas), reduced to what the loss needs:
#     times (dataset.jl:19-23), replica mean and mean square per row, de-duplicated as load_exp does (dataset.jl:7-10)
#   * theta = the deterministic initialiser of Cathode/src/network.jl:9-24 without its random part, mapped by
#     Cathode/src/network.jl:27-50 (the trained p_opt the UQ scripts load is not in the reference tree)
#   * golden HRR curves by SciPy Radau (rtol 1e-12) on a NumPy transliteration of crnn! / HRR_getter
#     (Cathode_NCM333_UQ/src_333/network.jl:152-175) and golden loss gradients from the continuous sensitivity ODE
# ============================================================================
def cathode_rhs(u, th, t, beta, lb=1e-16, T0=373.15):
    R = -1.0 / 8.314
    T = T0 + beta / 60.0 * t
    logX = np.log(cclamp(u, lb, 10.0))
    r = np.exp(np.log(T) * th[6:9] + (R / T) * (th[3:6] * 1e5) + th[12:15] * logX + th[0:3])
    du = -r
    du = du + np.array([0 * r[0], th[15] * r[0], th[16] * r[1]])
    return du, r


def cathode_main():
    out = {}
    p = np.array([1, 1, 1, 1.0, 1.1, 1.2, 0, 0, 0, 1, 0.2, 0.3, 1, 1, 1, 1, 1, 0.1])
    slope = p[17] * 10
    th = np.concatenate([np.clip(p[0:3] * slope * 20, 0, 50), np.clip(np.abs(p[3:6]), 0, 3), p[6:9],
                         np.clip(np.abs(p[9:12]) * 100, 10, 300), np.clip(p[12:15], 0.01, 10), np.clip(p[15:17], 0.01, 5)])
    out["theta"] = th.tolist()
    sets = []
    for beta in (2, 5, 10, 15, 20):
        raw = np.loadtxt(f"{REF}/Cathode_NCM333_UQ/exp_data/UNCERT_cath_1_{beta}.csv", delimiter=",")
        _, idx = np.unique(raw[:, 0], return_index=True)
        raw = raw[np.sort(idx)]
        ts = (raw[:, 0] - 100.0) * 60.0 / beta
        d = raw[:, 1:]
        sets.append(dict(beta=float(beta), ts=ts.tolist(), dbar=d.mean(axis=1).tolist(), d2bar=(d * d).mean(axis=1).tolist(),
                         n_replicas=int(d.shape[1])))
    for s in sets:
        ts = np.array(s["ts"]); beta = s["beta"]
        dbar = np.array(s["dbar"]); d2bar = np.array(s["d2bar"])
        u0 = np.array([1.0, 0.0, 0.0])

        def rhs_p(u, thc):
            return cathode_rhs(u, thc, rhs_p.t, beta)[0]

        # non-autonomous: integrate with t as an extra state so that sens_solve (autonomous interface) can be reused
        def rhs_aug(ua, thc):
            du, _ = cathode_rhs(ua[:3], thc, np.real(ua[3]), beta)
            return np.concatenate([du, [1.0 + 0 * ua[3]]])

        U, S = sens_solve(rhs_aug, th, np.concatenate([u0, [ts[0]]]), ts, t0=ts[0], method="Radau", rtol=1e-11, atol=1e-14)
        hrr = np.zeros(ts.size); dh = np.zeros((17, ts.size))
        for i, t in enumerate(ts):
            _, r = cathode_rhs(U[:3, i], th, t, beta)
            hrr[i] = r @ th[9:12]
            for k in range(17):
                thk = th.astype(complex); thk[k] += 1e-30j
                _, rk = cathode_rhs(U[:3, i] + 1e-30j * S[k, :3, i], thk, t, beta)
                dh[k, i] = np.imag(rk @ thk[9:12]) / 1e-30
        e = hrr - dbar
        loss = np.sum(e * e + d2bar - dbar * dbar) / ts.size
        grad = (2 * e[None, :] * dh).sum(axis=1) / ts.size
        s.update(hrr=hrr.tolist(), loss=float(loss), grad=grad.tolist(), u_end=U[:3, -1].tolist())
        print("cathode beta", beta, "D", ts.size, "loss", loss, "|grad|", np.linalg.norm(grad), flush=True)
    out["sets"] = sets
    with open(os.path.join(OUT, "fixtures_cathode.json"), "w") as f:
        json.dump(out, f)
    print("wrote fixtures_cathode.json", os.path.getsize(os.path.join(OUT, "fixtures_cathode.json")))


# ---------------------------------------------------------------------------------------------------------
# HyChem fixtures (python tests/golden/make_fixtures.py hychem): written to fixtures_hychem.json.
# The reference's data file is absent, so conditions come from crnn_amd/hychem.py's synthetic model (an element-
# balanced skeleton in CRNN form); the NumPy restatement of crnn! / p2vec there follows
# HyChem/crnn_pyrolysis_mass.jl:78-131.  Golden values: Radau (rtol 1e-11) trajectories of the CRNN at a perturbed
# parameter vector, its loss, and d loss/d p for a subset of parameters by continuous sensitivities (complex step).
# ---------------------------------------------------------------------------------------------------------
def hychem_main():
    sys.path.insert(0, os.path.join(OUT, "..", ".."))
    from crnn_amd import hychem as hy
    rng = np.random.Generator(np.random.PCG64(2024))
    B = 3
    ts, u0, Tt, Pt = hy.sample_conditions(B, rng)
    th_true = hy.true_theta()
    clean = np.zeros((B, hy.NS, ts.size))
    for b in range(B):
        f = lambda t, u: np.real(hy.crnn(u, th_true, hy.interp(t, ts, Tt[b]), hy.interp(t, ts, Pt[b]), hy.DYDT_SCALE))
        sol = solve_ivp(f, (0, ts[-1]), u0[b], method="Radau", rtol=1e-11, atol=1e-14, t_eval=ts)
        assert sol.success
        clean[b] = sol.y
    data = clean * (1.0 + 0.01 * rng.standard_normal(clean.shape))
    yscale = np.maximum((data.max(axis=2) - data.min(axis=2)).max(axis=0), hy.LB)
    p = hy.true_p() + 0.02 * rng.standard_normal(hy.NP)
    p[-1] = 0.1
    theta = hy.pack_theta(*hy.p2vec(p))
    # subset of parameters for the golden gradient: w_b, w_in_b, w_in_Ea, w_out_raw, w_in_raw entries, slope
    nr, ns = hy.NR, hy.NS
    sub = [0, 2, 5, nr + 1, nr + 5, 2 * nr + 0, 2 * nr + 3, 3 * nr + 0, 3 * nr + 6 + ns * 0, 3 * nr + 4 + ns * 1,
           nr * (ns + 3) + 0, nr * (ns + 3) + 6 + ns * 1, nr * (ns + 3) + 4 + ns * 3, nr * (ns + 3) + 1 + ns * 3, hy.NP - 1]
    out = dict(ts=ts.tolist(), u0=u0.tolist(), Ttab=Tt.tolist(), Ptab=Pt.tolist(), data=data.tolist(), yscale=yscale.tolist(),
               dydt_scale=hy.DYDT_SCALE.tolist(), p=p.tolist(), theta=theta.tolist(), sub=sub, traj=[])
    for b in range(B):
        def rhs_aug(ua, psub):
            pc = p.astype(complex)
            pc[sub] = psub
            thc = hy.pack_theta(*hy.p2vec(pc))
            t = np.real(ua[ns])
            du = hy.crnn(ua[:ns], thc, hy.interp(t, ts, Tt[b]), hy.interp(t, ts, Pt[b]), hy.DYDT_SCALE)
            return np.concatenate([du, [1.0 + 0 * ua[ns]]])
        U, S = sens_solve(rhs_aug, p[sub].copy(), np.concatenate([u0[b], [0.0]]), ts, t0=0.0, method="Radau", rtol=1e-11, atol=1e-14)
        pred = U[:ns]
        rr = (data[b] - pred) / yscale[:, None]
        loss = np.mean(np.abs(rr))
        w = -np.sign(rr) / yscale[:, None] / rr.size
        grad = np.array([(w * S[k, :ns, :]).sum() for k in range(len(sub))])
        out["traj"].append(dict(pred=pred.tolist(), loss=float(loss), grad_sub=grad.tolist()))
        print("hychem", b, "T0", Tt[b, 0], "loss", loss, "|grad_sub|", np.linalg.norm(grad), flush=True)
    with open(os.path.join(OUT, "fixtures_hychem.json"), "w") as f:
        json.dump(out, f)
    print("wrote fixtures_hychem.json", os.path.getsize(os.path.join(OUT, "fixtures_hychem.json")))


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else ""
    sys.exit(cathode_main() if mode == "cathode" else hychem_main() if mode == "hychem" else main())

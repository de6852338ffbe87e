"""CPU definitions of the HyChem pyrolysis script (HyChem/crnn_pyrolysis_mass.jl) -- constants, `p2vec`, `crnn!`,
and a synthetic data model (the reference's data file data/10atm_1300K_0.01.txt is absent from the reference).

    varnames, l_MW                      crnn_pyrolysis_mass.jl:57-58
    p2vec(p)                            :78-90
    Y2density, Y2C, crnn!(du,u,p,t)     :107-131   (T = itpT(t), P = itpP(t): piecewise linear on tsteps, :103-104)
    tsteps                              :38-39     (log-spaced, first point 0)

theta layout = [ w_in ((ns+2) x nr, column-major: rows 0..ns-1 species orders, row ns multiplies -1/(R T), row ns+1
multiplies log T) | w_b (nr) | w_out (ns x nr) ].  NumPy code works on complex arguments (complex-step tests).
"""
from __future__ import annotations

import numpy as np

VARNAMES = ["C10H16", "H2", "CH4", "C2H2", "C2H4", "N2", "C4H81", "H", "CH3"]
MW = np.array([136.238, 2.016, 16.043, 26.038, 28.054, 28.014, 56.108, 1.008, 15.035])
NS, NR = 9, 10
NP = NR * (2 * NS + 3) + 1             # 211
NTH = NR * (NS + 2 + 1 + NS)           # 210
RU = 8.31446261815324e3                # J/(kmol K)   :108
# R = 1.98720425864083f-3 is a Float32 literal and "-1 / R" is evaluated in Float32 before the division by T (:106,128)
INV_R = float(np.float32(-1.0) / np.float32(1.98720425864083e-3))
LB = 1e-8                              # lb = atol :27-28
UB = 10.0


def tsteps(t_end, ntotal=40):
    ts = 10.0 ** np.linspace(np.log10(t_end / 100), np.log10(t_end / 1.01), ntotal)
    ts[0] = 0.0
    return ts


def p2vec(p, ns=NS, nr=NR):
    """-> w_in [(ns+2), nr], w_b [nr], w_out [ns, nr]"""
    p = np.asarray(p)
    slope = p[-1] * 10.0
    w_b = p[0:nr] * slope
    w_in_b = p[nr:2 * nr]
    w_in_Ea = p[2 * nr:3 * nr] * slope
    w_out = p[3 * nr:nr * (ns + 3)].reshape((ns, nr), order="F")
    w_in = p[nr * (ns + 3):nr * (2 * ns + 3)].reshape((ns, nr), order="F")
    w_out = -w_in * 10.0 ** w_out
    w_in = np.vstack([_clamp(w_in, 0.0, 2.5), w_in_Ea[None, :], w_in_b[None, :]])
    return w_in, w_b, w_out


def pack_theta(w_in, w_b, w_out):
    return np.concatenate([np.asarray(w_in).flatten(order="F"), np.asarray(w_b), np.asarray(w_out).flatten(order="F")])


def unpack_theta(th, ns=NS, nr=NR):
    th = np.asarray(th)
    n = ns + 2
    return th[:n * nr].reshape((n, nr), order="F"), th[n * nr:(n + 1) * nr], th[(n + 1) * nr:].reshape((ns, nr), order="F")


def _clamp(x, lo, hi):
    re = np.real(x)
    return np.where(re < lo, lo + 0 * x, np.where(re > hi, hi + 0 * x, x))


def interp(t, ts, tab):
    """LinearInterpolation(tsteps, tab)(t)"""
    i = int(np.clip(np.searchsorted(ts, np.real(t), side="right") - 1, 0, len(ts) - 2))
    return tab[i] + (t - ts[i]) * (tab[i + 1] - tab[i]) / (ts[i + 1] - ts[i])


def crnn(u, theta, T, P, dydt_scale, lb=LB, mw=MW):
    """du of crnn! at temperature T [K], pressure P [Pa]."""
    ns = len(mw)
    w_in, w_b, w_out = unpack_theta(theta, ns, (len(theta)) // (2 * ns + 3))
    Y = _clamp(u, lb, UB)
    density = P / (RU * T * np.sum(Y / mw))
    C = density * (Y / mw) * 1e3
    logX = np.log(_clamp(C, lb, UB))
    x = np.concatenate([logX, [INV_R / T], [np.log(T)]])
    wdot = w_out @ np.exp(w_in.T @ x + w_b)
    return wdot * mw / density * dydt_scale


# ------------------------------------------------------------------ synthetic data model
# An element-balanced 6-step pyrolysis skeleton in CRNN form (orders = reactant stoichiometry), padded to nr = 10:
#   R1 C10H16 -> C4H8 + 3 C2H2 + H2     R2 C4H8 -> 2 C2H4        R3 C2H4 -> C2H2 + H2
#   R4 C2H4 + H2 -> 2 CH3               R5 CH3 + H2 -> CH4 + H   R6 2 H -> H2
_ORD = {0: {0: 1}, 1: {6: 1}, 2: {4: 1}, 3: {4: 1, 1: 1}, 4: {8: 1, 1: 1}, 5: {7: 2}}
_NU = {0: {0: -1, 6: 1, 3: 3, 1: 1}, 1: {6: -1, 4: 2}, 2: {4: -1, 3: 1, 1: 1}, 3: {4: -1, 1: -1, 8: 2},
       4: {8: -1, 1: -1, 2: 1, 7: 1}, 5: {7: -2, 1: 1}}
_LNA = np.array([20.2, 17.6, 15.9, 16.4, 14.8, 3.0])
_EA = np.array([60.0, 52.0, 50.0, 48.0, 42.0, 0.0])     # kcal/mol
_B = np.array([0.0, 0.0, 0.0, 0.0, 0.0, 0.5])


def true_theta():
    w_in = np.zeros((NS + 2, NR)); w_out = np.zeros((NS, NR)); w_b = np.full(NR, -30.0)
    for j in range(6):
        for i, o in _ORD[j].items():
            w_in[i, j] = o
        for i, nu in _NU[j].items():
            w_out[i, j] = nu
        w_in[NS, j] = _EA[j]; w_in[NS + 1, j] = _B[j]; w_b[j] = _LNA[j]
    return pack_theta(w_in, w_b, w_out)


T_END = 1e-2
DYDT_SCALE = np.full(NS, 0.05 / T_END)     # stands in for yscale / t_end (:120), fixed a priori for the synthetic model


def sample_conditions(B, rng, ntotal=40, ramp=0.03):
    """u0 [B, 9] (fuel in N2), T(t), P(t) tables [B, ntotal] (near-isothermal/isobaric with a linear drift)."""
    ts = tsteps(T_END, ntotal)
    u0 = np.full((B, NS), 0.0)
    yf = 0.01 + 0.09 * rng.random(B)
    u0[:, 0] = yf
    u0[:, 5] = 1.0 - yf
    T0 = 1100.0 + 400.0 * rng.random(B)
    P0 = (1.0 + 9.0 * rng.random(B)) * 101325.0
    drift = ramp * (rng.random((B, 2)) - 0.5) * 2
    Ttab = T0[:, None] * (1.0 - np.abs(drift[:, 0:1]) * ts[None, :] / T_END)       # endothermic cooling
    Ptab = P0[:, None] * (1.0 + drift[:, 1:2] * ts[None, :] / T_END)
    return ts, u0, Ttab, Ptab


def true_p():
    """A p with p2vec(p) == true_theta(): reactants carry their order in w_in_raw (w_out = -order), products carry
    -nu in w_in_raw (clamped to order 0, w_out = +nu); slope = 1."""
    w_in, w_b, w_out = unpack_theta(true_theta())
    p = np.zeros(NP)
    p[-1] = 0.1
    p[0:NR] = w_b
    p[NR:2 * NR] = w_in[NS + 1]
    p[2 * NR:3 * NR] = w_in[NS]
    raw_in = np.where(w_out > 0, -w_out, w_in[:NS])
    p[NR * (NS + 3):NR * (2 * NS + 3)] = raw_in.flatten(order="F")
    return p


def init_p(rng):
    """:74-76"""
    p = rng.standard_normal(NP) * 0.1
    p[-1] = 0.1
    return p

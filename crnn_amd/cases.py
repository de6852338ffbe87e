"""Problem constants and synthetic-data generators of the reference scripts
(the "L0 data synthesis" layer: case2/case2.jl:38-89, robertson/rober_crnn.jl:40-82,
case1/case1.jl:27-68).  Random streams are NumPy PCG64, not Julia's
MersenneTwister, so datasets are statistically, not bitwise, the reference's.

The ground-truth mechanisms are mass-action kinetics, i.e. exact instances of
the CRNN form, so the ensemble data is produced by the same gfx950 stepper at
tight tolerance (theta_true below) -- 65 536 SciPy solves would take minutes.
"""
from __future__ import annotations

import numpy as np



R_KCAL = 1.98720425864083e-3           # case2/case2.jl:56
# `inv_R = - 1 / 1.98720425864083f-3` (case2/case2.jl:113): R is a Float32 literal, so the quotient is formed in Float32.
# Its double value differs from -1/R_KCAL by 2e-8 relative -- 4e-7 in a rate at Ea = 14.5 kcal/mol, T = 333 K.
INV_R = float(np.float32(-1.0) / np.float32(R_KCAL))
# `lb = 1.f-5` (case1/case1.jl:34) and `lb = 1.f-6` (case2/case2.jl:34) are Float32 literals; `clamp(u, lb, ub)` and
# `max_min(...) .+ lb` promote them to the Float64 values 9.999999747378752e-06 / 9.999999974752427e-07 (robertson's
# `lb = 1e-8` and HyChem's `lb = atol = 1.e-8` are Float64 literals and stay decimal).
LB_CASE1 = float(np.float32(1e-5))
LB_CASE2 = float(np.float32(1e-6))


def pack_theta(w_in, w_b, w_out):
    return np.concatenate([np.asarray(w_in, float).flatten(order="F"), np.asarray(w_b, float),
                           np.asarray(w_out, float).flatten(order="F")])


# ----------------------------- case 2 ---------------------------------------
CASE2_LOGA = np.array([18.60, 19.13, 7.93])   # case2/case2.jl:52
CASE2_EA = np.array([14.54, 14.42, 6.47])     # case2/case2.jl:53 (kcal/mol)


def case2_tsteps(datasize=50, tstep=1.0):
    return np.linspace(0.0, datasize * tstep, datasize)   # range(0, 50, length=50), case2.jl:66-67


def case2_true_theta():
    """trueODEfunc (case2/case2.jl:38-50) + Arrhenius (:55-59) as CRNN weights.
    TG(1) ROH(2) DG(3) MG(4) GL(5) R'CO2R(6);  r1 = k1 TG ROH, r2 = k2 DG ROH, r3 = k3 MG ROH."""
    w_in = np.zeros((7, 3))
    w_in[[0, 1], 0] = 1.0
    w_in[[2, 1], 1] = 1.0
    w_in[[3, 1], 2] = 1.0
    w_in[6, :] = CASE2_EA                      # x_T = -1/(R T)
    w_out = np.array([[-1, 0, 0], [-1, -1, -1], [1, -1, 0], [0, 1, -1], [0, 0, 1], [1, 1, 1]], float)
    return pack_theta(w_in, CASE2_LOGA, w_out)


def case2_u0(B, rng):
    """u0_list (case2/case2.jl:62-65): TG, ROH ~ U(0.2, 2.2); others 0; T ~ U(323, 343) K."""
    u0 = rng.random((B, 7))
    u0[:, 0:2] = u0[:, 0:2] * 2.0 + 0.2
    u0[:, 2:6] = 0.0
    u0[:, 6] = u0[:, 6] * 20.0 + 323.0
    return u0


def case2_init_p(rng):
    """case2/case2.jl:85-89."""
    p = rng.standard_normal(25) * 0.1
    p[0:3] += 0.8
    p[21:24] += 0.8
    p[24] = 0.1
    return p


# ----------------------------- robertson ------------------------------------
ROBER_K = np.array([4e-2, 3e7, 1e4])           # rober_crnn.jl:52


def rober_tsteps(datasize=40):
    return 10.0 ** np.linspace(0.0, 5.0, datasize)   # rober_crnn.jl:49


def rober_true_theta():
    """trueODEfunc (rober_crnn.jl:56-63): r1 = k1 y1, r2 = k2 y2^2, r3 = k3 y2 y3."""
    w_in = np.array([[1, 0, 0], [0, 2, 1], [0, 0, 1]], float)
    w_out = np.array([[-1, 0, 1], [1, -1, -1], [0, 1, 0]], float)
    return pack_theta(w_in, np.log(ROBER_K), w_out)


def rober_u0(B, rng):
    """rober_crnn.jl:44-47 (uniform instead of Latin-hypercube sampling of y1, y3 in [0.5, 1.5])."""
    u0 = np.zeros((B, 3))
    u0[:, 1] = 1e-8
    u0[:, [0, 2]] = rng.random((B, 2)) + 0.5
    return u0


def rober_init_p(rng):
    """rober_crnn.jl:39-41: Glorot-uniform, slope 0.1."""
    ns, nr = 3, 6
    P = nr * (2 * ns + 1) + 1
    p = (rng.random(P) - 0.5) * 2 * np.sqrt(6 / (ns + nr))
    p[-1] = 0.1
    return p


# ----------------------------- case 1 ---------------------------------------
CASE1_K = np.array([0.1, 0.2, 0.13, 0.3])      # case1/case1.jl:27


def case1_tsteps(datasize=100, tstep=0.4):
    return np.linspace(0.0, datasize * tstep, datasize)


def case1_true_theta():
    """case1/case1.jl:38-44: 2A -> B (k1), A -> C (k2), C -> D (k3), B + D -> E (k4)."""
    w_in = np.zeros((5, 4))
    w_in[0, 0] = 2.0
    w_in[0, 1] = 1.0
    w_in[2, 2] = 1.0
    w_in[[1, 3], 3] = 1.0
    w_out = np.array([[-2, -1, 0, 0], [1, 0, 0, -1], [0, 1, -1, 0], [0, 0, 1, -1], [0, 0, 0, 1]], float)
    return pack_theta(w_in, np.log(CASE1_K), w_out)


def case1_u0(B, rng):
    u0 = np.zeros((B, 5))
    u0[:, 0:2] = rng.random((B, 2)) + 0.2
    return u0


def case1_init_p(rng):
    return rng.standard_normal(24) * 0.1


def max_min(ode_data, lb=0.0):
    """yscale: per species max over experiments of (max_t - min_t + lb)  (case2.jl:71-73,83)."""
    rng_ = ode_data.max(axis=2) - ode_data.min(axis=2) + lb   # [B, ns]
    return rng_.max(axis=0)


def add_noise(clean, noise, rng):
    """ode_data += randn(size) .* ode_data .* noise  (case2.jl:79)."""
    return clean + rng.standard_normal(clean.shape) * clean * noise

// crnn_amd/csrc/hychem_tan.hpp -- tangents of the HyChem right-hand side (HyChem/crnn_pyrolysis_mass.jl:121-131) in closed form and
// real arithmetic, split the way a dual-norm kernel with a COLUMN LOOP shares them.  Not wired into a kernel
// yet: hychem_sens_kernel.hpp evaluates hy_f over Du<Du<double>> with one column per lane; this header is the arithmetic core of its
// successor, compiled for host and device from one source and pinned on the host against the complex step (tests/test_hychem.py).
//
//   Y = clamp(u, lb, ub); S = sum Y / MW; rho = P / (Ru T S); C_m = rho Y_m / MW_m 1e3; x = [log clamp(C, lb, ub); inv_R / T; log T]
//   r_j = exp(w_b[j] + sum_m w_in[m, j] x_m);  f_i = gsc_i / rho  sum_j w_out[i, j] r_j
//
// What a Rosenbrock23 step with ForwardDiff partials asks for, per column (s = du/dp_k, dθ = dθ/dp_k):
//   f'  = d/dε f(u + ε s; θ + ε dθ)                         at three points of a step
//   (J v + τ ∂ₜf)' = d/dε [J(u + ε s; θ + ε dθ) v + τ ∂ₜ f(…)]   at the step's first point, for v = k1, k2 - k1, k3
// Three levels, by what they depend on:
//   HyTanPt   point (u, T, P, Ṫ, Ṗ) and θ            once per trajectory and point   hy_tan_point (= primal part + hy_tan_time)
//   HyTanV    + a primal direction v                   once per trajectory and v       hy_tan_v
//   HyTanCol  + a column (s, dθ): f', ∂ₜf'             once per column and point       hy_tan_col
//   mixed     + both                                   once per column and v           hy_tan_mixed
// Derivative conventions are ForwardDiff's: clamp' = 1 on the closed window, 0 outside.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define CRNN_HD __host__ __device__
#else
#define CRNN_HD
#endif

// On the device every weight is an LDS read; fully unrolled, the scheduler hoists the reads of all reactions to the top of a phase and
// the live set explodes (tools/ubench/hy_tan_probe.hip).  HYT_STEP closes one reaction's / one species' reads before the next one's.
#if !defined(HYT_STEP)
#if defined(__HIP_DEVICE_COMPILE__)
#define HYT_STEP asm volatile("" ::: "memory")
#else
#define HYT_STEP (void)0
#endif
#endif

namespace crnn {

struct HyTanConst {
    double lb, ub, inv_R, Ru;
    const double *imw;   // [NS] 1 / MW
    const double *gsc;   // [NS] MW .* dydt_scale
};

template <int NS, int NR>
struct HyTanLay {        // theta = [w_in (NS + 2) x NR | w_b NR | w_out NS x NR] (hychem_kernel.hpp: LayH)
    static constexpr int NF = NS + 2;
    static constexpr int NTH = NR * (NF + 1 + NS);
    CRNN_HD static constexpr int wi(int m, int j) { return m + NF * j; }
    CRNN_HD static constexpr int wb(int j) { return NF * NR + j; }
    CRNN_HD static constexpr int wo(int i, int j) { return (NF + 1) * NR + i + NS * j; }
};

template <int NS, int NR>
struct HyTanPt {
    double q[NS];        // [lb <= u <= ub] / Y: d log Y / du through the clamp on Y
    double a[NS];        // [lb <= C <= ub]: the clamp on C
    double x[NS + 2];
    double r[NR];        // rates
    double zt[NR];       // d z_j / dt along the table (Ṫ, Ṗ)
    double f[NS];
    double ft[NS];       // ∂ₜ f
    double K[NS];        // gsc / rho
    double iS, ld;       // 1 / S; d log rho / dt = Ṗ/P - Ṫ/T
    double e1, e2;       // d x[NS] / dt, d x[NS + 1] / dt
};

template <int NS, int NR>
struct HyTanV {
    double Lv;           // d log rho [v] = -S_v / S
    double xv[NS];       // d x [v]
    double zv[NR];       // d z [v]
    double Jv[NS];       // J v
};

template <int NS, int NR>
struct HyTanCol {
    double Lp;           // d log rho [s]
    double sq[NS];       // [window] s / Y
    double rp[NR];       // r'
    double ztp[NR];      // zt'
    double fp[NS];       // f'
    double ftp[NS];      // (∂ₜ f)'
};

// the time part of a point whose q, a, x, r, f, K, iS are filled (by hy_tan_point, or from a primal evaluation the caller holds)
template <int NS, int NR>
CRNN_HD inline void hy_tan_time(const double *th, const HyTanConst &k, const double T, const double P, const double Td, const double Pd,
                                HyTanPt<NS, NR> &pt) {
    using L_ = HyTanLay<NS, NR>;
    pt.ld = Pd / P - Td / T;
    pt.e1 = -k.inv_R * Td / (T * T);
    pt.e2 = Td / T;
    for (int j = 0; j < NR; ++j) {
        HYT_STEP;
        double sa = 0.0;
        for (int m = 0; m < NS; ++m) sa += th[L_::wi(m, j)] * pt.a[m];
        pt.zt[j] = th[L_::wi(NS, j)] * pt.e1 + th[L_::wi(NS + 1, j)] * pt.e2 + pt.ld * sa;
    }
    for (int i = 0; i < NS; ++i) {
        HYT_STEP;
        double B = 0.0;
        for (int j = 0; j < NR; ++j) B += th[L_::wo(i, j)] * pt.r[j] * pt.zt[j];
        pt.ft[i] = pt.K[i] * B - pt.f[i] * pt.ld;
    }
}

template <int NS, int NR>
CRNN_HD inline void hy_tan_point(const double *th, const HyTanConst &k, const double *u, const double T, const double P, const double Td,
                                 const double Pd, HyTanPt<NS, NR> &pt) {
    using L_ = HyTanLay<NS, NR>;
    double Y[NS], S = 0.0;
    for (int i = 0; i < NS; ++i) {
        const bool in = u[i] >= k.lb && u[i] <= k.ub;
        Y[i] = u[i] < k.lb ? k.lb : (u[i] > k.ub ? k.ub : u[i]);
        pt.q[i] = in ? 1.0 / Y[i] : 0.0;
        S += Y[i] * k.imw[i];
    }
    const double rho = P / (k.Ru * T * S), irho = 1.0 / rho;
    pt.iS = 1.0 / S;
    for (int m = 0; m < NS; ++m) {
        const double C = rho * (Y[m] * k.imw[m]) * 1e3;
        pt.a[m] = (C >= k.lb && C <= k.ub) ? 1.0 : 0.0;
        pt.x[m] = log(C < k.lb ? k.lb : (C > k.ub ? k.ub : C));
    }
    pt.x[NS] = k.inv_R / T;
    pt.x[NS + 1] = log(T);
    for (int j = 0; j < NR; ++j) {
        HYT_STEP;
        double z = th[L_::wb(j)];
        for (int m = 0; m < NS + 2; ++m) z += th[L_::wi(m, j)] * pt.x[m];
        pt.r[j] = exp(z);
    }
    for (int i = 0; i < NS; ++i) {
        HYT_STEP;
        double om = 0.0;
        for (int j = 0; j < NR; ++j) om += th[L_::wo(i, j)] * pt.r[j];
        pt.K[i] = k.gsc[i] * irho;
        pt.f[i] = pt.K[i] * om;
    }
    hy_tan_time<NS, NR>(th, k, T, P, Td, Pd, pt);
}

// J v at the point (the primal Rosenbrock stages need it anyway) and what the mixed derivative reuses of it
template <int NS, int NR>
CRNN_HD inline void hy_tan_v(const double *th, const HyTanPt<NS, NR> &pt, const HyTanConst &k, const double *v, HyTanV<NS, NR> &pv) {
    using L_ = HyTanLay<NS, NR>;
    double Sv = 0.0;
    for (int i = 0; i < NS; ++i) Sv += (pt.q[i] != 0.0 ? v[i] : 0.0) * k.imw[i];
    pv.Lv = -Sv * pt.iS;
    for (int m = 0; m < NS; ++m) pv.xv[m] = pt.a[m] * (pv.Lv + v[m] * pt.q[m]);
    for (int j = 0; j < NR; ++j) {
        HYT_STEP;
        double zv = 0.0;
        for (int m = 0; m < NS; ++m) zv += th[L_::wi(m, j)] * pv.xv[m];
        pv.zv[j] = zv;
    }
    for (int i = 0; i < NS; ++i) {
        HYT_STEP;
        double A = 0.0;
        for (int j = 0; j < NR; ++j) A += th[L_::wo(i, j)] * pt.r[j] * pv.zv[j];
        pv.Jv[i] = pt.K[i] * A - pt.f[i] * pv.Lv;
    }
}

// f' and (∂ₜ f)' of a column at the point
template <int NS, int NR>
CRNN_HD inline void hy_tan_col(const double *th, const double *dth, const HyTanPt<NS, NR> &pt, const HyTanConst &k, const double *s,
                               HyTanCol<NS, NR> &c) {
    using L_ = HyTanLay<NS, NR>;
    double Sp = 0.0;
    for (int i = 0; i < NS; ++i) {
        c.sq[i] = s[i] * pt.q[i];
        Sp += (pt.q[i] != 0.0 ? s[i] : 0.0) * k.imw[i];
    }
    c.Lp = -Sp * pt.iS;
    for (int j = 0; j < NR; ++j) {
        HYT_STEP;
        double z = dth[L_::wb(j)], dsa = 0.0;
        for (int m = 0; m < NS + 2; ++m) z += dth[L_::wi(m, j)] * pt.x[m];
        for (int m = 0; m < NS; ++m) {
            z += th[L_::wi(m, j)] * (pt.a[m] * (c.Lp + c.sq[m]));
            dsa += dth[L_::wi(m, j)] * pt.a[m];
        }
        c.rp[j] = pt.r[j] * z;
        c.ztp[j] = dth[L_::wi(NS, j)] * pt.e1 + dth[L_::wi(NS + 1, j)] * pt.e2 + pt.ld * dsa;
    }
    for (int i = 0; i < NS; ++i) {
        HYT_STEP;
        double omp = 0.0, B = 0.0, Bp = 0.0;
        for (int j = 0; j < NR; ++j) {
            const double w = th[L_::wo(i, j)], dw = dth[L_::wo(i, j)];
            omp += dw * pt.r[j] + w * c.rp[j];
            B += w * pt.r[j] * pt.zt[j];
            Bp += dw * pt.r[j] * pt.zt[j] + w * (c.rp[j] * pt.zt[j] + pt.r[j] * c.ztp[j]);
        }
        c.fp[i] = pt.K[i] * omp - pt.f[i] * c.Lp;
        c.ftp[i] = pt.K[i] * (Bp - B * c.Lp) - c.fp[i] * pt.ld;
    }
}

// (J v)' of a column for a primal direction; the caller adds tau * c.ftp
template <int NS, int NR>
CRNN_HD inline void hy_tan_mixed(const double *th, const double *dth, const HyTanPt<NS, NR> &pt, const HyTanV<NS, NR> &pv,
                                 const HyTanCol<NS, NR> &c, const double *v, double *out) {
    using L_ = HyTanLay<NS, NR>;
    const double Lvp = pv.Lv * c.Lp;
    double xvp[NS], zvp[NR];
    for (int m = 0; m < NS; ++m) xvp[m] = pt.a[m] * (Lvp - v[m] * c.sq[m] * pt.q[m]);
    for (int j = 0; j < NR; ++j) {
        HYT_STEP;
        double z = 0.0;
        for (int m = 0; m < NS; ++m) z += dth[L_::wi(m, j)] * pv.xv[m] + th[L_::wi(m, j)] * xvp[m];
        zvp[j] = z;
    }
    for (int i = 0; i < NS; ++i) {
        HYT_STEP;
        double A = 0.0, Ap = 0.0;
        for (int j = 0; j < NR; ++j) {
            const double w = th[L_::wo(i, j)], dw = dth[L_::wo(i, j)];
            A += w * pt.r[j] * pv.zv[j];
            Ap += dw * pt.r[j] * pv.zv[j] + w * (c.rp[j] * pv.zv[j] + pt.r[j] * zvp[j]);
        }
        out[i] = pt.K[i] * (Ap - A * c.Lp) - c.fp[i] * pv.Lv - pt.f[i] * Lvp;
    }
}

}  // namespace crnn

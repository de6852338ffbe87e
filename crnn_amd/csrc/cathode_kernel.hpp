// crnn_amd/csrc/cathode_kernel.hpp -- Bayesian cathode CRNN (BASELINE config 5) on gfx950.
//
// Reference (paths under Cathode_NCM333_UQ/src_333/):
//   crnn!           network.jl:152-165   r_j = exp(b_j log T - Ea_j 1e5/(8.314 T) + n_j log(clamp(u_j, lb, 10)) + lnA_j),
//                                        du = -r;  du_2 += nu_2 r_1;  du_3 += nu_3 r_2;   T = T0 + beta/60 t (:143-149)
//   HRR_getter      network.jl:167-175   hrr(t_i) = sum_j r_j(t_i, u(t_i)) dH_j
//   pred_n_ode      network.jl:196-218   u0 = (1,0,0), tspan = [ts[1], ts[end]], saveat = ts
//   loss_neuralode  network.jl:262-266   sum_{i,k} (hrr_i - data_ik)^2 / n_replicas / D
//   dlnprob         network.jl:222-260   per particle: loss and ForwardDiff.gradient of it
// theta (17 per particle, already multiplied by p_scales): [lnA(3) | Ea(3) | b(3) | dH(3) | n(3) | nu2, nu3].
//
// Mapping: the system is 3 x 3 with a lower-bidiagonal Jacobian, every particle has its own theta, and one
// particle is integrated for many heating rates -> one LANE per (particle, heating-rate) trajectory, no primal
// redundancy at all: W = I - gam*J is lower triangular (forward substitution, no pivoting), the 14 tangent
// columns that move the ODE (dH only enters the observable) are one-hot directions whose structure is resolved at
// compile time, S (14 x 3) lives in registers.  Lanes are persistent (global work queue), the replica statistics
// mean_k data_ik and mean_k data_ik^2 are all the loss needs and are staged per observation set in LDS.
// Stepper: non-autonomous Rosenbrock23 (the reference uses AutoTsit5(TRBDF2): same tolerances, different
// algorithm -- results agree to solver tolerance).
#pragma once
#include "ros23_kernel.hpp"

namespace crnn {

constexpr int kCathNP = 17;       // parameters per particle
constexpr int kCathNC = 14;       // tangent columns that act on the ODE
constexpr int kCathMaxD = 128;    // max rows of an observation set
constexpr int kCathMaxSets = 8;   // observation sets staged in LDS

struct CathodeParams {
    const double *theta;     // [n_part][17]
    const double *ts;        // [n_sets][Dmax]
    const double *dbar;      // [n_sets][Dmax]
    const double *d2bar;     // [n_sets][Dmax]
    const double *beta;      // [n_sets]  K/min
    const int32_t *D;        // [n_sets]
    double *loss;            // [n_traj]
    double *grad;            // [n_traj][17] or null
    double *hrr;             // [n_traj][Dmax] or null
    int32_t *retcode, *n_saved, *n_accept, *n_reject;   // [n_traj]
    unsigned long long *queue;
    int64_t n_traj;          // n_part * n_sets, trajectory tr = particle * n_sets + set
    int32_t n_sets, Dmax, maxiters, want_grad;
    double lb, T0, atol, rtol;
    double gamma, qmin, qmax, beta1, beta2, qsteady_min, qsteady_max, qoldinit;
};

// ODE-column k -> theta index (dH columns 9..11 are not ODE columns)
__device__ __forceinline__ constexpr int cath_col_theta(int k) { return k < 9 ? k : k + 3; }

struct CathPoint {   // everything the RHS / tangents need at one (u, t)
    double l[3], g[3], r[3];   // log clamp(u), d log/du, rates
    double lt, rt, it;         // log T, R/T, 1/T
};

__device__ __forceinline__ void cath_point(const double (&u)[3], double T, const double (&th)[kCathNP], double lb, CathPoint &p) {
    constexpr double Rg = -1.0 / 8.314;
    p.it = frcp(T);
    p.lt = flog(T);
    p.rt = Rg * p.it;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double uj = u[j];
        const bool inside = (uj >= lb) && (uj <= 10.0);
        p.l[j] = flog(clampv(uj, lb, 10.0));
        p.g[j] = inside ? frcp(uj) : 0.0;
        p.r[j] = exp(fma(th[6 + j], p.lt, fma(th[3 + j] * 1e5, p.rt, fma(th[12 + j], p.l[j], th[j]))));
    }
}

__device__ __forceinline__ void cath_f(const CathPoint &p, const double (&th)[kCathNP], double (&f)[3]) {
    f[0] = -p.r[0];
    f[1] = fma(th[15], p.r[0], -p.r[1]);
    f[2] = fma(th[16], p.r[1], -p.r[2]);
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void cathode_kernel(const CathodeParams prm) {
    __shared__ double ts_s[kCathMaxSets * kCathMaxD];
    __shared__ double db_s[kCathMaxSets * kCathMaxD];
    __shared__ double d2_s[kCathMaxSets * kCathMaxD];
    const int tid = threadIdx.x;
    // up to kCathMaxSets observation sets (the reference's five heating rates) are staged in LDS; larger ensembles of
    // heating rates (BASELINE config 5: 256) are read in place from HBM/L2 (rows of <= 1 KB, shared by all particles)
    const bool staged = prm.n_sets <= kCathMaxSets;
    if (staged) {
        for (int idx = tid; idx < prm.n_sets * prm.Dmax; idx += BLOCK) {
            const int s = idx / prm.Dmax, i = idx - s * prm.Dmax;
            ts_s[s * kCathMaxD + i] = prm.ts[idx];
            db_s[s * kCathMaxD + i] = prm.dbar[idx];
            d2_s[s * kCathMaxD + i] = prm.d2bar[idx];
        }
    }
    __syncthreads();

    constexpr double d_ = 0.29289321881345248, c32 = 7.4142135623730950, inv12d = 2.4142135623730950;
    constexpr double Rg = -1.0 / 8.314;
    const double lqinit = flog(prm.qoldinit);
    const int64_t nthreads = (int64_t)gridDim.x * BLOCK;
    int64_t traj = (int64_t)blockIdx.x * BLOCK + tid;
    int64_t traj_next = (int64_t)atomicAdd(prm.queue, 1ULL) + nthreads;

    double th[kCathNP];
    double u[3], f0[3];
    CathPoint P0;
    double S[kCathNC][3], gS[kCathNC], gD[kCathNP];
    double t = 0.0, dt = 0.0, lqold = 0.0, loss_sum = 0.0, Tdot = 0.0, tend = 0.0, t0 = 0.0;
    const double *tsv = ts_s, *dbv = db_s, *d2v = d2_s;
    int iter = 0, jsave = 0, nacc = 0, nrej = 0, D = 1;
    bool need_init = true;

    // HRR observable at a save point: loss term and gradient seeds
    //   w_j = 2 e dH_j r_j n_j g_j (acts on the state tangent), direct theta terms go to gD
    auto observe = [&](const double (&uu)[3], double tt, double (&w)[3]) {
        CathPoint q;
        cath_point(uu, fma(Tdot, tt - 0.0, prm.T0), th, prm.lb, q);
        const double hv = fma(q.r[0], th[9], fma(q.r[1], th[10], q.r[2] * th[11]));
        CRNN_CHK(jsave >= 0 && jsave < D && D <= prm.Dmax && traj < prm.n_traj, 60);
        const double db = dbv[jsave];
        const double e = hv - db;
        loss_sum += fma(e, e, d2v[jsave] - db * db);
        if (prm.hrr) prm.hrr[(size_t)traj * prm.Dmax + jsave] = hv;
        const double e2 = 2.0 * e;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double c = e2 * th[9 + j] * q.r[j];      // 2 e dH_j r_j
            gD[j] += c;                                    // d/d lnA_j
            gD[3 + j] = fma(c * 1e5, q.rt, gD[3 + j]);     // d/d Ea_j
            gD[6 + j] = fma(c, q.lt, gD[6 + j]);           // d/d b_j
            gD[9 + j] = fma(e2, q.r[j], gD[9 + j]);        // d/d dH_j
            gD[12 + j] = fma(c, q.l[j], gD[12 + j]);       // d/d n_j
            w[j] = c * th[12 + j] * q.g[j];
        }
    };

    while (true) {
        if (need_init) {
            if (traj >= prm.n_traj) break;
            need_init = false;
            const int64_t part = traj / prm.n_sets;
            const int set = (int)(traj - part * prm.n_sets);
#pragma unroll
            for (int k = 0; k < kCathNP; ++k) th[k] = prm.theta[(size_t)part * kCathNP + k];
            D = prm.D[set];
            if (staged) { tsv = ts_s + set * kCathMaxD; dbv = db_s + set * kCathMaxD; d2v = d2_s + set * kCathMaxD; }
            else { tsv = prm.ts + (size_t)set * prm.Dmax; dbv = prm.dbar + (size_t)set * prm.Dmax; d2v = prm.d2bar + (size_t)set * prm.Dmax; }
            Tdot = prm.beta[set] * (1.0 / 60.0);
            t0 = tsv[0];
            tend = tsv[D - 1];
            t = t0;
            u[0] = 1.0; u[1] = 0.0; u[2] = 0.0;          // network.jl:186-187
#pragma unroll
            for (int k = 0; k < kCathNC; ++k) { S[k][0] = 0.0; S[k][1] = 0.0; S[k][2] = 0.0; gS[k] = 0.0; }
#pragma unroll
            for (int k = 0; k < kCathNP; ++k) gD[k] = 0.0;
            loss_sum = 0.0; iter = 0; jsave = 0; nacc = 0; nrej = 0;
            lqold = lqinit;
            cath_point(u, fma(Tdot, t, prm.T0), th, prm.lb, P0);
            cath_f(P0, th, f0);
            {   // Hairer initial step, order 2
                double sk[3], d0 = 0.0, d1 = 0.0, d2 = 0.0, u1[3], f1[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    sk[i] = frcp(fma(fabs(u[i]), prm.rtol, prm.atol));
                    d0 = fma(u[i] * sk[i], u[i] * sk[i], d0);
                    d1 = fma(f0[i] * sk[i], f0[i] * sk[i], d1);
                }
                d0 = sqrt(d0 * (1.0 / 3.0)); d1 = sqrt(d1 * (1.0 / 3.0));
                const double dtmax = tend - t0;
                double dt0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * (d0 / d1);
                dt0 = fmin(dt0, dtmax);
#pragma unroll
                for (int i = 0; i < 3; ++i) u1[i] = fma(dt0, f0[i], u[i]);
                CathPoint q;
                cath_point(u1, fma(Tdot, t + dt0, prm.T0), th, prm.lb, q);
                cath_f(q, th, f1);
#pragma unroll
                for (int i = 0; i < 3; ++i) { const double e = (f1[i] - f0[i]) * sk[i]; d2 = fma(e, e, d2); }
                d2 = sqrt(d2 * (1.0 / 3.0)) / dt0;
                const double dm = fmax(d1, d2);
                const double dt1 = (dm <= 1e-15) ? fmax(1e-6, dt0 * 1e-3) : exp(-0.5 * (4.605170185988091368 + flog(dm)));
                dt = fmin(fmin(100.0 * dt0, dt1), dtmax);
            }
            {   // saveat contains tspan[1]
                double w[3];
                observe(u, t0, w);    // tangents are zero at t0: the state seed w is unused, direct terms are kept
                jsave = 1;
            }
        }

        int rc = -1;
        ++iter;
        bool last = false;
        if (jsave >= D) rc = 0;
        else if (iter > prm.maxiters) rc = 1;
        if (t + dt * (1.0 + 1e-13) >= tend) { dt = tend - t; last = true; }
        if (rc < 0 && (!(dt > 0.0) || t + dt == t)) rc = 2;

        if (rc < 0) {
            const double gam = d_ * dt;
            // point-0 quantities: a_j = dr_j/du_j, rho_j = dr_j/dt
            double a[3], sig[3], rho[3], iw[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                a[j] = P0.r[j] * th[12 + j] * P0.g[j];
                sig[j] = (th[6 + j] * P0.it - th[3 + j] * 1e5 * Rg * P0.it * P0.it) * Tdot;
                rho[j] = P0.r[j] * sig[j];
                iw[j] = frcp(fma(gam, a[j], 1.0));
            }
            const double l21 = gam * th[15] * a[0], l32 = gam * th[16] * a[1];   // -W[2][1], -W[3][2]
            auto wsolve = [&](double (&b)[3]) {
                b[0] *= iw[0];
                b[1] = fma(l21, b[0], b[1]) * iw[1];
                b[2] = fma(l32, b[1], b[2]) * iw[2];
            };
            double ft[3] = {-rho[0], fma(th[15], rho[0], -rho[1]), fma(th[16], rho[1], -rho[2])};
            double k1[3], dk[3], k3[3], u1[3], f1[3], unew[3], f2[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) k1[i] = fma(gam, ft[i], f0[i]);
            wsolve(k1);
#pragma unroll
            for (int i = 0; i < 3; ++i) u1[i] = fma(0.5 * dt, k1[i], u[i]);
            CathPoint P1, P2;
            cath_point(u1, fma(Tdot, t + 0.5 * dt, prm.T0), th, prm.lb, P1);
            cath_f(P1, th, f1);
#pragma unroll
            for (int i = 0; i < 3; ++i) dk[i] = f1[i] - k1[i];
            wsolve(dk);
#pragma unroll
            for (int i = 0; i < 3; ++i) unew[i] = fma(dt, k1[i] + dk[i], u[i]);
            const double tnew = last ? tend : t + dt;
            cath_point(unew, fma(Tdot, tnew, prm.T0), th, prm.lb, P2);
            cath_f(P2, th, f2);
            double es = 0.0;
            bool finite = true;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double k2i = k1[i] + dk[i];
                k3[i] = f2[i] - c32 * (k2i - f1[i]) - 2.0 * (k1[i] - f0[i]) + dt * ft[i];
            }
            wsolve(k3);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double k2i = k1[i] + dk[i];
                const double ev = dt * (1.0 / 6.0) * (k1[i] - 2.0 * k2i + k3[i]);
                const double m = fmax(fabs(u[i]), fabs(unew[i]));
                const double e = ev * frcp(fma(prm.rtol, m, prm.atol));
                es = fma(e, e, es);
                finite = finite && isfinite(unew[i]) && isfinite(ev);
            }
            es *= (1.0 / 3.0);
            if (!finite) rc = 3;
            else {
                const bool ee_zero = (es == 0.0);
                const double lEE = 0.5 * flog(ee_zero ? 1.0 : es);
                const double lq11 = prm.beta1 * lEE;
                double q = ee_zero ? 1.0 / prm.qmax
                                   : fmax(1.0 / prm.qmax, fmin(1.0 / prm.qmin, exp(lq11 - prm.beta2 * lqold) / prm.gamma));
                if (es <= 1.0) {
                    ++nacc;
                    // ---- save points: HRR observable, loss, seeds A, B1, B2 for the state tangents ----
                    double A_[3] = {0.0, 0.0, 0.0}, B1[3] = {0.0, 0.0, 0.0}, B2[3] = {0.0, 0.0, 0.0};
                    while (jsave < D) {
                        const double tsj = tsv[jsave];
                        if (!(tsj <= tnew)) break;
                        const bool at_end = (tsj == tnew);
                        const double Th = at_end ? 1.0 : (tsj - t) / dt;
                        const double c1 = at_end ? 0.0 : Th * (1.0 - Th) * inv12d;
                        const double c2 = at_end ? 1.0 : Th * (Th - 2.0 * d_) * inv12d;
                        double ui[3], w[3];
#pragma unroll
                        for (int i = 0; i < 3; ++i) ui[i] = at_end ? unew[i] : fma(dt, fma(c1, k1[i], c2 * (k1[i] + dk[i])), u[i]);
                        observe(ui, tsj, w);
#pragma unroll
                        for (int i = 0; i < 3; ++i) {
                            A_[i] += w[i];
                            B1[i] = fma(w[i], dt * c1, B1[i]);
                            B2[i] = fma(w[i], dt * c2, B2[i]);
                        }
                        ++jsave;
                    }
                    // ---- forward tangents: 14 one-hot directions, structure folded at compile time ----
                    if (prm.want_grad) {
#pragma unroll
                        for (int k = 0; k < kCathNC; ++k) {
                            const int m = cath_col_theta(k);           // theta index of this column (compile-time)
                            const int grp = m / 3, j0 = m % 3;         // 0 lnA, 1 Ea, 2 b, 4 n ; 5 -> nu (m = 15, 16)
                            const bool is_nu2 = (m == 15), is_nu3 = (m == 16);
                            double(&s)[3] = S[k];
                            // dz_j (direct part) at a point with (lt, rt, l)
                            auto dzdir = [&](const CathPoint &p, int j) -> double {
                                if (m >= 15 || j != j0) return 0.0;
                                return grp == 0 ? 1.0 : grp == 1 ? 1e5 * p.rt : grp == 2 ? p.lt : p.l[j];
                            };
                            auto fprime = [&](const CathPoint &p, const double (&ss)[3], double (&rp)[3], double (&fp)[3]) {
#pragma unroll
                                for (int j = 0; j < 3; ++j) rp[j] = p.r[j] * fma(th[12 + j] * p.g[j], ss[j], dzdir(p, j));
                                fp[0] = -rp[0];
                                fp[1] = fma(th[15], rp[0], -rp[1]) + (is_nu2 ? p.r[0] : 0.0);
                                fp[2] = fma(th[16], rp[1], -rp[2]) + (is_nu3 ? p.r[1] : 0.0);
                            };
                            double rp0[3], f0p[3];
                            fprime(P0, s, rp0, f0p);
                            // a'_j, rho'_j at point 0
                            double ap[3], rhop[3];
#pragma unroll
                            for (int j = 0; j < 3; ++j) {
                                const double dn = (grp == 4 && j == j0 && m < 15) ? 1.0 : 0.0;
                                ap[j] = fma(rp0[j], th[12 + j] * P0.g[j], P0.r[j] * P0.g[j] * (dn - th[12 + j] * P0.g[j] * s[j]));
                                const double dsig = (m < 15 && j == j0) ? (grp == 2 ? P0.it * Tdot : grp == 1 ? -1e5 * Rg * P0.it * P0.it * Tdot : 0.0) : 0.0;
                                rhop[j] = fma(rp0[j], sig[j], P0.r[j] * dsig);
                            }
                            const double ftp[3] = {-rhop[0], fma(th[15], rhop[0], -rhop[1]) + (is_nu2 ? rho[0] : 0.0),
                                                   fma(th[16], rhop[1], -rhop[2]) + (is_nu3 ? rho[1] : 0.0)};
                            auto jprime = [&](const double (&v)[3], double (&o)[3]) {
                                o[0] = -ap[0] * v[0];
                                o[1] = fma(th[15], ap[0] * v[0], -ap[1] * v[1]) + (is_nu2 ? a[0] * v[0] : 0.0);
                                o[2] = fma(th[16], ap[1] * v[1], -ap[2] * v[2]) + (is_nu3 ? a[1] * v[1] : 0.0);
                            };
                            double jk[3], k1p[3], dkp[3], s1[3], rp1[3], f1p[3];
                            jprime(k1, jk);
#pragma unroll
                            for (int i = 0; i < 3; ++i) k1p[i] = fma(gam, ftp[i] + jk[i], f0p[i]);
                            wsolve(k1p);
#pragma unroll
                            for (int i = 0; i < 3; ++i) s1[i] = fma(0.5 * dt, k1p[i], s[i]);
                            fprime(P1, s1, rp1, f1p);
                            jprime(dk, jk);
#pragma unroll
                            for (int i = 0; i < 3; ++i) dkp[i] = fma(gam, jk[i], f1p[i] - k1p[i]);
                            wsolve(dkp);
                            double acc = 0.0;
#pragma unroll
                            for (int i = 0; i < 3; ++i) {
                                const double k2p = k1p[i] + dkp[i];
                                acc = fma(A_[i], s[i], acc);
                                acc = fma(B1[i], k1p[i], acc);
                                acc = fma(B2[i], k2p, acc);
                                s[i] = fma(dt, k2p, s[i]);
                            }
                            gS[k] += acc;
                        }
                    }
                    // ---- advance (FSAL) ----
#pragma unroll
                    for (int i = 0; i < 3; ++i) { u[i] = unew[i]; f0[i] = f2[i]; }
                    P0 = P2;
                    t = tnew;
                    if (q >= prm.qsteady_min && q <= prm.qsteady_max) q = 1.0;
                    lqold = ee_zero ? lqinit : fmax(lEE, lqinit);
                    dt = fmin(dt / q, tend - t0);
                    if (jsave >= D) rc = 0;
                } else {
                    ++nrej;
                    dt = dt / fmin(1.0 / prm.qmin, exp(lq11) / prm.gamma);
                }
            }
        }

        if (rc >= 0) {
            // loss = sum(...)/n_replicas/size(exp_data)[1]: the FULL row count, also for a truncated solution (network.jl:266)
            const double invD = 1.0 / (double)D;
            prm.loss[traj] = loss_sum * invD;
            prm.retcode[traj] = rc;
            prm.n_saved[traj] = jsave;
            prm.n_accept[traj] = nacc;
            prm.n_reject[traj] = nrej;
            if (prm.grad) {
                double *go = prm.grad + (size_t)traj * kCathNP;
#pragma unroll
                for (int m = 0; m < kCathNP; ++m) go[m] = gD[m] * invD;
                if (prm.want_grad) {
#pragma unroll
                    for (int k = 0; k < kCathNC; ++k) {
                        const int m = cath_col_theta(k);
                        go[m] = (gD[m] + gS[k]) * invD;
                    }
                }
            }
            traj = traj_next;
            traj_next = (int64_t)atomicAdd(prm.queue, 1ULL) + nthreads;
            need_init = true;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Per-particle gradient by the discrete adjoint of the accepted steps (the method of ros23_adj_kernel.hpp applied to the
// cathode right-hand side).  f = S r with S = [[-1,0,0],[nu2,-1,0],[0,nu3,-1]], r_j = exp(z_j),
// z_j = lnA_j + Ea_j 1e5 Rg/T + b_j ln T + n_j log clamp(u_j);  for a direction (k, tau) in (u, t):
// z'_j = n_j g_j k_j + tau sigma_j, sigma_j = (b_j/T - Ea_j 1e5 Rg/T^2) dT/dt, and with A = S^T a, Psi_j = A_j r_j:
//     d(a.f)/d(lnA, Ea, b, n)_j = Psi_j (1, 1e5 Rg/T, ln T, l_j),  d/d nu2 = a_1 r_0,  d/d nu3 = a_2 r_1,  d/du_j = Psi_j n_j g_j
//     d(a.Df[(k,tau)])/d lnA_j = Psi_j y_j (y = z'),  d/d Ea_j = Psi_j (y_j 1e5 Rg/T - tau 1e5 Rg T'/T^2),
//     d/d b_j = Psi_j (y_j ln T + tau T'/T),  d/d n_j = Psi_j (y_j l_j + g_j k_j),  d/d nu2 = a_1 r_0 y_0,  d/d nu3 = a_2 r_1 y_1,
//     d/du_j = Psi_j n_j (y_j g_j - g_j^2 k_j)
// The time derivative df/dt of the non-autonomous Rosenbrock23 rides along as tau = 1 of the w-direction.  Cost: two
// primal sweeps instead of 1 + 14 tangent columns.  Work is handed out as 64 PARTICLES OF ONE HEATING RATE per wavefront
// (nearly identical step counts: the particles are a tight cloud), so the wave-synchronous sweeps lose little to
// divergence.  tape: (t, dt, u[3]) per accepted step and lane.
// ---------------------------------------------------------------------------------------------------------------
struct CathAdjParams {
    double *tape;              // [lanes][tape_cap][5]
    int32_t tape_cap;
    unsigned int *overflow;
    int64_t n_part;
};

#ifndef CRNN_CATH_ADJ_WAVES
#define CRNN_CATH_ADJ_WAVES 1   // waves per SIMD the register allocation targets (tools/kvariants.sh experiments)
#endif
// KCP > 1 (round 3): CHECKPOINTED tape.  The forward sweep records dt of every accepted step (8 B) and (t, u) of every KCP-th
// (32 B) instead of (t, dt, u) of every step (40 B); the reverse sweep works through the steps in blocks of KCP: from the
// block's checkpoint it forms the states of the block's steps again (KCP - 1 forward re-formations, parked per lane in LDS),
// then reverses them.  Same arithmetic as the forward sweep, so the re-formed states are the recorded ones; the tape
// shrinks from 40 to 8 + 32 / KCP bytes per step (KCP = 4: 16 B, 2.5x less HBM traffic) for KCP - 1 extra re-formations
// per KCP steps.
// The lane's 17 gradient accumulators: registers, or -- full-tape kernel -- a column of LDS (pitch 17, odd: conflict-free).  In LDS
// they cost a read-modify-write per update (same arithmetic, same bits) and free 34 registers: with them the full-tape kernel is
// built for two wavefronts per SIMD (256 registers; 61 spilled values left, none of them touched more than twice per step) and a
// CU holds two blocks: 4 096 x 256 trajectories 35.8 -> 32.1 ms.  With one wavefront per SIMD the LDS version is slower (37.8), and
// the checkpointed-tape kernels have no LDS to spare (their block states), so they keep registers.
template <bool IN_LDS, int N>
struct CathAcc;
template <int N>
struct CathAcc<false, N> {
    double v[N];
    __device__ __forceinline__ void bind(double *, int) {}
    __device__ __forceinline__ double &operator[](int k) { return v[k]; }
};
template <int N>
struct CathAcc<true, N> {
    double *p;
    __device__ __forceinline__ void bind(double *base, int tid) { p = base + tid * N; }
    __device__ __forceinline__ double &operator[](int k) { return p[k]; }
};

// PRIMAL = true: the forward sweep alone -- loss (and HRR at the measured temperatures) accumulated at the save points as they are
// passed, no tape, no reverse sweep: the primal calls of the UQ wrappers (loss_neuralode, pred_n_ode / HRR_getter, network.jl:167-275).
// cathode_kernel, which served them, maps consecutive LANES to consecutive heating rates of one particle (different grids and step
// counts side by side in a wavefront): 24.8 ms per 4 096 x 256 against 35.6 for the whole gradient here.
template <int BLOCK, int KCP, bool PRIMAL = false>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu((KCP <= 2 && !PRIMAL) ? 2 : CRNN_CATH_ADJ_WAVES, (KCP <= 2 || PRIMAL) ? 2 : CRNN_CATH_ADJ_WAVES))) void cathode_adj_kernel(const CathodeParams prm, const CathAdjParams adj) {
    __shared__ double ts_s[kCathMaxSets * kCathMaxD];
    __shared__ double db_s[kCathMaxSets * kCathMaxD];
    __shared__ double d2_s[kCathMaxSets * kCathMaxD];
    constexpr bool THB_LDS = (KCP <= 2 && !PRIMAL);   // two wavefronts per SIMD: accumulators in LDS (KCP = 2: 75 KB per block with the checkpoint block's states -- two blocks per CU still fit; KCP = 4 would need 91 KB)
    __shared__ double thb_lds[THB_LDS ? BLOCK * kCathNP : 1];
    __shared__ double blk_s[KCP > 1 ? KCP * 4 * BLOCK : 1];      // states (t, u) of the steps of the block being reversed, per lane
    const int tid = threadIdx.x;
    const bool staged = prm.n_sets <= kCathMaxSets;
    if (staged) {
        for (int idx = tid; idx < prm.n_sets * prm.Dmax; idx += BLOCK) {
            const int s = idx / prm.Dmax, i = idx - s * prm.Dmax;
            ts_s[s * kCathMaxD + i] = prm.ts[idx];
            db_s[s * kCathMaxD + i] = prm.dbar[idx];
            d2_s[s * kCathMaxD + i] = prm.d2bar[idx];
        }
    }
    __syncthreads();
    constexpr double d_ = 0.29289321881345248, c32 = 7.4142135623730950, inv12d = 2.4142135623730950;
    constexpr double Rg = -1.0 / 8.314;
    constexpr int RECW = 5;
    const double lqinit = flog(prm.qoldinit);
    const int lane = tid & 63;
    // KCP == 1: [tape_cap][5] records; KCP > 1: [tape_cap] step sizes, then [ceil(tape_cap / KCP)][4] checkpoints
    const size_t lane_stride = KCP > 1 ? (size_t)adj.tape_cap + 4 * (((size_t)adj.tape_cap + KCP - 1) / KCP) : (size_t)adj.tape_cap * RECW;
    double *const tape = adj.tape + (size_t)((size_t)blockIdx.x * BLOCK + tid) * lane_stride;
    // KCP > 1: the wavefront's step sizes and checkpoints are interleaved over its lanes ([step][lane], [checkpoint][field][lane]):
    // a wavefront's store of one dt per lane is 512 contiguous bytes (lane-strided 8-byte stores cost 4-8x their payload in
    // HBM write granules: measured 8.0 GB against 4.3 GB of payload), and the particles of one heating rate take nearly the
    // same number of steps, so the reverse sweep's reads fall into the same rows too
    double *const wtape = adj.tape + (size_t)(((size_t)blockIdx.x * BLOCK + tid) >> 6) * 64 * lane_stride + lane;
    double *const wck = wtape + (size_t)64 * adj.tape_cap;
#define CATH_DT(step) wtape[(size_t)(step) * 64]
#define CATH_CK(k, c) wck[((size_t)(k) * 4 + (c)) * 64]
    const int64_t per_set = (adj.n_part + 63) / 64;              // wave batches per heating rate
    const int64_t n_batches = per_set * prm.n_sets;

    while (true) {
        unsigned long long bq = 0;
        if (lane == 0) bq = atomicAdd(prm.queue, 1ULL);
        const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)bq);
        const unsigned bhi = __builtin_amdgcn_readfirstlane((unsigned)(bq >> 32));
        const int64_t batch = (int64_t)(((unsigned long long)bhi << 32) | blo);
        if (batch >= n_batches) break;
        const int set = (int)(batch / per_set);
        const int64_t part = (batch - (int64_t)set * per_set) * 64 + lane;
        const bool valid = part < adj.n_part;
        const int64_t traj = (valid ? part : 0) * prm.n_sets + set;

        double th[kCathNP];
#pragma unroll
        for (int k = 0; k < kCathNP; ++k) th[k] = prm.theta[(size_t)(valid ? part : 0) * kCathNP + k];
        const int D = prm.D[set];
        const double *tsv, *dbv, *d2v;
        if (staged) { tsv = ts_s + set * kCathMaxD; dbv = db_s + set * kCathMaxD; d2v = d2_s + set * kCathMaxD; }
        else { tsv = prm.ts + (size_t)set * prm.Dmax; dbv = prm.dbar + (size_t)set * prm.Dmax; d2v = prm.d2bar + (size_t)set * prm.Dmax; }
        const double Tdot = prm.beta[set] * (1.0 / 60.0);
        const double t0 = tsv[0], tend = tsv[D - 1];

        auto wfac = [&](const CathPoint &P, double gam, double (&a)[3], double (&iw)[3]) {
#pragma unroll
            for (int j = 0; j < 3; ++j) { a[j] = P.r[j] * th[12 + j] * P.g[j]; iw[j] = frcp(fma(gam, a[j], 1.0)); }
        };
        auto ftime = [&](const CathPoint &P, double (&sig)[3], double (&ft)[3]) {
            double rho[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) { sig[j] = (th[6 + j] * P.it - th[3 + j] * 1e5 * Rg * P.it * P.it) * Tdot; rho[j] = P.r[j] * sig[j]; }
            ft[0] = -rho[0]; ft[1] = fma(th[15], rho[0], -rho[1]); ft[2] = fma(th[16], rho[1], -rho[2]);
        };

        // ================================================================== forward sweep
        double u[3] = {1.0, 0.0, 0.0}, f0[3];
        CathPoint P0;
        double t = t0, dt = 0.0, lqold = lqinit;
        int iter = 0, jsave = 0, nacc = 0, nrej = 0;
        int rc = valid ? -1 : 0;
        cath_point(u, fma(Tdot, t0, prm.T0), th, prm.lb, P0);
        cath_f(P0, th, f0);
        {
            double sk[3], d0 = 0.0, d1 = 0.0, d2 = 0.0, u1[3], f1[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                sk[i] = frcp(fma(fabs(u[i]), prm.rtol, prm.atol));
                d0 = fma(u[i] * sk[i], u[i] * sk[i], d0);
                d1 = fma(f0[i] * sk[i], f0[i] * sk[i], d1);
            }
            d0 = sqrt(d0 * (1.0 / 3.0)); d1 = sqrt(d1 * (1.0 / 3.0));
            const double dtmax = tend - t0;
            double dt0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * (d0 / d1);
            dt0 = fmin(dt0, dtmax);
#pragma unroll
            for (int i = 0; i < 3; ++i) u1[i] = fma(dt0, f0[i], u[i]);
            CathPoint q;
            cath_point(u1, fma(Tdot, t + dt0, prm.T0), th, prm.lb, q);
            cath_f(q, th, f1);
#pragma unroll
            for (int i = 0; i < 3; ++i) { const double e = (f1[i] - f0[i]) * sk[i]; d2 = fma(e, e, d2); }
            d2 = sqrt(d2 * (1.0 / 3.0)) / dt0;
            const double dm = fmax(d1, d2);
            const double dt1 = (dm <= 1e-15) ? fmax(1e-6, dt0 * 1e-3) : exp(-0.5 * (4.605170185988091368 + flog(dm)));
            dt = fmin(fmin(100.0 * dt0, dt1), dtmax);
        }
        auto hrr_of = [&](const double (&uu)[3], double tt) -> double {
            CathPoint q;
            cath_point(uu, fma(Tdot, tt, prm.T0), th, prm.lb, q);
            return fma(q.r[0], th[9], fma(q.r[1], th[10], q.r[2] * th[11]));
        };
        if (valid && prm.hrr) prm.hrr[(size_t)traj * prm.Dmax + 0] = hrr_of(u, t0);
        jsave = 1;    // saveat contains tspan[1]
        double pf_loss = 0.0;   // PRIMAL: the loss, accumulated at the save points of the forward sweep

        while (__builtin_amdgcn_ballot_w64(rc < 0) != 0) {
            if (rc < 0) {
                ++iter;
                bool last = false;
                if (jsave >= D) rc = 0;
                else if (iter > prm.maxiters) rc = 1;
                if (t + dt * (1.0 + 1e-13) >= tend) { dt = tend - t; last = true; }
                if (rc < 0 && (!(dt > 0.0) || t + dt == t)) rc = 2;
                if (rc < 0) {
                    const double gam = d_ * dt;
                    double a[3], iw[3], sig[3], ft[3];
                    wfac(P0, gam, a, iw);
                    ftime(P0, sig, ft);
                    const double l21 = gam * th[15] * a[0], l32 = gam * th[16] * a[1];
                    auto wsolve = [&](double (&b)[3]) {
                        b[0] *= iw[0];
                        b[1] = fma(l21, b[0], b[1]) * iw[1];
                        b[2] = fma(l32, b[1], b[2]) * iw[2];
                    };
                    double k1[3], dk[3], k3[3], u1[3], f1[3], unew[3], f2[3];
#pragma unroll
                    for (int i = 0; i < 3; ++i) k1[i] = fma(gam, ft[i], f0[i]);
                    wsolve(k1);
#pragma unroll
                    for (int i = 0; i < 3; ++i) u1[i] = fma(0.5 * dt, k1[i], u[i]);
                    CathPoint P1, P2;
                    cath_point(u1, fma(Tdot, t + 0.5 * dt, prm.T0), th, prm.lb, P1);
                    cath_f(P1, th, f1);
#pragma unroll
                    for (int i = 0; i < 3; ++i) dk[i] = f1[i] - k1[i];
                    wsolve(dk);
#pragma unroll
                    for (int i = 0; i < 3; ++i) unew[i] = fma(dt, k1[i] + dk[i], u[i]);
                    const double tnew = last ? tend : t + dt;
                    cath_point(unew, fma(Tdot, tnew, prm.T0), th, prm.lb, P2);
                    cath_f(P2, th, f2);
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const double k2i = k1[i] + dk[i];
                        k3[i] = f2[i] - c32 * (k2i - f1[i]) - 2.0 * (k1[i] - f0[i]) + dt * ft[i];
                    }
                    wsolve(k3);
                    double es = 0.0;
                    bool finite = true;
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const double k2i = k1[i] + dk[i];
                        const double ev = dt * (1.0 / 6.0) * (k1[i] - 2.0 * k2i + k3[i]);
                        const double m = fmax(fabs(u[i]), fabs(unew[i]));
                        const double e = ev * frcp(fma(prm.rtol, m, prm.atol));
                        es = fma(e, e, es);
                        finite = finite && isfinite(unew[i]) && isfinite(ev);
                    }
                    es *= (1.0 / 3.0);
                    if (!finite) rc = 3;
                    else {
                        const bool ee_zero = (es == 0.0);
                        const double lEE = 0.5 * flog(ee_zero ? 1.0 : es);
                        const double lq11 = prm.beta1 * lEE;
                        double q = ee_zero ? 1.0 / prm.qmax
                                           : fmax(1.0 / prm.qmax, fmin(1.0 / prm.qmin, exp(lq11 - prm.beta2 * lqold) / prm.gamma));
                        if (es <= 1.0) {
                            if (!PRIMAL && nacc >= adj.tape_cap) {
                                rc = 5;
                                atomicAdd(adj.overflow, 1u);
                            } else {
                                if constexpr (PRIMAL) {
                                } else if constexpr (KCP > 1) {
                                    CRNN_CHK(nacc >= 0 && nacc < adj.tape_cap, 46);
                                    CATH_DT(nacc) = dt;
                                    if (nacc % KCP == 0) {
                                        const int kk = nacc / KCP;
                                        CATH_CK(kk, 0) = t; CATH_CK(kk, 1) = u[0]; CATH_CK(kk, 2) = u[1]; CATH_CK(kk, 3) = u[2];
                                    }
                                } else {
                                    CRNN_CHK(nacc >= 0 && nacc < adj.tape_cap, 40);
                                    double *rec = tape + (size_t)nacc * RECW;
                                    rec[0] = t; rec[1] = dt; rec[2] = u[0]; rec[3] = u[1]; rec[4] = u[2];
                                }
                                ++nacc;
                                while (jsave < D) {
                                    const double tsj = tsv[jsave];
                                    if (!(tsj <= tnew)) break;
                                    if (prm.hrr || PRIMAL) {
                                        const bool at_end = (tsj == tnew);
                                        const double Th = at_end ? 1.0 : (tsj - t) / dt;
                                        const double c1 = at_end ? 0.0 : Th * (1.0 - Th) * inv12d;
                                        const double c2 = at_end ? 1.0 : Th * (Th - 2.0 * d_) * inv12d;
                                        double ui[3];
#pragma unroll
                                        for (int i = 0; i < 3; ++i) ui[i] = at_end ? unew[i] : fma(dt, fma(c1, k1[i], c2 * (k1[i] + dk[i])), u[i]);
                                        const double hv = hrr_of(ui, tsj);
                                        if (prm.hrr) prm.hrr[(size_t)traj * prm.Dmax + jsave] = hv;
                                        if (PRIMAL) {
                                            CRNN_CHK(jsave < D && D <= prm.Dmax && traj < prm.n_traj, 41);
                                            const double db = dbv[jsave], e = hv - db;
                                            pf_loss += fma(e, e, d2v[jsave] - db * db);
                                        }
                                    }
                                    ++jsave;
                                }
#pragma unroll
                                for (int i = 0; i < 3; ++i) { u[i] = unew[i]; f0[i] = f2[i]; }
                                P0 = P2;
                                t = tnew;
                                if (q >= prm.qsteady_min && q <= prm.qsteady_max) q = 1.0;
                                lqold = ee_zero ? lqinit : fmax(lEE, lqinit);
                                dt = fmin(dt / q, tend - t0);
                                if (jsave >= D) rc = 0;
                            }
                        } else {
                            ++nrej;
                            dt = dt / fmin(1.0 / prm.qmin, exp(lq11) / prm.gamma);
                        }
                    }
                }
            }
        }

        // ================================================================== reverse sweep
        const int n_saved = jsave;
        CathAcc<THB_LDS, kCathNP> thb;
        thb.bind(thb_lds, (int)threadIdx.x);
        double lam[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < kCathNP; ++k) thb[k] = 0.0;
        double loss_sum = PRIMAL ? pf_loss : 0.0, tnew = t;
        int s = (valid && !PRIMAL) ? nacc - 1 : -1;
        // observation at a save point: loss term, direct theta gradients (into thb), state seed w
        auto observe = [&](const double (&uu)[3], double tt, int j, double (&w)[3]) {
            CathPoint q;
            cath_point(uu, fma(Tdot, tt, prm.T0), th, prm.lb, q);
            const double hv = fma(q.r[0], th[9], fma(q.r[1], th[10], q.r[2] * th[11]));
            CRNN_CHK(j >= 0 && j < D && D <= prm.Dmax, 42);
            const double db = dbv[j];
            const double e = hv - db;
            loss_sum += fma(e, e, d2v[j] - db * db);
            const double e2 = 2.0 * e;
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) {
                const double c = e2 * th[9 + jj] * q.r[jj];
                thb[jj] += c;
                thb[3 + jj] = fma(c * 1e5, q.rt, thb[3 + jj]);
                thb[6 + jj] = fma(c, q.lt, thb[6 + jj]);
                thb[9 + jj] = fma(e2, q.r[jj], thb[9 + jj]);
                thb[12 + jj] = fma(c, q.l[jj], thb[12 + jj]);
                w[jj] = c * th[12 + jj] * q.g[jj];
            }
        };
        double rt = 0.0, rdt = 0.0, ru[3] = {0.0, 0.0, 0.0};
        if constexpr (KCP == 1 && !PRIMAL) {   // (a context that has only made primal calls has no tape at all)
            CRNN_CHK(s < adj.tape_cap, 43);
            const double *rec = tape + (size_t)(s > 0 ? s : 0) * RECW;
            rt = rec[0]; rdt = rec[1]; ru[0] = rec[2]; ru[1] = rec[3]; ru[2] = rec[4];
        }
        double *const bs = blk_s + (KCP > 1 ? tid : 0);    // state c of the block's step i: bs[(i * 4 + c) * BLOCK]
        int s_lo = s + 1;                                   // first step of the block being reversed (KCP > 1)
        // Block boundaries must coincide across the lanes of the wavefront: otherwise SOME lane re-forms a block in every
        // iteration and the wave pays the KCP - 1 re-formations every time (measured: +63 % at KCP = 4).  The top block of a lane
        // holds 1..KCP steps; a lane whose top block is short sits out the first KCP - (its size) iterations, after which all
        // lanes start their blocks in the same iterations (at most KCP - 1 idle iterations per wavefront).
        const int lag = (KCP > 1 && s >= 0) ? KCP - (s - (s / KCP) * KCP + 1) : 0;
        int it_rev = 0;
        while (__builtin_amdgcn_ballot_w64(s >= 0) != 0) {
            const bool go = it_rev >= lag;
            ++it_rev;
            if constexpr (KCP > 1) {
                // a new block: re-form the states of its steps from the checkpoint
                if (go && s >= 0 && s < s_lo) {
                    const int blk = s / KCP;
                    s_lo = blk * KCP;
                    const int ns = s - s_lo + 1;
                    double tt = CATH_CK(blk, 0), uu[3] = {CATH_CK(blk, 1), CATH_CK(blk, 2), CATH_CK(blk, 3)};
                    for (int i = 0; i < ns; ++i) {
                        bs[(i * 4 + 0) * BLOCK] = tt; bs[(i * 4 + 1) * BLOCK] = uu[0]; bs[(i * 4 + 2) * BLOCK] = uu[1]; bs[(i * 4 + 3) * BLOCK] = uu[2];
                        if (i + 1 < ns) {   // the forward sweep's arithmetic, step s_lo + i
                            const double hh = CATH_DT(s_lo + i);
                            const double gam_ = d_ * hh;
                            CathPoint Pa, Pb;
                            cath_point(uu, fma(Tdot, tt, prm.T0), th, prm.lb, Pa);
                            double fa[3], a_[3], iw_[3], sig_[3], ft_[3], k1_[3], dk_[3], fb[3], ua[3];
                            cath_f(Pa, th, fa);
                            wfac(Pa, gam_, a_, iw_);
                            ftime(Pa, sig_, ft_);
                            const double m21 = gam_ * th[15] * a_[0], m32 = gam_ * th[16] * a_[1];
#pragma unroll
                            for (int c = 0; c < 3; ++c) k1_[c] = fma(gam_, ft_[c], fa[c]);
                            k1_[0] *= iw_[0]; k1_[1] = fma(m21, k1_[0], k1_[1]) * iw_[1]; k1_[2] = fma(m32, k1_[1], k1_[2]) * iw_[2];
#pragma unroll
                            for (int c = 0; c < 3; ++c) ua[c] = fma(0.5 * hh, k1_[c], uu[c]);
                            cath_point(ua, fma(Tdot, tt + 0.5 * hh, prm.T0), th, prm.lb, Pb);
                            cath_f(Pb, th, fb);
#pragma unroll
                            for (int c = 0; c < 3; ++c) dk_[c] = fb[c] - k1_[c];
                            dk_[0] *= iw_[0]; dk_[1] = fma(m21, dk_[0], dk_[1]) * iw_[1]; dk_[2] = fma(m32, dk_[1], dk_[2]) * iw_[2];
#pragma unroll
                            for (int c = 0; c < 3; ++c) uu[c] = fma(hh, k1_[c] + dk_[c], uu[c]);
                            tt = tt + hh;
                        }
                    }
                }
            }
            if (go && s >= 0) {
                double tn_, h_, un_[3];
                if constexpr (KCP > 1) {
                    const int i = s - s_lo;
                    tn_ = bs[(i * 4 + 0) * BLOCK]; un_[0] = bs[(i * 4 + 1) * BLOCK]; un_[1] = bs[(i * 4 + 2) * BLOCK]; un_[2] = bs[(i * 4 + 3) * BLOCK];
                    h_ = CATH_DT(s);
                } else {
                    tn_ = rt; h_ = rdt; un_[0] = ru[0]; un_[1] = ru[1]; un_[2] = ru[2];
                    CRNN_CHK(s - 1 < adj.tape_cap, 44);
                    const double *rec = tape + (size_t)(s > 0 ? s - 1 : 0) * RECW;
                    rt = rec[0]; rdt = rec[1]; ru[0] = rec[2]; ru[1] = rec[3]; ru[2] = rec[4];
                }
                const double tn = tn_, h = h_;
                const double un[3] = {un_[0], un_[1], un_[2]};
                const double gam = d_ * h;
                CathPoint Pn, Pm;
                cath_point(un, fma(Tdot, tn, prm.T0), th, prm.lb, Pn);
                double fn[3], a[3], iw[3], sig[3], ft[3];
                cath_f(Pn, th, fn);
                wfac(Pn, gam, a, iw);
                ftime(Pn, sig, ft);
                const double l21 = gam * th[15] * a[0], l32 = gam * th[16] * a[1];
                auto wsolve = [&](double (&b)[3]) {
                    b[0] *= iw[0];
                    b[1] = fma(l21, b[0], b[1]) * iw[1];
                    b[2] = fma(l32, b[1], b[2]) * iw[2];
                };
                auto wsolveT = [&](double (&b)[3]) {   // W^T is upper bidiagonal
                    b[2] *= iw[2];
                    b[1] = fma(l32, b[2], b[1]) * iw[1];
                    b[0] = fma(l21, b[1], b[0]) * iw[0];
                };
                double k1[3], dk[3], f1[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) k1[i] = fma(gam, ft[i], fn[i]);
                wsolve(k1);
                {
                    double u1[3];
#pragma unroll
                    for (int i = 0; i < 3; ++i) u1[i] = fma(0.5 * h, k1[i], un[i]);
                    cath_point(u1, fma(Tdot, tn + 0.5 * h, prm.T0), th, prm.lb, Pm);
                    cath_f(Pm, th, f1);
                }
#pragma unroll
                for (int i = 0; i < 3; ++i) dk[i] = f1[i] - k1[i];
                wsolve(dk);
                // ---- save points inside (tn, tnew]
                double A_[3] = {0.0, 0.0, 0.0}, B1[3] = {0.0, 0.0, 0.0}, B2[3] = {0.0, 0.0, 0.0};
                while (jsave > 1) {
                    const double tsj = tsv[jsave - 1];
                    if (!(tsj > tn)) break;
                    const bool at_end = (tsj == tnew);
                    const double Th = at_end ? 1.0 : (tsj - tn) / h;
                    const double c1 = at_end ? 0.0 : Th * (1.0 - Th) * inv12d;
                    const double c2 = at_end ? 1.0 : Th * (Th - 2.0 * d_) * inv12d;
                    double ui[3], w[3];
#pragma unroll
                    for (int i = 0; i < 3; ++i) ui[i] = at_end ? fma(h, k1[i] + dk[i], un[i]) : fma(h, fma(c1, k1[i], c2 * (k1[i] + dk[i])), un[i]);
                    observe(ui, tsj, jsave - 1, w);
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        A_[i] += w[i];
                        B1[i] = fma(w[i], h * c1, B1[i]);
                        B2[i] = fma(w[i], h * c2, B2[i]);
                    }
                    --jsave;
                }
                if (prm.want_grad) {
                    double v[3], kb1[3], ub[3];
#pragma unroll
                    for (int i = 0; i < 3; ++i) { v[i] = fma(h, lam[i], B2[i]); ub[i] = lam[i] + A_[i]; kb1[i] = B1[i] + v[i]; }
                    wsolveT(v);
#pragma unroll
                    for (int i = 0; i < 3; ++i) kb1[i] -= v[i];
                    // A = S^T a
                    const double Av[3] = {fma(th[15], v[1], -v[0]), fma(th[16], v[2], -v[1]), -v[2]};
                    {   // point u_mid
                        double um[3];
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            const double Psi = Av[j] * Pm.r[j];
                            thb[j] += Psi;
                            thb[3 + j] = fma(Psi * 1e5, Pm.rt, thb[3 + j]);
                            thb[6 + j] = fma(Psi, Pm.lt, thb[6 + j]);
                            thb[12 + j] = fma(Psi, Pm.l[j], thb[12 + j]);
                            um[j] = Psi * th[12 + j] * Pm.g[j];
                        }
                        thb[15] = fma(v[1], Pm.r[0], thb[15]);
                        thb[16] = fma(v[2], Pm.r[1], thb[16]);
#pragma unroll
                        for (int j = 0; j < 3; ++j) { ub[j] += um[j]; kb1[j] = fma(0.5 * h, um[j], kb1[j]); }
                    }
                    wsolveT(kb1);   // kb1 = w
                    const double Aw[3] = {fma(th[15], kb1[1], -kb1[0]), fma(th[16], kb1[2], -kb1[1]), -kb1[2]};
                    double yv0 = 0.0, yw0 = 0.0, yv1 = 0.0, yw1 = 0.0;
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const double ng = th[12 + j] * Pn.g[j];
                        const double yv = ng * dk[j], yw = fma(ng, k1[j], sig[j]);
                        if (j == 0) { yv0 = yv; yw0 = yw; }
                        if (j == 1) { yv1 = yv; yw1 = yw; }
                        const double Pv = Av[j] * Pn.r[j], Pw = Aw[j] * Pn.r[j];
                        const double E = fma(Pw, fma(gam, yw, 1.0), gam * Pv * yv);
                        const double gPw = gam * Pw;
                        const double mix = fma(Pw, k1[j], Pv * dk[j]);                 // Pw k1_j + Pv dk_j
                        thb[j] += E;
                        thb[3 + j] = fma(E * 1e5, Pn.rt, fma(gPw, -1e5 * Rg * Pn.it * Pn.it * Tdot, thb[3 + j]));
                        thb[6 + j] = fma(E, Pn.lt, fma(gPw, Pn.it * Tdot, thb[6 + j]));
                        thb[12 + j] = fma(E, Pn.l[j], fma(gam * Pn.g[j], mix, thb[12 + j]));
                        // d/du_j: E n g + gam n h (Pw k1 + Pv dk),  h = -g^2
                        lam[j] = ub[j] + th[12 + j] * Pn.g[j] * (E - gam * Pn.g[j] * mix);
                    }
                    thb[15] = fma(Pn.r[0], fma(kb1[1], fma(gam, yw0, 1.0), gam * v[1] * yv0), thb[15]);
                    thb[16] = fma(Pn.r[1], fma(kb1[2], fma(gam, yw1, 1.0), gam * v[2] * yv1), thb[16]);
                }
                tnew = tn;
                --s;
            }
        }
        if (valid) {
            double w0[3];
            const double u00[3] = {1.0, 0.0, 0.0};
            observe(u00, t0, 0, w0);           // the saved initial point: loss term and its direct theta gradients
            const double invD = 1.0 / (double)D;   // the FULL row count, also for a truncated solution (network.jl:266)
            prm.loss[traj] = loss_sum * invD;
            prm.retcode[traj] = rc;
            prm.n_saved[traj] = n_saved;
            prm.n_accept[traj] = nacc;
            prm.n_reject[traj] = nrej;
            if (prm.grad) {
                double *go = prm.grad + (size_t)traj * kCathNP;
#pragma unroll
                for (int m = 0; m < kCathNP; ++m) go[m] = thb[m] * invD;
            }
        }
    }
#undef CATH_DT
#undef CATH_CK
}

}  // namespace crnn

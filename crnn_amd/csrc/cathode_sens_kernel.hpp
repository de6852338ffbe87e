// crnn_amd/csrc/cathode_sens_kernel.hpp -- the cathode gradient as the reference evaluates it (gfx950).
//
// Reference: Cathode_NCM333_UQ/src_333/network.jl:232  grad_curr = ForwardDiff.gradient(x -> loss_neuralode(x, i_exp), p_temp):
// Duals through the adaptive solve.  ForwardDiff works through the 17 normalised parameters in chunks (pickchunksize(17) = 9:
// p[1:9], then p[10:17] and one zero partial), EVERY CHUNK IS ITS OWN ADAPTIVE SOLVE, and DiffEqBase's norm of a Dual-valued state
// weighs the partials with the value -- the step-size controller sees the chunk's tangents (ros23_sens_kernel.hpp has the norm;
// [UNVERIFIED-DEP] like it; mode 1: / length(u), mode 2: / totallength(u) = 3 (1 + 9), the form of the DiffEqBase 6.189 the
// cathode Manifest pins).  crnn_cathode_set_errnorm_sens(ctx, mode, p_scales) selects it for gradient launches.
//
// One launch = one chunk (CH = 0: lnA, Ea, b -- nine ODE columns; CH = 1: dH, n, nu2, nu3 -- five ODE columns, dH enters the
// observable only).  cathode_kernel's frame (one lane per (particle, heating rate) trajectory, persistent lanes, W lower
// bidiagonal, one-hot directions resolved at compile time) with what the dual-inclusive norm needs: the tangents go through
// EVERY ATTEMPT (the accept / reject decision needs them), including the third stage's W k3' = f2' - c32 (k2' - f1') -
// 2 (k1' - f0') + dt ft' + gam J' k3 that only the error estimate uses; an attempt's new tangent columns and gradient increments
// are committed on acceptance.  Partials are taken with respect to p (theta_m = p_m p_scales[m], network.jl:152-157): a column's
// contribution to the norm carries p_scales[m]^2.  Stepper: Rosenbrock23 (the reference: AutoTsit5(TRBDF2) -- the composite is a
// primal-launch option, cathode_auto_kernel.hpp).  The launch writes its chunk's entries of the gradient rows and the step
// counts; loss, heat-release curve and return codes of a gradient call are those of the plain solve that follows (what
// loss_neuralode(p) evaluates, network.jl:229).
#pragma once
#include "cathode_kernel.hpp"

namespace crnn {

struct CathSensParams {
    const double *dir_scale;   // [17] d theta_m / d p_m = p_scales[m]
    int32_t mode;              // 1: squared norm / length(u); 2: / totallength(u)
    int32_t dual_partials;     // partials per Dual (9 for both chunks: the second one carries a zero partial)
};

template <int BLOCK, int CH>
__global__ __launch_bounds__(BLOCK) void cathode_sens_kernel(const CathodeParams prm, const CathSensParams sp) {
    constexpr int K0 = CH == 0 ? 0 : 9, NCOL = CH == 0 ? 9 : 5;        // this chunk's ODE columns: k in [K0, K0 + NCOL)
    constexpr int M0 = CH == 0 ? 0 : 9, M1 = CH == 0 ? 9 : 17;          // and its range of theta / p
    __shared__ double ts_s[kCathMaxSets * kCathMaxD];
    __shared__ double db_s[kCathMaxSets * kCathMaxD];
    __shared__ double d2_s[kCathMaxSets * kCathMaxD];
    // the attempt's stage tangents k1', k2' of every column wait for the decision (commit: gradient increments, new columns).  Nine
    // columns of them are 108 registers of a file that is full (292 B of scratch per lane, ~100 scratch instructions per attempt):
    // the first chunk parks them in LDS (lane-contiguous, conflict-free; 108 KB per block of 256, one block per CU either way)
    constexpr bool KP_LDS = (CH == 0);
    __shared__ double kp_s[KP_LDS ? 2 * NCOL * 3 * BLOCK : 1];
    const int tid = threadIdx.x;
    double *const kpl = kp_s + (KP_LDS ? tid : 0);
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
    double S[NCOL][3], gS[NCOL], gD[kCathNP];
    double sc2[NCOL];                 // (d theta_m / d p_m)^2 of the chunk's columns: the norm weighs partials with respect to p
#pragma unroll
    for (int kk = 0; kk < NCOL; ++kk) { const double v = sp.dir_scale[cath_col_theta(K0 + kk)]; sc2[kk] = v * v; }
    const double inv_div = sp.mode == 2 ? 1.0 / (3.0 * (1.0 + (double)sp.dual_partials)) : 1.0 / 3.0;
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

    // f'(point; state tangent ss) of ODE column k (direction theta_m, m = cath_col_theta(k)); rp = r'
    auto col_fprime = [&](const int k, const CathPoint &p, const double (&ss)[3], double (&rp)[3], double (&fp)[3]) {
        const int m = cath_col_theta(k);
        const int grp = m / 3, j0 = m % 3;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double dz = (m >= 15 || j != j0) ? 0.0 : (grp == 0 ? 1.0 : grp == 1 ? 1e5 * p.rt : grp == 2 ? p.lt : p.l[j]);
            rp[j] = p.r[j] * fma(th[12 + j] * p.g[j], ss[j], dz);
        }
        fp[0] = -rp[0];
        fp[1] = fma(th[15], rp[0], -rp[1]) + (m == 15 ? p.r[0] : 0.0);
        fp[2] = fma(th[16], rp[1], -rp[2]) + (m == 16 ? p.r[1] : 0.0);
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
            for (int k = 0; k < NCOL; ++k) { S[k][0] = 0.0; S[k][1] = 0.0; S[k][2] = 0.0; gS[k] = 0.0; }
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
                // ode_determine_initdt with Duals: u0 has zero partials, f0 and f1 = f(u0 + dt0 f0) carry those of p (ros23_sens_kernel.hpp: sens_init_dt)
                double f0p_[NCOL][3];
#pragma unroll
                for (int kk = 0; kk < NCOL; ++kk) {
                    const double zero3[3] = {0.0, 0.0, 0.0};
                    double rp_[3];
                    col_fprime(K0 + kk, P0, zero3, rp_, f0p_[kk]);
#pragma unroll
                    for (int i = 0; i < 3; ++i) { const double e = f0p_[kk][i] * sk[i]; d1 = fma(sc2[kk] * e, e, d1); }
                }
                d0 = sqrt(d0 * inv_div); d1 = sqrt(d1 * inv_div);
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
#pragma unroll
                for (int kk = 0; kk < NCOL; ++kk) {
                    double s1_[3], rp_[3], f1p_[3];
#pragma unroll
                    for (int i = 0; i < 3; ++i) s1_[i] = dt0 * f0p_[kk][i];
                    col_fprime(K0 + kk, q, s1_, rp_, f1p_);
#pragma unroll
                    for (int i = 0; i < 3; ++i) { const double e = (f1p_[i] - f0p_[kk][i]) * sk[i]; d2 = fma(sc2[kk] * e, e, d2); }
                }
                d2 = sqrt(d2 * inv_div) / dt0;
                const double dm = fmax(d1, d2);
                const double dt1 = (dm <= 1e-15) ? fmax(1e-6, dt0 * 1e-3) : fexp_ctl(-0.5 * (4.605170185988091368 + flog_ctl(dm)));
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
            double na[3], nb[3], ee[3];       // value^2 + partials^2 of u, u+ and of the error estimate, per component
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double k2i = k1[i] + dk[i];
                const double ev = dt * (1.0 / 6.0) * (k1[i] - 2.0 * k2i + k3[i]);
                na[i] = u[i] * u[i]; nb[i] = unew[i] * unew[i]; ee[i] = ev * ev;
                finite = finite && isfinite(unew[i]) && isfinite(ev);
            }
            // ---- the chunk's tangents through THIS ATTEMPT (the decision needs them), third stage included ----
            double K1P[KP_LDS ? 1 : NCOL][3], K2P[KP_LDS ? 1 : NCOL][3];
#pragma unroll
            for (int kk = 0; kk < NCOL; ++kk) {
                const int k = K0 + kk;
                const int m = cath_col_theta(k);
                const int grp = m / 3, j0 = m % 3;
                const bool is_nu2 = (m == 15), is_nu3 = (m == 16);
                const double(&s)[3] = S[kk];
                double rp0[3], f0p[3];
                col_fprime(k, P0, s, rp0, f0p);
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
                double jk[3], k1p[3], dkp[3], k3p[3], s1[3], sn[3], rp1[3], f1p[3], rp2[3], f2p[3];
                jprime(k1, jk);
#pragma unroll
                for (int i = 0; i < 3; ++i) k1p[i] = fma(gam, ftp[i] + jk[i], f0p[i]);
                wsolve(k1p);
#pragma unroll
                for (int i = 0; i < 3; ++i) s1[i] = fma(0.5 * dt, k1p[i], s[i]);
                col_fprime(k, P1, s1, rp1, f1p);
                jprime(dk, jk);
#pragma unroll
                for (int i = 0; i < 3; ++i) dkp[i] = fma(gam, jk[i], f1p[i] - k1p[i]);
                wsolve(dkp);
#pragma unroll
                for (int i = 0; i < 3; ++i) sn[i] = fma(dt, k1p[i] + dkp[i], s[i]);
                col_fprime(k, P2, sn, rp2, f2p);
                // W k3' = f2' - c32 (k2' - f1') - 2 (k1' - f0') + dt ft' + gam J' k3
                jprime(k3, jk);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const double k2p = k1p[i] + dkp[i];
                    k3p[i] = f2p[i] - c32 * (k2p - f1p[i]) - 2.0 * (k1p[i] - f0p[i]) + dt * ftp[i] + gam * jk[i];
                }
                wsolve(k3p);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const double k2p = k1p[i] + dkp[i];
                    const double de = dt * (1.0 / 6.0) * (k1p[i] - 2.0 * k2p + k3p[i]);
                    na[i] = fma(sc2[kk] * s[i], s[i], na[i]);
                    nb[i] = fma(sc2[kk] * sn[i], sn[i], nb[i]);
                    ee[i] = fma(sc2[kk] * de, de, ee[i]);
                    if constexpr (KP_LDS) { kpl[((0 * NCOL + kk) * 3 + i) * BLOCK] = k1p[i]; kpl[((1 * NCOL + kk) * 3 + i) * BLOCK] = k2p; }
                    else { K1P[kk][i] = k1p[i]; K2P[kk][i] = k2p; }
                }
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double scl = fma(prm.rtol, sqrt(fmax(na[i], nb[i])), prm.atol);
                es += ee[i] / (scl * scl);
            }
            es *= inv_div;
            finite = finite && isfinite(es);
            if (!finite) rc = 3;
            else {
                const bool ee_zero = (es == 0.0);
                const double lEE = 0.5 * flog_ctl(ee_zero ? 1.0 : es);
                const double lq11 = prm.beta1 * lEE;
                double q = ee_zero ? 1.0 / prm.qmax
                                   : fmax(1.0 / prm.qmax, fmin(1.0 / prm.qmin, fexp_ctl(lq11 - prm.beta2 * lqold) / prm.gamma));
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
                    // ---- commit the attempt's tangents: gradient increments and the new tangent columns ----
#pragma unroll
                    for (int kk = 0; kk < NCOL; ++kk) {
                        double acc = 0.0;
#pragma unroll
                        for (int i = 0; i < 3; ++i) {
                            const double k1p_ = KP_LDS ? kpl[((0 * NCOL + kk) * 3 + i) * BLOCK] : K1P[KP_LDS ? 0 : kk][i];
                            const double k2p_ = KP_LDS ? kpl[((1 * NCOL + kk) * 3 + i) * BLOCK] : K2P[KP_LDS ? 0 : kk][i];
                            acc = fma(A_[i], S[kk][i], acc);
                            acc = fma(B1[i], k1p_, acc);
                            acc = fma(B2[i], k2p_, acc);
                            S[kk][i] = fma(dt, k2p_, S[kk][i]);
                        }
                        gS[kk] += acc;
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
                    dt = dt / fmin(1.0 / prm.qmin, fexp_ctl(lq11) / prm.gamma);
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
            {   // this chunk's entries of the gradient row (the other chunk's launch writes the rest: its own adaptive solve)
                double *go = prm.grad + (size_t)traj * kCathNP;
#pragma unroll
                for (int m = M0; m < M1; ++m) go[m] = gD[m] * invD;
#pragma unroll
                for (int kk = 0; kk < NCOL; ++kk) {
                    const int m = cath_col_theta(K0 + kk);
                    go[m] = (gD[m] + gS[kk]) * invD;
                }
            }
            traj = traj_next;
            traj_next = (int64_t)atomicAdd(prm.queue, 1ULL) + nthreads;
            need_init = true;
        }
    }
}


}  // namespace crnn

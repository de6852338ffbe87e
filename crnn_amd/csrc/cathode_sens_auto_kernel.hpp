// crnn_amd/csrc/cathode_sens_auto_kernel.hpp -- gfx950 (MI355X): the cathode gradient as the reference evaluates it, THROUGH THE
// REFERENCE'S OWN STEPPER.
//
// Reference: Cathode_NCM333_UQ/src_333/network.jl:232  grad_curr = ForwardDiff.gradient(x -> loss_neuralode(x, i_exp), p_temp), whose solve is
// :195's alg = AutoTsit5(TRBDF2(autodiff = true)) (used by pred_n_ode, :205-212): Duals through the adaptive COMPOSITE, in ForwardDiff's chunks of
// the 17 normalised parameters (9, then 8 and a zero partial), every chunk its own adaptive solve, the error estimate of BOTH algorithms carrying
// the chunk's partials (cathode_sens_kernel.hpp has the same chunks and norm on Rosenbrock23; this kernel closes rows A4 x A7 of SURVEY section 8
// for config 5).  Selected by crnn_cathode_set_solver(ctx, CRNN_CATH_SOLVER_AUTOTSIT5_TRBDF2) together with crnn_cathode_set_errnorm_sens.
//
// What is restated (all [UNVERIFIED-DEP] like cathode_auto_kernel.hpp, whose primal statement this is, operation for operation; the test
// suite's CPU statement of the same composite with the same chunks and norm is the checker -- tests/test_cathode.py):
//   Tsit5       stage s at t + c_s dt; a direction's slope k_s' = f'(g_s; g_s'), g_s' = s + dt sum_j a_sj k_j'; the embedded estimate's partial
//               dt sum_j bt_j k_j'; free 4th-order interpolant for u and u' at the save points
//   AutoSwitch  on the PRIMAL's stiffness estimates (|k7 - k6| / |g7 - g6| after a Tsit5 attempt; opnorm(J, Inf) of the last formed Jacobian in
//               TRBDF2), > 10 stiff / > 3 non-stiff verdicts in a row, dt * 2 and dt / 2 at the switches, the running algorithm's PI exponents
//   TRBDF2      two Newton-solved stages in z = dt f form with OrdinaryDiffEq's Jacobian / W reuse; the tangent copies ride through the iteration
//               with the primal's W, its number of iterations and its decisions, the partial of W on the right-hand side:
//                   W dz' = (dt f'(tmp + d z; tmp' + d z') - z') / (d dt) - J' dz,   J' = the direction's derivative of J WHERE J WAS FORMED
//               (held with J and, stale, reused with it); smoothed estimate's partial W^-1 (btilde . z' - J' est); fsallast' = z' / dt; Hermite
//               interpolation of u and u' at the save points; a switch back to Tsit5 re-evaluates f and f' at (u, t).
//   norm        DiffEqBase's norm of Dual-valued arrays: value^2 + sum over the chunk's partials^2 per component (partials with respect to
//               p: each weighs with p_scales[m]^2), divided by length(u) (mode 1) or totallength(u) = 3 (1 + 9) (mode 2).
//
// Mapping: a GROUP OF NINE LANES per trajectory, one direction of the chunk per lane (seven groups per wavefront, one lane idle; the second
// chunk has eight directions: its ninth lane carries zeros).  The system has three species: every lane carries the primal redundantly -- the
// same operations in the same order, so a group never diverges and takes one decision -- and ITS direction through every attempt and every
// Newton iteration; the group sums the lanes' contributions to the norm by ds_bpermute in lane order.  No LDS beyond the staged observations.
// A parity kernel: it exists so that a gradient call can be the reference's algorithm; the Rosenbrock23 statement of the same chunks and norm
// (cathode_sens_kernel.hpp) is the fast one.
#pragma once
#include "cathode_auto_kernel.hpp"
#include "cathode_sens_kernel.hpp"

namespace crnn {

template <int BLOCK, int CH>
__global__ __launch_bounds__(BLOCK) void cathode_sens_auto_kernel(const CathodeParams prm, const CathSensParams sp) {
    constexpr int G = 9, GPW = 64 / G;                                 // lanes per trajectory, groups per wavefront
    constexpr int M0 = CH == 0 ? 0 : 9, NDIR = CH == 0 ? 9 : 8;        // this chunk's range of theta / p
    __shared__ double ts_s[kCathMaxSets * kCathMaxD];
    __shared__ double db_s[kCathMaxSets * kCathMaxD];
    __shared__ double d2_s[kCathMaxSets * kCathMaxD];
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
    constexpr double b1_ts = 7.0 / 50.0, b2_ts = 2.0 / 25.0, b1_rb = 7.0 / 20.0, b2_rb = 2.0 / 10.0;
    constexpr double s2_ = 1.4142135623730951;
    constexpr double tb_g = 2.0 - s2_, tb_d = 1.0 - s2_ / 2.0, tb_w = s2_ / 4.0;
    constexpr double tb_bt1 = (1.0 - s2_) / 3.0, tb_bt2 = 1.0 / 3.0, tb_bt3 = (s2_ - 2.0) / 3.0, tb_a1 = -s2_ / 2.0, tb_a2 = 1.0 + s2_ / 2.0;
    constexpr int NL_CONV = 1, NL_DIV = -2, NL_TRYAGAIN = -4;
    const double lqinit = flog(prm.qoldinit);
    const int lane = tid & 63, wave = tid >> 6;
    const int grp = lane / G, col = lane - grp * G;
    const bool lane_on = grp < GPW;
    const int gbase = grp * G;
    // this lane's direction: theta_m (m = -1: the zero partial that pads the second chunk)
    const int m = (lane_on && col < NDIR) ? M0 + col : -1;
    const int mg = m >= 0 ? m / 3 : -1, mj = m >= 0 ? m % 3 : -1;      // parameter family (0 lnA, 1 Ea, 2 b, 3 dH, 4 n, 5 nu) and its reaction
    const double dsc = m >= 0 ? sp.dir_scale[m] : 0.0;
    const double sc2 = dsc * dsc;                                       // the norm weighs partials with respect to p
    const double inv_div = sp.mode == 2 ? 1.0 / (3.0 * (1.0 + (double)sp.dual_partials)) : 1.0 / 3.0;
    auto group_sum = [&](const double v) -> double {
        double a = 0.0;
#pragma unroll
        for (int q = 0; q < G; ++q) a += __shfl(v, gbase + q);
        return a;
    };
    const int64_t groups_total = (int64_t)gridDim.x * (BLOCK / 64) * GPW;
    int64_t traj = ((int64_t)blockIdx.x * (BLOCK / 64) + wave) * GPW + grp;
    if (!lane_on) traj = prm.n_traj;

    while (traj < prm.n_traj) {
        const int64_t part = traj / prm.n_sets;
        const int set = (int)(traj - part * prm.n_sets);
        double th[kCathNP];
#pragma unroll
        for (int k = 0; k < kCathNP; ++k) th[k] = prm.theta[(size_t)part * kCathNP + k];
        const int D = prm.D[set];
        const double *tsv, *dbv, *d2v;
        if (staged) { tsv = ts_s + set * kCathMaxD; dbv = db_s + set * kCathMaxD; d2v = d2_s + set * kCathMaxD; }
        else { tsv = prm.ts + (size_t)set * prm.Dmax; dbv = prm.dbar + (size_t)set * prm.Dmax; d2v = prm.d2bar + (size_t)set * prm.Dmax; }
        const double Tdot = prm.beta[set] * (1.0 / 60.0);
        const double t0 = tsv[0], tend = tsv[D - 1];

        auto point_at = [&](const double (&uu)[3], double tt, CathPoint &P) { cath_point(uu, fma(Tdot, tt, prm.T0), th, prm.lb, P); };
        // this lane's direction at a point: the direct part of d z_j / d theta_m, then r', f', a' (a_j = d r_j / d u_j = r_j n_j g_j)
        auto dzdir = [&](const CathPoint &p, const int j) -> double {
            if (mg < 0 || mg == 3 || mg == 5 || j != mj) return 0.0;
            return mg == 0 ? 1.0 : mg == 1 ? 1e5 * p.rt : mg == 2 ? p.lt : p.l[j];
        };
        auto fprime = [&](const CathPoint &p, const double (&ss)[3], double (&rp)[3], double (&fp)[3]) {
#pragma unroll
            for (int j = 0; j < 3; ++j) rp[j] = p.r[j] * fma(th[12 + j] * p.g[j], ss[j], dzdir(p, j));
            fp[0] = -rp[0];
            fp[1] = fma(th[15], rp[0], -rp[1]) + (m == 15 ? p.r[0] : 0.0);
            fp[2] = fma(th[16], rp[1], -rp[2]) + (m == 16 ? p.r[1] : 0.0);
        };
        auto aprime = [&](const CathPoint &p, const double (&ss)[3], const double (&rp)[3], double (&ap)[3]) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double dn = (mg == 4 && j == mj) ? 1.0 : 0.0;
                ap[j] = fma(rp[j], th[12 + j] * p.g[j], p.r[j] * p.g[j] * (dn - th[12 + j] * p.g[j] * ss[j]));
            }
        };
        // the heat-release observable and its partial along the direction at (v, v')
        auto hrr_at = [&](const CathPoint &q) -> double { return fma(q.r[0], th[9], fma(q.r[1], th[10], q.r[2] * th[11])); };
        auto hrr_prime = [&](const CathPoint &q, const double (&vp)[3]) -> double {
            double a = 0.0;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double rpj = q.r[j] * fma(th[12 + j] * q.g[j], vp[j], dzdir(q, j));
                a = fma(rpj, th[9 + j], a);
                a += (mg == 3 && j == mj) ? q.r[j] : 0.0;
            }
            return a;
        };

        double u[3] = {1.0, 0.0, 0.0}, f0[3], s[3] = {0.0, 0.0, 0.0}, f0p[3];
        CathPoint P0;
        double t = t0, dt = 0.0, lqold = lqinit, loss_sum = 0.0, gsum = 0.0;
        int iter = 0, jsave = 0, nacc = 0, nrej = 0, rc = -1;
        int alg = 0, cnt = 0;
        double eig = 0.0;
        bool have_eig = false;
        // TRBDF2's nonlinear-solver cache: J = S diag(nl_a) formed at nl_Jt, its partial along this lane's direction (nl_ap and the nu terms), W
        double nl_a[3] = {0.0, 0.0, 0.0}, nl_ap[3] = {0.0, 0.0, 0.0}, nl_Jt = -INFINITY, nl_Wgdt = 0.0, ee_prev = 1.0;
        int nl_status = NL_DIV;
        bool nl_first = true;
        point_at(u, t0, P0);
        cath_f(P0, th, f0);
        {
            double rp_[3];
            fprime(P0, s, rp_, f0p);
        }
        {   // Hairer initial step with the dual-inclusive norms (cathode_sens_kernel.hpp), the order of the STARTING algorithm (Tsit5)
            double sk[3], d0 = 0.0, d1 = 0.0, d2 = 0.0, u1[3], f1[3], d1p = 0.0, d2p = 0.0;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                sk[i] = frcp(fma(fabs(u[i]), prm.rtol, prm.atol));
                d0 = fma(u[i] * sk[i], u[i] * sk[i], d0);
                d1 = fma(f0[i] * sk[i], f0[i] * sk[i], d1);
                const double e = f0p[i] * sk[i];
                d1p = fma(sc2 * e, e, d1p);
            }
            d1 += group_sum(d1p);
            d0 = sqrt(d0 * inv_div); d1 = sqrt(d1 * inv_div);
            const double dtmax = tend - t0;
            double dt0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * (d0 / d1);
            dt0 = fmin(dt0, dtmax);
            double s1_[3], rp_[3], f1p_[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) { u1[i] = fma(dt0, f0[i], u[i]); s1_[i] = dt0 * f0p[i]; }
            CathPoint q;
            point_at(u1, t + dt0, q);
            cath_f(q, th, f1);
            fprime(q, s1_, rp_, f1p_);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double e = (f1[i] - f0[i]) * sk[i], ep = (f1p_[i] - f0p[i]) * sk[i];
                d2 = fma(e, e, d2);
                d2p = fma(sc2 * ep, ep, d2p);
            }
            d2 += group_sum(d2p);
            d2 = sqrt(d2 * inv_div) / dt0;
            const double dm = fmax(d1, d2);
            const double dt1 = (dm <= 1e-15) ? fmax(1e-6, dt0 * 1e-3) : exp(-0.2 * (4.605170185988091368 + flog(dm)));
            dt = fmin(fmin(100.0 * dt0, dt1), dtmax);
        }
        // a save point: loss term of the primal, the direction's gradient term
        auto save_at = [&](const double (&v)[3], const double (&vp)[3], const double tsj, const int j) {
            CathPoint q;
            point_at(v, tsj, q);
            const double hv = hrr_at(q);
            CRNN_CHK(j >= 0 && j < D && D <= prm.Dmax && traj < prm.n_traj, 46);
            const double db = dbv[j], e = hv - db;
            loss_sum += fma(e, e, d2v[j] - db * db);
            gsum = fma(2.0 * e, hrr_prime(q, vp), gsum);
        };
        save_at(u, s, t0, 0);      // saveat contains tspan[1]
        jsave = 1;

        while (rc < 0) {
            ++iter;
            bool last = false;
            if (jsave >= D) { rc = 0; break; }
            if (iter > prm.maxiters) { rc = 1; break; }
            if (have_eig) {   // choose_algorithm! at the loop header
                const bool stiff = fabs(eig * dt * (1.0 / AutoSw::stability_size)) > AutoSw::tol;
                cnt = stiff ? (cnt < 0 ? 1 : cnt + 1) : (cnt > 0 ? -1 : cnt - 1);
                if (alg == 0 && cnt > AutoSw::maxstiffstep) { dt *= AutoSw::dtfac; alg = 1; }
                else if (alg == 1 && cnt < -AutoSw::maxnonstiffstep) {
                    dt *= 1.0 / AutoSw::dtfac; alg = 0;
                    // initialize!(Tsit5 cache): fsalfirst = f(uprev, t) afresh (TRBDF2 left z / dt there) -- and its partial
                    double rp_[3];
                    point_at(u, t, P0);
                    cath_f(P0, th, f0);
                    fprime(P0, s, rp_, f0p);
                }
            }
            if (t + dt * (1.0 + 1e-13) >= tend) { dt = tend - t; last = true; }
            if (!(dt > 0.0) || t + dt == t) { rc = 2; break; }
            const double tnew = last ? tend : t + dt;
            double unew[3], f2[3], snew[3], f2p[3], es = 0.0;
            CathPoint P2;
            bool finite = true, accepted = false, ee_zero = false, stepfail = false;
            double q = 1.0, lEE = 0.0, lq11 = 0.0;
            auto controller = [&](double b1, double b2) -> bool {
                ee_zero = (es == 0.0);
                lEE = 0.5 * flog(ee_zero ? 1.0 : es);
                lq11 = b1 * lEE;
                q = ee_zero ? 1.0 / prm.qmax : fmax(1.0 / prm.qmax, fmin(1.0 / prm.qmin, exp(lq11 - b2 * lqold) / prm.gamma));
                return es <= 1.0;
            };
            // the dual-inclusive norm of an attempt: ev = the primal's estimate, de = this direction's
            auto dual_norm = [&](const double (&ev)[3], const double (&de)[3]) {
                double ssum = 0.0;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const double na = fma(u[i], u[i], group_sum(sc2 * s[i] * s[i]));
                    const double nb = fma(unew[i], unew[i], group_sum(sc2 * snew[i] * snew[i]));
                    const double ee = fma(ev[i], ev[i], group_sum(sc2 * de[i] * de[i]));
                    const double scl = fma(prm.rtol, sqrt(fmax(na, nb)), prm.atol);
                    ssum += ee / (scl * scl);
                    finite = finite && isfinite(unew[i]) && isfinite(ev[i]);
                }
                es = ssum * inv_div;
                finite = finite && isfinite(es);
            };
            if (alg == 0) {
                // ---------------------------------------------------------------- Tsit5 attempt, the direction through its stages
                double k[7][3], kp[7][3], g6[3] = {0.0, 0.0, 0.0};
#pragma unroll
                for (int i = 0; i < 3; ++i) { k[0][i] = f0[i]; kp[0][i] = f0p[i]; }
#pragma unroll
                for (int st = 1; st < 7; ++st) {
                    double g[3], gp[3], rp_[3];
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        double a = 0.0, ap_ = 0.0;
#pragma unroll
                        for (int j = 0; j < 6; ++j)
                            if (j < st) { a = fma(Ts5::a(st - 1, j), k[j][i], a); ap_ = fma(Ts5::a(st - 1, j), kp[j][i], ap_); }
                        g[i] = fma(dt, a, u[i]);
                        gp[i] = fma(dt, ap_, s[i]);
                    }
                    if (st == 5) {
#pragma unroll
                        for (int i = 0; i < 3; ++i) g6[i] = g[i];
                    }
                    const double tq = st == 6 ? tnew : st == 5 ? t + dt : fma(st == 1 ? Ts5::c2 : st == 2 ? Ts5::c3 : st == 3 ? Ts5::c4 : Ts5::c5, dt, t);
                    if (st == 6) {
#pragma unroll
                        for (int i = 0; i < 3; ++i) { unew[i] = g[i]; snew[i] = gp[i]; }
                        point_at(g, tq, P2);
                        cath_f(P2, th, k[6]);
                        fprime(P2, gp, rp_, kp[6]);
                    } else {
                        CathPoint Ps;
                        point_at(g, tq, Ps);
                        cath_f(Ps, th, k[st]);
                        fprime(Ps, gp, rp_, kp[st]);
                    }
                }
                double ev[3], de[3], est = 0.0;
                bool isnan_ = false;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    f2[i] = k[6][i]; f2p[i] = kp[6][i];
                    double a = 0.0, ap_ = 0.0;
#pragma unroll
                    for (int j = 0; j < 7; ++j) { a = fma(Ts5::bt(j), k[j][i], a); ap_ = fma(Ts5::bt(j), kp[j][i], ap_); }
                    ev[i] = dt * a; de[i] = dt * ap_;
                    const double qq = fabs((k[6][i] - k[5][i]) / (unew[i] - g6[i]));   // Hairer II p.22, Inf norm, the primal's; NaN propagates
                    isnan_ = isnan_ || (qq != qq);
                    est = fmax(est, qq);
                }
                eig = isnan_ ? __longlong_as_double(0x7ff8000000000000LL) : est;
                have_eig = true;
                dual_norm(ev, de);
                if (!finite) { rc = 3; break; }
                if (controller(b1_ts, b2_ts)) {
                    accepted = true;
                    ++nacc;
                    while (jsave < D) {
                        const double tsj = tsv[jsave];
                        if (!(tsj <= tnew)) break;
                        const bool at_end = (tsj == tnew);
                        double bth[7], vi[3], vpi[3];
                        Ts5::dense(at_end ? 1.0 : (tsj - t) / dt, bth);
#pragma unroll
                        for (int i = 0; i < 3; ++i) {
                            double a = 0.0, ap_ = 0.0;
#pragma unroll
                            for (int j = 0; j < 7; ++j) { a = fma(bth[j], k[j][i], a); ap_ = fma(bth[j], kp[j][i], ap_); }
                            vi[i] = at_end ? unew[i] : fma(dt, a, u[i]);
                            vpi[i] = at_end ? snew[i] : fma(dt, ap_, s[i]);
                        }
                        save_at(vi, vpi, tsj, jsave);
                        ++jsave;
                    }
                } else {
                    ++nrej;
                    dt = dt / fmin(1.0 / prm.qmin, exp(lq11) / prm.gamma);
                }
            } else {
                // ---------------------------------------------------------------- TRBDF2 attempt, the direction through the Newton iterations
                have_eig = true;
                const double gW = tb_d * dt;
                // J' v of this lane's direction with the Jacobian the cache holds: J = S diag(a), S = [[-1, 0, 0], [nu2, -1, 0], [0, nu3, -1]]
                auto jp_times = [&](const double (&v)[3], double (&o)[3]) {
                    o[0] = -nl_ap[0] * v[0];
                    o[1] = fma(th[15], nl_ap[0] * v[0], -nl_ap[1] * v[1]) + (m == 15 ? nl_a[0] * v[0] : 0.0);
                    o[2] = fma(th[16], nl_ap[1] * v[1], -nl_ap[2] * v[2]) + (m == 16 ? nl_a[1] * v[1] : 0.0);
                };
                auto wsolve = [&](double (&b)[3]) {      // W = J - I / (gamma dt) with the cache's J and gamma dt
                    const double wi = 1.0 / nl_Wgdt;
                    b[0] = b[0] / (-nl_a[0] - wi);
                    b[1] = (b[1] - th[15] * nl_a[0] * b[0]) / (-nl_a[1] - wi);
                    b[2] = (b[2] - th[16] * nl_a[1] * b[1]) / (-nl_a[2] - wi);
                };
                // one nlsolve! call: z = dt f(tmp + d z, t + cst dt), the direction's (tmp', z') riding along; false = the step fails
                auto nlsolve = [&](bool isfs, double cst, const double (&tmp)[3], double (&z)[3], const double (&tmpp)[3], double (&zp_)[3]) -> bool {
                    const double inv_gdt = 1.0 / gW, tstep = last && cst == 1.0 ? tnew : fma(cst, dt, t);
                    for (int redo = 0; redo < 3; ++redo) {
                        bool new_jac, new_W;
                        if (iter <= 1 || nl_first) { new_jac = true; new_W = true; }
                        else {
                            const bool errorfail = ee_prev > 1.0;
                            const bool freshJ = (t == nl_Jt) && !errorfail;
                            bool jbad = false, small = true;
                            if (!freshJ) {
                                small = fabs(inv_gdt / (1.0 / nl_Wgdt) - 1.0) <= 0.2;
                                jbad = (nl_status == NL_TRYAGAIN) && small;
                            }
                            const bool wbad = (!small) || (isfs && errorfail) || nl_status == NL_DIV;
                            new_jac = jbad; new_W = jbad || wbad;
                        }
                        if (new_jac) {   // J = df/du at (uprev, t) and its partial along the direction at (uprev, t; s)
                            CathPoint Pj;
                            double rp_[3];
                            point_at(u, t, Pj);
#pragma unroll
                            for (int j = 0; j < 3; ++j) { nl_a[j] = Pj.r[j] * th[12 + j] * Pj.g[j]; rp_[j] = Pj.r[j] * fma(th[12 + j] * Pj.g[j], s[j], dzdir(Pj, j)); }
                            aprime(Pj, s, rp_, nl_ap);
                            nl_Jt = t;
                            eig = fmax(fabs(nl_a[0]), fmax(fabs(th[15] * nl_a[0]) + fabs(nl_a[1]), fabs(th[16] * nl_a[1]) + fabs(nl_a[2])));
                        }
                        if (new_W) nl_Wgdt = gW;
                        nl_status = NL_DIV;   // check_div: what a loop that runs out of iterations leaves behind
                        double ndz = 0.0, ndzprev = 0.0;
                        for (int it = 1; it <= 10; ++it) {
                            double us[3], fs[3], dz[3], usp[3], rp_[3], fsp[3], dzp[3], jd[3];
#pragma unroll
                            for (int i = 0; i < 3; ++i) { us[i] = fma(tb_d, z[i], tmp[i]); usp[i] = fma(tb_d, zp_[i], tmpp[i]); }
                            CathPoint Ps;
                            point_at(us, tstep, Ps);
                            cath_f(Ps, th, fs);
                            fprime(Ps, usp, rp_, fsp);
#pragma unroll
                            for (int i = 0; i < 3; ++i) dz[i] = (dt * fs[i] - z[i]) * inv_gdt;
                            wsolve(dz);
                            jp_times(dz, jd);
#pragma unroll
                            for (int i = 0; i < 3; ++i) dzp[i] = (dt * fsp[i] - zp_[i]) * inv_gdt - jd[i];
                            wsolve(dzp);
                            double ss = 0.0;
#pragma unroll
                            for (int i = 0; i < 3; ++i) {
                                const double e = dz[i] / fma(prm.rtol, fmax(fabs(u[i]), fabs(us[i])), prm.atol);
                                ss = fma(e, e, ss);
                            }
                            ndzprev = ndz;
                            ndz = sqrt(ss * (1.0 / 3.0));
                            if (!isfinite(ndz)) { nl_status = NL_DIV; break; }
                            double theta = 0.0;
                            if (it > 1) {
                                theta = ndz / ndzprev;
                                if (fabs(theta - 1.0) <= 10.0 * 2.220446049250313e-16) { nl_status = ndz <= 1.0 ? NL_CONV : NL_DIV; break; }
                                if (theta > 2.0) { nl_status = NL_DIV; break; }
                            }
#pragma unroll
                            for (int i = 0; i < 3; ++i) { z[i] -= dz[i]; zp_[i] -= dzp[i]; }   // apply_step!, the copy with it
                            const double eta = theta / (1.0 - theta);
                            if ((it == 1 && ndz < 1e-5) || (it > 1 && eta >= 0.0 && eta * ndz < 1.0 / 100.0)) { nl_status = NL_CONV; break; }
                        }
                        if (nl_status == NL_DIV && !(t == nl_Jt)) { nl_status = NL_TRYAGAIN; continue; }   // @goto REDO
                        break;
                    }
                    nl_first = false;   // postamble!
                    return nl_status >= 0;
                };
                double zp[3], zg[3], z[3], tmp[3], zpp[3], zgp[3], zq[3], tmpp[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    zp[i] = dt * f0[i]; zg[i] = zp[i]; tmp[i] = fma(tb_d, zp[i], u[i]);
                    zpp[i] = dt * f0p[i]; zgp[i] = zpp[i]; tmpp[i] = fma(tb_d, zpp[i], s[i]);
                    z[i] = 0.0; zq[i] = 0.0;
                }
                stepfail = !nlsolve(true, tb_g, tmp, zg, tmpp, zgp);
                if (!stepfail) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        z[i] = tb_a1 * zp[i] + tb_a2 * zg[i]; tmp[i] = u[i] + tb_w * zp[i] + tb_w * zg[i];
                        zq[i] = tb_a1 * zpp[i] + tb_a2 * zgp[i]; tmpp[i] = s[i] + tb_w * zpp[i] + tb_w * zgp[i];
                    }
                    stepfail = !nlsolve(false, 1.0, tmp, z, tmpp, zq);
                }
                if (stepfail) {   // force_stepfail: dt / failfactor, no controller call, EEst as it was
                    ++nrej;
                    dt *= 0.5;
                } else {
                    double est[3], estp[3], jd[3];
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        unew[i] = fma(tb_d, z[i], tmp[i]); snew[i] = fma(tb_d, zq[i], tmpp[i]);
                        f2[i] = z[i] / dt; f2p[i] = zq[i] / dt;                              // fsallast = z ./ dt
                        est[i] = tb_bt1 * zp[i] + tb_bt2 * zg[i] + tb_bt3 * z[i];
                        estp[i] = tb_bt1 * zpp[i] + tb_bt2 * zgp[i] + tb_bt3 * zq[i];
                    }
                    wsolve(est);                              // smooth_est: get_W(nlsolver) \ tmp
                    jp_times(est, jd);
#pragma unroll
                    for (int i = 0; i < 3; ++i) estp[i] -= jd[i];
                    wsolve(estp);                             // its partial: W^-1 (tmp' - J' est)
                    dual_norm(est, estp);
                    if (!finite) { rc = 3; break; }
                    if (controller(b1_rb, b2_rb)) {
                        accepted = true;
                        ++nacc;
                        while (jsave < D) {
                            const double tsj = tsv[jsave];
                            if (!(tsj <= tnew)) break;
                            const bool at_end = (tsj == tnew);
                            const double Th = at_end ? 1.0 : (tsj - t) / dt;
                            double vi[3], vpi[3];
#pragma unroll
                            for (int i = 0; i < 3; ++i) {   // Hermite on (uprev, u, fsalfirst, fsallast), value and partial
                                const double dy = unew[i] - u[i], dyp = snew[i] - s[i];
                                const double hm = (1.0 - Th) * u[i] + Th * unew[i] + Th * (Th - 1.0) * ((1.0 - 2.0 * Th) * dy + (Th - 1.0) * dt * f0[i] + Th * dt * f2[i]);
                                const double hp = (1.0 - Th) * s[i] + Th * snew[i] + Th * (Th - 1.0) * ((1.0 - 2.0 * Th) * dyp + (Th - 1.0) * dt * f0p[i] + Th * dt * f2p[i]);
                                vi[i] = at_end ? unew[i] : hm;
                                vpi[i] = at_end ? snew[i] : hp;
                            }
                            save_at(vi, vpi, tsj, jsave);
                            ++jsave;
                        }
                    } else {
                        ++nrej;
                        dt = dt / fmin(1.0 / prm.qmin, exp(lq11) / prm.gamma);
                    }
                }
            }
            if (!stepfail) ee_prev = sqrt(es);   // integrator.EEst of this attempt (do_newJW's errorfail)
            if (accepted) {
#pragma unroll
                for (int i = 0; i < 3; ++i) { u[i] = unew[i]; f0[i] = f2[i]; s[i] = snew[i]; f0p[i] = f2p[i]; }
                if (alg == 0) P0 = P2;
                t = tnew;
                if (q >= prm.qsteady_min && q <= prm.qsteady_max) q = 1.0;
                lqold = ee_zero ? lqinit : fmax(lEE, lqinit);
                dt = fmin(dt / q, tend - t0);
                if (jsave >= D) rc = 0;
            }
        }
        {
            const double invD = 1.0 / (double)D;   // the FULL row count, also for a truncated solution (network.jl:266)
            if (m >= 0) prm.grad[(size_t)traj * kCathNP + m] = gsum * invD;     // this chunk's entries of the gradient row
            if (col == 0) {
                prm.loss[traj] = loss_sum * invD;
                prm.retcode[traj] = rc;
                prm.n_saved[traj] = jsave;
                prm.n_accept[traj] = nacc;
                prm.n_reject[traj] = nrej;
            }
        }
        traj += groups_total;
    }
}

}  // namespace crnn

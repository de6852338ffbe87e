// crnn_amd/csrc/hychem_auto_kernel.hpp -- gfx950 (MI355X): the HyChem pyrolysis CRNN through the reference's COMPOSITE stepper,
// primal launches (predict_n_ode / loss_n_ode: HyChem/crnn_pyrolysis_mass.jl:135-147).
//
// Reference: crnn_pyrolysis_mass.jl:29  ode_solver = AutoTsit5(Rosenbrock23(autodiff=false)), used by predict_n_ode (:138-139).
// A context created with crnn_config_set_solver(cfg, CRNN_SOLVER_AUTOTSIT5) runs its primal launches here:
//   * Tsit5 with the stage times t + c_s dt on the T(t), P(t) tables (piecewise linear, :103-104), free 4th-order interpolant;
//   * OrdinaryDiffEq's AutoSwitch rule as auto_adj_kernel.hpp restates it (eigen estimate of every attempt -- Hairer's
//     |k7 - k6| / |g7 - g6| in the Inf norm after a Tsit5 attempt, opnorm(J, Inf) after a Rosenbrock23 attempt --, more than 10
//     stiff / 3 non-stiff verdicts in a row, dt * 2 and dt / 2 at the switches, PI exponents of the running algorithm);
//   * non-autonomous Rosenbrock23 with the analytic Jacobian as the stiff algorithm (the reference's autodiff=false takes
//     FiniteDiff increments: INTEGRATION.md) -- hychem2_kernel's step, operation for operation.
// The tests compare it with a CPU statement of the same composite.  All of it [UNVERIFIED-DEP] like the other
// steppers (no Manifest for HyChem; the packages are not in the reference tree).
//
// Mapping: hychem2_kernel's -- a lane PAIR per trajectory, everything of length ns distributed over the pair (species 2 i + m in
// slot i of lane m), W's rows in registers, the point's rates in the lane's LDS frame -- without tape, reverse sweep and MFMA stage:
// the loss is accumulated at the save points as they are passed.  Gradient launches of such a context run the Rosenbrock23
// adjoint (hychem2_kernel) with Rosenbrock23's controller constants: the adjoint through explicit steps at their stability limit
// is not something to hand to an optimiser (cathode_auto_kernel.hpp has the measurements).
#pragma once
#include "auto_adj_kernel.hpp"
#include "hychem2_kernel.hpp"

namespace crnn {

// Rosenbrock23(autodiff = false) -- what the reference configures (crnn_pyrolysis_mass.jl:29): this lane's rows of W = I - gam J with J from
// FiniteDiff's forward differences of the right-hand side, column c = (f(u + eps_c e_c, t) - f(u, t)) / eps_c, eps_c = max(sqrt(eps) |u_c|,
// sqrt(eps)), against the FSAL value f(u, t), and the time derivative dT = (f(u, t + e_t) - f(u, t)) / e_t, e_t = max(sqrt(eps) |t|, sqrt(eps)),
// on the T(t), P(t) tables (Te, Pe = the tables at t + e_t).  NS + 1 point evaluations per attempt instead of one analytic pass: a parity
// mode (primal launches; crnn_ctx_set_jacobian), checked against a CPU statement of the same increments (tests/test_hychem.py).  [UNVERIFIED-DEP]: FiniteDiff.jl is not in the reference tree.
template <int NS, int NR>
__device__ __forceinline__ void hy_jac_ft2_fd(const double *th, const KConst *kc, const double inv_R, const double (&uo)[(NS + 1) / 2],
                                              const double (&f0)[(NS + 1) / 2], const double T, const double P, const double Te, const double Pe,
                                              const double et, const double gam, const bool m1, const HyLane<(NS + 1) / 2> &ln,
                                              double (&A)[(NS + 1) / 2][NS], double (&fto)[(NS + 1) / 2]) {
    constexpr int H = (NS + 1) / 2;
    constexpr double rel = 1.4901161193847656e-08;   // sqrt(eps(Float64)): FiniteDiff's default relative step of forward differences
#pragma unroll
    for (int c = 0; c < NS; ++c) {
        const bool codd = (c & 1) != 0;
        const double uc = pair_pick(uo[c >> 1], codd, m1);
        const double eps = fmax(rel * fabs(uc), rel);
        double up[H];
#pragma unroll
        for (int i = 0; i < H; ++i) up[i] = uo[i];
        if (m1 == codd) up[c >> 1] = uc + eps;
        HyPoint2<NS, NR> pp;
        hy_point2<NS, NR, 1>(th, kc, inv_R, up, T, P, m1, ln, pp, nullptr);
        const double ie = 1.0 / eps;
#pragma unroll
        for (int i = 0; i < H; ++i) {
            const bool diag = (c == 2 * i) ? !m1 : ((c == 2 * i + 1) ? m1 : false);
            A[i][c] = (diag ? 1.0 : 0.0) - gam * ((pp.fo[i] - f0[i]) * ie);
        }
#pragma unroll
        for (int i = 0; i < H; ++i) opaque(A[i][c]);   // column by column: the NS evaluations are not interleaved (registers)
        CRNN_SCHED_FENCE();
    }
    HyPoint2<NS, NR> pe;
    hy_point2<NS, NR, 1>(th, kc, inv_R, uo, Te, Pe, m1, ln, pe, nullptr);
    const double iet = 1.0 / et;
#pragma unroll
    for (int i = 0; i < H; ++i) fto[i] = ln.ow[i] ? (pe.fo[i] - f0[i]) * iet : 0.0;
}

// JFD: the stiff algorithm's J and dT by forward differences (above).  STIFF_ONLY: no Tsit5 branch and no switching -- plain Rosenbrock23 with the
// context's PI exponents (hychem2_kernel's primal launch operation for operation; instantiated with JFD for CRNN_SOLVER_ROSENBROCK23 contexts
// in finite-difference mode).
template <int NS, int NR, int BLOCK, bool JFD = false, bool STIFF_ONLY = false>
__global__ __launch_bounds__(BLOCK) void hychem_auto_kernel(const SolveParams prm, const double *__restrict__ theta, const HyParams hp) {
    using L_ = LayH<NS, NR>;
    constexpr int NTH = L_::NTH;
    constexpr int H = (NS + 1) / 2;
    __shared__ double kc_lds[kNConst];
    __shared__ double ts_lds[kMaxSave];
    __shared__ double th_lds[NTH];
    // the lane's LDS frame as in hychem2_kernel: rates of the FSAL point (0-9) and of the new point (10-19); the FSAL point's Y, irho,
    // iS (58-64) -- what a Rosenbrock23 attempt needs of the point the previous attempt (of either algorithm) ended on
    constexpr int NFR = 65;
    __shared__ double fr_lds[NFR * BLOCK];
    const int tid = threadIdx.x;
    double *const fr = fr_lds + tid * NFR;
#define FRA(k_) fr[(k_)]
#define HYA_FRESH_THETA(ptr)                    \
    do {                                        \
        unsigned z_ = 0;                        \
        asm volatile("" : "+s"(z_));            \
        (ptr) = th_lds + z_;                    \
    } while (0)
#define HYA_FRESH_KC(ptr)                                            \
    do {                                                             \
        unsigned z_ = 0;                                             \
        asm volatile("" : "+s"(z_));                                 \
        (ptr) = reinterpret_cast<const KConst *>(kc_lds + z_);       \
    } while (0)
    const int lane = tid & 63;
    const bool m1 = (lane & 1) != 0;
    const int giw = lane >> 1;
    for (int idx = tid; idx < kNConst; idx += BLOCK) kc_lds[idx] = reinterpret_cast<const double *>(prm.kc)[idx];
    for (int idx = tid; idx < hp.n_save_total; idx += BLOCK) ts_lds[idx] = prm.tsave[idx];
    for (int idx = tid; idx < NTH; idx += BLOCK) th_lds[idx] = theta[idx];
    __syncthreads();
    const KConst *kc = reinterpret_cast<const KConst *>(kc_lds);
    const double *th = th_lds;
    HyLane<H> ln;
#pragma unroll
    for (int i = 0; i < H; ++i) {
        const int c = 2 * i + (m1 ? 1 : 0);
        ln.ow[i] = c < NS;
        ln.ci[i] = ln.ow[i] ? c : 0;
    }
    const double d_ = 0.29289321881345248, c32 = 7.4142135623730950, inv12d = 2.4142135623730950;
    constexpr double b1_ts = 7.0 / 50.0, b2_ts = 2.0 / 25.0, b1_rb = 7.0 / 20.0, b2_rb = 2.0 / 10.0;
    const int nsave = prm.n_save, Dfull = hp.n_save_total;
    const double tend = to_sgpr(ts_lds[nsave - 1]), ts0 = to_sgpr(ts_lds[0]), t0 = to_sgpr(kc->t0);
    const double dtmax = to_sgpr(tend - t0);
    const double lqinit = to_sgpr(flog(kc->qoldinit));
    const double inv_qmax = to_sgpr(1.0 / kc->qmax), inv_qmin = to_sgpr(1.0 / kc->qmin);
    const bool start_saved = (ts0 == t0);

    while (true) {
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(prm.queue, 32ULL);
        const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)base);
        const unsigned bhi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
        const int64_t wave_base = (int64_t)(((unsigned long long)bhi << 32) | blo);
        if (wave_base >= prm.count) break;
        const int64_t traj = wave_base + giw;
        const bool valid = traj < prm.count;
        const int64_t b = prm.first + (valid ? (hp.perm ? (int64_t)hp.perm[traj] : traj) : 0);
        CRNN_CHK(b >= prm.first && b < prm.first + prm.count && b < prm.B, 21);
        const double *const tabT = hp.tabs + (size_t)b * 2 * Dfull;
        const double *const tabP = tabT + Dfull;

        int seg = -1;
        double Ta = 0, Tb = 0, Pa = 0, Pb = 0, tsa = 0, idts = 0;
        auto tab = [&](const double tq, double &T, double &P, double &Td, double &Pd) {
            int sg = seg < 0 ? 0 : seg;
            while (sg + 1 < Dfull - 1 && ts_lds[sg + 1] <= tq) ++sg;
            while (sg > 0 && ts_lds[sg] > tq) --sg;
            if (sg != seg) {
                seg = sg;
                CRNN_CHK(sg >= 0 && sg + 1 < Dfull, 20);
                Ta = tabT[sg]; Tb = tabT[sg + 1]; Pa = tabP[sg]; Pb = tabP[sg + 1];
                tsa = ts_lds[sg];
                idts = frcp(ts_lds[sg + 1] - tsa);
            }
            Td = (Tb - Ta) * idts;
            Pd = (Pb - Pa) * idts;
            T = fma(tq - tsa, Td, Ta);
            P = fma(tq - tsa, Pd, Pa);
        };

        double u[H], f0[H];
        HyPoint2<NS, NR> p0;
        unsigned f0cY = 0, f0cC = 0;
        double t = t0, dt = 0.0, lqold = lqinit;
        int iter = 0, jsave = 0, nacc = 0, nrej = 0;
        int rc = valid ? -1 : 0;
        int alg = STIFF_ONLY ? 1 : 0, cnt = 0;          // 0 Tsit5, 1 Rosenbrock23; signed run length of the stiffness test
        double eig = 0.0;
        bool have_eig = false;
#pragma unroll
        for (int i = 0; i < H; ++i) u[i] = ln.ow[i] ? prm.u0[(size_t)ln.ci[i] * prm.B + b] : 0.0;
        {
            double T, P, Td, Pd;
            tab(t0, T, P, Td, Pd);
            hy_point2<NS, NR, 1>(th, kc, hp.inv_R, u, T, P, m1, ln, p0, fr);
#pragma unroll
            for (int i = 0; i < H; ++i) FRA(58 + i) = p0.Yo[i];
            FRA(63) = p0.irho; FRA(64) = p0.iS;
            f0cY = p0.cY; f0cC = p0.cC;
            double d0 = 0.0, d1 = 0.0, sk[H];
#pragma unroll
            for (int i = 0; i < H; ++i) {
                sk[i] = ln.ow[i] ? frcp(fma(fabs(u[i]), kc->rtol[ln.ci[i]], kc->atol[ln.ci[i]])) : 0.0;
                const double a = u[i] * sk[i], c = p0.fo[i] * sk[i];
                d0 = fma(a, a, d0);
                d1 = fma(c, c, d1);
            }
            d0 = sqrt(pair_sum(d0) * (1.0 / NS));
            d1 = sqrt(pair_sum(d1) * (1.0 / NS));
            double dt0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * (d0 / d1);
            dt0 = fmin(dt0, dtmax);
            double u1[H];
#pragma unroll
            for (int i = 0; i < H; ++i) u1[i] = fma(dt0, p0.fo[i], u[i]);
            HyPoint2<NS, NR> p1;
            tab(t0 + dt0, T, P, Td, Pd);
            hy_point2<NS, NR, 1>(th, kc, hp.inv_R, u1, T, P, m1, ln, p1, nullptr);
            double d2 = 0.0;
#pragma unroll
            for (int i = 0; i < H; ++i) { const double e = (p1.fo[i] - p0.fo[i]) * sk[i]; d2 = fma(e, e, d2); }
            d2 = sqrt(pair_sum(d2) * (1.0 / NS)) / dt0;
            const double dm = fmax(d1, d2);
            // the order of the STARTING algorithm (5): 10^(-(2 + log10 dm) / 5)
            const double dt1 = (dm <= 1e-15) ? fmax(1e-6, dt0 * 1e-3) : exp((STIFF_ONLY ? -0.5 : -0.2) * (4.605170185988091368 + flog(dm)));
            dt = fmax(kc->dtmin, fmin(fmin(100.0 * dt0, dt1), dtmax));
#pragma unroll
            for (int i = 0; i < H; ++i) f0[i] = p0.fo[i];
        }
        double pf_loss = 0.0;
        // a save point: prediction out, its loss term in (this lane's species)
        auto save_point = [&](const double (&v_)[H], const int j) {
            CRNN_CHK(j >= 0 && (int64_t)(j + 1) * prm.n_obs <= prm.row_stride && j < Dfull, 24);
            const double *prow = prm.data + (size_t)b * prm.row_stride + (size_t)j * prm.n_obs;
#pragma unroll
            for (int i = 0; i < H; ++i) {
                if (ln.ow[i]) {
                    double v = v_[i];
                    if (prm.clamp_pred) v = clampv(v, -kc->ub, kc->ub);
                    if (prm.pred) prm.pred[((size_t)j * NS + ln.ci[i]) * prm.B + b] = v;
                    const int dr = (int)kc->drow[ln.ci[i]];
                    if (dr >= 0) {
                        const double rr = (prow[dr] - v) * kc->inv_yscale[ln.ci[i]];
                        pf_loss = (prm.loss_kind == 0) ? pf_loss + fabs(rr) : fma(rr, rr, pf_loss);
                    }
                }
            }
        };
        if (start_saved) {   // saveat contains tspan[1]: the initial point is a save point (prediction and loss term)
            if (valid) save_point(u, 0);
            jsave = 1;
        }

        while (__builtin_amdgcn_ballot_w64(rc < 0) != 0) {
            if (rc < 0) {
                ++iter;
                bool last = false;
                if (jsave >= nsave) rc = 0;
                else if (iter > prm.maxiters) rc = 1;
                if (!STIFF_ONLY && rc < 0 && have_eig) {   // choose_algorithm! at the loop header
                    const bool stiff = fabs(eig * dt * (1.0 / AutoSw::stability_size)) > AutoSw::tol;
                    cnt = stiff ? (cnt < 0 ? 1 : cnt + 1) : (cnt > 0 ? -1 : cnt - 1);
                    if (alg == 0 && cnt > AutoSw::maxstiffstep) { dt *= AutoSw::dtfac; alg = 1; }
                    else if (alg == 1 && cnt < -AutoSw::maxnonstiffstep) { dt *= 1.0 / AutoSw::dtfac; alg = 0; }
                }
                if (t + dt * (1.0 + 1e-13) >= tend) { dt = tend - t; last = true; }
                if (rc < 0 && (!(dt > kc->dtmin) || t + dt == t)) rc = 2;
                if (rc < 0) {
                    HYA_FRESH_THETA(th); HYA_FRESH_KC(kc);
                    const double tnew = last ? tend : t + dt;
                    double unew[H], es = 0.0;
                    int fin = 1;
                    HyPoint2<NS, NR> p2;
                    bool accepted = false, ee_zero = false;
                    double q = 1.0, lEE = 0.0, lq11 = 0.0;
                    auto controller = [&](const double b1, const double b2) -> bool {
                        ee_zero = (es == 0.0);
                        lEE = 0.5 * flog_ctl(ee_zero ? 1.0 : es);
                        lq11 = b1 * lEE;
                        q = ee_zero ? inv_qmax : fmax(inv_qmax, fmin(inv_qmin, fexp_ctl(lq11 - b2 * lqold) / kc->gamma));
                        return es <= 1.0;
                    };
                    if (!STIFF_ONLY && alg == 0) {
                        // ---------------------------------------------------------------- Tsit5 attempt
                        double k[7][H], g6[H];
#pragma unroll
                        for (int i = 0; i < H; ++i) { k[0][i] = f0[i]; g6[i] = 0.0; }
#pragma unroll
                        for (int s = 1; s < 7; ++s) {
                            double g[H];
#pragma unroll
                            for (int i = 0; i < H; ++i) {
                                double a = 0.0;
#pragma unroll
                                for (int j = 0; j < 6; ++j)
                                    if (j < s) a = fma(Ts5::a(s - 1, j), k[j][i], a);
                                g[i] = fma(dt, a, u[i]);
                            }
                            const double tq = s == 6 ? tnew : s == 5 ? t + dt : fma(s == 1 ? Ts5::c2 : s == 2 ? Ts5::c3 : s == 3 ? Ts5::c4 : Ts5::c5, dt, t);
                            double Tq, Pq, a_, b_;
                            tab(tq, Tq, Pq, a_, b_);
                            HYA_FRESH_THETA(th); HYA_FRESH_KC(kc);
                            if (s == 6) {
#pragma unroll
                                for (int i = 0; i < H; ++i) unew[i] = g[i];
                                hy_point2<NS, NR, 1>(th, kc, hp.inv_R, g, Tq, Pq, m1, ln, p2, fr + 10);
#pragma unroll
                                for (int i = 0; i < H; ++i) k[6][i] = p2.fo[i];
                            } else {
                                if (s == 5) {
#pragma unroll
                                    for (int i = 0; i < H; ++i) g6[i] = g[i];
                                }
                                HyPoint2<NS, NR> ps;
                                hy_point2<NS, NR, 1>(th, kc, hp.inv_R, g, Tq, Pq, m1, ln, ps, nullptr);
#pragma unroll
                                for (int i = 0; i < H; ++i) k[s][i] = ps.fo[i];
                            }
                            opaque(k[s]);
                            CRNN_SCHED_FENCE();
                        }
                        double est = 0.0;
                        int nan_ = 0;
#pragma unroll
                        for (int i = 0; i < H; ++i) {
                            double a = 0.0;
#pragma unroll
                            for (int j = 0; j < 7; ++j) a = fma(Ts5::bt(j), k[j][i], a);
                            const double ev = dt * a;
                            const double mx = fmax(fabs(u[i]), fabs(unew[i]));
                            const double e = ln.ow[i] ? ev * frcp(fma(kc->rtol[ln.ci[i]], mx, kc->atol[ln.ci[i]])) : 0.0;
                            es = fma(e, e, es);
                            fin &= (!ln.ow[i] || (isfinite(unew[i]) && isfinite(ev))) ? 1 : 0;
                            const double qq = fabs((k[6][i] - k[5][i]) / (unew[i] - g6[i]));   // Hairer II p.22, Inf norm; NaN propagates
                            if (ln.ow[i]) { nan_ |= (qq != qq) ? 1 : 0; est = fmax(est, qq); }
                        }
                        es = pair_sum(es) * (1.0 / NS);
                        est = fmax(est, pair_other(est));
                        nan_ |= pair_other_i(nan_);
                        eig = nan_ ? __longlong_as_double(0x7ff8000000000000LL) : est;
                        have_eig = true;
                        if (pair_and(fin) == 0) rc = 3;
                        else if (controller(b1_ts, b2_ts)) {
                            accepted = true;
                            ++nacc;
                            while (jsave < nsave) {
                                const double tsj = ts_lds[jsave];
                                if (!(tsj <= tnew)) break;
                                const bool at_end = (tsj == tnew);
                                double bth[7], v[H];
                                Ts5::dense(at_end ? 1.0 : (tsj - t) / dt, bth);
#pragma unroll
                                for (int i = 0; i < H; ++i) {
                                    double a = 0.0;
#pragma unroll
                                    for (int j = 0; j < 7; ++j) a = fma(bth[j], k[j][i], a);
                                    v[i] = at_end ? unew[i] : fma(dt, a, u[i]);
                                }
                                save_point(v, jsave);
                                ++jsave;
                            }
                        } else {
                            ++nrej;
                            dt = dt / fmin(inv_qmin, fexp_ctl(lq11) / kc->gamma);
                        }
                    } else {
                        // ---------------------------------------------------------------- Rosenbrock23 attempt (hychem2_kernel's)
                        const double gam = d_ * dt;
                        double T, P, Td, Pd;
                        tab(t, T, P, Td, Pd);
                        double A[H][NS], dinv[NS], ft[H];
                        unsigned long long piv;
                        bool anyp;
                        {
                            unsigned zf_ = 0;
                            asm volatile("" : "+v"(zf_));
                            const double *const fq = fr + zf_;
#pragma unroll
                            for (int i = 0; i < H; ++i) { p0.Yo[i] = fq[58 + i]; p0.fo[i] = f0[i]; }
                            p0.irho = fq[63]; p0.iS = fq[64];
                            p0.cY = f0cY; p0.cC = f0cC;
                        }
                        if constexpr (JFD) {
                            const double et = fmax(1.4901161193847656e-08 * fabs(t), 1.4901161193847656e-08);
                            double Te, Pe, a_, b_;
                            tab(t + et, Te, Pe, a_, b_);
                            hy_jac_ft2_fd<NS, NR>(th, kc, hp.inv_R, u, f0, T, P, Te, Pe, et, gam, m1, ln, A, ft);
                            HYA_FRESH_THETA(th); HYA_FRESH_KC(kc);
                        } else
                        hy_jac_ft2<NS, NR, 1>(th, kc, p0, fr, gam, Pd * frcp(P) - Td * frcp(T), -hp.inv_R * Td * frcp(T * T), Td * frcp(T), m1, ln, A, ft);
                        CRNN_SCHED_FENCE();
                        {   // eigen_est = opnorm(J, Inf): J = (I - A) / gam, this lane's rows, then the pair's maximum
                            const double ig = 1.0 / gam;
                            double est = 0.0;
#pragma unroll
                            for (int i = 0; i < H; ++i) {
                                double a = 0.0;
#pragma unroll
                                for (int c = 0; c < NS; ++c) {
                                    const bool diag = (c == 2 * i) ? !m1 : ((c == 2 * i + 1) ? m1 : false);
                                    a += fabs(((diag ? 1.0 : 0.0) - A[i][c]) * ig);
                                }
                                if (ln.ow[i]) est = fmax(est, a);
                            }
                            eig = fmax(est, pair_other(est));
                            have_eig = true;
                        }
                        const bool okf = lu2_factor<NS>(A, m1, dinv, piv, anyp);
                        const bool wp = __builtin_amdgcn_ballot_w64(anyp) != 0;
                        double k1[H], dk[H], f1[H];
#pragma unroll
                        for (int i = 0; i < H; ++i) k1[i] = fma(gam, ft[i], f0[i]);
                        lu2_solve<NS>(A, dinv, piv, wp, m1, k1);
                        CRNN_SCHED_FENCE();
                        HYA_FRESH_THETA(th); HYA_FRESH_KC(kc);
                        {
                            double u1[H];
#pragma unroll
                            for (int i = 0; i < H; ++i) u1[i] = fma(0.5 * dt, k1[i], u[i]);
                            HyPoint2<NS, NR> p1;
                            double T1, P1, a_, b_;
                            tab(t + 0.5 * dt, T1, P1, a_, b_);
                            hy_point2<NS, NR, 1>(th, kc, hp.inv_R, u1, T1, P1, m1, ln, p1, nullptr);
#pragma unroll
                            for (int i = 0; i < H; ++i) f1[i] = p1.fo[i];
                            opaque(f1);
                        }
#pragma unroll
                        for (int i = 0; i < H; ++i) dk[i] = f1[i] - k1[i];
                        lu2_solve<NS>(A, dinv, piv, wp, m1, dk);
#pragma unroll
                        for (int i = 0; i < H; ++i) unew[i] = fma(dt, k1[i] + dk[i], u[i]);
                        CRNN_SCHED_FENCE();
                        HYA_FRESH_THETA(th); HYA_FRESH_KC(kc);
                        {
                            double T2, P2, a_, b_;
                            tab(tnew, T2, P2, a_, b_);
                            hy_point2<NS, NR, 1>(th, kc, hp.inv_R, unew, T2, P2, m1, ln, p2, fr + 10);
                        }
                        CRNN_SCHED_FENCE();
                        double k3[H];
#pragma unroll
                        for (int i = 0; i < H; ++i) {
                            const double k2i = k1[i] + dk[i];
                            k3[i] = fma(dt, ft[i], p2.fo[i] - c32 * (k2i - f1[i]) - 2.0 * (k1[i] - f0[i]));
                        }
                        lu2_solve<NS>(A, dinv, piv, wp, m1, k3);
                        fin = okf ? 1 : 0;
#pragma unroll
                        for (int i = 0; i < H; ++i) {
                            const double k2i = k1[i] + dk[i];
                            const double ev = dt * (1.0 / 6.0) * (k1[i] - 2.0 * k2i + k3[i]);
                            const double mx = fmax(fabs(u[i]), fabs(unew[i]));
                            const double e = ln.ow[i] ? ev * frcp(fma(kc->rtol[ln.ci[i]], mx, kc->atol[ln.ci[i]])) : 0.0;
                            es = fma(e, e, es);
                            fin &= (!ln.ow[i] || (isfinite(unew[i]) && isfinite(ev))) ? 1 : 0;
                        }
                        es = pair_sum(es) * (1.0 / NS);
                        if (pair_and(fin) == 0) rc = 3;
                        else if (STIFF_ONLY ? controller(kc->beta1, kc->beta2) : controller(b1_rb, b2_rb)) {
                            accepted = true;
                            ++nacc;
                            while (jsave < nsave) {
                                const double tsj = ts_lds[jsave];
                                if (!(tsj <= tnew)) break;
                                const bool at_end = (tsj == tnew);
                                const double Th = at_end ? 1.0 : (tsj - t) / dt;
                                const double c1 = at_end ? 0.0 : Th * (1.0 - Th) * inv12d;
                                const double c2 = at_end ? 1.0 : Th * (Th - 2.0 * d_) * inv12d;
                                double v[H];
#pragma unroll
                                for (int i = 0; i < H; ++i) {
                                    const double k2i = k1[i] + dk[i];
                                    v[i] = at_end ? unew[i] : fma(dt, fma(c1, k1[i], c2 * k2i), u[i]);
                                }
                                save_point(v, jsave);
                                ++jsave;
                            }
                        } else {
                            ++nrej;
                            dt = dt / fmin(inv_qmin, fexp_ctl(lq11) / kc->gamma);
                        }
                    }
                    if (accepted) {
#pragma unroll
                        for (int i = 0; i < H; ++i) u[i] = unew[i];
#pragma unroll
                        for (int i = 0; i < H; ++i) { f0[i] = p2.fo[i]; FRA(58 + i) = p2.Yo[i]; }
                        FRA(63) = p2.irho; FRA(64) = p2.iS; f0cY = p2.cY; f0cC = p2.cC;
                        {   // the new point's rates (slots 10-19) become the FSAL point's (0-9)
                            double rr_[NR];
#pragma unroll
                            for (int j = 0; j < NR; ++j) rr_[j] = FRA(10 + j);
#pragma unroll
                            for (int j = 0; j < NR; ++j) FRA(j) = rr_[j];
                        }
                        t = tnew;
                        if (q >= kc->qsteady_min && q <= kc->qsteady_max) q = 1.0;
                        lqold = ee_zero ? lqinit : fmax(lEE, lqinit);
                        dt = fmin(dt / q, dtmax);
                        if (jsave >= nsave) rc = 0;
                    }
                }
            }
        }

        const double loss_tot = pair_sum(pf_loss);
        if (valid && !m1) {
            const double denom = (double)prm.n_obs * (double)jsave;
            prm.loss[b] = jsave > 0 ? loss_tot / denom : 0.0;
            prm.retcode[b] = rc;
            prm.n_saved[b] = jsave;
            prm.n_accept[b] = nacc;
            prm.n_reject[b] = nrej;
        }
    }
#undef FRA
#undef HYA_FRESH_THETA
#undef HYA_FRESH_KC
}

}  // namespace crnn

// crnn_amd/csrc/auto_adj_kernel.hpp -- gfx950 (MI355X): Tsit5 and the AutoTsit5(Rosenbrock23()) composite with the
// loss gradient by the DISCRETE ADJOINT of the accepted steps.
//
// Reference algorithms: `alg = Tsit5()` (case1/case1.jl:28,94-95) and `alg = AutoTsit5(Rosenbrock23(autodiff=false))`
// (case2/case2.jl:26, HyChem/crnn_pyrolysis_mass.jl:29); the gradient is ForwardDiff's (case2/case2.jl:195): the
// derivative of the solver's arithmetic with dt, the accept/reject decisions, the algorithm choices and the saveat
// weights held fixed.  Same wave-synchronous forward sweep / reverse sweep over a per-lane tape (t_n, dt_n, u_n) as
// ros23_adj_kernel.hpp; a Tsit5 step is recorded with -dt_n, so the tape layout does not change.
//
//   Tsit5 forward   k_1 = f(u_n) (FSAL),  g_s = u_n + dt sum_{j<s} a_sj k_j,  k_s = f(g_s)  (s = 2..7),  u_{n+1} = g_7
//                   saveat: u(t_n + Th dt) = u_n + dt sum_j b_j(Th) k_j  (j = 1..7, the method's free interpolant)
//   Tsit5 reverse   kb_j = dt a_7j lam + B_j (j <= 6),  kb_7 = B_7,  ub = lam + A     (A, B_j: loss seeds in the step)
//                   for s = 7..1:  gb = f_u(g_s)^T kb_s,  thb += f_theta(g_s)^T kb_s,  ub += gb,  kb_j += dt a_sj gb (j < s)
//                   lam = ub
//   Rosenbrock23    as in ros23_adj_kernel.hpp
//
// Switching (COMPOSITE): OrdinaryDiffEq's AutoSwitch with its defaults, restated from the published algorithm -- the
// package is not in the reference tree [UNVERIFIED-DEP].  Before every attempt after the first,
//     stiffness = |eigen_est dt| / 3.5068  >  9/10
// is tested with the dt about to be tried and eigen_est of the previous attempt; more than 10 positives in a row on
// Tsit5: dt *= 2 and switch to Rosenbrock23; more than 3 negatives in a row on Rosenbrock23: dt /= 2 and switch back.
// eigen_est is max_i |k7_i - k6_i| / |g7_i - g6_i| after a Tsit5 attempt and the Inf-norm of J(u_n) after a
// Rosenbrock23 attempt.  A state vector with a component that never moves (case2 carries its constant temperature as
// the 7th state) gives 0/0 = NaN there, Julia's `maximum` propagates it and NaN > 9/10 is false ([UNVERIFIED-DEP]): such a problem stays
// on Tsit5 for ever, so HAS_T shapes are instantiated with COMPOSITE = false.  The PI exponents follow the running
// algorithm (beta1 = 7/(10 order), beta2 = 2/(5 order)); gamma, qmin, qmax, the steady band and qold are shared.
#pragma once
#define CRNN_AUTO_FENCE() ((void)0)   // (a scheduling-fence experiment of round 3, measured slower: the call sites mark where it stood)
#include "ros23_adj_kernel.hpp"
#include "tsit5_kernel.hpp"

namespace crnn {

struct AutoSw {   // AutoSwitch(nonstiffalg, stiffalg) defaults
    static constexpr int maxstiffstep = 10, maxnonstiffstep = 3;
    static constexpr double tol = 0.9, dtfac = 2.0, stability_size = 3.5068;   // alg_stability_size(Tsit5())
};

// PRIMAL = true: the forward sweep alone (predictions, loss accumulated at the save points, no tape, no reverse sweep) -- the primal
// calls of the Tsit5 / AutoTsit5 problems; without it they ran the whole tape kernel with no directions (AutoTsit5: 0.75 ms per
// 65 536 case2 trajectories, the price of the gradient) or tsit5_kernel's lane groups (Tsit5: 0.47 ms).
template <int NS, int NR, bool HAS_T, bool USE_SCALE, int BLOCK, bool COMPOSITE, bool PRIMAL = false>
__global__ __launch_bounds__(BLOCK) void auto_adj_kernel(const SolveParams prm, const double *__restrict__ theta,
                                                         const AdjParams adj) {
    using L_ = Lay<NS, NR, HAS_T>;
    constexpr int N = L_::N;
    constexpr int NTH = L_::NTH;
    constexpr int RECW = NS + 2;
    static_assert(NTH + kExtra <= 64, "the per-batch sums use one lane per column");
    static_assert(!(COMPOSITE && HAS_T), "a constant state component never switches (see the header)");
    constexpr bool kKeepStages = 7 * (2 * NS + NR) <= 84;   // reverse Tsit5 step: keep the features of all 7 stages in registers
    using Solver = typename SolverSel<(NR < NS), NS, NR, HAS_T, USE_SCALE>::type;

    __shared__ double kc_lds[kNConst];
    __shared__ double ts_lds[kMaxSave];
    __shared__ double thb_lds[NTH * BLOCK];     // gradient accumulators [m][lane] (also the staging area of the batch sums)
    __shared__ double ex_lds[kExtra * BLOCK];
    const int tid = threadIdx.x;
    for (int idx = tid; idx < kNConst; idx += BLOCK) kc_lds[idx] = reinterpret_cast<const double *>(prm.kc)[idx];
    for (int idx = tid; idx < prm.n_save; idx += BLOCK) ts_lds[idx] = prm.tsave[idx];
    __syncthreads();
    const KConst *kc = reinterpret_cast<const KConst *>(kc_lds);
    // Where theta lives: wave-uniform scalar loads RE-ISSUED per phase (a handful of s_load_dwordx16 per evaluation through a pointer with an opaque
    // SGPR zero offset, issued ahead of the logarithms that precede theta's first use), theta as SGPR operands of the FMAs.  Round 3 measured this
    // fastest and leanest of four placements (hoisted scalar loads: ~88 SGPRs for the whole kernel; LDS hoisted into VGPRs / AGPRs; LDS re-read per
    // phase); the switch that selected among them was deleted in round 5.
    const double *th = theta;
#define CRNN_TH_FRESH() (theta + opaque_zero_s())
    double *const thb_s = thb_lds + tid;
#define THB_ADD(m, val) unsafeAtomicAdd(&thb_s[(m) * BLOCK], (val))

    const double d_ = 0.29289321881345248;    // 1/(2+sqrt 2)
    const double c32 = 7.4142135623730950;    // 6+sqrt 2
    const double inv12d = 2.4142135623730950; // 1/(1-2d)
    const int nsave = prm.n_save;
    const double tend = ts_lds[nsave - 1];
    const double ts0 = ts_lds[0];
    const double t0 = kc->t0;
    const double dtmax = tend - t0;
    const double lqinit = flog(kc->qoldinit);
    const bool start_saved = (ts0 == t0);
    // PI exponents of the two algorithms (Tsit5-only: whatever the configuration says, normally the same numbers)
    const double b1_ts = COMPOSITE ? 7.0 / 50.0 : kc->beta1, b2_ts = COMPOSITE ? 2.0 / 25.0 : kc->beta2;
    const double b1_rb = 7.0 / 20.0, b2_rb = 2.0 / 10.0;

    const int lane = tid & 63;
    double *const tape = adj.tape + (size_t)((size_t)blockIdx.x * BLOCK + tid) * adj.tape_cap * RECW;

    while (true) {
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(prm.queue, 64ULL);
        const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)base);
        const unsigned bhi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
        const int64_t wave_base = (int64_t)(((unsigned long long)bhi << 32) | blo);
        if (wave_base >= prm.count) break;
        const int64_t traj = wave_base + lane;
        const bool valid = traj < prm.count;
        // queue order by last known step counts for ensembles larger than the resident lanes (sort_steps_kernel)
        const int64_t b = prm.first + (valid ? (adj.perm ? (int64_t)adj.perm[traj] : traj) : 0);
        CRNN_CHK(b >= 0 && b < prm.B && traj >= 0, 1);

        double bT[NR];
        double xT = 0.0, Tconst = 0.0;
        auto eval_point = [&](const double (&uu)[NS], double (&x)[NS], double (&g)[NS], double (&r)[NR], double (&f)[NS]) {
            const double *tq = CRNN_TH_FRESH();
            features<NS>(uu, kc->lb, kc->ub, x, g);
            rates<NS, NR, HAS_T>(tq, x, bT, r);
            rhs_from_rates<NS, NR, HAS_T, USE_SCALE>(tq, r, kc->scale, f);
        };
        auto write_pred = [&](int j, const double (&v)[NS]) {
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                double w = v[i];
                if (prm.clamp_pred) w = clampv(w, -kc->ub, kc->ub);
                CRNN_CHK((int64_t)j * prm.n_obs < prm.row_stride && j >= 0, 9);
                prm.pred[((size_t)j * N + i) * prm.B + b] = w;
            }
            if (HAS_T) {
                double w = Tconst;
                if (prm.clamp_pred) w = clampv(w, -kc->ub, kc->ub);
                prm.pred[((size_t)j * N + NS) * prm.B + b] = w;
            }
        };

        // a save point of the forward sweep: prediction out and, PRIMAL, its loss term (ascending in the save index)
        double pf_loss = 0.0;
        auto save_point = [&](int j, const double (&v)[NS]) {
            if (prm.pred) write_pred(j, v);
            if (PRIMAL) {
                const double *row = prm.data + (size_t)b * prm.row_stride + (size_t)j * prm.n_obs;
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const int dr = (int)kc->drow[i];
                    if (dr >= 0) {
                        double w = v[i];
                        if (prm.clamp_pred) w = clampv(w, -kc->ub, kc->ub);
                        const double rr = (row[dr] - w) * kc->inv_yscale[i];
                        pf_loss = (prm.loss_kind == 0) ? pf_loss + fabs(rr) : fma(rr, rr, pf_loss);
                    }
                }
            }
        };

        // ================================================================== forward sweep
        double u[NS], f0[NS], g0[NS], r0[NR];
        double t = t0, dt = 0.0, lqold = lqinit;
        int iter = 0, jsave = 0, nacc = 0, nrej = 0;
        int rc = valid ? -1 : 0;
        int alg = 0, cnt = 0;          // 0 Tsit5, 1 Rosenbrock23; signed run length of the stiffness test
        double eig = 0.0;
        bool have_eig = false;
#pragma unroll
        for (int i = 0; i < NS; ++i) u[i] = prm.u0[(size_t)i * prm.B + b];
        if (HAS_T) {
            Tconst = prm.u0[(size_t)NS * prm.B + b];
            xT = kc->inv_R * frcp(Tconst);
        }
#pragma unroll
        for (int j = 0; j < NR; ++j) bT[j] = HAS_T ? fma(th[L_::wi(NS, j)], xT, th[L_::wb(j)]) : th[L_::wb(j)];
        {
            double x0[NS];
            eval_point(u, x0, g0, r0, f0);
            // Hairer initial step (OrdinaryDiffEq ode_determine_initdt) with the order of the starting algorithm, 5
            double d0 = 0.0, d1 = 0.0, sk[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                sk[i] = frcp(fma(fabs(u[i]), kc->rtol[i], kc->atol[i]));
                double a = u[i] * sk[i], c = f0[i] * sk[i];
                d0 = fma(a, a, d0);
                d1 = fma(c, c, d1);
            }
            if (HAS_T) { double a = Tconst * frcp(fma(fabs(Tconst), kc->rtol[NS], kc->atol[NS])); d0 = fma(a, a, d0); }
            d0 = sqrt(d0 * (1.0 / N));
            d1 = sqrt(d1 * (1.0 / N));
            double dt0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * (d0 / d1);
            dt0 = fmin(dt0, dtmax);
            double u1[NS], x1[NS], g1[NS], r1[NR], f1[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) u1[i] = fma(dt0, f0[i], u[i]);
            eval_point(u1, x1, g1, r1, f1);
            double d2 = 0.0;
#pragma unroll
            for (int i = 0; i < NS; ++i) { double e = (f1[i] - f0[i]) * sk[i]; d2 = fma(e, e, d2); }
            d2 = sqrt(d2 * (1.0 / N)) / dt0;
            double dm = fmax(d1, d2);
            double dt1 = (dm <= 1e-15) ? fmax(1e-6, dt0 * 1e-3) : exp(-0.2 * (4.605170185988091368 + flog(dm)));
            dt = fmax(kc->dtmin, fmin(fmin(100.0 * dt0, dt1), dtmax));
        }
        if (start_saved) {
            if (valid && prm.pred) write_pred(0, u);
            jsave = 1;
        }

        while (__builtin_amdgcn_ballot_w64(rc < 0) != 0) {
            if (rc < 0) {
                ++iter;
                bool last = false;
                if (jsave >= nsave) rc = 0;
                else if (iter > prm.maxiters) rc = 1;
                if (COMPOSITE && rc < 0 && have_eig) {   // choose_algorithm! at the loop header
                    const double stiffness = fabs(eig * dt * (1.0 / AutoSw::stability_size));
                    const bool stiff = stiffness > AutoSw::tol;
                    cnt = stiff ? (cnt < 0 ? 1 : cnt + 1) : (cnt > 0 ? -1 : cnt - 1);
                    if (alg == 0 && cnt > AutoSw::maxstiffstep) { dt *= AutoSw::dtfac; alg = 1; }
                    else if (alg == 1 && cnt < -AutoSw::maxnonstiffstep) { dt *= 1.0 / AutoSw::dtfac; alg = 0; }
                }
                if (t + dt * (1.0 + 1e-13) >= tend) { dt = tend - t; last = true; }
                if (rc < 0 && (!(dt > kc->dtmin) || t + dt == t)) rc = 2;
                if (rc < 0) {
                    double unew[NS], f2[NS], g2[NS], r2[NR];
                    double es = 0.0;
                    bool finite = true, accepted = false;
                    double q = 1.0, lEE = 0.0, lq11 = 0.0;
                    bool ee_zero = false;
                    const double tnew = last ? tend : t + dt;
                    // error test + PI controller (OrdinaryDiffEq PIController, in log space); true = accept
                    auto controller = [&](double b1, double b2) -> bool {
                        ee_zero = (es == 0.0);
                        lEE = 0.5 * flog(ee_zero ? 1.0 : es);
                        lq11 = b1 * lEE;
                        q = ee_zero ? 1.0 / kc->qmax
                                    : fmax(1.0 / kc->qmax, fmin(1.0 / kc->qmin, exp(lq11 - b2 * lqold) / kc->gamma));
                        return es <= 1.0;
                    };
                    // room on the tape?  then record the step (Tsit5 steps with -dt)
                    auto record = [&](double dt_signed) -> bool {
                        if (PRIMAL) { ++nacc; return true; }
                        if (nacc >= adj.tape_cap) {
                            rc = 5;
                            atomicAdd(adj.overflow, 1u);
                            return false;
                        }
                        CRNN_CHK(nacc >= 0 && nacc < adj.tape_cap, 5);
                        double *rec = tape + (size_t)nacc * RECW;
                        rec[0] = t;
                        rec[1] = dt_signed;
#pragma unroll
                        for (int i = 0; i < NS; ++i) rec[2 + i] = u[i];
                        ++nacc;
                        return true;
                    };
                    if (!COMPOSITE || alg == 0) {
                        // ---------------------------------------------------------------- Tsit5 attempt
                        double k[7][NS], g6[NS];
#pragma unroll
                        for (int i = 0; i < NS; ++i) k[0][i] = f0[i];
#pragma unroll
                        for (int s = 1; s < 7; ++s) {
                            CRNN_AUTO_FENCE();
                            double g[NS];
#pragma unroll
                            for (int i = 0; i < NS; ++i) {
                                double a = 0.0;
#pragma unroll
                                for (int j = 0; j < s; ++j) a = fma(Ts5::a(s - 1, j), k[j][i], a);
                                g[i] = fma(dt, a, u[i]);
                            }
                            if (s == 5) {
#pragma unroll
                                for (int i = 0; i < NS; ++i) g6[i] = g[i];
                            }
                            if (s == 6) {
#pragma unroll
                                for (int i = 0; i < NS; ++i) unew[i] = g[i];
                                double x2[NS];
                                eval_point(g, x2, g2, r2, k[6]);
                            } else {
                                double x[NS], gg[NS], r[NR];
                                eval_point(g, x, gg, r, k[s]);
                            }
                        }
#pragma unroll
                        for (int i = 0; i < NS; ++i) f2[i] = k[6][i];
#pragma unroll
                        for (int i = 0; i < NS; ++i) {
                            double a = 0.0;
#pragma unroll
                            for (int j = 0; j < 7; ++j) a = fma(Ts5::bt(j), k[j][i], a);
                            const double ev = dt * a;
                            const double m = fmax(fabs(u[i]), fabs(unew[i]));
                            const double e = ev * frcp1(fma(kc->rtol[i], m, kc->atol[i]));
                            es = fma(e, e, es);
                            finite = finite && isfinite(unew[i]) && isfinite(ev);
                        }
                        es = es * (1.0 / N);
                        if (COMPOSITE) {   // Hairer II p.22 in the Inf norm; NaN (0/0) propagates as in Julia's maximum
                            double est = 0.0;
                            bool isnan_ = false;
#pragma unroll
                            for (int i = 0; i < NS; ++i) {
                                const double qq = fabs((k[6][i] - k[5][i]) / (unew[i] - g6[i]));
                                isnan_ = isnan_ || (qq != qq);
                                est = fmax(est, qq);    // fmax drops a NaN operand; isnan_ keeps track of it
                            }
                            eig = isnan_ ? __longlong_as_double(0x7ff8000000000000LL) : est;
                            have_eig = true;
                        }
                        if (!finite) rc = 3;
                        else if (controller(b1_ts, b2_ts)) {
                            if (record(-dt)) {
                                accepted = true;
                                while (jsave < nsave) {
                                    CRNN_CHK(jsave >= 0 && jsave < nsave, 6);
                                    const double ts = ts_lds[jsave];
                                    if (!(ts <= tnew)) break;
                                    if (prm.pred || PRIMAL) {
                                        const bool at_end = (ts == tnew);
                                        double bth[7], v[NS];
                                        Ts5::dense(at_end ? 1.0 : (ts - t) / dt, bth);
#pragma unroll
                                        for (int i = 0; i < NS; ++i) {
                                            double a = 0.0;
#pragma unroll
                                            for (int j = 0; j < 7; ++j) a = fma(bth[j], k[j][i], a);
                                            v[i] = at_end ? unew[i] : fma(dt, a, u[i]);
                                        }
                                        save_point(jsave, v);
                                    }
                                    ++jsave;
                                }
                            }
                        } else {
                            ++nrej;
                            dt = dt / fmin(1.0 / kc->qmin, exp(lq11) / kc->gamma);
                        }
                    } else if (COMPOSITE) {
                        // ---------------------------------------------------------------- Rosenbrock23 attempt
                        Solver W;
                        const double *tr = CRNN_TH_FRESH();
                        const double gam = d_ * dt;
                        double gr0[NR];
#pragma unroll
                        for (int j = 0; j < NR; ++j) gr0[j] = gam * r0[j];
                        {   // eigen_est = opnorm(J(u_n), Inf)
                            double est = 0.0;
#pragma unroll
                            for (int i = 0; i < NS; ++i) {
                                double row = 0.0;
#pragma unroll
                                for (int c = 0; c < NS; ++c) {
                                    double a = 0.0;
#pragma unroll
                                    for (int j = 0; j < NR; ++j) a = fma(tr[L_::wo(i, j)] * r0[j], tr[L_::wi(c, j)], a);
                                    row += fabs(a * g0[c]);
                                }
                                est = fmax(est, USE_SCALE ? row * fabs(kc->scale[i]) : row);
                            }
                            eig = est;
                            have_eig = true;
                        }
                        double k1[NS], dk[NS], f1[NS];
                        const bool okf = W.factor(tr, g0, r0, gam, kc->scale);
#pragma unroll
                        for (int i = 0; i < NS; ++i) k1[i] = f0[i];
                        W.solve(tr, g0, gr0, kc->scale, k1);
                        {
                            double u1[NS], x1[NS], g1[NS], r1[NR];
#pragma unroll
                            for (int i = 0; i < NS; ++i) u1[i] = fma(0.5 * dt, k1[i], u[i]);
                            eval_point(u1, x1, g1, r1, f1);
                        }
#pragma unroll
                        for (int i = 0; i < NS; ++i) dk[i] = f1[i] - k1[i];
                        W.solve(tr, g0, gr0, kc->scale, dk);
#pragma unroll
                        for (int i = 0; i < NS; ++i) unew[i] = fma(dt, k1[i] + dk[i], u[i]);
                        {
                            double x2[NS];
                            eval_point(unew, x2, g2, r2, f2);
                        }
                        double k3[NS];
#pragma unroll
                        for (int i = 0; i < NS; ++i) {
                            const double k2i = k1[i] + dk[i];
                            k3[i] = f2[i] - c32 * (k2i - f1[i]) - 2.0 * (k1[i] - f0[i]);
                        }
                        W.solve(tr, g0, gr0, kc->scale, k3);
                        finite = okf;
#pragma unroll
                        for (int i = 0; i < NS; ++i) {
                            const double k2i = k1[i] + dk[i];
                            const double ev = dt * (1.0 / 6.0) * (k1[i] - 2.0 * k2i + k3[i]);
                            const double m = fmax(fabs(u[i]), fabs(unew[i]));
                            const double e = ev * frcp1(fma(kc->rtol[i], m, kc->atol[i]));
                            es = fma(e, e, es);
                            finite = finite && isfinite(unew[i]) && isfinite(ev);
                        }
                        es = es * (1.0 / N);
                        if (!finite) rc = 3;
                        else if (controller(b1_rb, b2_rb)) {
                            if (record(dt)) {
                                accepted = true;
                                while (jsave < nsave) {
                                    CRNN_CHK(jsave >= 0 && jsave < nsave, 6);
                                    const double ts = ts_lds[jsave];
                                    if (!(ts <= tnew)) break;
                                    if (prm.pred || PRIMAL) {
                                        const bool at_end = (ts == tnew);
                                        const double Th = at_end ? 1.0 : (ts - t) / dt;
                                        const double c1 = at_end ? 0.0 : Th * (1.0 - Th) * inv12d;
                                        const double c2 = at_end ? 1.0 : Th * (Th - 2.0 * d_) * inv12d;
                                        double v[NS];
#pragma unroll
                                        for (int i = 0; i < NS; ++i) {
                                            const double k2i = k1[i] + dk[i];
                                            v[i] = at_end ? unew[i] : fma(dt, fma(c1, k1[i], c2 * k2i), u[i]);
                                        }
                                        save_point(jsave, v);
                                    }
                                    ++jsave;
                                }
                            }
                        } else {
                            ++nrej;
                            dt = dt / fmin(1.0 / kc->qmin, exp(lq11) / kc->gamma);
                        }
                    }
                    if (accepted) {
#pragma unroll
                        for (int i = 0; i < NS; ++i) { u[i] = unew[i]; f0[i] = f2[i]; g0[i] = g2[i]; }
#pragma unroll
                        for (int j = 0; j < NR; ++j) r0[j] = r2[j];
                        t = tnew;
                        // step_accept_controller
                        if (q >= kc->qsteady_min && q <= kc->qsteady_max) q = 1.0;
                        lqold = ee_zero ? lqinit : fmax(lEE, lqinit);
                        dt = fmin(dt / q, dtmax);
                        if (jsave >= nsave) rc = 0;
                    }
                }
            }
        }

        // ================================================================== reverse sweep
        const int n_saved = jsave;
        const int jlo = start_saved ? 1 : 0;
#pragma unroll
        for (int m = 0; m < NTH; ++m) thb_s[m * BLOCK] = 0.0;
        double lam[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) lam[i] = 0.0;
        double wbb[NR];              // d/d w_b of THIS trajectory, in registers; its temperature row is xT times the same sum
#pragma unroll
        for (int j = 0; j < NR; ++j) wbb[j] = 0.0;
        double loss_sum = PRIMAL ? pf_loss : 0.0;
        double tnew = t;             // end time of the step being reversed
        int s = (valid && !PRIMAL) ? nacc - 1 : -1;

        const double *const drows = prm.data + (size_t)b * prm.row_stride;
        int doff[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) { const int dr = (int)kc->drow[i]; doff[i] = dr >= 0 ? dr : 0; }
        auto load_row = [&](int j, double (&d)[NS]) {
            CRNN_CHK((int64_t)(j > 0 ? j : 0) * prm.n_obs < prm.row_stride, 2);
            const double *row = drows + (size_t)(j > 0 ? j : 0) * prm.n_obs;
#pragma unroll
            for (int i = 0; i < NS; ++i) d[i] = row[doff[i]];
        };
        // residual of one save point: adds its loss term, returns the seed weight w_i = d loss_term / d v_i
        // (round 5, as the Rosenbrock23 adjoint kernels: no clamp is an infinite clamp -- v_max / v_min, v is finite on accepted steps --, the
        // mask is "the clamp changed nothing"; the same values as the compare-and-select form it replaces)
        const double ubc = prm.clamp_pred ? kc->ub : __builtin_inf();
        auto residual = [&](int i, const double v, double dobs) -> double {
            const double vc = fmin(fmax(v, -ubc), ubc);
            const double iy = kc->inv_yscale[i];
            const double rr = (dobs - vc) * iy;
            double w;
            if (prm.loss_kind == 0) { loss_sum += fabs(rr); w = signbit(rr) ? iy : -iy; }
            else { loss_sum = fma(rr, rr, loss_sum); w = (-2.0 * rr) * iy; }
            return (vc == v) ? w : 0.0;
        };
        // reverse accumulation through one right-hand-side evaluation k = f(point) with features (x, g, r):
        //   gb = f_u^T kb,   thb += f_theta^T kb
        // The theta terms are NOT added where they arise: an LDS atomic costs a wavefront ~40 cycles here, so vjp_core only
        // returns the two factors (rho_j = (kb.w_out[:,j]) r_j, vs = sc .* kb) and the caller adds the terms of TWO points
        // with one ds_add_f64 per accumulator (vjp_flush2); the w_b terms go to the trajectory's register sums.
        auto vjp_core = [&](const double (&g)[NS], const double (&r)[NR], const double (&kb)[NS], double (&gb)[NS],
                            double (&rho)[NR], double (&vs)[NS]) {
            double um[NS];
            const double *tq = CRNN_TH_FRESH();
#pragma unroll
            for (int i = 0; i < NS; ++i) { vs[i] = USE_SCALE ? kb[i] * kc->scale[i] : kb[i]; um[i] = 0.0; }
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                double a = 0.0;
#pragma unroll
                for (int i = 0; i < NS; ++i) a = fma(vs[i], tq[L_::wo(i, j)], a);
                rho[j] = a * r[j];
                wbb[j] += rho[j];
#pragma unroll
                for (int c = 0; c < NS; ++c) um[c] = fma(rho[j], tq[L_::wi(c, j)], um[c]);
            }
#pragma unroll
            for (int c = 0; c < NS; ++c) gb[c] = um[c] * g[c];
        };
        auto vjp_flush1 = [&](const double (&x)[NS], const double (&r)[NR], const double (&rho)[NR], const double (&vs)[NS]) {
#pragma unroll
            for (int j = 0; j < NR; ++j) {
#pragma unroll
                for (int c = 0; c < NS; ++c) THB_ADD(L_::wi(c, j), rho[j] * x[c]);
#pragma unroll
                for (int i = 0; i < NS; ++i) THB_ADD(L_::wo(i, j), vs[i] * r[j]);
            }
        };
        auto vjp_flush2 = [&](const double (&xa)[NS], const double (&ra)[NR], const double (&rhoa)[NR], const double (&vsa)[NS],
                              const double (&xb)[NS], const double (&rb)[NR], const double (&rhob)[NR], const double (&vsb)[NS]) {
#pragma unroll
            for (int j = 0; j < NR; ++j) {
#pragma unroll
                for (int c = 0; c < NS; ++c) THB_ADD(L_::wi(c, j), fma(rhoa[j], xa[c], rhob[j] * xb[c]));
#pragma unroll
                for (int i = 0; i < NS; ++i) THB_ADD(L_::wo(i, j), fma(vsa[i], ra[j], vsb[i] * rb[j]));
            }
        };

        // the times of the next two save points (backwards) sit in registers: the test "is it inside this step" and the seed
        // itself do not wait for LDS (ros23_adj_kernel.hpp)
        double ts_cur = (jsave - 1 >= jlo) ? ts_lds[jsave - 1] : -INFINITY;   // none left: -inf, never inside a step
        double ts_nxt = (jsave - 2 >= jlo) ? ts_lds[jsave - 2] : -INFINITY;
        double rt = 0.0, rdt = 0.0, ru[NS];   // tape record s, prefetched
        if constexpr (!PRIMAL) {   // a context that has only made primal calls has no tape at all: nothing may touch it
            CRNN_CHK(s < adj.tape_cap, 3);
            const double *rec = tape + (size_t)(s > 0 ? s : 0) * RECW;
            rt = rec[0]; rdt = rec[1];
#pragma unroll
            for (int i = 0; i < NS; ++i) ru[i] = rec[2 + i];
        } else {
#pragma unroll
            for (int i = 0; i < NS; ++i) ru[i] = 0.0;
        }

        while (__builtin_amdgcn_ballot_w64(s >= 0) != 0) {
            if (s >= 0) {
                const double tn = rt;
                const bool is_ts = !COMPOSITE || rdt < 0.0;
                const double h = fabs(rdt);
                double un[NS];
#pragma unroll
                for (int i = 0; i < NS; ++i) un[i] = ru[i];
                double dA[NS], dB[NS], dC[NS];
                load_row(jsave - 1, dA);
                load_row(jsave - 2, dB);
                load_row(jsave - 3, dC);
                {   // prefetch the next record (s-1)
                    CRNN_CHK(s - 1 < adj.tape_cap, 7);
                    const double *rec = tape + (size_t)(s > 0 ? s - 1 : 0) * RECW;
                    rt = rec[0]; rdt = rec[1];
#pragma unroll
                    for (int i = 0; i < NS; ++i) ru[i] = rec[2 + i];
                }
                auto in_step = [&]() -> bool { return ts_cur > tn; };
                const double inv_h = frcp(h);      // one reciprocal per step instead of a division per save point
                auto next_ts = [&]() -> double {   // consumes save point jsave-1 (the caller decrements jsave afterwards)
                    const double ts = ts_cur;
                    CRNN_CHK(jsave - 1 >= jlo && jsave - 1 < nsave, 8);
                    ts_cur = ts_nxt;
                    ts_nxt = (jsave - 3 >= jlo) ? ts_lds[jsave - 3] : -INFINITY;
                    return ts;
                };

                if (is_ts) {
                    // ------------------------------------------------------------ Tsit5 step: re-form the stages
                    // (kKeepStages = false: only the stage slopes are kept and the features of a stage are formed again where its
                    //  adjoint needs them -- seven more evaluations, independent of each other, against ~300 spilled registers
                    //  for case2; 1.69 -> 0.94 ms.  The robertson shape holds all 84 values in registers: 1.42 vs 1.66 ms.)
                    constexpr int KS = kKeepStages ? 7 : 1;
                    double k[7][NS], unew[NS], xs[KS][NS], gs[KS][NS], rs[KS][NR];
                    if constexpr (kKeepStages) eval_point(un, xs[0], gs[0], rs[0], k[0]);
                    else {
                        double x[NS], g[NS], r[NR];
                        eval_point(un, x, g, r, k[0]);
                    }
#pragma unroll
                    for (int st = 1; st < 7; ++st) {
                        CRNN_AUTO_FENCE();
                        double gp[NS];
#pragma unroll
                        for (int i = 0; i < NS; ++i) {
                            double a = 0.0;
#pragma unroll
                            for (int j = 0; j < st; ++j) a = fma(Ts5::a(st - 1, j), k[j][i], a);
                            gp[i] = fma(h, a, un[i]);
                        }
                        if (st == 6) {
#pragma unroll
                            for (int i = 0; i < NS; ++i) unew[i] = gp[i];
                        }
                        if constexpr (kKeepStages) eval_point(gp, xs[st], gs[st], rs[st], k[st]);
                        else {
                            double x[NS], g[NS], r[NR];
                            eval_point(gp, x, g, r, k[st]);
                        }
                    }
                    // loss and its seeds: v = u_n + h sum_j b_j(Th) k_j  ->  A += w, kb_j += w h b_j(Th)
                    double ub[NS], kb[7][NS];
#pragma unroll
                    for (int i = 0; i < NS; ++i) {
                        ub[i] = lam[i];
#pragma unroll
                        for (int j = 0; j < 6; ++j) kb[j][i] = h * Ts5::a(5, j) * lam[i];
                        kb[6][i] = 0.0;
                    }
                    auto seed_point = [&](const double (&dobs)[NS]) {
                        const double ts = next_ts();
                        const bool at_end = (ts == tnew);
                        double bth[7];
                        Ts5::dense(at_end ? 1.0 : (ts - tn) * inv_h, bth);
#pragma unroll
                        for (int i = 0; i < NS; ++i) {
                            const int dr = (int)kc->drow[i];
                            if (dr >= 0) {
                                double a = 0.0;
#pragma unroll
                                for (int j = 0; j < 7; ++j) a = fma(bth[j], k[j][i], a);
                                const double v = at_end ? unew[i] : fma(h, a, un[i]);
                                const double w = residual(i, v, dobs[i]);
                                ub[i] += w;
                                // at the end point v = u_{n+1} = u_n + h sum_j a_7j k_j  (b_j(1) = a_7j, b_7(1) = 0)
#pragma unroll
                                for (int j = 0; j < 7; ++j)
                                    kb[j][i] = fma(w * h, at_end ? (j < 6 ? Ts5::a(5, j) : 0.0) : bth[j], kb[j][i]);
                            }
                        }
                        --jsave;
                    };
                    if (in_step()) {
                        seed_point(dA);
                        if (in_step()) {
                            seed_point(dB);
                            if (in_step()) {
                                seed_point(dC);
                                while (in_step()) {
                                    double dD[NS];
                                    load_row(jsave - 1, dD);
                                    seed_point(dD);
                                }
                            }
                        }
                    }
                    // adjoint of the stages, last to first.  Without kKeepStages the stage features are formed again, with h
                    // behind an empty asm: otherwise the compiler recognises the stage points of the re-formation above and
                    // keeps all their features alive after all.
                    double h2 = h;
                    if (!kKeepStages) asm volatile("" : "+v"(h2));
                    double rho_p[NR], vs_p[NS], x_p[NS], r_p[NR];   // the stage whose theta terms are still to be added
#pragma unroll
                    for (int st = 6; st >= 0; --st) {
                        CRNN_AUTO_FENCE();
                        double gb[NS], rho[NR], vs[NS];
                        double xl[NS], gl[NS], rl[NR];
                        if constexpr (!kKeepStages) {
                            double gp[NS], fdump[NS];
#pragma unroll
                            for (int i = 0; i < NS; ++i) {
                                double a = 0.0;
#pragma unroll
                                for (int j = 0; j < st; ++j) a = fma(Ts5::a(st > 0 ? st - 1 : 0, j), k[j][i], a);
                                gp[i] = st > 0 ? fma(h2, a, un[i]) : un[i];
                            }
                            eval_point(gp, xl, gl, rl, fdump);
                        }
                        const double (&x)[NS] = kKeepStages ? xs[kKeepStages ? st : 0] : xl;
                        const double (&g)[NS] = kKeepStages ? gs[kKeepStages ? st : 0] : gl;
                        const double (&r)[NR] = kKeepStages ? rs[kKeepStages ? st : 0] : rl;
                        vjp_core(g, r, kb[st], gb, rho, vs);
                        if (st == 0) vjp_flush1(x, r, rho, vs);                       // seven stages: the first one is left over
                        else if ((st & 1) == 0) {                                     // st = 6, 4, 2: wait for the next stage
#pragma unroll
                            for (int j = 0; j < NR; ++j) { rho_p[j] = rho[j]; r_p[j] = r[j]; }
#pragma unroll
                            for (int i = 0; i < NS; ++i) { vs_p[i] = vs[i]; x_p[i] = x[i]; }
                        } else vjp_flush2(x_p, r_p, rho_p, vs_p, x, r, rho, vs);      // st = 5, 3, 1
#pragma unroll
                        for (int i = 0; i < NS; ++i) {
                            ub[i] += gb[i];
#pragma unroll
                            for (int j = 0; j < st; ++j) kb[j][i] = fma(h * Ts5::a(st > 0 ? st - 1 : 0, j), gb[i], kb[j][i]);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < NS; ++i) lam[i] = ub[i];
                } else if (COMPOSITE) {
                    // ------------------------------------------------------------ Rosenbrock23 step (ros23_adj_kernel.hpp)
                    double x0[NS], gg0[NS], rr0[NR], ff0[NS];
                    const double *tr = CRNN_TH_FRESH();
                    Solver W;
                    const double gam = d_ * h;
                    double gr0[NR], x1[NS], g1[NS], r1[NR];
                    eval_point(un, x0, gg0, rr0, ff0);
#pragma unroll
                    for (int j = 0; j < NR; ++j) gr0[j] = gam * rr0[j];
                    (void)W.factor(tr, gg0, rr0, gam, kc->scale);
                    double k1[NS], dk[NS];
#pragma unroll
                    for (int i = 0; i < NS; ++i) k1[i] = ff0[i];
                    W.solve(tr, gg0, gr0, kc->scale, k1);
                    {
                        double u1[NS], f1[NS];
#pragma unroll
                        for (int i = 0; i < NS; ++i) u1[i] = fma(0.5 * h, k1[i], un[i]);
                        eval_point(u1, x1, g1, r1, f1);
#pragma unroll
                        for (int i = 0; i < NS; ++i) dk[i] = f1[i] - k1[i];
                    }
                    W.solve(tr, gg0, gr0, kc->scale, dk);

                    double A_[NS], B1[NS], B2[NS];
#pragma unroll
                    for (int i = 0; i < NS; ++i) { A_[i] = 0.0; B1[i] = 0.0; B2[i] = 0.0; }
                    auto seed_point = [&](const double (&dobs)[NS]) {
                        const double ts = next_ts();
                        const bool at_end = (ts == tnew);
                        const double Th = at_end ? 1.0 : (ts - tn) * inv_h;
                        const double c1 = at_end ? 0.0 : Th * (1.0 - Th) * inv12d;
                        const double c2 = at_end ? 1.0 : Th * (Th - 2.0 * d_) * inv12d;
#pragma unroll
                        for (int i = 0; i < NS; ++i) {
                            const int dr = (int)kc->drow[i];
                            if (dr >= 0) {
                                const double k2i = k1[i] + dk[i];
                                const double v = at_end ? fma(h, k2i, un[i]) : fma(h, fma(c1, k1[i], c2 * k2i), un[i]);
                                const double w = residual(i, v, dobs[i]);
                                A_[i] += w;
                                B1[i] = fma(w, h * c1, B1[i]);
                                B2[i] = fma(w, h * c2, B2[i]);
                            }
                        }
                        --jsave;
                    };
                    if (in_step()) {
                        seed_point(dA);
                        if (in_step()) {
                            seed_point(dB);
                            if (in_step()) {
                                seed_point(dC);
                                while (in_step()) {
                                    double dD[NS];
                                    load_row(jsave - 1, dD);
                                    seed_point(dD);
                                }
                            }
                        }
                    }

                    double kb1[NS], v[NS], ub[NS];
#pragma unroll
                    for (int i = 0; i < NS; ++i) { v[i] = fma(h, lam[i], B2[i]); ub[i] = lam[i] + A_[i]; }
#pragma unroll
                    for (int i = 0; i < NS; ++i) kb1[i] = B1[i] + v[i];
                    solve_T<NS, NR, HAS_T, USE_SCALE>(W, tr, gg0, gr0, kc->scale, v);        // v = W^-T kb2
#pragma unroll
                    for (int i = 0; i < NS; ++i) kb1[i] -= v[i];
                    double vs[NS];
#pragma unroll
                    for (int i = 0; i < NS; ++i) vs[i] = USE_SCALE ? v[i] * kc->scale[i] : v[i];
                    double av[NR];
#pragma unroll
                    for (int j = 0; j < NR; ++j) {
                        double a = 0.0;
#pragma unroll
                        for (int i = 0; i < NS; ++i) a = fma(vs[i], tr[L_::wo(i, j)], a);
                        av[j] = a;
                    }
                    double rho1[NR], vs_dump[NS];   // the u_mid point's theta terms are folded into the u_n point's addends
                    {   // point u_mid
                        double gb[NS];
                        vjp_core(g1, r1, v, gb, rho1, vs_dump);
#pragma unroll
                        for (int c = 0; c < NS; ++c) {
                            ub[c] += gb[c];
                            kb1[c] = fma(0.5 * h, gb[c], kb1[c]);
                        }
                    }
                    solve_T<NS, NR, HAS_T, USE_SCALE>(W, tr, gg0, gr0, kc->scale, kb1);      // kb1 = w = W^-T kb1
                    {   // point u_n: d/d(u, theta) [ w.f + gam (v.J dk + w.J k1) ]
                        double ws[NS];
#pragma unroll
                        for (int i = 0; i < NS; ++i) ws[i] = USE_SCALE ? kb1[i] * kc->scale[i] : kb1[i];
                        double s1[NS], s2[NS];
#pragma unroll
                        for (int c = 0; c < NS; ++c) { s1[c] = 0.0; s2[c] = 0.0; }
#pragma unroll
                        for (int j = 0; j < NR; ++j) {
                            double aw = 0.0, q1 = 0.0, qd = 0.0;
#pragma unroll
                            for (int i = 0; i < NS; ++i) aw = fma(ws[i], tr[L_::wo(i, j)], aw);
#pragma unroll
                            for (int c = 0; c < NS; ++c) {
                                const double wg = tr[L_::wi(c, j)] * gg0[c];
                                q1 = fma(wg, k1[c], q1);
                                qd = fma(wg, dk[c], qd);
                            }
                            const double c1j = fma(gam, q1, 1.0), czd = gam * qd;
                            const double pv = av[j] * gr0[j];
                            const double pw = aw * rr0[j];
                            const double gpw = gam * pw;
                            const double beta = fma(pw, c1j, pv * qd);
                            wbb[j] += beta;
#pragma unroll
                            for (int c = 0; c < NS; ++c) {
                                const double m = fma(pv, dk[c], gpw * k1[c]);
                                THB_ADD(L_::wi(c, j), fma(rho1[j], x1[c], fma(beta, x0[c], gg0[c] * m)));
                                const double wi = tr[L_::wi(c, j)];
                                s1[c] = fma(beta, wi, s1[c]);
                                s2[c] = fma(wi, m, s2[c]);
                            }
                            const double ca = fma(rr0[j], czd, r1[j]), cb = rr0[j] * c1j;
#pragma unroll
                            for (int i = 0; i < NS; ++i) THB_ADD(L_::wo(i, j), fma(vs[i], ca, ws[i] * cb));
                        }
#pragma unroll
                        for (int c = 0; c < NS; ++c) {
                            const double g = gg0[c];
                            lam[c] = ub[c] + g * (s1[c] - g * s2[c]);
                        }
                    }
                }
                tnew = tn;
                --s;
            }
        }

        // ---- outputs
        if (valid) {
            if (start_saved && n_saved >= 1) {  // the saved initial point: a loss term without gradient
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const int dr = (int)kc->drow[i];
                    if (dr >= 0) {
                        double v = prm.u0[(size_t)i * prm.B + b];
                        if (prm.clamp_pred) v = clampv(v, -kc->ub, kc->ub);
                        const double rr = (drows[doff[i]] - v) * kc->inv_yscale[i];
                        loss_sum += (prm.loss_kind == 0) ? fabs(rr) : rr * rr;
                    }
                }
            }
            const double denom = (double)prm.n_obs * (double)n_saved;
            const double inv_den = n_saved > 0 ? 1.0 / denom : 0.0;
            prm.loss[b] = loss_sum * inv_den;
            prm.retcode[b] = rc;
            prm.n_saved[b] = n_saved;
            prm.n_accept[b] = nacc;
            prm.n_reject[b] = nrej;
        }
        // ---- sums over the 64 trajectories of this batch (as in ros23_adj_kernel.hpp: through LDS, in lane order)
        {
            const double denom_ = (double)prm.n_obs * (double)n_saved;
            const double scale_ = (valid && n_saved > 0) ? 1.0 / denom_ : 0.0;
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                THB_ADD(L_::wb(j), wbb[j]);
                if (HAS_T) THB_ADD(L_::wi(NS, j), wbb[j] * xT);
            }
#pragma unroll
            for (int m = 0; m < NTH; ++m) thb_s[m * BLOCK] = thb_s[m * BLOCK] * scale_;
            ex_lds[0 * BLOCK + tid] = valid ? loss_sum * scale_ : 0.0;
            ex_lds[1 * BLOCK + tid] = (valid && rc == 0) ? 1.0 : 0.0;
            ex_lds[2 * BLOCK + tid] = valid ? (double)nacc : 0.0;
            ex_lds[3 * BLOCK + tid] = valid ? (double)nrej : 0.0;
            ex_lds[4 * BLOCK + tid] = valid ? 1.0 : 0.0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            CRNN_CHK((wave_base >> 6) < ((prm.count + 63) >> 6), 4);
            double *prow = adj.batch_partials + (size_t)(wave_base >> 6) * (NTH + kExtra);
            const int w0 = tid & ~63;
            if (lane < NTH + kExtra) {
                const double *src = lane < NTH ? thb_lds + lane * BLOCK + w0 : ex_lds + (lane - NTH) * BLOCK + w0;
                double a = 0.0;
                for (int k = 0; k < 64; ++k) a += src[k];
                prow[lane] = a;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
#undef THB_ADD
#undef CRNN_TH_FRESH
}

}  // namespace crnn

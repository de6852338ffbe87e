// crnn_amd/csrc/tsit5_kernel.hpp -- explicit Tsit5 stepper (+ forward tangents) for the CRNN ensemble.
//
// The reference integrates case1 with `Tsit5()` (case1/case1.jl:28,94-95) and case2 / HyChem with
// `AutoTsit5(Rosenbrock23(...))` (case2/case2.jl:26), which stays on Tsit5 while the problem is non-stiff
// (SURVEY F6).  This kernel is the explicit member of the family; it shares the lane-group / LDS-record /
// persistent-queue design and every helper of ros23_kernel.hpp:
//   PRIMAL : stages k2..k7 (7 = FSAL) in registers; for every stage the features (x, g, r) go to the group's LDS
//            record; error estimate, PI controller (beta1 = 7/50, beta2 = 2/25), saveat through the method's free
//            4th-order interpolant u + dt sum_j b_j(Theta) k_j; loss seeds A, B_1..B_7.
//   TANGENT: per column k'_s = f_u(g_s)(s + dt sum_j a_sj k'_j) + f_theta(g_s) dtheta for s = 1..7, the column's
//            dtheta entries held in registers for the whole column, s_new = s + dt sum_j a_7j k'_j,
//            g += A.s + sum_j B_j.k'_j.
// Tableau and interpolant: Tsitouras 2011 as in OrdinaryDiffEq's Tsit5ConstantCache; the same numbers are checked
// against the Runge-Kutta order conditions by the test-suite.
#pragma once
#include "ros23_kernel.hpp"

namespace crnn {

struct Ts5 {
    static constexpr double c2 = 0.161, c3 = 0.327, c4 = 0.9, c5 = 0.9800255409045097;
    static constexpr double a21 = 0.161;
    static constexpr double a31 = -0.008480655492356989, a32 = 0.335480655492357;
    static constexpr double a41 = 2.8971530571054935, a42 = -6.359448489975075, a43 = 4.3622954328695815;
    static constexpr double a51 = 5.325864828439257, a52 = -11.748883564062828, a53 = 7.4955393428898365, a54 = -0.09249506636175525;
    static constexpr double a61 = 5.86145544294642, a62 = -12.92096931784711, a63 = 8.159367898576159, a64 = -0.071584973281401,
                            a65 = -0.028269050394068383;
    static constexpr double a71 = 0.09646076681806523, a72 = 0.01, a73 = 0.4798896504144996, a74 = 1.379008574103742,
                            a75 = -3.290069515436081, a76 = 2.324710524099774;
    static constexpr double bt1 = -0.00178001105222577714, bt2 = -0.0008164344596567469, bt3 = 0.007880878010261995,
                            bt4 = -0.1447110071732629, bt5 = 0.5823571654525552, bt6 = -0.45808210592918697,
                            bt7 = 0.015151515151515152;
    // a[s][j], s = 1..6 -> stages 2..7
    __device__ __forceinline__ static constexpr double a(int s, int j) {
        constexpr double A[6][6] = {{a21, 0, 0, 0, 0, 0},         {a31, a32, 0, 0, 0, 0},       {a41, a42, a43, 0, 0, 0},
                                    {a51, a52, a53, a54, 0, 0},   {a61, a62, a63, a64, a65, 0}, {a71, a72, a73, a74, a75, a76}};
        return A[s][j];
    }
    __device__ __forceinline__ static constexpr double bt(int j) {
        constexpr double B[7] = {bt1, bt2, bt3, bt4, bt5, bt6, bt7};
        return B[j];
    }
    __device__ __forceinline__ static void dense(double T, double (&b)[7]) {
        const double T2 = T * T;
        b[0] = -1.0530884977290216 * T * (T - 1.3299890189751412) * (T2 - 1.4364028541716351 * T + 0.7139816917074209);
        b[1] = 0.1017 * T2 * (T2 - 2.1966568338249754 * T + 1.2949852507374631);
        b[2] = 2.490627285651252793 * T2 * (T2 - 2.38535645472061657 * T + 1.57803468208092486);
        b[3] = -16.54810288924490272 * (T - 1.21712927295533244) * (T - 0.61620406037800089) * T2;
        b[4] = 47.37952196281928122 * (T - 1.203071208372362603) * (T - 0.658047292653547382) * T2;
        b[5] = -34.87065786149660974 * (T - 1.2) * (T - 0.666666666666666667) * T2;
        b[6] = 2.5 * (T - 1.0) * (T - 0.6) * T2;
    }
};

// record: 7 stage areas (x, g, r) = stages 1..7 of the current step, addressed modulo a rotating base: stage 7 of
// an accepted step becomes stage 1 of the next one without a copy (the new stages 2..7 overwrite the dead areas of
// the old stages 1..6); then the FSAL point (u, k7) and the loss seeds A, B_1..B_7.
template <int NS, int NR>
struct RecT {
    static constexpr int SA = 2 * NS + NR;          // one stage area: X, G, R
    static constexpr int XO = 0, GO = NS, RO = 2 * NS;
    static constexpr int NAREA = 7;
    static constexpr int UP = NAREA * SA;           // parked u_{n+1}
    static constexpr int FP = UP + NS;              // parked k7
    static constexpr int AA = FP + NS;
    static constexpr int BB = AA + NS;              // B_j at BB + j*NS
    static constexpr int NREC = BB + 7 * NS;
};

template <int NS, int NR, bool HAS_T, bool USE_SCALE, int C, int L, int BLOCK>
__global__ __launch_bounds__(BLOCK) void tsit5_kernel(const SolveParams prm, const double *__restrict__ theta,
                                                      const double *__restrict__ dtheta) {
    using L_ = Lay<NS, NR, HAS_T>;
    using R_ = RecT<NS, NR>;
    constexpr int N = L_::N;
    constexpr int NTH = L_::NTH;
    constexpr int NTHP = L_::NTHP;
    constexpr int CC = (C > 0) ? C : 1;
    constexpr int NREC = R_::NREC;
    constexpr int WAVES = BLOCK / 64;
    constexpr int GPW = 64 / L;
    constexpr int PPAD = L * CC;
    static_assert(C > 0 || L == 1, "primal-only variant uses one lane per trajectory");

    __shared__ double kc_lds[kNConst];
    __shared__ double ts_lds[kMaxSave];
    __shared__ double dth_lds[C > 0 ? PPAD * NTHP : 1];
    __shared__ double S_lds[C > 0 ? WAVES * C * NS * 64 : 1];
    __shared__ double rec_lds[C > 0 ? WAVES * NREC * GPW : 1];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int grp = lane / L;
    const int chunk = lane - grp * L;
    const bool lane_active = grp < GPW;
    const bool lead = lane_active && chunk == 0;

    double *S_s = S_lds + (C > 0 ? wave * C * NS * 64 + lane : 0);
    double *rec = rec_lds + (C > 0 ? wave * NREC * GPW + (lane_active ? grp : 0) : 0);

    for (int idx = tid; idx < kNConst; idx += BLOCK) kc_lds[idx] = reinterpret_cast<const double *>(prm.kc)[idx];
    for (int idx = tid; idx < prm.n_save; idx += BLOCK) ts_lds[idx] = prm.tsave[idx];
    if (C > 0) {
        for (int idx = tid; idx < PPAD * NTHP; idx += BLOCK) {
            int k = idx / NTHP, m = idx - k * NTHP;
            dth_lds[idx] = (k < prm.P && m < NTH) ? dtheta[(size_t)k * NTH + m] : 0.0;
        }
    }
    __syncthreads();
    const KConst *kc = reinterpret_cast<const KConst *>(kc_lds);

    const int64_t ngroups = (int64_t)gridDim.x * WAVES * GPW;
    int64_t traj = ((int64_t)blockIdx.x * WAVES + wave) * GPW + grp;
    if (!lane_active) traj = prm.count;
    auto fetch_next = [&]() -> int64_t {
        unsigned long long v = 0;
        if (lead) v = atomicAdd(prm.queue, 1ULL);
        const int src = grp * L;
        const unsigned lo = (unsigned)__shfl((int)(unsigned)v, src);
        const unsigned hi = (unsigned)__shfl((int)(unsigned)(v >> 32), src);
        return (int64_t)(((unsigned long long)hi << 32) | lo) + ngroups;
    };
    int64_t traj_next = lane_active ? fetch_next() : prm.count;

    const double *__restrict__ th = theta;
    const int nsave = prm.n_save;
    const double tend = prm.tsave[nsave - 1];
    const double ts0 = prm.tsave[0];
    const double t0 = kc->t0;
    const double dtmax = tend - t0;
    const double lqinit = flog(kc->qoldinit);

    double u[NS], k1[NS], bT[NR];
    double gtr[CC];
    double dA[NS], dB[NS];
    double xT = 0.0;
    double t = 0.0, dt = 0.0, lqold = 0.0, loss_sum = 0.0;
    int iter = 0, jsave = 0, base = 0, nacc = 0, nrej = 0;   // base: record area of stage 1 (rotates by 6 per accepted step)
    int64_t b = 0;
    bool need_init = true;

    auto load_row = [&](int j, double (&d)[NS]) {
        const int jj = j < nsave ? j : nsave - 1;
        const double *row = prm.data + (size_t)b * prm.row_stride + (size_t)jj * prm.n_obs;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int dr = (int)kc->drow[i];
            d[i] = row[dr >= 0 ? dr : 0];
        }
    };
    // record area of stage s (0-based 0..6) of the current step
    auto area = [&](int s) -> int { int a = base + s; return (a >= R_::NAREA ? a - R_::NAREA : a) * R_::SA; };
    auto publish_stage = [&](int s, const double (&x)[NS], const double (&g)[NS], const double (&r)[NR]) {
        if (C > 0 && lead) {
            const int o = area(s);
#pragma unroll
            for (int i = 0; i < NS; ++i) { rec[(o + R_::XO + i) * GPW] = x[i]; rec[(o + R_::GO + i) * GPW] = g[i]; }
#pragma unroll
            for (int j = 0; j < NR; ++j) rec[(o + R_::RO + j) * GPW] = r[j];
        }
    };

    while (true) {
        if (need_init) {
            if (traj >= prm.count) break;
            need_init = false;
            b = prm.first + traj;
#pragma unroll
            for (int i = 0; i < NS; ++i) u[i] = prm.u0[(size_t)i * prm.B + b];
            load_row(0, dA);
            load_row(1, dB);
            double Tconst = 0.0;
            if (HAS_T) {
                Tconst = prm.u0[(size_t)NS * prm.B + b];
                xT = kc->inv_R * frcp(Tconst);
            }
#pragma unroll
            for (int j = 0; j < NR; ++j) bT[j] = HAS_T ? fma(th[L_::wi(NS, j)], xT, th[L_::wb(j)]) : th[L_::wb(j)];
            base = 0;
            {
                double x0[NS], g0[NS], r0[NR];
                features<NS>(u, kc->lb, kc->ub, x0, g0);
                rates<NS, NR, HAS_T>(th, x0, bT, r0);
                rhs_from_rates<NS, NR, HAS_T, USE_SCALE>(th, r0, kc->scale, k1);
                publish_stage(0, x0, g0, r0);
            }
            // Hairer initial step (ode_determine_initdt), order 5
            {
                double d0 = 0.0, d1 = 0.0, sk[NS];
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    sk[i] = frcp(fma(fabs(u[i]), kc->rtol[i], kc->atol[i]));
                    double a = u[i] * sk[i], c = k1[i] * sk[i];
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
                for (int i = 0; i < NS; ++i) u1[i] = fma(dt0, k1[i], u[i]);
                features<NS>(u1, kc->lb, kc->ub, x1, g1);
                rates<NS, NR, HAS_T>(th, x1, bT, r1);
                rhs_from_rates<NS, NR, HAS_T, USE_SCALE>(th, r1, kc->scale, f1);
                double d2 = 0.0;
#pragma unroll
                for (int i = 0; i < NS; ++i) { double e = (f1[i] - k1[i]) * sk[i]; d2 = fma(e, e, d2); }
                d2 = sqrt(d2 * (1.0 / N)) / dt0;
                double dm = fmax(d1, d2);
                // 10^(-(2 + log10 dm)/5)
                double dt1 = (dm <= 1e-15) ? fmax(1e-6, dt0 * 1e-3) : exp(-0.2 * (4.605170185988091368 + flog(dm)));
                dt = fmax(kc->dtmin, fmin(fmin(100.0 * dt0, dt1), dtmax));
            }
            t = t0;
            lqold = lqinit;
            iter = 0; jsave = 0; nacc = 0; nrej = 0;
            loss_sum = 0.0;
#pragma unroll
            for (int q = 0; q < CC; ++q) gtr[q] = 0.0;
            if (C > 0) {
#pragma unroll
                for (int q = 0; q < C * NS; ++q) S_s[q * 64] = 0.0;
            }
            if (ts0 == t0) {  // save_start
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    double v = u[i];
                    if (prm.clamp_pred) v = clampv(v, -kc->ub, kc->ub);
                    if (prm.pred && chunk == 0) prm.pred[((size_t)0 * N + i) * prm.B + b] = v;
                    int dr = (int)kc->drow[i];
                    if (dr >= 0) {
                        double rr = (dA[i] - v) * kc->inv_yscale[i];
                        loss_sum += (prm.loss_kind == 0) ? fabs(rr) : rr * rr;
                    }
                }
#pragma unroll
                for (int i = 0; i < NS; ++i) dA[i] = dB[i];
                load_row(2, dB);
                if (HAS_T && prm.pred && chunk == 0) {
                    double v = Tconst;
                    if (prm.clamp_pred) v = clampv(v, -kc->ub, kc->ub);
                    prm.pred[((size_t)0 * N + NS) * prm.B + b] = v;
                }
                jsave = 1;
            }
        }

        // ================================================================== PRIMAL: one Tsit5 attempt
        int rc = -1;
        ++iter;
        bool last = false;
        if (jsave >= nsave) rc = 0;
        else if (iter > prm.maxiters) rc = 1;
        if (t + dt * (1.0 + 1e-13) >= tend) { dt = tend - t; last = true; }
        if (rc < 0 && (!(dt > kc->dtmin) || t + dt == t)) rc = 2;

        bool accept = false;
        double q = 1.0, lq11 = 0.0, lEE = 0.0;
        bool ee_zero = false;
        if (rc < 0) {
            double k[7][NS];   // k[0] = k1 (FSAL) ... k[6] = k7
            double unew[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) k[0][i] = k1[i];
#pragma unroll
            for (int s = 1; s < 7; ++s) {
                double g[NS], x[NS], gg[NS], r[NR];
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    double a = 0.0;
#pragma unroll
                    for (int j = 0; j < s; ++j) a = fma(Ts5::a(s - 1, j), k[j][i], a);
                    g[i] = fma(dt, a, u[i]);
                }
                if (s == 6) {
#pragma unroll
                    for (int i = 0; i < NS; ++i) unew[i] = g[i];
                }
                features<NS>(g, kc->lb, kc->ub, x, gg);
                rates<NS, NR, HAS_T>(th, x, bT, r);
                rhs_from_rates<NS, NR, HAS_T, USE_SCALE>(th, r, kc->scale, k[s]);
                publish_stage(s, x, gg, r);
            }
            double es = 0.0;
            bool finite = true;
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
            if (!finite) rc = 3;
            else {
                ee_zero = (es == 0.0);
                lEE = 0.5 * flog(ee_zero ? 1.0 : es);
                lq11 = kc->beta1 * lEE;
                q = ee_zero ? 1.0 / kc->qmax
                            : fmax(1.0 / kc->qmax, fmin(1.0 / kc->qmin, exp(lq11 - kc->beta2 * lqold) / kc->gamma));
                accept = (es <= 1.0);
            }
            if (rc < 0 && accept) {
                const double tnew = last ? tend : t + dt;
                double A_[NS], Bs[7][NS];
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    A_[i] = 0.0;
#pragma unroll
                    for (int j = 0; j < 7; ++j) Bs[j][i] = 0.0;
                }
                while (jsave < nsave) {
                    const double ts = ts_lds[jsave];
                    if (!(ts <= tnew)) break;
                    const bool at_end = (ts == tnew);
                    double bth[7];
                    Ts5::dense(at_end ? 1.0 : (ts - t) / dt, bth);
#pragma unroll
                    for (int i = 0; i < NS; ++i) {
                        double a = 0.0;
#pragma unroll
                        for (int j = 0; j < 7; ++j) a = fma(bth[j], k[j][i], a);
                        double v = at_end ? unew[i] : fma(dt, a, u[i]);
                        double mask = 1.0;
                        if (prm.clamp_pred) {
                            mask = (v > kc->ub || v < -kc->ub) ? 0.0 : 1.0;
                            v = clampv(v, -kc->ub, kc->ub);
                        }
                        if (prm.pred && chunk == 0) prm.pred[((size_t)jsave * N + i) * prm.B + b] = v;
                        int dr = (int)kc->drow[i];
                        if (dr >= 0) {
                            const double iy = kc->inv_yscale[i];
                            const double rr = (dA[i] - v) * iy;
                            double w;
                            if (prm.loss_kind == 0) { loss_sum += fabs(rr); w = signbit(rr) ? 1.0 : -1.0; }
                            else { loss_sum = fma(rr, rr, loss_sum); w = -2.0 * rr; }
                            w *= mask * iy;
                            A_[i] += w;
                            // at the end point the tangent is s_new = s + dt sum_j a_7j k'_j  (b_j(1) = a_7j, b_7(1) = 0)
#pragma unroll
                            for (int j = 0; j < 7; ++j) Bs[j][i] = fma(w * dt, at_end ? (j < 6 ? Ts5::a(5, j) : 0.0) : bth[j], Bs[j][i]);
                        }
                    }
                    if (HAS_T && prm.pred && chunk == 0) {
                        double v = prm.u0[(size_t)NS * prm.B + b];
                        if (prm.clamp_pred) v = clampv(v, -kc->ub, kc->ub);
                        prm.pred[((size_t)jsave * N + NS) * prm.B + b] = v;
                    }
                    ++jsave;
#pragma unroll
                    for (int i = 0; i < NS; ++i) dA[i] = dB[i];
                    load_row(jsave + 1, dB);
                }
                if (C > 0 && lead) {
#pragma unroll
                    for (int i = 0; i < NS; ++i) {
                        rec[(R_::UP + i) * GPW] = unew[i];
                        rec[(R_::FP + i) * GPW] = k[6][i];
                        rec[(R_::AA + i) * GPW] = A_[i];
#pragma unroll
                        for (int j = 0; j < 7; ++j) rec[(R_::BB + j * NS + i) * GPW] = Bs[j][i];
                    }
                }
                if (C == 0) {
#pragma unroll
                    for (int i = 0; i < NS; ++i) { u[i] = unew[i]; k1[i] = k[6][i]; }
                }
                t = tnew;
            }
        }
        __builtin_amdgcn_wave_barrier();

        if (rc < 0 && accept) {
            ++nacc;
            // ============================================================== TANGENT phase
            if (C > 0) {
#pragma unroll 1
                for (int qc = 0; qc < C; ++qc) {
                    const double *dcol = dth_lds + (chunk * C + qc) * NTHP;
                    double *Sq = S_s + qc * NS * 64;
                    double dth_r[NTH];          // the column's direction, held for all 7 stages
#pragma unroll
                    for (int m = 0; m < NTH; ++m) dth_r[m] = dcol[m];
                    double s[NS], kp[6][NS];    // k'_1..k'_6 (k'_7 is only needed for the interpolation seeds)
#pragma unroll
                    for (int i = 0; i < NS; ++i) s[i] = Sq[i * 64];
                    double acc = 0.0;
#pragma unroll
                    for (int i = 0; i < NS; ++i) acc = fma(rec[(R_::AA + i) * GPW], s[i], acc);
#pragma unroll
                    for (int st = 0; st < 7; ++st) {
                        // keep the stages apart in the instruction schedule: without this fence the scheduler hoists the
                        // record loads of all seven stages to the top of the column (210 live registers -> scratch)
                        __builtin_amdgcn_sched_barrier(0);
                        const int o = area(st);
                        // g'_st = s + dt sum_j a_{st,j} k'_j
                        double gs[NS];
#pragma unroll
                        for (int c = 0; c < NS; ++c) {
                            double a = 0.0;
#pragma unroll
                            for (int j = 0; j < st; ++j) a = fma(Ts5::a(st - 1, j), kp[j][c], a);
                            gs[c] = rec[(o + R_::GO + c) * GPW] * (st == 0 ? s[c] : fma(dt, a, s[c]));
                        }
                        double kps[NS];
#pragma unroll
                        for (int i = 0; i < NS; ++i) kps[i] = 0.0;
#pragma unroll
                        for (int j = 0; j < NR; ++j) {
                            double e = dth_r[L_::wb(j)];
                            if (HAS_T) e = fma(dth_r[L_::wi(NS, j)], xT, e);
#pragma unroll
                            for (int c = 0; c < NS; ++c) {
                                e = fma(dth_r[L_::wi(c, j)], rec[(o + R_::XO + c) * GPW], e);
                                e = fma(th[L_::wi(c, j)], gs[c], e);
                            }
                            const double rj = rec[(o + R_::RO + j) * GPW];
                            const double er = e * rj;
#pragma unroll
                            for (int i = 0; i < NS; ++i) {
                                kps[i] = fma(dth_r[L_::wo(i, j)], rj, kps[i]);
                                kps[i] = fma(th[L_::wo(i, j)], er, kps[i]);
                            }
                        }
#pragma unroll
                        for (int i = 0; i < NS; ++i) {
                            if (USE_SCALE) kps[i] *= kc->scale[i];
                            acc = fma(rec[(R_::BB + st * NS + i) * GPW], kps[i], acc);
                            if (st < 6) kp[st][i] = kps[i];
                        }
                    }
#pragma unroll
                    for (int i = 0; i < NS; ++i) {
                        double a = 0.0;
#pragma unroll
                        for (int j = 0; j < 6; ++j) a = fma(Ts5::a(5, j), kp[j][i], a);
                        Sq[i * 64] = fma(dt, a, s[i]);
                    }
                    gtr[qc] += acc;
                }
                // advance: stage 7 of this step is stage 1 of the next (rotate the record base), reload the FSAL point
                base += 6;
                if (base >= R_::NAREA) base -= R_::NAREA;
#pragma unroll
                for (int i = 0; i < NS; ++i) { u[i] = rec[(R_::UP + i) * GPW]; k1[i] = rec[(R_::FP + i) * GPW]; }
            }
            if (q >= kc->qsteady_min && q <= kc->qsteady_max) q = 1.0;
            lqold = ee_zero ? lqinit : fmax(lEE, lqinit);
            dt = fmin(dt / q, dtmax);
            if (jsave >= nsave) rc = 0;
        } else if (rc < 0) {
            ++nrej;
            dt = dt / fmin(1.0 / kc->qmin, exp(lq11) / kc->gamma);
        }
        __builtin_amdgcn_wave_barrier();

        if (rc >= 0) {
            const double denom = (double)prm.n_obs * (double)jsave;
            const double inv_den = jsave > 0 ? 1.0 / denom : 0.0;
            if (chunk == 0) {
                prm.loss[b] = loss_sum * inv_den;
                prm.retcode[b] = rc;
                prm.n_saved[b] = jsave;
                prm.n_accept[b] = nacc;
                prm.n_reject[b] = nrej;
            }
            if (C > 0) {
                double *grow = prm.gtraj + (size_t)traj * PPAD + chunk * C;
#pragma unroll
                for (int q_ = 0; q_ < C; ++q_) grow[q_] = gtr[q_] * inv_den;
            }
            traj = traj_next;
            traj_next = fetch_next();
            need_init = true;
        }
    }
}

}  // namespace crnn

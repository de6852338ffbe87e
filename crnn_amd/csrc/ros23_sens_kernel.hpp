// crnn_amd/csrc/ros23_sens_kernel.hpp -- gfx950 (MI355X): Rosenbrock23 + forward tangents with the step-size controller
// driven by ForwardDiff's DUAL-INCLUSIVE error norm (crnn_config.errnorm_sens = 1): the reference-faithful gradient mode.
//
// The reference forms its gradient by pushing ForwardDiff.Dual numbers through the adaptive solver
// (case2/case2.jl:195, robertson/rober_crnn.jl:219, case1/case1.jl:147).  DiffEqBase's error norm of a Dual-valued state
// weighs the partials together with the value, so the accept / reject decisions and the step sizes of a GRADIENT call
// differ from those of a plain solve -- and, because ForwardDiff works through the P parameters in chunks
// (pickchunksize: 25 -> 9 + 9 + 7, 43 -> 11 + 11 + 11 + 10, 24 -> 12 + 12), every chunk is its own adaptive solve whose
// norm sees only that chunk's partials.  One launch of this kernel is ONE chunk: primal + the chunk's tangent columns
// through every ATTEMPT (not only through accepted steps: the decision needs them), including the third stage's
// tangent k3' that only the error estimate uses:
//
//     W k1' = f0' + gam J' k1          W (k2-k1)' = f1' - k1' + gam J' (k2-k1)         s+ = s + dt k2'
//     W k3' = f2' - c32 (k2' - f1') - 2 (k1' - f0') + gam J' k3        e' = dt/6 (k1' - 2 k2' + k3')
//     EEst^2 = 1/n sum_i (e_i^2 + sum_k e'_ik^2) / (atol_i + rtol_i sqrt(max(u_i^2 + sum_k s_ik^2, u+_i^2 + sum_k s+_ik^2)))^2
//
// ([UNVERIFIED-DEP] DiffEqBase.ODE_DEFAULT_NORM on Dual arrays -- DiffEqBase is not vendored with the reference.  The
// initial step size uses the same norm: f0 and f1 carry the partials of p -- sens_init_dt below.)
//
// MI355X mapping: the lane-group layout of ros23_kernel.hpp (L lanes own one trajectory, C columns each, the primal
// step computed redundantly, a per-group LDS step record that lane 0 of the group publishes) with three differences:
// the seeds of the save points inside the attempted step, the new tangent columns (second LDS slot) and the gradient
// increments are PROVISIONAL until the group has summed its lanes' norm contributions (three ds_bpermute rounds per
// species) and taken the decision; the record also carries the u+ point and k3; trajectories come in wave-synchronous
// batches from a queue.  This mode costs about (1 + 1.6 C L) primal attempts per attempt and three or four launches per
// gradient: it exists for parity with the reference's step sequence, not for speed (bench.py reports its cost).
#pragma once
#include "ros23_kernel.hpp"

// phase timing (tools/kvariants.sh build prof="-DCRNN_SENS_PROF=1"; the library then prints the shares of wave 0 of block 0 to
// stderr after every chunk launch): s_memtime deltas per phase summed in scalar registers (32-bit: launches of a few ms), no
// fences at the phase boundaries -- the shares are a guide, not the measurement
#ifdef CRNN_SENS_PROF
namespace crnn { __device__ unsigned long long g_sens_prof[16]; }
#define SENS_T(k) do { const unsigned now_ = (unsigned)__builtin_readcyclecounter(); prof_acc[(k)] += now_ - prof_last; prof_last = now_; } while (0)
#else
#define SENS_T(k) do { } while (0)
#endif

// scheduling fences of the tangent column (CRNN_SCHED_FENCE), individually switchable for measurements
#ifndef SENS_FENCE_1
#define SENS_FENCE_1() CRNN_SCHED_FENCE()
#endif
#ifndef SENS_FENCE_2
#define SENS_FENCE_2() CRNN_SCHED_FENCE()
#endif
#ifndef SENS_FENCE_3
#define SENS_FENCE_3() CRNN_SCHED_FENCE()
#endif
#ifndef SENS_FENCE_4
#define SENS_FENCE_4() CRNN_SCHED_FENCE()
#endif
#ifndef SENS_FENCE_5
#define SENS_FENCE_5() CRNN_SCHED_FENCE()
#endif

namespace crnn {

template <int NS, int NR>
struct RecS {
    static constexpr int X0 = 0;
    static constexpr int G0 = X0 + NS;
    static constexpr int R0 = G0 + NS;
    static constexpr int X1 = R0 + NR;
    static constexpr int G1 = X1 + NS;
    static constexpr int R1 = G1 + NS;
    static constexpr int X2 = R1 + NR;
    static constexpr int G2 = X2 + NS;
    static constexpr int R2 = G2 + NS;
    static constexpr int K1 = R2 + NR;
    static constexpr int DK = K1 + NS;
    static constexpr int K3 = DK + NS;
    static constexpr int C1J = K3 + NS;
    static constexpr int CZD = C1J + NR;
    static constexpr int CZ3 = CZD + NR;
    static constexpr int GR0 = CZ3 + NR;
    static constexpr int AA = GR0 + NR;
    static constexpr int B1 = AA + NS;
    static constexpr int B2 = B1 + NS;
    static constexpr int NREC = B2 + NS;
};

// features() / rates() of a point that the L lanes of a group all evaluate (the primal attempt is redundant across the group): the
// logarithms and exponentials -- half of a primal attempt's arithmetic -- are divided among the lanes, lane q takes species
// q, q + L, ... and reactions q, q + L, ..., and gathered with ds_bpermute (the same bits on every lane: one lane formed them).
// flog_vec / fexp_vec work element by element, so the values are those of features() / rates().
template <int NS, int L>
__device__ __forceinline__ void features_grp(const double (&u)[NS], const double lb, const double ub, double (&x)[NS], double (&g)[NS],
                                             const int q, const int gbase) {
    constexpr int PER = (NS + L - 1) / L;
    double c[NS], cm[PER], xm[PER];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        c[i] = fmin(fmax(u[i], lb), ub);
        g[i] = (c[i] == u[i]) ? frcp1(c[i]) : 0.0;
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        cm[k] = c[k * L < NS ? k * L : 0];
#pragma unroll
        for (int m = 1; m < L; ++m)
            if (k * L + m < NS) cm[k] = (q == m) ? c[k * L + m] : cm[k];
    }
    flog_vec<PER>(cm, xm);
#pragma unroll
    for (int i = 0; i < NS; ++i) x[i] = __shfl(xm[i / L], gbase + i % L);
}
template <int NS, int NR, bool HAS_T, int L>
__device__ __forceinline__ void rates_grp(const double *__restrict__ th, const double (&x)[NS], const double (&bT)[NR], double (&r)[NR],
                                          const int q, const int gbase) {
    using L_ = Lay<NS, NR, HAS_T>;
    constexpr int PER = (NR + L - 1) / L;
    double z[NR], zm[PER], em[PER];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        double zz = bT[j];
#pragma unroll
        for (int i = 0; i < NS; ++i) zz = fma(th[L_::wi(i, j)], x[i], zz);
        z[j] = zz;
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        zm[k] = z[k * L < NR ? k * L : 0];
#pragma unroll
        for (int m = 1; m < L; ++m)
            if (k * L + m < NR) zm[k] = (q == m) ? z[k * L + m] : zm[k];
    }
    fexp_vec<PER>(zm, em);
#pragma unroll
    for (int j = 0; j < NR; ++j) r[j] = __shfl(em[j / L], gbase + j % L);
}

// sqrt(m) for finite m >= 0: v_rsq_f64 seed (~4e-8 like v_rcp_f64), two coupled Newton steps s += r/2 (m - s^2): 1.5 delta^2 after the
// first, rounding after the second (1-2 ulp); m = 0 gives 0
__device__ __forceinline__ double fsqrt_pos(const double m) {
    const double r = __builtin_amdgcn_rsq(m), hr = 0.5 * r;
    double s_ = m * r;
    s_ = fma(hr, fma(-s_, s_, m), s_);
    s_ = fma(hr, fma(-s_, s_, m), s_);
    return m > 0.0 ? s_ : 0.0;
}

// Hairer's initial step (OrdinaryDiffEq ode_determine_initdt) when the state carries partials: u0 is promoted to Duals with
// zero partials, f0 = f(u0, p) and f1 = f(u0 + dt0 f0, p) carry the partials of p, and every internalnorm is the dual-
// inclusive one -- d1 and d2 grow by the partials, d0 only shares the divisor ([UNVERIFIED-DEP] like the norm itself).
// Every lane of the group calls this with the same primal point (x0, r0, f0) and its own C columns; the partial sums are
// combined over the group's L lanes in lane order.  Stmp: the lane's LDS slot (C * NS doubles, stride 64) parks f0'.
// The lane's columns qc < nvalid are the rows dcols + qc * pitch of d theta / d p, the others read the all-zero row zcol -- or, MASKED
// (rows read from global memory, where there is no zero row to point at: tsit5_sens_kernel), row 0 with the result replaced by zero.
template <int NS, int NR, bool HAS_T, bool USE_SCALE, int C, int L, int ORDER, bool MASKED = false>
__device__ __forceinline__ double sens_init_dt(const double *__restrict__ th, const KConst *kc, const double *dcols, const int pitch,
                                               double *Stmp, const double (&u)[NS], const double (&f0)[NS], const double (&x0)[NS],
                                               const double (&r0)[NR], const double (&bT)[NR], const double xT, const double Tconst,
                                               const double dtmax, const int norm_cols, const int gbase, const int nvalid = C,
                                               const double *zcol = nullptr) {
    using L_ = Lay<NS, NR, HAS_T>;
    constexpr int N = L_::N;
    auto group_sum = [&](double v) -> double {
        double a = 0.0;
#pragma unroll
        for (int q = 0; q < L; ++q) a += __shfl(v, gbase + q);
        return a;
    };
    const double idiv = 1.0 / ((double)N * (1.0 + (double)norm_cols));
    double sk[NS], d0 = 0.0, d1 = 0.0;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        sk[i] = frcp(fma(fabs(u[i]), kc->rtol[i], kc->atol[i]));
        const double a = u[i] * sk[i], c = f0[i] * sk[i];
        d0 = fma(a, a, d0);
        d1 = fma(c, c, d1);
    }
    if (HAS_T) { const double a = Tconst * frcp(fma(fabs(Tconst), kc->rtol[NS], kc->atol[NS])); d0 = fma(a, a, d0); }
    double d1p = 0.0;
#pragma unroll 1
    for (int qc = 0; qc < C; ++qc) {
        const bool cvalid = qc < nvalid;
        const double *dcol = MASKED ? dcols + (cvalid ? qc : 0) * pitch : (cvalid ? dcols + qc * pitch : zcol);
        double f0p[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) f0p[i] = 0.0;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            double e = dcol[L_::wb(j)];
            if (HAS_T) e = fma(dcol[L_::wi(NS, j)], xT, e);
#pragma unroll
            for (int c = 0; c < NS; ++c) e = fma(dcol[L_::wi(c, j)], x0[c], e);
#pragma unroll
            for (int i = 0; i < NS; ++i) f0p[i] = fma(fma(th[L_::wo(i, j)], e, dcol[L_::wo(i, j)]), r0[j], f0p[i]);
        }
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            if (USE_SCALE) f0p[i] *= kc->scale[i];
            if (MASKED && !cvalid) f0p[i] = 0.0;
            Stmp[(qc * NS + i) * 64] = f0p[i];
            const double c = f0p[i] * sk[i];
            d1p = fma(c, c, d1p);
        }
    }
    d1 += group_sum(d1p);
    d0 = sqrt(d0 * idiv);
    d1 = sqrt(d1 * idiv);
    double dt0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * (d0 / d1);
    dt0 = fmin(dt0, dtmax);
    double u1[NS], x1[NS], g1[NS], r1[NR], f1[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) u1[i] = fma(dt0, f0[i], u[i]);
    features<NS>(u1, kc->lb, kc->ub, x1, g1);
    rates<NS, NR, HAS_T>(th, x1, bT, r1);
    rhs_from_rates<NS, NR, HAS_T, USE_SCALE>(th, r1, kc->scale, f1);
    double d2 = 0.0, d2p = 0.0;
#pragma unroll
    for (int i = 0; i < NS; ++i) { const double e = (f1[i] - f0[i]) * sk[i]; d2 = fma(e, e, d2); }
#pragma unroll 1
    for (int qc = 0; qc < C; ++qc) {
        const bool cvalid = qc < nvalid;
        const double *dcol = MASKED ? dcols + (cvalid ? qc : 0) * pitch : (cvalid ? dcols + qc * pitch : zcol);
        double gs1[NS], f1p[NS];
#pragma unroll
        for (int c = 0; c < NS; ++c) { gs1[c] = g1[c] * (dt0 * Stmp[(qc * NS + c) * 64]); f1p[c] = 0.0; }
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            double e = dcol[L_::wb(j)];
            if (HAS_T) e = fma(dcol[L_::wi(NS, j)], xT, e);
#pragma unroll
            for (int c = 0; c < NS; ++c) { e = fma(dcol[L_::wi(c, j)], x1[c], e); e = fma(th[L_::wi(c, j)], gs1[c], e); }
#pragma unroll
            for (int i = 0; i < NS; ++i) f1p[i] = fma(fma(th[L_::wo(i, j)], e, dcol[L_::wo(i, j)]), r1[j], f1p[i]);
        }
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            if (USE_SCALE) f1p[i] *= kc->scale[i];
            if (MASKED && !cvalid) f1p[i] = 0.0;
            const double e = (f1p[i] - Stmp[(qc * NS + i) * 64]) * sk[i];
            d2p = fma(e, e, d2p);
        }
    }
    d2 = sqrt((d2 + group_sum(d2p)) * idiv) / dt0;
    const double dm = fmax(d1, d2);
    // 10^(-(2 + log10 dm)/order) = exp(-(ln 100 + ln dm)/order)
    const double dt1 = (dm <= 1e-15) ? fmax(1e-6, dt0 * 1e-3) : fexp_ctl((-1.0 / ORDER) * (4.605170185988091368 + flog(dm)));
    return fmax(kc->dtmin, fmin(fmin(100.0 * dt0, dt1), dtmax));
}

// DROWS: rows of d theta / d p staged in LDS (the last one all zero).  L * C + 1: one chunk per launch.  P + 1 or more: ALL
// chunks of a P-parameter gradient in ONE launch (SolveParams::n_chunks > 1) -- a batch is (chunk, GPW trajectories), batches
// of the chunks alternate in the queue, a lane's columns are rows cid * chunk_size + chunk * C + qc while those lie inside the
// chunk and inside P (the surplus partials of ForwardDiff's last chunk are zero: the zero row), gradient rows come out
// compact [count][P].  The chunks are independent adaptive solves either way; one launch has one tail instead of one per
// chunk, and no batch of a later chunk waits for the stragglers of an earlier one.
template <int NS, int NR, bool HAS_T, bool USE_SCALE, int C, int L, int BLOCK, int DROWS = L * C + 1>
__global__ __launch_bounds__(BLOCK) void ros23_sens_kernel(const SolveParams prm, const double *__restrict__ theta,
                                                           const double *__restrict__ dtheta) {
    using L_ = Lay<NS, NR, HAS_T>;
    using R_ = RecS<NS, NR>;
    constexpr int N = L_::N;
    constexpr int NTH = L_::NTH;
    constexpr int NTHP = L_::NTHP;
    constexpr int NREC = R_::NREC;
    constexpr int WAVES = BLOCK / 64;
    constexpr int GPW = 64 / L;
    constexpr int PPAD = L * C;
    static_assert(C > 0 && L >= 1 && L <= 64 && DROWS > PPAD, "lane-group shape");
    using Solver = typename SolverSel<(NR < NS), NS, NR, HAS_T, USE_SCALE>::type;

    __shared__ double kc_lds[kNConst];
    __shared__ double ts_lds[kMaxSave];
    __shared__ double dth_lds[DROWS * NTHP];
    __shared__ double S_lds[2 * WAVES * C * NS * 64];   // two slots per lane: committed columns / columns of the attempt
    __shared__ double rec_lds[WAVES * NREC * GPW];
#ifdef CRNN_SENS_PROF
    unsigned prof_acc[12] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};   // wave-uniform: scalar registers
    unsigned prof_last = (unsigned)__builtin_readcyclecounter();
#endif

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = lane / L, chunk = lane - grp * L;
    const bool lane_active = grp < GPW;
    const bool lead = lane_active && chunk == 0;
    const int gbase = grp * L;                           // first lane of this group
    double *const S_base = S_lds + (size_t)wave * 2 * C * NS * 64 + lane;   // slot s, (qc, i): S_base[(s*C*NS + qc*NS + i) * 64]
    double *const rec = rec_lds + wave * NREC * GPW + (lane_active ? grp : 0);

    for (int idx = tid; idx < kNConst; idx += BLOCK) kc_lds[idx] = reinterpret_cast<const double *>(prm.kc)[idx];
    for (int idx = tid; idx < prm.n_save; idx += BLOCK) ts_lds[idx] = prm.tsave[idx];
    for (int idx = tid; idx < DROWS * NTHP; idx += BLOCK) {
        const int k = idx / NTHP, m = idx - k * NTHP;
        dth_lds[idx] = (k < prm.P && k < DROWS - 1 && m < NTH) ? dtheta[(size_t)k * NTH + m] : 0.0;
    }
    const double *const zcol = dth_lds + (DROWS - 1) * NTHP;
    const int nch = prm.n_chunks > 1 ? prm.n_chunks : 1;     // chunks in this launch
    const int cs_eff = nch > 1 ? prm.chunk_size : PPAD;      // partials of a (full) chunk
    const int ncols = nch > 1 ? prm.P : PPAD;                // columns in a gradient row
    __syncthreads();
    const KConst *kc = reinterpret_cast<const KConst *>(kc_lds);
    const double *__restrict__ th = theta;

    const double d_ = 0.29289321881345248, c32 = 7.4142135623730950, inv12d = 2.4142135623730950;
    const int nsave = prm.n_save;
    const double tend = ts_lds[nsave - 1], ts0 = ts_lds[0], t0 = kc->t0;
    const double dtmax = tend - t0;
    const double lqinit = flog(kc->qoldinit);

    // wave-synchronous batches of GPW trajectories from a queue (one atomic per batch, fetched a batch ahead); positions map to
    // trajectories through SolveParams::perm when the context has an order by step counts (its last plain solve): a batch
    // lasts as long as its longest trajectory, so it should be homogeneous, and the queue longest-first
    const unsigned nwaves = gridDim.x * WAVES;
    const int64_t nbatch = ((prm.count + GPW - 1) / GPW) * nch;
    int64_t bi = (int64_t)blockIdx.x * WAVES + wave;

    // sum of v over the L lanes of this lane's group, in lane order (identical on every lane of the group)
    auto group_sum = [&](double v) -> double {
        double a = 0.0;
#pragma unroll
        for (int q = 0; q < L; ++q) a += __shfl(v, gbase + q);
        return a;
    };

    for (; bi < nbatch;) {
        unsigned nx = 0;
        if (lane == 0) nx = (unsigned)atomicAdd(prm.queue, 1ULL);
        const int cid = nch > 1 ? (int)(bi % nch) : 0;
        const int64_t pos = (nch > 1 ? bi / nch : bi) * GPW + grp;
        const int col0 = cid * cs_eff + chunk * C;        // this lane's first column; the valid ones are a prefix
        const int nvalid = max(0, min(C, min(cs_eff - chunk * C, ncols - col0)));
        const double *const dcols = dth_lds + (nvalid > 0 ? col0 : 0) * NTHP;
        if (lane_active && pos < prm.count) {
        CRNN_CHK(!prm.perm || ((int64_t)prm.perm[pos] >= 0 && (int64_t)prm.perm[pos] < prm.count), 0x5001);
        const int64_t traj = prm.perm ? (int64_t)prm.perm[pos] : pos;
        const int64_t b = prm.first + traj;
        const double *const drows = prm.data + (size_t)b * prm.row_stride;
        double u[NS], f0[NS], g0[NS], x0[NS], r0[NR], bT[NR], gtr[C];
        double xT = 0.0, Tconst = 0.0;
#pragma unroll
        for (int i = 0; i < NS; ++i) u[i] = prm.u0[(size_t)i * prm.B + b];
        if (HAS_T) {
            Tconst = prm.u0[(size_t)NS * prm.B + b];
            xT = kc->inv_R * frcp(Tconst);
        }
#pragma unroll
        for (int j = 0; j < NR; ++j) bT[j] = HAS_T ? fma(th[L_::wi(NS, j)], xT, th[L_::wb(j)]) : th[L_::wb(j)];
        features_grp<NS, L>(u, kc->lb, kc->ub, x0, g0, chunk, gbase);
        rates_grp<NS, NR, HAS_T, L>(th, x0, bT, r0, chunk, gbase);
        rhs_from_rates<NS, NR, HAS_T, USE_SCALE>(th, r0, kc->scale, f0);
        // Hairer initial step with the dual-inclusive norms (the Sn slot parks f0' meanwhile; it is zeroed below)
        const double dt0_ = sens_init_dt<NS, NR, HAS_T, USE_SCALE, C, L, 2>(th, kc, dcols, NTHP,
                                                                              S_base + (size_t)C * NS * 64, u, f0, x0, r0, bT, xT,
                                                                              Tconst, dtmax, prm.norm_cols, gbase, nvalid, zcol);
        double dt = dt0_;
        double t = t0, lqold = lqinit, loss_sum = 0.0;
        int iter = 0, jsave = 0, nacc = 0, nrej = 0, cur = 0, rc = -1;
        // sum_k s_ik^2 over the group's committed columns: the accepted attempt's sum_k s+_ik^2 (the same sums of the same numbers), kept
        double nas[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) nas[i] = 0.0;
#pragma unroll
        for (int q = 0; q < C; ++q) gtr[q] = 0.0;
#pragma unroll
        for (int q = 0; q < 2 * C * NS; ++q) S_base[q * 64] = 0.0;
        if (ts0 == t0) {   // save_start: a loss term without gradient
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                double v = u[i];
                if (prm.clamp_pred) v = clampv(v, -kc->ub, kc->ub);
                if (prm.pred && chunk == 0) prm.pred[((size_t)0 * N + i) * prm.B + b] = v;
                const int dr = (int)kc->drow[i];
                if (dr >= 0) {
                    const double rr = (drows[dr] - v) * kc->inv_yscale[i];
                    loss_sum += (prm.loss_kind == 0) ? fabs(rr) : rr * rr;
                }
            }
            if (HAS_T && prm.pred && chunk == 0) {
                double v = Tconst;
                if (prm.clamp_pred) v = clampv(v, -kc->ub, kc->ub);
                prm.pred[((size_t)0 * N + NS) * prm.B + b] = v;
            }
            jsave = 1;
        }

        SENS_T(0);
        while (rc < 0) {
            ++iter;
            bool last = false;
            if (jsave >= nsave) { rc = 0; break; }
            if (iter > prm.maxiters) { rc = 1; break; }
            if (t + dt * (1.0 + 1e-13) >= tend) { dt = tend - t; last = true; }
            if (!(dt > kc->dtmin) || t + dt == t) { rc = 2; break; }

            // the observed rows of the next two save points, requested before the stage arithmetic (an attempt covers 1.5 save
            // points on case2; read where they are used, every one of them was an exposed HBM / L2 round trip)
            double dA[NS], dB[NS];
            {
                const int ja = jsave < nsave ? jsave : nsave - 1, jb = jsave + 1 < nsave ? jsave + 1 : nsave - 1;
                const double *const ra = drows + (size_t)ja * prm.n_obs, *const rb = drows + (size_t)jb * prm.n_obs;
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const int dr = (int)kc->drow[i];
                    CRNN_CHK((size_t)b * prm.row_stride + (size_t)jb * prm.n_obs + (dr >= 0 ? dr : 0) < (size_t)prm.B * prm.row_stride, 0x5002);
                    dA[i] = ra[dr >= 0 ? dr : 0];
                    dB[i] = rb[dr >= 0 ? dr : 0];
                }
            }
            double tsc = ts_lds[jsave < nsave ? jsave : nsave - 1];   // the next save time, requested ahead like the rows
            // ============================================================ PRIMAL: one Rosenbrock23 attempt
            Solver W;
            const double gam = d_ * dt;
            double gr0[NR];
#pragma unroll
            for (int j = 0; j < NR; ++j) gr0[j] = gam * r0[j];
            double k1[NS], dk[NS], k3[NS], unew[NS], f1[NS], f2[NS], g2[NS], x2[NS], r2[NR], ev[NS];
            const bool okf = W.factor(th, g0, r0, gam, kc->scale);
#pragma unroll
            for (int i = 0; i < NS; ++i) k1[i] = f0[i];
            W.solve(th, g0, gr0, kc->scale, k1);
            {
                double u1[NS], x1[NS], g1[NS], r1[NR];
#pragma unroll
                for (int i = 0; i < NS; ++i) u1[i] = fma(0.5 * dt, k1[i], u[i]);
                features_grp<NS, L>(u1, kc->lb, kc->ub, x1, g1, chunk, gbase);
                rates_grp<NS, NR, HAS_T, L>(th, x1, bT, r1, chunk, gbase);
                rhs_from_rates<NS, NR, HAS_T, USE_SCALE>(th, r1, kc->scale, f1);
                if (lead) {
#pragma unroll
                    for (int i = 0; i < NS; ++i) { rec[(R_::X1 + i) * GPW] = x1[i]; rec[(R_::G1 + i) * GPW] = g1[i]; }
#pragma unroll
                    for (int j = 0; j < NR; ++j) rec[(R_::R1 + j) * GPW] = r1[j];
                }
            }
#pragma unroll
            for (int i = 0; i < NS; ++i) dk[i] = f1[i] - k1[i];
            W.solve(th, g0, gr0, kc->scale, dk);
#pragma unroll
            for (int i = 0; i < NS; ++i) unew[i] = fma(dt, k1[i] + dk[i], u[i]);
            features_grp<NS, L>(unew, kc->lb, kc->ub, x2, g2, chunk, gbase);
            rates_grp<NS, NR, HAS_T, L>(th, x2, bT, r2, chunk, gbase);
            rhs_from_rates<NS, NR, HAS_T, USE_SCALE>(th, r2, kc->scale, f2);
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const double k2i = k1[i] + dk[i];
                k3[i] = f2[i] - c32 * (k2i - f1[i]) - 2.0 * (k1[i] - f0[i]);
            }
            W.solve(th, g0, gr0, kc->scale, k3);
            bool finite = okf;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const double k2i = k1[i] + dk[i];
                ev[i] = dt * (1.0 / 6.0) * (k1[i] - 2.0 * k2i + k3[i]);
                finite = finite && isfinite(unew[i]) && isfinite(ev[i]);
            }
            if (!finite) { rc = 3; break; }

            SENS_T(1);
            // ---- PROVISIONAL save points of (t, tnew]: loss terms and seeds as if the attempt were accepted.  Straight-line per
            //      save point: an unobserved species is a zero weight, an unclamped prediction an infinite clamp, the loss kind a
            //      select -- per-species branches (forty small blocks per save point) cost a quarter of the kernel's time; the
            //      save time and the observed row of the NEXT save points are requested while this one is worked on.
            const double tnew = last ? tend : t + dt;
            double A_[NS], B1[NS], B2[NS], iym[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                A_[i] = 0.0; B1[i] = 0.0; B2[i] = 0.0;
                iym[i] = kc->drow[i] >= 0.0 ? kc->inv_yscale[i] : 0.0;
            }
            const double ubc = prm.clamp_pred ? kc->ub : __builtin_inf();
            const bool lk0 = prm.loss_kind == 0;
            double loss_new = loss_sum;
            int jnew = jsave;
            while (jnew < nsave) {
                const double ts = tsc;
                if (!(ts <= tnew)) break;
                tsc = ts_lds[jnew + 1 < nsave ? jnew + 1 : nsave - 1];
                double ob[NS];
                {   // rotate the row queue: this point's row, the next one's, and a request for the one after
                    const int jc = jnew + 2 < nsave ? jnew + 2 : nsave - 1;
                    const double *const rc_ = drows + (size_t)jc * prm.n_obs;
#pragma unroll
                    for (int i = 0; i < NS; ++i) {
                        const int dr = (int)kc->drow[i];
                        ob[i] = dA[i]; dA[i] = dB[i]; dB[i] = rc_[dr >= 0 ? dr : 0];
                    }
                }
                const bool at_end = (ts == tnew);
                const double Th = at_end ? 1.0 : (ts - t) / dt;
                const double c1 = at_end ? 0.0 : Th * (1.0 - Th) * inv12d;
                const double c2 = at_end ? 1.0 : Th * (Th - 2.0 * d_) * inv12d;
                double vv[NS];
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const double k2i = k1[i] + dk[i];
                    double v = at_end ? unew[i] : fma(dt, fma(c1, k1[i], c2 * k2i), u[i]);
                    const double mask = (v > ubc || v < -ubc) ? 0.0 : 1.0;
                    v = clampv(v, -ubc, ubc);
                    vv[i] = v;
                    const double iy = iym[i];
                    const double rr = (ob[i] - v) * iy;
                    loss_new = lk0 ? loss_new + fabs(rr) : fma(rr, rr, loss_new);
                    double w = lk0 ? (signbit(rr) ? 1.0 : -1.0) : -2.0 * rr;
                    w *= mask * iy;
                    A_[i] += w;
                    B1[i] = fma(w, dt * c1, B1[i]);
                    B2[i] = fma(w, dt * c2, B2[i]);
                }
                if (prm.pred && chunk == 0) {   // a rejected attempt's values are overwritten by the accepted step that covers this save point
#pragma unroll
                    for (int i = 0; i < NS; ++i) prm.pred[((size_t)jnew * N + i) * prm.B + b] = vv[i];
                    if (HAS_T) prm.pred[((size_t)jnew * N + NS) * prm.B + b] = clampv(Tconst, -ubc, ubc);
                }
                ++jnew;
            }
            SENS_T(2);
            // ---- publish the step record
            {
                double c1j[NR], czd[NR], cz3[NR];
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    double z1 = 0.0, zd = 0.0, z3 = 0.0;
#pragma unroll
                    for (int c = 0; c < NS; ++c) {
                        const double wg = th[L_::wi(c, j)] * g0[c];
                        z1 = fma(wg, k1[c], z1);
                        zd = fma(wg, dk[c], zd);
                        z3 = fma(wg, k3[c], z3);
                    }
                    c1j[j] = fma(gam, z1, 1.0);
                    czd[j] = gam * zd;
                    cz3[j] = gam * z3;
                }
                if (lead) {
#pragma unroll
                    for (int i = 0; i < NS; ++i) {
                        rec[(R_::X0 + i) * GPW] = x0[i]; rec[(R_::G0 + i) * GPW] = g0[i];
                        rec[(R_::X2 + i) * GPW] = x2[i]; rec[(R_::G2 + i) * GPW] = g2[i];
                        rec[(R_::K1 + i) * GPW] = k1[i]; rec[(R_::DK + i) * GPW] = dk[i]; rec[(R_::K3 + i) * GPW] = k3[i];
                        rec[(R_::AA + i) * GPW] = A_[i]; rec[(R_::B1 + i) * GPW] = B1[i]; rec[(R_::B2 + i) * GPW] = B2[i];
                    }
#pragma unroll
                    for (int j = 0; j < NR; ++j) {
                        rec[(R_::R0 + j) * GPW] = r0[j]; rec[(R_::R2 + j) * GPW] = r2[j];
                        rec[(R_::C1J + j) * GPW] = c1j[j]; rec[(R_::CZD + j) * GPW] = czd[j]; rec[(R_::CZ3 + j) * GPW] = cz3[j];
                        rec[(R_::GR0 + j) * GPW] = gr0[j];
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            SENS_T(3);
            // ============================================================ TANGENTS of the attempt, C columns per lane
            double *const Sc = S_base + (size_t)cur * C * NS * 64;         // committed columns
            double *const Sn = S_base + (size_t)(cur ^ 1) * C * NS * 64;   // columns after this attempt
            double ee[NS], nb[NS], gnew[C];
#pragma unroll
            for (int i = 0; i < NS; ++i) { ee[i] = 0.0; nb[i] = 0.0; }
            const double hdt = 0.5 * dt;
#pragma unroll 1
            for (int qc = 0; qc < C; ++qc) {
                const double *dcol = qc < nvalid ? dcols + qc * NTHP : zcol;
                const double *Sq = Sc + qc * NS * 64;
                double *Sqn = Sn + qc * NS * 64;
                // The LDS operands of a loop step are requested one step ahead of their use (double-buffered by hand, the fences keep
                // the requests above the arithmetic of the current step): the lane's wavefront has its SIMD to itself, so nothing
                // else hides an LDS round trip -- left to the scheduler, ~100 of them per column were waited for one at a time
                // (36 % of the wave cycles waiting).  The arithmetic and its order are unchanged.
                struct P1 { double sc, g, x0, x1, x2, k1, dk, k3, dwi[NR]; };
                struct P2 { double r0, r1, r2, gr, c1, cz, c3, dwo[NS]; };
                auto load1 = [&](const int c, P1 &v) {
                    v.sc = Sq[c * 64];
                    v.g = rec[(R_::G0 + c) * GPW];
                    v.x0 = rec[(R_::X0 + c) * GPW]; v.x1 = rec[(R_::X1 + c) * GPW]; v.x2 = rec[(R_::X2 + c) * GPW];
                    v.k1 = rec[(R_::K1 + c) * GPW]; v.dk = rec[(R_::DK + c) * GPW]; v.k3 = rec[(R_::K3 + c) * GPW];
#pragma unroll
                    for (int j = 0; j < NR; ++j) v.dwi[j] = dcol[L_::wi(c, j)];
                };
                auto load2 = [&](const int j, P2 &v) {
                    v.r0 = rec[(R_::R0 + j) * GPW]; v.r1 = rec[(R_::R1 + j) * GPW]; v.r2 = rec[(R_::R2 + j) * GPW];
                    v.gr = rec[(R_::GR0 + j) * GPW];
                    v.c1 = rec[(R_::C1J + j) * GPW]; v.cz = rec[(R_::CZD + j) * GPW]; v.c3 = rec[(R_::CZ3 + j) * GPW];
#pragma unroll
                    for (int i = 0; i < NS; ++i) v.dwo[i] = dcol[L_::wo(i, j)];
                };
                // ---- pass 1 (species-major)
                double e0[NR], e1d[NR], e2d[NR], zp1[NR], zpd[NR], zp3[NR];
                double gq[NS], grq[NR], sv[NS];
                P1 p1a, p1b;
                P2 p2a, p2b;
                {
                    double dwb[NR], dwT[NR];
#pragma unroll
                    for (int j = 0; j < NR; ++j) { dwb[j] = dcol[L_::wb(j)]; dwT[j] = HAS_T ? dcol[L_::wi(NS, j)] : 0.0; }
                    load1(0, p1a);
                    SENS_FENCE_1();
#pragma unroll
                    for (int j = 0; j < NR; ++j) {
                        double e = dwb[j];
                        if (HAS_T) e = fma(dwT[j], xT, e);
                        e0[j] = e; e1d[j] = e; e2d[j] = e; zp1[j] = 0.0; zpd[j] = 0.0; zp3[j] = 0.0;
                    }
                }
#pragma unroll
                for (int c = 0; c < NS; ++c) {
                    P1 &cur1 = (c & 1) ? p1b : p1a;
                    P1 &nxt1 = (c & 1) ? p1a : p1b;
                    if (c + 1 < NS) load1(c + 1, nxt1); else load2(0, p2a);
                    SENS_FENCE_1();
                    const double sc_ = cur1.sc;
                    const double g = cur1.g;
                    gq[c] = g;
                    sv[c] = sc_;
                    const double x0c = cur1.x0, x1c = cur1.x1, x2c = cur1.x2;
                    const double k1c = cur1.k1, dkc = cur1.dk, k3c = cur1.k3;
                    const double gsv = g * sc_;
                    const double hsv = -g * gsv;   // g' = -g^2 s inside the window (g = 1/u), 0 outside
#pragma unroll
                    for (int j = 0; j < NR; ++j) {
                        const double dwi = cur1.dwi[j];
                        const double wi = th[L_::wi(c, j)];
                        e0[j] = fma(dwi, x0c, e0[j]);
                        e0[j] = fma(wi, gsv, e0[j]);
                        e1d[j] = fma(dwi, x1c, e1d[j]);
                        e2d[j] = fma(dwi, x2c, e2d[j]);
                        const double m = fma(dwi, g, wi * hsv);
                        zp1[j] = fma(m, k1c, zp1[j]);
                        zpd[j] = fma(m, dkc, zpd[j]);
                        zp3[j] = fma(m, k3c, zp3[j]);
                    }
                    SENS_FENCE_1();
                }
                SENS_T(4);
                // ---- pass 2 (reaction-major)
                double rhs1[NS], w2[NS], w3[NS], f0p[NS], f1d[NS], f2d[NS];
                double g1v[NS], r1v[NR];      // requested during the last step of pass 2, used behind the first solve
#pragma unroll
                for (int i = 0; i < NS; ++i) { rhs1[i] = 0.0; w2[i] = 0.0; w3[i] = 0.0; f0p[i] = 0.0; f1d[i] = 0.0; f2d[i] = 0.0; }
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    P2 &cur2 = (j & 1) ? p2b : p2a;
                    P2 &nxt2 = (j & 1) ? p2a : p2b;
                    if (j + 1 < NR) load2(j + 1, nxt2);
                    else {
#pragma unroll
                        for (int c = 0; c < NS; ++c) g1v[c] = rec[(R_::G1 + c) * GPW];
#pragma unroll
                        for (int jj = 0; jj < NR; ++jj) r1v[jj] = rec[(R_::R1 + jj) * GPW];
                    }
                    SENS_FENCE_2();
                    const double r0j = cur2.r0, r1j = cur2.r1, r2j = cur2.r2;
                    const double gr = cur2.gr;
                    grq[j] = gr;
                    const double c1 = cur2.c1, cz = cur2.cz, c3 = cur2.c3;
                    const double y1 = gr * zp1[j], yd = gr * zpd[j], y3 = gr * zp3[j];
#pragma unroll
                    for (int i = 0; i < NS; ++i) {
                        const double wo = th[L_::wo(i, j)];
                        const double dwo = cur2.dwo[i];
                        const double H = fma(wo, e0[j], dwo) * r0j;
                        rhs1[i] = fma(H, c1, rhs1[i]);
                        rhs1[i] = fma(wo, y1, rhs1[i]);
                        w2[i] = fma(H, cz, w2[i]);
                        w2[i] = fma(wo, yd, w2[i]);
                        w3[i] = fma(H, c3, w3[i]);
                        w3[i] = fma(wo, y3, w3[i]);
                        f0p[i] += H;
                        f1d[i] = fma(dwo, r1j, f1d[i]);
                        f2d[i] = fma(dwo, r2j, f2d[i]);
                    }
                    SENS_FENCE_2();
                }
                if (USE_SCALE) {
#pragma unroll
                    for (int i = 0; i < NS; ++i) { const double sc = kc->scale[i]; rhs1[i] *= sc; w2[i] *= sc; w3[i] *= sc; f0p[i] *= sc; }
                }
                SENS_T(5);
                W.solve(th, gq, grq, kc->scale, rhs1);   // k1'
                SENS_FENCE_3();
                SENS_T(6);
                // f1' at u1 with s1 = s + dt/2 k1'
                double f1p[NS];
                double av[NS], b1v[NS], b2v[NS];   // the save-point seeds: requested here, used behind the second solve
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    av[i] = rec[(R_::AA + i) * GPW]; b1v[i] = rec[(R_::B1 + i) * GPW]; b2v[i] = rec[(R_::B2 + i) * GPW];
                }
                SENS_FENCE_3();
                {
                    double gs1[NS];
#pragma unroll
                    for (int c = 0; c < NS; ++c) gs1[c] = g1v[c] * fma(hdt, rhs1[c], sv[c]);
#pragma unroll
                    for (int i = 0; i < NS; ++i) f1p[i] = f1d[i];
#pragma unroll
                    for (int j = 0; j < NR; ++j) {
                        double e = e1d[j];
#pragma unroll
                        for (int c = 0; c < NS; ++c) e = fma(th[L_::wi(c, j)], gs1[c], e);
                        const double er = e * r1v[j];
#pragma unroll
                        for (int i = 0; i < NS; ++i) f1p[i] = fma(th[L_::wo(i, j)], er, f1p[i]);
                    }
                    if (USE_SCALE) {
#pragma unroll
                        for (int i = 0; i < NS; ++i) f1p[i] *= kc->scale[i];
                    }
                }
                SENS_FENCE_4();
                SENS_T(7);
                double rhs2[NS];
                double g2v[NS], r2v[NR];      // requested before the second solve, used behind it
#pragma unroll
                for (int c = 0; c < NS; ++c) g2v[c] = rec[(R_::G2 + c) * GPW];
#pragma unroll
                for (int j = 0; j < NR; ++j) r2v[j] = rec[(R_::R2 + j) * GPW];
                SENS_FENCE_4();
#pragma unroll
                for (int i = 0; i < NS; ++i) rhs2[i] = f1p[i] - rhs1[i] + w2[i];
                W.solve(th, gq, grq, kc->scale, rhs2);   // (k2 - k1)'
                double k2p[NS], snv[NS], acc = 0.0;
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const double si = sv[i];
                    k2p[i] = rhs1[i] + rhs2[i];
                    acc = fma(av[i], si, acc);
                    acc = fma(b1v[i], rhs1[i], acc);
                    acc = fma(b2v[i], k2p[i], acc);
                    const double sn = fma(dt, k2p[i], si);
                    Sqn[i * 64] = sn;
                    snv[i] = sn;
                    nb[i] = fma(sn, sn, nb[i]);
                }
                gnew[qc] = acc;
                SENS_FENCE_5();
                SENS_T(8);
                // f2' at u+ with s+ ; then W k3' = f2' - c32 (k2' - f1') - 2 (k1' - f0') + gam J' k3
                double rhs3[NS];
                {
                    double gs2[NS];
#pragma unroll
                    for (int c = 0; c < NS; ++c) gs2[c] = g2v[c] * snv[c];
#pragma unroll
                    for (int i = 0; i < NS; ++i) rhs3[i] = f2d[i];
#pragma unroll
                    for (int j = 0; j < NR; ++j) {
                        double e = e2d[j];
#pragma unroll
                        for (int c = 0; c < NS; ++c) e = fma(th[L_::wi(c, j)], gs2[c], e);
                        const double er = e * r2v[j];
#pragma unroll
                        for (int i = 0; i < NS; ++i) rhs3[i] = fma(th[L_::wo(i, j)], er, rhs3[i]);
                    }
                }
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const double f2p = USE_SCALE ? rhs3[i] * kc->scale[i] : rhs3[i];
                    rhs3[i] = f2p - c32 * (k2p[i] - f1p[i]) - 2.0 * (rhs1[i] - f0p[i]) + w3[i];
                }
                SENS_T(9);
                W.solve(th, gq, grq, kc->scale, rhs3);   // k3'
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const double de = dt * (1.0 / 6.0) * (rhs1[i] - 2.0 * k2p[i] + rhs3[i]);
                    ee[i] = fma(de, de, ee[i]);
                }
            }
            SENS_T(10);
            // ---- the group's dual-inclusive error norm and the decision (identical on the L lanes of the group)
            double es = 0.0;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                nb[i] = group_sum(nb[i]);
                const double nai = fma(u[i], u[i], nas[i]);
                const double nbi = fma(unew[i], unew[i], nb[i]);
                const double eei = fma(ev[i], ev[i], group_sum(ee[i]));
                // sqrt and 1 / sc to 1-2 ulp (hardware seeds + one / two Newton steps: a third of the instructions of the IEEE
                // routines, six of each per attempt); EEst^2 moves in its last digits, the decisions only if it sits within them of 1
                const double sc = fma(kc->rtol[i], fsqrt_pos(fmax(nai, nbi)), kc->atol[i]);
                const double isc = frcp(sc);
                es = fma(eei * isc, isc, es);
            }
            // / length(u) (errnorm_sens = 1) or / totallength(u) = n (1 + partials per Dual) (errnorm_sens = 2)
            es = es / ((double)N * (1.0 + (double)prm.norm_cols));
            if (!isfinite(es)) { rc = 3; break; }
            const bool ee_zero = (es == 0.0);
            const double lEE = 0.5 * flog_ctl(ee_zero ? 1.0 : es);
            const double lq11 = kc->beta1 * lEE;
            // one exponential and one division for both outcomes (with 21 trajectories in a wavefront some lane rejects in most
            // iterations, so both branches used to run): accepted: q = clamp(exp(lq11 - beta2 lqold) / gamma), rejected: exp(lq11) / gamma
            const bool accept = es <= 1.0;
            // x, g, r of the point the trajectory stands on after this attempt come back from the step record (u+ if accepted, u_n
            // if not: the three areas lie 2 (2 NS + NR) fields apart) -- held in registers across the tangent columns they were
            // thirty doubles of a register file that is full; requested here, used by the next attempt
            {
                const int po = accept ? (R_::X2 - R_::X0) : 0;
                static_assert(R_::X2 - R_::X0 == R_::G2 - R_::G0 && R_::X2 - R_::X0 == R_::R2 - R_::R0, "record layout");
#pragma unroll
                for (int i = 0; i < NS; ++i) { x0[i] = rec[(R_::X0 + po + i) * GPW]; g0[i] = rec[(R_::G0 + po + i) * GPW]; }
#pragma unroll
                for (int j = 0; j < NR; ++j) r0[j] = rec[(R_::R0 + po + j) * GPW];
            }
            const double qe = fexp_ctl(accept ? lq11 - kc->beta2 * lqold : lq11) / kc->gamma;
            double q = ee_zero ? 1.0 / kc->qmax : fmax(1.0 / kc->qmax, fmin(1.0 / kc->qmin, qe));
            if (q >= kc->qsteady_min && q <= kc->qsteady_max) q = 1.0;
            const double dtq = dt / (accept ? q : fmin(1.0 / kc->qmin, qe));
            if (accept) {   // commit
                ++nacc;
#pragma unroll
                for (int i = 0; i < NS; ++i) { u[i] = unew[i]; f0[i] = f2[i]; nas[i] = nb[i]; }
#pragma unroll
                for (int qc = 0; qc < C; ++qc) gtr[qc] += gnew[qc];
                cur ^= 1;
                loss_sum = loss_new;
                jsave = jnew;
                t = tnew;
                lqold = ee_zero ? lqinit : fmax(lEE, lqinit);
                dt = fmin(dtq, dtmax);
                if (jsave >= nsave) rc = 0;
            } else {
                ++nrej;
                dt = dtq;
            }
            SENS_T(11);
            __builtin_amdgcn_wave_barrier();   // the record is rewritten by the next attempt
        }

        const double denom = (double)prm.n_obs * (double)jsave;
        const double inv_den = jsave > 0 ? 1.0 / denom : 0.0;
        if (chunk == 0 && nch == 1) {   // a launch of all chunks: loss and statistics are those of the plain solve that follows
            prm.loss[b] = loss_sum * inv_den;
            prm.retcode[b] = rc;
            prm.n_saved[b] = jsave;
            prm.n_accept[b] = nacc;
            prm.n_reject[b] = nrej;
        }
        double *grow = prm.gtraj + (size_t)traj * ncols + col0;
#pragma unroll
        for (int q_ = 0; q_ < C; ++q_)
            if (q_ < nvalid) grow[q_] = gtr[q_] * inv_den;
        }
        bi = (int64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)nx) + nwaves;
    }
#ifdef CRNN_SENS_PROF
    if (blockIdx.x == 0 && tid == 0)
        for (int k = 0; k < 12; ++k) g_sens_prof[k] = prof_acc[k];
#endif
}

}  // namespace crnn

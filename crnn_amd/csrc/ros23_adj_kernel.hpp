// crnn_amd/csrc/ros23_adj_kernel.hpp -- gfx950 (MI355X): loss gradient by the DISCRETE ADJOINT of the accepted
// Rosenbrock23 steps.
//
// ForwardDiff.gradient(x -> loss_neuralode(x, i_exp), p) (case2/case2.jl:195, rober_crnn.jl:219) differentiates the
// solver's arithmetic with the step sizes held fixed (dt, the accept/reject decisions and the saveat interpolation
// weights are plain Float64 in OrdinaryDiffEq).  ros23_kernel.hpp reproduces that by pushing P forward tangents through
// every accepted step -- (1 + P) x the primal work.  The same derivative is obtained here by running the accepted
// steps BACKWARDS once (reverse accumulation over the identical computational graph):
//
//   forward   u_{n+1} = u_n + dt k2,   k1 = W^-1 f(u_n),   k2 = W^-1 (f(u_n + dt/2 k1) - k1) + k1,   W = I - d dt J(u_n)
//             saveat:  u(t_n + Th dt) = u_n + dt (c1 k1 + c2 k2)
//   reverse   kb2 = dt lam + B2,  kb1 = B1,  ub = lam + A          (A, B1, B2: loss seeds of the save points in the step)
//             v = W^-T kb2;  kb1 += kb2 - v
//             at u_mid:   ub_mid = J^T v,         thb += d(v.f)/dtheta
//             ub += ub_mid;  kb1 += dt/2 ub_mid;  w = W^-T kb1
//             at u_n:     ub += J^T w + d/du [gam (v.J dk + w.J k1)],   thb += d/dtheta [w.f + gam (v.J dk + w.J k1)]
//             lam = ub
//
// so a trajectory + gradient costs about two primal solves, independent of the number of parameters, and the result
// equals the forward-tangent gradient up to rounding (tests/test_gpu_parity.py).  The gradient
// is produced in theta space (w_in | w_b | w_out); the chain rule through p2vec is a [P x n_theta] product applied to
// the batch sum (reduce_project_kernel).
//
// MI355X mapping: ONE LANE PER TRAJECTORY (no lane groups, no redundant primal, no LDS step record).  A wavefront takes
// 64 trajectories from the global queue, runs the forward sweep for all of them, then the reverse sweep.  The forward
// sweep appends (t_n, dt_n, u_n) of every accepted step to a per-lane tape in HBM (lane-contiguous: the partial lines
// of a wavefront's store merge in L2; a [wavefront][step][field][lane] layout with 512-byte stores measured slower);
// the reverse sweep re-forms J, W and the stages from the tape record instead of storing them.  The observed data are
// read once, by the reverse sweep, where loss and seeds are formed together.  theta sits in LDS (see below), the 42 (case2)
// gradient accumulators of a lane in LDS ([m][lane], ds_add_f64), the per-batch sums are formed in the kernel.
#pragma once
#include <type_traits>
#include "ros23_kernel.hpp"

// kernel-timing ablations (tools/kvariants.sh): 1 = no reverse sweep, 2 = no observed-data loads, 4 = no tape stores,
// 8 = no loss / seeds in the reverse sweep.  Round 2 (queued by step count, case2 65 536, 0.514 ms): forward sweep 57 % of an
// index-order launch, tape stores 8 %, observed rows 4.5 %, loss + seeds 17 % (robertson 9 %)
#ifndef CRNN_ADJ_DBG
#define CRNN_ADJ_DBG 0
#endif
// Settled experiments (their switches were deleted in round 5; measurements in DESIGN.md 3.1 / docs/HISTORY.md): the n_theta gradient accumulators
// of a lane live in LDS ([m][lane], conflict-free) and are updated with ds_add_f64 -- fire and forget; read + add + write was 2-8 % slower inside the
// kernels although cheaper in isolation (tools/ubench/lds_acc.hip), registers (84 VGPRs for case2) parked them in AGPRs --; the tape record holds
// (t, dt, u) only -- with the stages k1, k2 - k1 on it as well the reverse sweep skips two right-hand sides and two solves but the 2.3x tape traffic
// eats the saved arithmetic (case2 0.612 vs 0.618 ms, robertson 0.706 vs 0.704).

// phase timing (tools/kvariants.sh build prof="-DCRNN_ADJ_PROF=1"; the library then prints the shares of wave 0 of block 0 to
// stderr after every launch): s_memtime deltas per phase, with scheduling fences at the phase boundaries (the fenced
// kernel is ~30 % slower than the real one: the shares are a guide, the ablations above the measurement)
#ifdef CRNN_ADJ_PROF
#define ADJ_T(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_readcyclecounter(); prof_acc[k] += now_ - prof_last; prof_last = now_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define ADJ_T(k) do { } while (0)
#endif

namespace crnn {

// Bounds-checked build (-DCRNN_BOUNDS_CHECK; tools/crossbuild.py variant "O3chk"): every indexed access of the adjoint kernels
// -- tape records, save times in LDS, observed rows, u0 / pred / per-trajectory outputs, the queue permutation, the batch
// partial rows -- is checked against its extent on the lanes that perform it; violations are counted in g_bounds[0], the site
// code of the first one is kept in g_bounds[1] (crnn_debug_bounds reads the pair).  Release builds compile the checks away.
// (g_bounds / CRNN_CHK live in ros23_kernel.hpp since round 4: the HyChem, cathode and forward-tangent kernels carry checks too --
//  site codes 1-19 adjoint kernels, 20-39 HyChem, 40-59 cathode, 60-69 forward tangents)

struct AdjParams {
    double *tape;                // [lanes][tape_cap][NS + 2]
    int32_t tape_cap;            // accepted steps a lane can record
    unsigned int *overflow;      // incremented by every trajectory that ran out of tape (host then falls back)
    int32_t align_rev;           // ros23_adj_kernel: reverse sweep aligned by step index (see the kernel)
    double *batch_partials;      // [ceil(count/64)][NTH + kExtra]: per 64-trajectory batch sums
    const int32_t *perm;         // ros23_adj_kernel: position in the queue -> trajectory (relative to first); null = identity
    unsigned long long *prof;    // CRNN_ADJ_PROF builds: 16 phase totals in s_memtime ticks
};

// Solve A^T x = b with the factors of lu_factor (P A = L U): x = P^T L^-T U^-T b
template <int NS>
__device__ __forceinline__ void lu_solve_T(const double (&A)[NS][NS], const double (&dinv)[NS], const int (&piv)[NS],
                                           const bool wave_pivots, double (&b)[NS]) {
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        b[k] *= dinv[k];
        const double a = b[k];
#pragma unroll
        for (int i = k + 1; i < NS; ++i) b[i] = fma(-A[k][i], a, b[i]);
    }
#pragma unroll
    for (int k = NS - 1; k >= 0; --k) {
        const double a = b[k];
#pragma unroll
        for (int i = 0; i < k; ++i) b[i] = fma(-A[k][i], a, b[i]);
    }
    if (wave_pivots) {
#pragma unroll
        for (int k = NS - 1; k >= 0; --k) {
            const int p = piv[k];
#pragma unroll
            for (int i = k + 1; i < NS; ++i) {
                const bool sw = (p == i);
                const double bk = b[k], bi = b[i];
                b[k] = sw ? bi : bk;
                b[i] = sw ? bk : bi;
            }
        }
    }
}

// an integer zero the optimiser cannot see through (no instruction): `ptr + opaque_zero()` re-derives an LDS pointer so that
// loads through it are neither hoisted out of loops nor merged with earlier ones
__device__ __forceinline__ int opaque_zero() {
    int z = 0;
    asm volatile("" : "+v"(z));
    return z;
}

__device__ __forceinline__ int opaque_zero_s() {   // the same in a scalar register: the derived pointer stays wave-uniform
    int z = 0;
    asm volatile("" : "+s"(z));
    return z;
}

// b <- W^-T b for the two W representations of ros23_kernel.hpp
template <int NS, int NR, bool HAS_T, bool USE_SCALE>
__device__ __forceinline__ void solve_T(const DenseLU<NS, NR, HAS_T, USE_SCALE> &W, const double *__restrict__,
                                        const double (&)[NS], const double (&)[NR], const double *, double (&b)[NS]) {
    lu_solve_T<NS>(W.A, W.dinv, W.piv, W.wave_pivots, b);
}
// W^-1 = I + D_sc Wo D_gr M^-1 Wi^T D_g   =>   W^-T = I + D_g Wi M^-T D_gr Wo^T D_sc
template <int NS, int NR, bool HAS_T, bool USE_SCALE>
__device__ __forceinline__ void solve_T(const Woodbury<NS, NR, HAS_T, USE_SCALE> &W, const double *__restrict__ th,
                                        const double (&g)[NS], const double (&gr)[NR], const double *sc_s, double (&b)[NS]) {
    using L_ = Lay<NS, NR, HAS_T>;
    double y[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) y[j] = 0.0;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const double t = USE_SCALE ? b[i] * sc_s[i] : b[i];
#pragma unroll
        for (int j = 0; j < NR; ++j) y[j] = fma(th[L_::wo(i, j)], t, y[j]);
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) y[j] *= gr[j];
    lu_solve_T<NR>(W.M, W.dinv, W.piv, W.wave_pivots, y);
#pragma unroll
    for (int c = 0; c < NS; ++c) {
        double a = 0.0;
#pragma unroll
        for (int j = 0; j < NR; ++j) a = fma(th[L_::wi(c, j)], y[j], a);
        b[c] = fma(a, g[c], b[c]);
    }
}

// PRIMAL = true: the forward sweep alone -- solution at the save points (prm.pred), loss accumulated there, no tape, no reverse
// sweep.  This is the primal path of crnn_solve (predict_neuralode, the epoch-end loss loop, case2.jl:124-128 / 199-203): the
// same wave-synchronous batches and longest-first queue as the gradient launch, where ros23_kernel's one-lane-group-per-trajectory
// scheme (built for tangent columns) lets the 64 lanes of a wavefront drift apart -- 0.55 ms per 65 536 against 0.50 for the
// whole gradient.
// JFD = true (PRIMAL only): Rosenbrock23(autodiff = false) -- W from forward differences of the right-hand side
// (DenseLU::factor_fd, ros23_kernel.hpp; crnn_ctx_set_jacobian): NS more right-hand sides per attempt, a dense NS x NS factorisation.
template <int NS, int NR, bool HAS_T, bool USE_SCALE, int BLOCK, bool PRIMAL = false, bool JFD = false>
__global__ __launch_bounds__(BLOCK) void ros23_adj_kernel(const SolveParams prm, const double *__restrict__ theta,
                                                          const AdjParams adj) {
    using L_ = Lay<NS, NR, HAS_T>;
    constexpr int N = L_::N;
    constexpr int NTH = L_::NTH;
    constexpr int RECW = NS + 2;
    static_assert(NTH + kExtra <= 64, "the per-batch sums use one lane per column");
    static_assert(!JFD || PRIMAL, "the finite-difference W exists for primal launches");
    using Solver = typename SolverSel<(NR < NS) && !JFD, NS, NR, HAS_T, USE_SCALE>::type;

    __shared__ double kc_lds[kNConst];
    __shared__ double ts_lds[kMaxSave + 4];                       // four +inf slots behind the last save time: gradient launches count save points four at a time
    __shared__ double thb_lds[NTH * BLOCK];     // gradient accumulators [m][lane] (also the staging area of the batch sums)
    __shared__ double ex_lds[kExtra * BLOCK];
    const int tid = threadIdx.x;
    for (int idx = tid; idx < kNConst; idx += BLOCK) kc_lds[idx] = reinterpret_cast<const double *>(prm.kc)[idx];
    for (int idx = tid; idx < prm.n_save; idx += BLOCK) ts_lds[idx] = prm.tsave[idx];
    if (tid < 4) ts_lds[prm.n_save + tid] = INFINITY;
    const KConst *kc = reinterpret_cast<const KConst *>(kc_lds);
    // theta is read from LDS (broadcast ds_read), not through scalar loads: with 42-43 parameters live across the forward
    // AND the reverse sweep the SGPR file overflows (247 SGPR spills -> v_readlane/v_writelane churn, s_load + s_waitcnt
    // in the adjoint contraction).  Measured: case2 -1.5 %, robertson -10 %, B = 131 072 -4 % (identical results).  The
    // forward-tangent kernels are the other way round (theta * C columns: +35 % with LDS theta) and keep the scalar path.
    // The compiler loads theta once and parks it in AGPRs (two v_accvgpr_read per use); forcing a fresh LDS read per
    // phase instead (pointer laundered through an empty asm) removes 10 % of the VALU instructions and is SLOWER (case2
    // +4 %, robertson +15 %): at one wavefront per SIMD the exposed lgkmcnt waits cost more than the issue slots saved.
    // (Round 3 also measured wave-uniform scalar loads re-issued at the top of every attempt / reverse step -- what auto_adj_kernel.hpp ships -- here:
    // slower for this kernel; its switch was deleted in round 5.)
    __shared__ double th_lds[NTH];
    for (int idx = tid; idx < NTH; idx += BLOCK) th_lds[idx] = theta[idx];
    __syncthreads();
    const double *th = th_lds;
#define CRNN_ADJ_TH_FRESH() th_outer
    const double *const th_outer = th;
    double *const thb_s = thb_lds + tid;   // accumulator m of this lane: thb_s[m * BLOCK]

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

    const int lane = tid & 63;
#ifdef CRNN_ADJ_PROF
    unsigned long long prof_acc[16] = {0}, prof_last = __builtin_readcyclecounter();
#endif
    double *const tape = adj.tape + (size_t)((size_t)blockIdx.x * BLOCK + tid) * adj.tape_cap * RECW;

    while (true) {
        // ---- next 64 trajectories for this wavefront
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(prm.queue, 64ULL);
        const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)base);
        const unsigned bhi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
        const int64_t wave_base = (int64_t)(((unsigned long long)bhi << 32) | blo);
        if (wave_base >= prm.count) break;
        const int64_t traj = wave_base + lane;
        const bool valid = traj < prm.count;
        // Ensembles larger than the resident lanes are queued in the order of their last known step counts, longest
        // first (sort_steps_kernel): the 64 trajectories of a batch then take nearly the same number of steps, and a
        // wavefront is busy for its batch's mean rather than for its slowest member.
        const int64_t b = prm.first + (valid ? (adj.perm ? (int64_t)adj.perm[traj] : traj) : 0);
        CRNN_CHK(b >= 0 && b < prm.B && traj >= 0, 1);

        // ================================================================== forward sweep
        double u[NS], f0[NS], g0[NS], r0[NR], bT[NR];
        double xT = 0.0, Tconst = 0.0;
        double t = t0, dt = 0.0, lqold = lqinit;
        int iter = 0, jsave = 0, nacc = 0, nrej = 0;
        int rc = valid ? -1 : 0;
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
            features<NS>(u, kc->lb, kc->ub, x0, g0);
            rates<NS, NR, HAS_T>(th, x0, bT, r0);
            rhs_from_rates<NS, NR, HAS_T, USE_SCALE>(th, r0, kc->scale, f0);
            // Hairer initial step (OrdinaryDiffEq ode_determine_initdt, order 2)
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
            features<NS>(u1, kc->lb, kc->ub, x1, g1);
            rates<NS, NR, HAS_T>(th, x1, bT, r1);
            rhs_from_rates<NS, NR, HAS_T, USE_SCALE>(th, r1, kc->scale, f1);
            double d2 = 0.0;
#pragma unroll
            for (int i = 0; i < NS; ++i) { double e = (f1[i] - f0[i]) * sk[i]; d2 = fma(e, e, d2); }
            d2 = sqrt(d2 * (1.0 / N)) / dt0;
            double dm = fmax(d1, d2);
            double dt1 = (dm <= 1e-15) ? fmax(1e-6, dt0 * 1e-3) : exp(-0.5 * (4.605170185988091368 + flog(dm)));
            dt = fmax(kc->dtmin, fmin(fmin(100.0 * dt0, dt1), dtmax));
        }
        if (start_saved) {  // save_start: saveat contains tspan[1]
            if (valid && prm.pred) {
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    double v = u[i];
                    if (prm.clamp_pred) v = clampv(v, -kc->ub, kc->ub);
                    prm.pred[((size_t)0 * N + i) * prm.B + b] = v;
                }
                if (HAS_T) {
                    double v = Tconst;
                    if (prm.clamp_pred) v = clampv(v, -kc->ub, kc->ub);
                    prm.pred[((size_t)0 * N + NS) * prm.B + b] = v;
                }
            }
            jsave = 1;
        }

        // PRIMAL: the loss is accumulated at the save points of the forward sweep (ascending, as ros23_kernel does); the observed row
        // of the next save point is requested one save point ahead
        double pf_loss = 0.0, pf_row[NS];
        int pf_off[NS];
        bool pf_obs[NS];
        const double *const pf_rows = prm.data + (size_t)b * prm.row_stride;
        if (PRIMAL) {
#pragma unroll
            for (int i = 0; i < NS; ++i) { const int dr = (int)kc->drow[i]; pf_obs[i] = dr >= 0; pf_off[i] = dr >= 0 ? dr : 0; }
            const double *row = pf_rows + (size_t)(jsave < nsave ? jsave : 0) * prm.n_obs;
#pragma unroll
            for (int i = 0; i < NS; ++i) pf_row[i] = row[pf_off[i]];
        }
        while (__builtin_amdgcn_ballot_w64(rc < 0) != 0) {
            if (rc < 0) {
                ++iter;
                bool last = false;
                if (jsave >= nsave) rc = 0;            // horizon = tspan[1]: nothing to integrate
                else if (iter > prm.maxiters) rc = 1;
                if (t + dt * (1.0 + 1e-13) >= tend) { dt = tend - t; last = true; }
                if (rc < 0 && (!(dt > kc->dtmin) || t + dt == t)) rc = 2;
                if (rc < 0) {
                    ADJ_T(0);   // loop control, step-size bookkeeping
                    const double *th = CRNN_ADJ_TH_FRESH();
                    Solver W;
                    const double gam = d_ * dt;
                    double gr0[NR];
#pragma unroll
                    for (int j = 0; j < NR; ++j) gr0[j] = gam * r0[j];
                    double k1[NS], dk[NS], unew[NS], f1[NS], f2[NS], g2[NS], r2[NR];
                    bool okf;
                    if constexpr (JFD) {
                        okf = W.factor_fd(u, f0, gam, [&](const double (&up)[NS], double (&fp)[NS]) {
                            double xp[NS], gp[NS], rp[NR];
                            features<NS>(up, kc->lb, kc->ub, xp, gp);
                            rates<NS, NR, HAS_T>(th, xp, bT, rp);
                            rhs_from_rates<NS, NR, HAS_T, USE_SCALE>(th, rp, kc->scale, fp);
                        });
                    } else okf = W.factor(th, g0, r0, gam, kc->scale);
#pragma unroll
                    for (int i = 0; i < NS; ++i) k1[i] = f0[i];
                    W.solve(th, g0, gr0, kc->scale, k1);
                    ADJ_T(1);   // factor + first solve
                    {
                        double u1[NS], x1[NS], g1[NS], r1[NR];
#pragma unroll
                        for (int i = 0; i < NS; ++i) u1[i] = fma(0.5 * dt, k1[i], u[i]);
                        features<NS>(u1, kc->lb, kc->ub, x1, g1);
                        rates<NS, NR, HAS_T>(th, x1, bT, r1);
                        rhs_from_rates<NS, NR, HAS_T, USE_SCALE>(th, r1, kc->scale, f1);
                    }
                    ADJ_T(2);   // evaluation at u_mid
#pragma unroll
                    for (int i = 0; i < NS; ++i) dk[i] = f1[i] - k1[i];
                    W.solve(th, g0, gr0, kc->scale, dk);
#pragma unroll
                    for (int i = 0; i < NS; ++i) unew[i] = fma(dt, k1[i] + dk[i], u[i]);
                    ADJ_T(3);   // second solve
                    {
                        double x2[NS];
                        features<NS>(unew, kc->lb, kc->ub, x2, g2);
                        rates<NS, NR, HAS_T>(th, x2, bT, r2);
                        rhs_from_rates<NS, NR, HAS_T, USE_SCALE>(th, r2, kc->scale, f2);
                    }
                    ADJ_T(4);   // evaluation at u_new
                    double k3[NS];
#pragma unroll
                    for (int i = 0; i < NS; ++i) {
                        double k2i = k1[i] + dk[i];
                        k3[i] = f2[i] - c32 * (k2i - f1[i]) - 2.0 * (k1[i] - f0[i]);
                    }
                    W.solve(th, g0, gr0, kc->scale, k3);
                    ADJ_T(5);   // third solve
                    double es = 0.0;
                    bool finite = okf;
#pragma unroll
                    for (int i = 0; i < NS; ++i) {
                        double k2i = k1[i] + dk[i];
                        double ev = dt * (1.0 / 6.0) * (k1[i] - 2.0 * k2i + k3[i]);
                        double m = fmax(fabs(u[i]), fabs(unew[i]));
                        double e = ev * frcp1(fma(kc->rtol[i], m, kc->atol[i]));
                        es = fma(e, e, es);
                        finite = finite && isfinite(unew[i]) && isfinite(ev);
                    }
                    es = es * (1.0 / N);
                    ADJ_T(6);   // error norm
                    if (!finite) rc = 3;
                    else {
                        // PI controller (OrdinaryDiffEq PIController), in log space
                        const bool ee_zero = (es == 0.0);
                        const double lEE = 0.5 * flog(ee_zero ? 1.0 : es);
                        const double lq11 = kc->beta1 * lEE;
                        double q = ee_zero ? 1.0 / kc->qmax
                                           : fmax(1.0 / kc->qmax, fmin(1.0 / kc->qmin, exp(lq11 - kc->beta2 * lqold) / kc->gamma));
                        ADJ_T(7);   // controller (log, exp, divisions)
                        if (es <= 1.0) {
                            if (!PRIMAL && nacc >= adj.tape_cap) {
                                rc = 5;  // out of tape: the host re-runs the call with forward tangents
                                atomicAdd(adj.overflow, 1u);
                            } else {
                                if (!PRIMAL) {
                                    CRNN_CHK(nacc >= 0 && nacc < adj.tape_cap, 5);
                                    double *rec = tape + (size_t)((CRNN_ADJ_DBG & 4) ? 0 : nacc) * RECW;
                                    rec[0] = t;
                                    rec[1] = dt;
#pragma unroll
                                    for (int i = 0; i < NS; ++i) rec[2 + i] = u[i];
                                }
                                ++nacc;
                                const double tnew = last ? tend : t + dt;
                                if (!PRIMAL && !prm.pred) {
                                    // a gradient launch only COUNTS the save points inside the step (the reverse sweep evaluates them): four save
                                    // times per LDS round trip instead of a dependent read, a compare and a branch per point (ascending times,
                                    // +inf behind the last one)
                                    while (true) {
                                        CRNN_CHK(jsave >= 0 && jsave <= nsave, 6);
                                        const double a0 = ts_lds[jsave], a1 = ts_lds[jsave + 1], a2 = ts_lds[jsave + 2], a3 = ts_lds[jsave + 3];
                                        const int c = (a0 <= tnew ? 1 : 0) + (a1 <= tnew ? 1 : 0) + (a2 <= tnew ? 1 : 0) + (a3 <= tnew ? 1 : 0);
                                        jsave += c;
                                        if (c < 4) break;
                                    }
                                } else
                                while (jsave < nsave) {
                                    CRNN_CHK(jsave >= 0 && jsave < nsave, 6);
                                    const double ts = ts_lds[jsave];
                                    if (!(ts <= tnew)) break;
                                    if (prm.pred || PRIMAL) {
                                        const bool at_end = (ts == tnew);
                                        const double Th = at_end ? 1.0 : (ts - t) / dt;
                                        const double c1 = at_end ? 0.0 : Th * (1.0 - Th) * inv12d;
                                        const double c2 = at_end ? 1.0 : Th * (Th - 2.0 * d_) * inv12d;
                                        double drow_[NS];
                                        if (PRIMAL) {   // this save point's observations; the next one's are requested below
#pragma unroll
                                            for (int i = 0; i < NS; ++i) drow_[i] = pf_row[i];
                                            const double *nrow = pf_rows + (size_t)(jsave + 1 < nsave ? jsave + 1 : jsave) * prm.n_obs;
#pragma unroll
                                            for (int i = 0; i < NS; ++i) pf_row[i] = nrow[pf_off[i]];
                                        }
#pragma unroll
                                        for (int i = 0; i < NS; ++i) {
                                            double k2i = k1[i] + dk[i];
                                            double v = at_end ? unew[i] : fma(dt, fma(c1, k1[i], c2 * k2i), u[i]);
                                            if (prm.clamp_pred) v = clampv(v, -kc->ub, kc->ub);
                                            if (prm.pred) prm.pred[((size_t)jsave * N + i) * prm.B + b] = v;
                                            if (PRIMAL && pf_obs[i]) {
                                                const double rr = (drow_[i] - v) * kc->inv_yscale[i];
                                                pf_loss = (prm.loss_kind == 0) ? pf_loss + fabs(rr) : fma(rr, rr, pf_loss);
                                            }
                                        }
                                        if (HAS_T && prm.pred) {
                                            double v = Tconst;
                                            if (prm.clamp_pred) v = clampv(v, -kc->ub, kc->ub);
                                            prm.pred[((size_t)jsave * N + NS) * prm.B + b] = v;
                                        }
                                    }
                                    ++jsave;
                                }
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
                                ADJ_T(8);   // tape record, save-point loop, FSAL copy
                            }
                        } else {
                            ++nrej;
                            dt = dt / fmin(1.0 / kc->qmin, exp(lq11) / kc->gamma);
                        }
                    }
                }
            }
        }

        // ================================================================== reverse sweep
        const int n_saved = jsave;
        const int jlo = start_saved ? 1 : 0;
#pragma unroll
        for (int m = 0; m < NTH; ++m) thb_s[m * BLOCK] = 0.0;
#define THB_ADD(m, val) unsafeAtomicAdd(&thb_s[(m) * BLOCK], (val))
#define THB_REG(m) 0.0
        double lam[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) lam[i] = 0.0;
        // d/d w_b[j] summed over the steps of THIS trajectory, in registers: T is a constant of the trajectory, so the
        // temperature row of w_in is xT times the same sum -- 2 NR fewer LDS atomics per step (time-neutral, measured).
        double wbb[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) wbb[j] = 0.0;
        double loss_sum = PRIMAL ? pf_loss : 0.0;
        double tnew = t;             // end time of the step being reversed
        int s = (!PRIMAL && valid && !(CRNN_ADJ_DBG & 1)) ? nacc - 1 : -1;

        // Observed rows are fetched at the top of a reverse step, a whole step re-formation (~2 us of arithmetic) ahead
        // of their use: rows jsave-1, jsave-2, jsave-3 cover the save points one step usually spans.
        const double *const drows = prm.data + (size_t)b * prm.row_stride;
        int doff[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) { const int dr = (int)kc->drow[i]; doff[i] = dr >= 0 ? dr : 0; }
        auto load_row = [&](int j, double (&d)[NS]) {
            CRNN_CHK((int64_t)(j > 0 ? j : 0) * prm.n_obs < prm.row_stride, 2);
            const double *row = drows + (size_t)(j > 0 ? j : 0) * prm.n_obs;
#pragma unroll
            for (int i = 0; i < NS; ++i) d[i] = (CRNN_ADJ_DBG & 2) ? 0.5 : row[doff[i]];
        };
        double ts_cur = (jsave - 1 >= jlo) ? ts_lds[jsave - 1] : -INFINITY;   // save point jsave-1 (none: -inf, never inside a step)
        // (the one after it too where registers allow: with ns >= 5 the allocator spills that second value to scratch and reloads
        //  it three times per step -- a memory latency each for a wavefront alone on its SIMD: case2 at 65 536 0.512 -> 0.493 ms
        //  without it; robertson's shape keeps it, +1 %)
        constexpr bool TS_PF = NS < 5;
        double ts_nxt = (TS_PF && jsave - 2 >= jlo) ? ts_lds[jsave - 2] : -INFINITY;
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

        // Which lanes reverse together.  All lanes start with their LAST step (lag 0: the lanes of an iteration are at about
        // the same physical time) or, align_rev, a trajectory with fewer steps than the longest of its wavefront starts
        // later, so that the lanes of an iteration are at the same step INDEX.  The wavefront runs max(n_accept) iterations
        // either way; what differs is how many save points the lanes of one iteration have inside their steps -- every
        // iteration pays for the maximum over its lanes.  Steps grow geometrically behind a stiff transient: on a uniform
        // save grid the count follows the physical time (case2: lag 0 is 24 % faster), on a geometric grid the step index
        // (robertson: aligned is 7 % faster).  The host picks by the shape of the grid (crnn_capi.hip).
        int lag = 0;
        if (adj.align_rev) {
            int smax = s;
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) smax = max(smax, __shfl_xor(smax, m));
            lag = smax - s;
        }
        int it_rev = 0;

        while (__builtin_amdgcn_ballot_w64(s >= 0) != 0) {
            const bool go = (it_rev >= lag);
            ++it_rev;
            if (s >= 0 && go) {
                const double tn = rt, h = rdt;
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
                // ---- re-form the step
                const double *th = CRNN_ADJ_TH_FRESH();
                double x0[NS], gg0[NS], rr0[NR];
                Solver W;
                const double gam = d_ * h;
                double gr0[NR], x1[NS], g1[NS], r1[NR];
                double ff0[NS];
                features<NS>(un, kc->lb, kc->ub, x0, gg0);
                rates<NS, NR, HAS_T>(th, x0, bT, rr0);
                rhs_from_rates<NS, NR, HAS_T, USE_SCALE>(th, rr0, kc->scale, ff0);
#pragma unroll
                for (int j = 0; j < NR; ++j) gr0[j] = gam * rr0[j];
                (void)W.factor(th, gg0, rr0, gam, kc->scale);
                double k1[NS], dk[NS];
#pragma unroll
                for (int i = 0; i < NS; ++i) k1[i] = ff0[i];
                W.solve(th, gg0, gr0, kc->scale, k1);
                {
                    double u1[NS], f1[NS];
#pragma unroll
                    for (int i = 0; i < NS; ++i) u1[i] = fma(0.5 * h, k1[i], un[i]);
                    features<NS>(u1, kc->lb, kc->ub, x1, g1);
                    rates<NS, NR, HAS_T>(th, x1, bT, r1);
                    rhs_from_rates<NS, NR, HAS_T, USE_SCALE>(th, r1, kc->scale, f1);
#pragma unroll
                    for (int i = 0; i < NS; ++i) dk[i] = f1[i] - k1[i];
                }
                W.solve(th, gg0, gr0, kc->scale, dk);

                ADJ_T(9);    // reverse: prefetches + re-formation of the step
                // ---- loss and its seeds at the save points inside (tn, tnew]
                // Straight-line per point (round 5, as ros23_adj2_kernel.hpp): the loss kind is one wave-uniform branch around the phase, no
                // clamp is an infinite clamp, the end-of-step point needs no select on v (c1 = 0, c2 = 1 give h k2 + u_n bit for bit), the
                // mask is "the clamp changed nothing" -- the values of the branchy form it replaces.
                double A_[NS], B1[NS], B2[NS];
#pragma unroll
                for (int i = 0; i < NS; ++i) { A_[i] = 0.0; B1[i] = 0.0; B2[i] = 0.0; }
                // the time of the next save point (backwards) sits in a register -- and, TS_PF, the one after it: the test "is it
                // inside this step" and the seed itself do not wait for LDS (the same in the forward sweep's save-point loop gains
                // nothing: two more registers live across that loop)
                const double inv_h = frcp(h);      // one reciprocal per step instead of an IEEE division per save point (1e-16)
                const double ubc = prm.clamp_pred ? kc->ub : __builtin_inf();
                auto in_step = [&]() -> bool { return ts_cur > tn; };
                auto seed_point = [&](const double (&dobs)[NS], auto lk_) {
                    constexpr bool LK0 = decltype(lk_)::value;
                    const double ts = ts_cur;
                    CRNN_CHK(jsave - 1 >= jlo && jsave - 1 < nsave, 8);
                    if (TS_PF) {
                        ts_cur = ts_nxt;
                        ts_nxt = (jsave - 3 >= jlo) ? ts_lds[jsave - 3] : -INFINITY;
                    } else {
                        ts_cur = (jsave - 2 >= jlo) ? ts_lds[jsave - 2] : -INFINITY;
                    }
                    const bool at_end = (ts == tnew);
                    const double Th = at_end ? 1.0 : (ts - tn) * inv_h;
                    const double c1 = Th * (1.0 - Th) * inv12d;        // at the end of the step: 1 * 0 * inv12d = 0 exactly
                    const double c2 = at_end ? 1.0 : Th * (Th - 2.0 * d_) * inv12d;
                    const double hc1 = h * c1, hc2 = h * c2;
#pragma unroll
                    for (int i = 0; i < NS; ++i) {
                        const int dr = (int)kc->drow[i];
                        if (dr >= 0) {
                            const double v = fma(h, fma(c1, k1[i], c2 * (k1[i] + dk[i])), un[i]);
                            const double vc = fmin(fmax(v, -ubc), ubc);      // v is finite here (accepted steps only): = clampv
                            const double iy = kc->inv_yscale[i];
                            const double rr = (dobs[i] - vc) * iy;
                            double w;
                            if constexpr (LK0) { loss_sum += fabs(rr); w = signbit(rr) ? iy : -iy; }
                            else { loss_sum = fma(rr, rr, loss_sum); w = (-2.0 * rr) * iy; }
                            w = (vc == v) ? w : 0.0;
                            A_[i] += w;
                            B1[i] = fma(w, hc1, B1[i]);
                            B2[i] = fma(w, hc2, B2[i]);
                        }
                    }
                    --jsave;
                };
                auto seeds = [&](auto lk_) {
                    if (in_step()) {
                        seed_point(dA, lk_);
                        if (in_step()) {
                            seed_point(dB, lk_);
                            if (in_step()) {
                                seed_point(dC, lk_);
                                while (in_step()) {  // more than three save points inside one step: fetch on demand
                                    double dD[NS];
                                    load_row(jsave - 1, dD);
                                    seed_point(dD, lk_);
                                }
                            }
                        }
                    }
                };
                if (!(CRNN_ADJ_DBG & 8)) {   // ablation 8: no loss / seeds at all (timing only)
                    if (prm.loss_kind == 0) seeds(std::true_type{}); else seeds(std::false_type{});
                }

                ADJ_T(10);   // reverse: loss + seeds
                // ---- adjoint of the step
                double kb1[NS], v[NS], ub[NS];
#pragma unroll
                for (int i = 0; i < NS; ++i) { v[i] = fma(h, lam[i], B2[i]); ub[i] = lam[i] + A_[i]; }
#pragma unroll
                for (int i = 0; i < NS; ++i) kb1[i] = B1[i] + v[i];
                solve_T<NS, NR, HAS_T, USE_SCALE>(W, th, gg0, gr0, kc->scale, v);        // v = W^-T kb2
#pragma unroll
                for (int i = 0; i < NS; ++i) kb1[i] -= v[i];
                double vs[NS];   // sc .* v
#pragma unroll
                for (int i = 0; i < NS; ++i) vs[i] = USE_SCALE ? v[i] * kc->scale[i] : v[i];
                double av[NR];
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    double a = 0.0;
#pragma unroll
                    for (int i = 0; i < NS; ++i) a = fma(vs[i], th[L_::wo(i, j)], a);
                    av[j] = a;
                }
                // point u_mid: d(v.f)/d(u, theta)
                // The theta terms of the u_mid point (rho_j, rho_j x1_c, vs_i r1_j) are not added on their own: they are folded
                // into the u_n point's addends below -- one ds_add_f64 per accumulator and step instead of two.  An LDS
                // atomic costs ~40 cycles of a wavefront's time here; halving their number: case2 -6 %, robertson -7.5 %.
                double rho1[NR];   // av_j r_j(u_mid)
                {
                    double um[NS];
#pragma unroll
                    for (int c = 0; c < NS; ++c) um[c] = 0.0;
#pragma unroll
                    for (int j = 0; j < NR; ++j) {
                        const double rho = av[j] * r1[j];
                        rho1[j] = rho;
#pragma unroll
                        for (int c = 0; c < NS; ++c) um[c] = fma(rho, th[L_::wi(c, j)], um[c]);
                    }
#pragma unroll
                    for (int c = 0; c < NS; ++c) {
                        const double m = um[c] * g1[c];
                        ub[c] += m;
                        kb1[c] = fma(0.5 * h, m, kb1[c]);
                    }
                }
                solve_T<NS, NR, HAS_T, USE_SCALE>(W, th, gg0, gr0, kc->scale, kb1);      // kb1 = w = W^-T kb1
                // point u_n: d/d(u, theta) [ w.f + gam (v.J dk + w.J k1) ]
                {
                    double ws[NS];
#pragma unroll
                    for (int i = 0; i < NS; ++i) ws[i] = USE_SCALE ? kb1[i] * kc->scale[i] : kb1[i];
                    double s1[NS], s2[NS];   // sum_j beta_j w_in[c,j];  sum_j w_in[c,j] (pv_j dk_c + gam pw_j k1_c)
#pragma unroll
                    for (int c = 0; c < NS; ++c) { s1[c] = 0.0; s2[c] = 0.0; }
#pragma unroll
                    for (int j = 0; j < NR; ++j) {
                        double aw = 0.0, q1 = 0.0, qd = 0.0;
#pragma unroll
                        for (int i = 0; i < NS; ++i) aw = fma(ws[i], th[L_::wo(i, j)], aw);
#pragma unroll
                        for (int c = 0; c < NS; ++c) {
                            const double wg = th[L_::wi(c, j)] * gg0[c];
                            q1 = fma(wg, k1[c], q1);
                            qd = fma(wg, dk[c], qd);
                        }
                        const double c1j = fma(gam, q1, 1.0), czd = gam * qd;
                        const double pv = av[j] * gr0[j];        // gam a^v_j r_j
                        const double pw = aw * rr0[j];           // a^w_j r_j
                        const double gpw = gam * pw;
                        const double beta = fma(pw, c1j, pv * qd);
                        wbb[j] += beta + rho1[j];
#pragma unroll
                        for (int c = 0; c < NS; ++c) {
                            const double m = fma(pv, dk[c], gpw * k1[c]);
                            THB_ADD(L_::wi(c, j), fma(rho1[j], x1[c], fma(beta, x0[c], gg0[c] * m)));
                            const double wi = th[L_::wi(c, j)];
                            s1[c] = fma(beta, wi, s1[c]);
                            s2[c] = fma(wi, m, s2[c]);
                        }
                        const double ca = fma(rr0[j], czd, r1[j]), cb = rr0[j] * c1j;   // vs_i (r1_j + r0_j czd) + ws_i r0_j c1j
#pragma unroll
                        for (int i = 0; i < NS; ++i) THB_ADD(L_::wo(i, j), fma(vs[i], ca, ws[i] * cb));
                    }
#pragma unroll
                    for (int c = 0; c < NS; ++c) {
                        const double g = gg0[c];
                        lam[c] = ub[c] + g * (s1[c] - g * s2[c]);   // g' = -g^2 inside the window, 0 outside
                    }
                }
                tnew = tn;
                --s;
                ADJ_T(11);   // reverse: adjoint of the step
            }
        }
        ADJ_T(12);

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
        // ---- sums over the 64 trajectories of this batch, through LDS: every lane parks its scaled accumulators and its
        //      five scalars, then lane m of the wavefront adds up row m over the 64 lanes in lane order.  The result depends
        //      on the batch only (not on which wavefront processed it), so the gradient stays bitwise reproducible.
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
            const int w0 = tid & ~63;   // first lane of this wavefront within the block
            if (lane < NTH + kExtra) {
                const double *src = lane < NTH ? thb_lds + lane * BLOCK + w0 : ex_lds + (lane - NTH) * BLOCK + w0;
                double a = 0.0;
                for (int k = 0; k < 64; ++k) a += src[k];
                prow[lane] = a;
            }
            __builtin_amdgcn_wave_barrier();
        }
#undef THB_ADD
#undef THB_REG
        ADJ_T(13);   // outputs + batch sums
    }
#ifdef CRNN_ADJ_PROF
    if (adj.prof && blockIdx.x == 0 && tid == 0)
        for (int k = 0; k < 16; ++k) adj.prof[k] = prof_acc[k];
#endif
}

// Queue order for ensembles larger than the resident lanes.  Homogeneous 64-trajectory batches are all that is needed, not
// a global order: every block sorts its own run of 1024 consecutive trajectories by the step count (accepted + rejected)
// of their previous launch, longest first, ties by index (the sort key carries the index, so the order -- and with it the
// composition of every batch and the rounding of the batch sums -- is a deterministic function of the previous launch),
// the runs' batches are then interleaved (see the end of the kernel).  Bitonic sort of 1024 keys in LDS, ~10 us for any
// ensemble size (one block per run).
__device__ __forceinline__ void sort_steps_run(const int32_t *__restrict__ n_accept, const int32_t *__restrict__ n_reject,
                                               int64_t first, int count, int32_t *__restrict__ perm, const int r, const int NR,
                                               unsigned *key) {   // run r of NR; key: 1024 words of LDS
    const int tid = threadIdx.x;
    const int base = r * 1024;
    const int e = base + tid;
    unsigned k = 0xFFFFFFFFu;                                  // beyond the ensemble: sorts to the end of the run
    if (e < count) {
        const int steps = n_accept[first + e] + n_reject[first + e];
        k = ((unsigned)(1023 - min(max(steps, 0), 1023)) << 10) | (unsigned)tid;     // descending in steps, ascending in index
    }
    key[tid] = k;
    __syncthreads();
    for (int size = 2; size <= 1024; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const int partner = tid ^ stride;
            if (partner > tid) {
                const unsigned a = key[tid], b = key[partner];
                const bool up = (tid & size) == 0;             // ascending sub-sequence
                if ((a > b) == up) { key[tid] = b; key[partner] = a; }
            }
            __syncthreads();
        }
    }
    // Queue position of the k-th trajectory of run r: batches (64 consecutive ranks) of all runs interleaved, batch-major --
    // the longest batch of every run first, then every run's second longest ... -- so the queue is approximately
    // longest-first as a whole (with a handful of batches per wavefront the tail of the launch matters).  Only the last run
    // can be short; its last batch can be partial, the positions behind it close the gap.
    const int nv_last = count - (NR - 1) * 1024;               // trajectories in the last run (1 .. 1024)
    const int nb_last = (nv_last + 63) / 64, rem = nv_last & 63;
    if (e < count) {                                            // rank tid of this run is a real trajectory
        const int j = tid >> 6, l = tid & 63;
        const int idx = (j < nb_last) ? j * NR + r : nb_last * NR + (j - nb_last) * (NR - 1) + r;
        int pos = 64 * idx + l;
        if (rem != 0 && idx > nb_last * NR - 1) pos -= 64 - rem;
        perm[pos] = base + (int)(key[tid] & 1023u);
    }
}
// Spread of a launch's step counts (accepted + rejected, clipped like the sort key): { longest, lower median, count }.  What
// lanes_per_traj = AUTO needs to choose between one lane and a lane pair per trajectory when the pairs need a second generation
// (crnn_capi.hip: launch_adjoint) -- a deterministic function of the launch, formed by one extra block of the sort launch.
__device__ __forceinline__ void step_spread_block(const int32_t *__restrict__ n_accept, const int32_t *__restrict__ n_reject,
                                                  int64_t first, int count, int32_t *__restrict__ spread, unsigned *hist /* 1024 words */) {
    const int tid = threadIdx.x;
    __shared__ int mx_s;
    hist[tid] = 0u;
    if (tid == 0) mx_s = 0;
    __syncthreads();
    for (int e = tid; e < count; e += 1024) atomicAdd(&hist[min(max(n_accept[first + e] + n_reject[first + e], 0), 1023)], 1u);
    __syncthreads();
    const unsigned own = hist[tid];
    if (own) atomicMax(&mx_s, tid);
    // inclusive prefix sums over the 1 024 bins (Hillis-Steele in place: ten rounds; a serial scan by one thread cost 40 us)
    for (int off = 1; off < 1024; off <<= 1) {
        const unsigned v = hist[tid] + (tid >= off ? hist[tid - off] : 0u);
        __syncthreads();
        hist[tid] = v;
        __syncthreads();
    }
    const unsigned incl = hist[tid], rank = (unsigned)((count - 1) / 2);
    if (own && incl - own <= rank && rank < incl) spread[1] = tid;     // the bin that holds the lower median
    if (tid == 0) { spread[0] = mx_s; spread[2] = count; }
}
__global__ __launch_bounds__(1024) void sort_steps_kernel(const int32_t *__restrict__ n_accept, const int32_t *__restrict__ n_reject,
                                                          int64_t first, int count, int32_t *__restrict__ perm) {
    __shared__ unsigned key[1024];
    sort_steps_run(n_accept, n_reject, first, count, perm, (int)blockIdx.x, (int)gridDim.x, key);
}

// Fixed-order reduction of the per-block partials (theta space) followed by the chain rule through p2vec:
//   red_theta[m] = sum_blk partials[blk][m]   (m < nth + kExtra; the order over blk is fixed)
//   out = [ dtheta[k,:] . red_theta[0:nth], k < P | n_overflow | extras ]      (the common layout, ros23_kernel.hpp kTail)
// One block; nth + kExtra <= 256.
__device__ __forceinline__ void reduce_project_body(const double *__restrict__ partials, int nblk,
                                                    const double *__restrict__ dtheta, int nth, int P,
                                                    double *__restrict__ red_theta, double *__restrict__ out,
                                                    const unsigned int *__restrict__ overflow, double *sh, double (*part)[64]) {
    const int npart = nth + kExtra;
    const int tid = threadIdx.x;
    // columns in chunks of 64, sixteen row lanes per column; the combination order is fixed
    const int col = tid & 63, rl = tid >> 6;
    for (int c0 = 0; c0 < npart; c0 += 64) {
        const int k = c0 + col;
        double a0 = 0.0, a1 = 0.0;
        if (k < npart) {
            // rows rl, rl + 16, rl + 32, ... ; sixteen loads in flight per step of the loop (thirty-two spill: a 1 024-thread block has 128 registers per lane) (the sum order stays fixed: even
            // positions of the row sequence into a0, odd ones into a1).  This block is alone on the critical path between two solve
            // launches and every step of this loop is one exposed memory latency: with eight loads in flight the 2 048 batch rows of a
            // lane-pair launch took 50 us (1 024 rows: 18 us).
            for (int bI = rl; bI < nblk; bI += 256) {
                double v[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) v[q] = (bI + 16 * q < nblk) ? partials[(size_t)(bI + 16 * q) * npart + k] : 0.0;
#pragma unroll
                for (int q = 0; q < 16; q += 2) { a0 += v[q]; a1 += v[q + 1]; }
            }
        }
        part[rl][col] = a0 + a1;
        __syncthreads();
        if (rl == 0 && k < npart) {
            double a = 0.0;
#pragma unroll
            for (int r = 0; r < 16; ++r) a += part[r][col];
            sh[k] = a;
            red_theta[k] = a;
        }
        __syncthreads();
    }
    // A trajectory that outran its tape makes this launch's gradient unusable.  The count travels with the gradient
    // (slot P) -- through the all-reduce to every rank -- and the optimiser kernel of every rank skips a step whose
    // summed count is not zero in the same way; the host repeats the step with forward tangents when it next looks
    // (crnn_capi.hip: check_pending).  A NaN gradient from any other cause is NOT intercepted: it reaches p, as it
    // would in the reference.
    for (int k = tid; k < P; k += 1024) {
        double a = 0.0;
        for (int m = 0; m < nth; ++m) a = fma(dtheta[(size_t)k * nth + m], sh[m], a);
        out[k] = a;
    }
    if (tid == 0) out[P] = overflow ? (double)*overflow : 0.0;
    if (tid < kExtra) out[P + 1 + tid] = sh[nth + tid];
}
__global__ __launch_bounds__(1024) void reduce_project_kernel(const double *__restrict__ partials, int nblk,
                                                              const double *__restrict__ dtheta, int nth, int P,
                                                              double *__restrict__ red_theta, double *__restrict__ out,
                                                              const unsigned int *__restrict__ overflow) {
    __shared__ double sh[256];
    __shared__ double part[16][64];
    reduce_project_body(partials, nblk, dtheta, nth, P, red_theta, out, overflow, sh, part);
}
// The two in one launch: block 0 reduces this launch's partial sums, blocks 1 .. nruns sort its step counts into the queue
// order of the NEXT launch over the same range (the solve kernel that read `perm` has finished: same stream).  One launch
// and one inter-kernel gap less per training step (~16 us of 0.53 ms).
__global__ __launch_bounds__(1024) void reduce_project_sort_kernel(const double *__restrict__ partials, int nblk,
                                                                   const double *__restrict__ dtheta, int nth, int P,
                                                                   double *__restrict__ red_theta, double *__restrict__ out,
                                                                   const unsigned int *__restrict__ overflow,
                                                                   const int32_t *__restrict__ n_accept,
                                                                   const int32_t *__restrict__ n_reject, int64_t first, int count,
                                                                   int32_t *__restrict__ perm, int32_t *__restrict__ spread) {
    __shared__ double sh[256];
    __shared__ double part[16][64];
    __shared__ unsigned key[1024];
    const int extra = spread ? 1 : 0;        // one more block at the end of the grid: the spread of the step counts
    if (blockIdx.x == 0) reduce_project_body(partials, nblk, dtheta, nth, P, red_theta, out, overflow, sh, part);
    else if (extra && blockIdx.x == gridDim.x - 1) step_spread_block(n_accept, n_reject, first, count, spread, key);
    else sort_steps_run(n_accept, n_reject, first, count, perm, (int)blockIdx.x - 1, (int)gridDim.x - 1 - extra, key);
}

}  // namespace crnn

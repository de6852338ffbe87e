// crnn_amd/csrc/ros23_adj2_kernel.hpp -- gfx950 (MI355X): the Rosenbrock23 discrete-adjoint gradient kernel with TWO LANES
// PER TRAJECTORY (round 3).
//
// Same mathematics, same tape, same outputs as ros23_adj_kernel.hpp (reference: solve + loss + ForwardDiff.gradient,
// case2/case2.jl:124-137,195; case1/case1.jl) -- what changes is the mapping.  ros23_adj_kernel gives a trajectory one
// lane: at 65 536 trajectories that is exactly one wavefront per SIMD, a launch lasts as long as the longest wavefront's
// chain of ~2 900 instructions per step pair, and below 65 536 trajectories the chip is partly empty while the launch
// takes just as long (0.466 ms from 4 096 to 32 768 trajectories: DESIGN.md 5) -- a shard of a strongly-scaled batch
// gains nothing from its idle lanes.  Here an adjacent lane pair (2g, 2g+1) owns one trajectory:
//
//   * lane m of the pair holds species m*H .. m*H+H-1 (H = ceil(NS/2)): u, f, k1, dk, lambda, the rows of w_in / w_out that
//     belong to them (in registers: the two lanes of a pair need different weights, so they cannot be scalar operands) and
//     the gradient accumulators of exactly those rows -- 2*H*NR registers-doubles, no LDS atomics at all;
//   * the logarithms (one per species) and the O(NS*NR) contractions split in half; everything that couples the species
//     goes through ONE cross-lane step: s = a + dpp_quad_perm[1,0,3,2](a) -- both lanes form a+b / b+a, the same bits, so
//     reaction rates, the NR x NR Woodbury matrix, the error norm, the step-size controller and every branch decision are
//     replicated exactly and the pair never diverges;
//   * the wave takes 32 trajectories from the queue; its batch sums (one partial row per 32 trajectories) go through an
//     LDS staging area in fixed order like the one-lane kernel's.
//
// Per step pair a lane executes ~55-60 % of the one-lane kernel's instructions (the replicated part -- LU of the NR x NR
// matrix, controller, save-point bookkeeping -- does not shrink) with about half its registers.  The host uses it where the
// pairs still fit the resident lanes (crnn_capi.hip launch_adjoint; crnn_ctx_set_lanes_per_traj).
// Shapes: nr < ns (Woodbury form of W: case1, case2), no rate scaling.  Robertson (ns = 3 < nr = 6, dense W) keeps one lane.
#pragma once
#include <type_traits>
#include "ros23_adj_kernel.hpp"

// phase timing (tools/kvariants.sh build prof2="-DCRNN_ADJ2_PROF=1"; the library prints the shares of wave 0 of block 0 to stderr after
// every lane-pair launch): s_memtime deltas per phase summed in scalar registers, no fences -- a guide, not the measurement
#ifdef CRNN_ADJ2_PROF
namespace crnn { __device__ unsigned long long g_adj2_prof[16]; }
#if CRNN_ADJ2_PROF == 2   /* with scheduling fences at the phase boundaries: true attribution, slower kernel */
#define ADJ2_T(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned now_ = (unsigned)__builtin_readcyclecounter(); prof_acc[(k)] += now_ - prof_last; prof_last = now_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define ADJ2_T(k) do { const unsigned now_ = (unsigned)__builtin_readcyclecounter(); prof_acc[(k)] += now_ - prof_last; prof_last = now_; } while (0)
#endif
#else
#define ADJ2_T(k) do { } while (0)
#endif

namespace crnn {

// a + (the other lane of the pair's a): one DPP step per 32-bit half; identical bits in both lanes (a+b == b+a)
__device__ __forceinline__ double pair_sum(double a) {
    const int lo = __double2loint(a), hi = __double2hiint(a);
    // quad_perm [1,0,3,2]; bound_ctrl (every source lane exists) so that no "old" value has to be moved into the destination first
    const int plo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true);
    const int phi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true);
    return a + __hiloint2double(phi, plo);
}
__device__ __forceinline__ int pair_and(int a) { return a & __builtin_amdgcn_update_dpp(0, a, 0xB1, 0xF, 0xF, true); }

// (Round 4 measured a register-lean build of this kernel -- weight rows re-read from LDS, accumulators in LDS cells, two wavefronts per
// SIMD -- at break-even with this one; its switches CRNN_ADJ2_LEAN / CRNN_ADJ2_OCC were deleted in round 5.)
template <int NS, int NR, bool HAS_T, int BLOCK, int OCC>
__global__ __launch_bounds__(BLOCK, OCC) void ros23_adj2_kernel(const SolveParams prm, const double *__restrict__ theta,
                                                                const AdjParams adj) {
    using L_ = Lay<NS, NR, HAS_T>;
    constexpr int N = L_::N;
    constexpr int NTH = L_::NTH;
    constexpr int H = (NS + 1) / 2;        // species per lane (the second lane's last one is padding when NS is odd)
    constexpr int RECW = NS + 2;
    constexpr int GPB = BLOCK / 2;         // trajectories (lane pairs) per block
    static_assert(NR < NS, "Woodbury form of W only (nr < ns)");
    static_assert(NTH + kExtra <= 64, "the per-batch sums use one lane per column");

    __shared__ double kc_lds[kNConst];
    // two -inf slots in front (the reverse sweep reads two save times back unconditionally), four +inf slots behind the last save time (the
    // forward sweep counts the save points a step passes four at a time)
    __shared__ double tsp_lds[kMaxSave + 2 + 4];
    double *const ts_lds = tsp_lds + 2;
    __shared__ double stage_lds[(NTH + kExtra) * GPB];    // batch sums: [column][pair of this block]
    const int tid = threadIdx.x;
    for (int idx = tid; idx < kNConst; idx += BLOCK) kc_lds[idx] = reinterpret_cast<const double *>(prm.kc)[idx];
    for (int idx = tid; idx < prm.n_save; idx += BLOCK) ts_lds[idx] = prm.tsave[idx];
    if (tid < 2) tsp_lds[tid] = -INFINITY;
    if (tid >= 2 && tid < 6) ts_lds[prm.n_save + tid - 2] = INFINITY;
    __syncthreads();
    const KConst *kc = reinterpret_cast<const KConst *>(kc_lds);

    const int lane = tid & 63;
    const int m = lane & 1;                 // which half of the species this lane owns
    const int gib = tid >> 1;               // pair index within the block
    const int giw = lane >> 1;              // pair index within the wavefront (0..31)

    // ---- this lane's weights and per-species constants, in registers for the whole kernel
    double wi_[H][NR], wo_[H][NR], wT[NR], wb_[NR];
    double atl_[H], rtl_[H], iys_[H];
    int dro[H];                             // data column of the species, -1 if unobserved (or padding)
    bool own[H];
#pragma unroll
    for (int i = 0; i < H; ++i) {
        const int c = m * H + i;
        own[i] = c < NS;
        const int cc = own[i] ? c : 0;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            wi_[i][j] = own[i] ? theta[L_::wi(cc, j)] : 0.0;
            wo_[i][j] = own[i] ? theta[L_::wo(cc, j)] : 0.0;
        }
        atl_[i] = own[i] ? kc->atol[cc] : 1.0;
        rtl_[i] = own[i] ? kc->rtol[cc] : 0.0;
        iys_[i] = own[i] ? kc->inv_yscale[cc] : 0.0;
        dro[i] = own[i] ? (int)kc->drow[cc] : -1;
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        wT[j] = HAS_T ? theta[L_::wi(NS, j)] : 0.0;
        wb_[j] = theta[L_::wb(j)];
    }
#define WI(i, j) wi_[(i)][(j)]
#define WO(i, j) wo_[(i)][(j)]
#define ATL(i) atl_[(i)]
#define RTL(i) rtl_[(i)]
#define IYS(i) iys_[(i)]

    double iyz[H];                          // 1/yscale of this lane's observed species, 0 for unobserved / padding ones (reverse sweep)
#pragma unroll
    for (int i = 0; i < H; ++i) iyz[i] = dro[i] >= 0 ? kc->inv_yscale[m * H + i < NS ? m * H + i : 0] : 0.0;
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
    const double ubc = prm.clamp_pred ? kc->ub : __builtin_inf();   // no clamp = an infinite clamp
    const bool lk0 = prm.loss_kind == 0;

    double *const tape = adj.tape + (size_t)((size_t)blockIdx.x * GPB + gib) * adj.tape_cap * RECW;
#ifdef CRNN_ADJ2_PROF
    unsigned prof_acc[14] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    unsigned prof_last = (unsigned)__builtin_readcyclecounter();
#endif

    while (true) {
        // ---- next 32 trajectories for this wavefront
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(prm.queue, 32ULL);
        const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)base);
        const unsigned bhi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
        const int64_t wave_base = (int64_t)(((unsigned long long)bhi << 32) | blo);
        if (wave_base >= prm.count) break;
        const int64_t traj = wave_base + giw;
        const bool valid = traj < prm.count;
        const int64_t b = prm.first + (valid ? (adj.perm ? (int64_t)adj.perm[traj] : traj) : 0);
        CRNN_CHK(b >= 0 && b < prm.B && traj >= 0, 21);

        double bT[NR];
        double xT = 0.0, Tconst = 0.0;
        // point evaluation: x = log clamp(u), g = dx/du (this lane's species); r (replicated); f (this lane's species)
        auto eval_point = [&](const double (&uu)[H], double (&x)[H], double (&g)[H], double (&r)[NR], double (&f)[H]) {
            features<H>(uu, kc->lb, kc->ub, x, g);
            double z[NR];
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                double a = 0.0;
#pragma unroll
                for (int i = 0; i < H; ++i) a = fma(WI(i, j), x[i], a);
                z[j] = pair_sum(a) + bT[j];
            }
            fexp_vec<NR>(z, r);
#pragma unroll
            for (int i = 0; i < H; ++i) {
                double a = 0.0;
#pragma unroll
                for (int j = 0; j < NR; ++j) a = fma(WO(i, j), r[j], a);
                f[i] = a;
            }
        };
        // W = I - gam J in Woodbury form: M = I_nr - gam B^T A (replicated), pivoted LU
        double M[NR][NR], dinv[NR];
        int piv[NR];
        bool wave_pivots = false;
        auto factor = [&](const double (&g)[H], const double (&r)[NR], const double gam) -> bool {
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                double tj[H];
#pragma unroll
                for (int i = 0; i < H; ++i) tj[i] = WI(i, j) * g[i];
#pragma unroll
                for (int l = 0; l < NR; ++l) {
                    double a = 0.0;
#pragma unroll
                    for (int i = 0; i < H; ++i) a = fma(tj[i], WO(i, l), a);
                    M[j][l] = ((j == l) ? 1.0 : 0.0) - (gam * r[l]) * pair_sum(a);
                }
            }
            bool anyp;
            const bool ok = lu_factor<NR>(M, dinv, piv, anyp);
            wave_pivots = __builtin_amdgcn_ballot_w64(anyp) != 0;
            return ok;
        };
        // b <- W^-1 b = b + w_out (gr .* M^-1 (w_in^T (g .* b)))
        auto solve = [&](const double (&g)[H], const double (&gr)[NR], double (&bb)[H]) {
            double y[NR];
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                double a = 0.0;
#pragma unroll
                for (int i = 0; i < H; ++i) a = fma(WI(i, j), g[i] * bb[i], a);
                y[j] = pair_sum(a);
            }
            lu_solve<NR>(M, dinv, piv, wave_pivots, y);
#pragma unroll
            for (int j = 0; j < NR; ++j) y[j] *= gr[j];
#pragma unroll
            for (int i = 0; i < H; ++i) {
                double a = 0.0;
#pragma unroll
                for (int j = 0; j < NR; ++j) a = fma(WO(i, j), y[j], a);
                bb[i] += a;
            }
        };
        // b <- W^-T b = b + g .* (w_in (M^-T (gr .* (w_out^T b))))
        auto solve_Tr = [&](const double (&g)[H], const double (&gr)[NR], double (&bb)[H]) {
            double y[NR];
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                double a = 0.0;
#pragma unroll
                for (int i = 0; i < H; ++i) a = fma(WO(i, j), bb[i], a);
                y[j] = pair_sum(a) * gr[j];
            }
            lu_solve_T<NR>(M, dinv, piv, wave_pivots, y);
#pragma unroll
            for (int i = 0; i < H; ++i) {
                double a = 0.0;
#pragma unroll
                for (int j = 0; j < NR; ++j) a = fma(WI(i, j), y[j], a);
                bb[i] = fma(a, g[i], bb[i]);
            }
        };

        // ================================================================== forward sweep
        double u[H], f0[H], g0[H], r0[NR];
        double t = t0, dt = 0.0, lqold = lqinit;
        int iter = 0, jsave = 0, nacc = 0, nrej = 0;
        int rc = valid ? -1 : 0;
#pragma unroll
        for (int i = 0; i < H; ++i) u[i] = own[i] ? prm.u0[(size_t)(m * H + i) * prm.B + b] : 1.0;
        if (HAS_T) {
            Tconst = prm.u0[(size_t)NS * prm.B + b];
            xT = kc->inv_R * frcp(Tconst);
        }
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            bT[j] = HAS_T ? fma(wT[j], xT, wb_[j]) : wb_[j];
        }
        {
            double x0[H];
            eval_point(u, x0, g0, r0, f0);
            // Hairer initial step (OrdinaryDiffEq ode_determine_initdt, order 2)
            double d0 = 0.0, d1 = 0.0, sk[H];
#pragma unroll
            for (int i = 0; i < H; ++i) {
                sk[i] = own[i] ? frcp(fma(fabs(u[i]), RTL(i), ATL(i))) : 0.0;
                const double a = u[i] * sk[i], c = f0[i] * sk[i];
                d0 = fma(a, a, d0);
                d1 = fma(c, c, d1);
            }
            d0 = pair_sum(d0);
            d1 = pair_sum(d1);
            if (HAS_T) { const double a = Tconst * frcp(fma(fabs(Tconst), kc->rtol[NS], kc->atol[NS])); d0 = fma(a, a, d0); }
            d0 = sqrt(d0 * (1.0 / N));
            d1 = sqrt(d1 * (1.0 / N));
            double dt0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * (d0 / d1);
            dt0 = fmin(dt0, dtmax);
            double u1[H], x1[H], g1[H], r1[NR], f1[H];
#pragma unroll
            for (int i = 0; i < H; ++i) u1[i] = fma(dt0, f0[i], u[i]);
            eval_point(u1, x1, g1, r1, f1);
            double d2 = 0.0;
#pragma unroll
            for (int i = 0; i < H; ++i) { const double e = (f1[i] - f0[i]) * sk[i]; d2 = fma(e, e, d2); }
            d2 = sqrt(pair_sum(d2) * (1.0 / N)) / dt0;
            const double dm = fmax(d1, d2);
            const double dt1 = (dm <= 1e-15) ? fmax(1e-6, dt0 * 1e-3) : exp(-0.5 * (4.605170185988091368 + flog(dm)));
            dt = fmax(kc->dtmin, fmin(fmin(100.0 * dt0, dt1), dtmax));
        }
        auto write_pred = [&](int j, const double (&v)[H]) {
            CRNN_CHK((int64_t)j * prm.n_obs < prm.row_stride && j >= 0, 29);
#pragma unroll
            for (int i = 0; i < H; ++i) {
                if (own[i]) {
                    double w = v[i];
                    if (prm.clamp_pred) w = clampv(w, -kc->ub, kc->ub);
                    prm.pred[((size_t)j * N + (m * H + i)) * prm.B + b] = w;
                }
            }
            if (HAS_T && m == 0) {
                double w = Tconst;
                if (prm.clamp_pred) w = clampv(w, -kc->ub, kc->ub);
                prm.pred[((size_t)j * N + NS) * prm.B + b] = w;
            }
        };
        if (start_saved) {  // save_start: saveat contains tspan[1]
            if (valid && prm.pred) write_pred(0, u);
            jsave = 1;
        }

        while (__builtin_amdgcn_ballot_w64(rc < 0) != 0) {
            if (rc < 0) {
                ++iter;
                bool last = false;
                if (jsave >= nsave) rc = 0;
                else if (iter > prm.maxiters) rc = 1;
                if (t + dt * (1.0 + 1e-13) >= tend) { dt = tend - t; last = true; }
                if (rc < 0 && (!(dt > kc->dtmin) || t + dt == t)) rc = 2;
                if (rc < 0) {
                    ADJ2_T(0);   // loop control
                    const double gam = d_ * dt;
                    double gr0[NR];
#pragma unroll
                    for (int j = 0; j < NR; ++j) gr0[j] = gam * r0[j];
                    double k1[H], dk[H], unew[H], f1[H], f2[H], g2[H], r2[NR];
                    const bool okf = factor(g0, r0, gam);
#pragma unroll
                    for (int i = 0; i < H; ++i) k1[i] = f0[i];
                    solve(g0, gr0, k1);
                    ADJ2_T(1);   // factor + solve
                    {
                        double u1[H], x1[H], g1[H], r1[NR];
#pragma unroll
                        for (int i = 0; i < H; ++i) u1[i] = fma(0.5 * dt, k1[i], u[i]);
                        eval_point(u1, x1, g1, r1, f1);
                    }
                    ADJ2_T(2);   // evaluation at u_mid
#pragma unroll
                    for (int i = 0; i < H; ++i) dk[i] = f1[i] - k1[i];
                    solve(g0, gr0, dk);
#pragma unroll
                    for (int i = 0; i < H; ++i) unew[i] = fma(dt, k1[i] + dk[i], u[i]);
                    ADJ2_T(3);   // solve
                    {
                        double x2[H];
                        eval_point(unew, x2, g2, r2, f2);
                    }
                    ADJ2_T(4);   // evaluation at u_new
                    double k3[H];
#pragma unroll
                    for (int i = 0; i < H; ++i) {
                        const double k2i = k1[i] + dk[i];
                        k3[i] = f2[i] - c32 * (k2i - f1[i]) - 2.0 * (k1[i] - f0[i]);
                    }
                    solve(g0, gr0, k3);
                    double es = 0.0;
                    bool finite = okf;
#pragma unroll
                    for (int i = 0; i < H; ++i) {
                        const double k2i = k1[i] + dk[i];
                        const double ev = dt * (1.0 / 6.0) * (k1[i] - 2.0 * k2i + k3[i]);
                        const double mx = fmax(fabs(u[i]), fabs(unew[i]));
                        const double e = ev * frcp1(fma(RTL(i), mx, ATL(i)));
                        es = fma(e, e, es);
                        finite = finite && isfinite(unew[i]) && isfinite(ev);
                    }
                    es = pair_sum(es) * (1.0 / N);
                    finite = pair_and(finite ? 1 : 0) != 0;
                    ADJ2_T(5);   // solve + error norm
                    if (!finite) rc = 3;
                    else {
                        // PI controller (OrdinaryDiffEq PIController), in log space
                        const bool ee_zero = (es == 0.0);
                        const double lEE = 0.5 * flog(ee_zero ? 1.0 : es);
                        const double lq11 = kc->beta1 * lEE;
                        double q = ee_zero ? 1.0 / kc->qmax
                                           : fmax(1.0 / kc->qmax, fmin(1.0 / kc->qmin, exp(lq11 - kc->beta2 * lqold) / kc->gamma));
                        ADJ2_T(6);   // controller
                        if (es <= 1.0) {
                            if (nacc >= adj.tape_cap) {
                                rc = 5;  // out of tape: the host re-runs the call with forward tangents
                                if (m == 0) atomicAdd(adj.overflow, 1u);
                            } else {
                                CRNN_CHK(nacc >= 0 && nacc < adj.tape_cap, 25);
                                double *rec = tape + (size_t)nacc * RECW;
                                if (m == 0) { rec[0] = t; rec[1] = dt; }
#pragma unroll
                                for (int i = 0; i < H; ++i)
                                    if (own[i]) rec[2 + m * H + i] = u[i];
                                ++nacc;
                                const double tnew = last ? tend : t + dt;
                                if (!prm.pred) {
                                    // a gradient launch only COUNTS the save points inside the step (the reverse sweep evaluates them): four save
                                    // times per LDS round trip instead of a dependent read, a compare and a branch per point (the times ascend;
                                    // +inf behind the last one stops the count at nsave)
                                    while (true) {
                                        CRNN_CHK(jsave >= 0 && jsave <= nsave, 26);
                                        const double a0 = ts_lds[jsave], a1 = ts_lds[jsave + 1], a2 = ts_lds[jsave + 2], a3 = ts_lds[jsave + 3];
                                        const int c = (a0 <= tnew ? 1 : 0) + (a1 <= tnew ? 1 : 0) + (a2 <= tnew ? 1 : 0) + (a3 <= tnew ? 1 : 0);
                                        jsave += c;
                                        if (c < 4) break;
                                    }
                                } else
                                while (jsave < nsave) {
                                    CRNN_CHK(jsave >= 0 && jsave < nsave, 26);
                                    const double ts = ts_lds[jsave];
                                    if (!(ts <= tnew)) break;
                                    if (prm.pred) {
                                        const bool at_end = (ts == tnew);
                                        const double Th = at_end ? 1.0 : (ts - t) / dt;
                                        const double c1 = at_end ? 0.0 : Th * (1.0 - Th) * inv12d;
                                        const double c2 = at_end ? 1.0 : Th * (Th - 2.0 * d_) * inv12d;
                                        double v[H];
#pragma unroll
                                        for (int i = 0; i < H; ++i) {
                                            const double k2i = k1[i] + dk[i];
                                            v[i] = at_end ? unew[i] : fma(dt, fma(c1, k1[i], c2 * k2i), u[i]);
                                        }
                                        write_pred(jsave, v);
                                    }
                                    ++jsave;
                                }
#pragma unroll
                                for (int i = 0; i < H; ++i) { u[i] = unew[i]; f0[i] = f2[i]; g0[i] = g2[i]; }
#pragma unroll
                                for (int j = 0; j < NR; ++j) r0[j] = r2[j];
                                t = tnew;
                                // step_accept_controller
                                if (q >= kc->qsteady_min && q <= kc->qsteady_max) q = 1.0;
                                lqold = ee_zero ? lqinit : fmax(lEE, lqinit);
                                dt = fmin(dt / q, dtmax);
                                if (jsave >= nsave) rc = 0;
                                ADJ2_T(7);   // tape record, save-point loop, FSAL copy
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
        [[maybe_unused]] const int jlo = start_saved ? 1 : 0;      // (read by the bounds checks only: -DCRNN_BOUNDS_CHECK)
        double awi[H][NR], awo[H][NR];     // d loss / d (this lane's rows of w_in, w_out): registers, no atomics
#pragma unroll
        for (int i = 0; i < H; ++i)
#pragma unroll
            for (int j = 0; j < NR; ++j) { awi[i][j] = 0.0; awo[i][j] = 0.0; }
#define AWI_ADD(i, j, val) awi[(i)][(j)] += (val)
#define AWO_ADD(i, j, val) awo[(i)][(j)] += (val)
        double lam[H];
#pragma unroll
        for (int i = 0; i < H; ++i) lam[i] = 0.0;
        double wbb[NR];                    // d/d w_b (replicated); the temperature row of w_in is xT times the same sum
#pragma unroll
        for (int j = 0; j < NR; ++j) wbb[j] = 0.0;
        double loss_sum = 0.0;             // this lane's species only; the pair's sum is formed at the end
        double tnew = t;
        int s = valid ? nacc - 1 : -1;

        const double *const drows = prm.data + (size_t)b * prm.row_stride;
        int doff[H];
#pragma unroll
        for (int i = 0; i < H; ++i) doff[i] = dro[i] >= 0 ? dro[i] : 0;
        auto load_row = [&](int j, double (&d)[H]) {
            CRNN_CHK((int64_t)(j > 0 ? j : 0) * prm.n_obs < prm.row_stride, 22);
            const double *row = drows + (size_t)(j > 0 ? j : 0) * prm.n_obs;
#pragma unroll
            for (int i = 0; i < H; ++i) d[i] = row[doff[i]];
        };
        // the save times the reverse sweep is about to pass.  No guard on the index: below zero sit the -inf slots, and with save_start
        // (jlo = 1) slot 0 holds t0 itself, which no step begins before -- "ts > tn" is false for it as it would be for -inf
        double ts_cur = ts_lds[jsave - 1];
        double ts_nxt = ts_lds[jsave - 2];
        double rt = 0.0, rdt = 0.0, ru[H];   // tape record s, prefetched
        auto load_rec = [&](int idx) {
            CRNN_CHK(idx < adj.tape_cap, 23);
            const double *rec = tape + (size_t)(idx > 0 ? idx : 0) * RECW;
            rt = rec[0]; rdt = rec[1];
#pragma unroll
            for (int i = 0; i < H; ++i) ru[i] = own[i] ? rec[2 + m * H + i] : 1.0;
        };
        load_rec(s);

        while (__builtin_amdgcn_ballot_w64(s >= 0) != 0) {
            if (s >= 0) {
                const double tn = rt, h = rdt;
                double un[H];
#pragma unroll
                for (int i = 0; i < H; ++i) un[i] = ru[i];
                double dA[H], dB[H], dC[H];
                load_row(jsave - 1, dA);
                load_row(jsave - 2, dB);
                load_row(jsave - 3, dC);
                load_rec(s - 1);   // prefetch the next record
                ADJ2_T(8);   // reverse: loop control + prefetches
                // ---- re-form the step
                double x0[H], gg0[H], rr0[NR], ff0[H];
                const double gam = d_ * h;
                double gr0[NR], x1[H], g1[H], r1[NR];
                eval_point(un, x0, gg0, rr0, ff0);
#pragma unroll
                for (int j = 0; j < NR; ++j) gr0[j] = gam * rr0[j];
                (void)factor(gg0, rr0, gam);
                double k1[H], dk[H];
#pragma unroll
                for (int i = 0; i < H; ++i) k1[i] = ff0[i];
                solve(gg0, gr0, k1);
                {
                    double u1[H], f1[H];
#pragma unroll
                    for (int i = 0; i < H; ++i) u1[i] = fma(0.5 * h, k1[i], un[i]);
                    eval_point(u1, x1, g1, r1, f1);
#pragma unroll
                    for (int i = 0; i < H; ++i) dk[i] = f1[i] - k1[i];
                }
                solve(gg0, gr0, dk);
                ADJ2_T(9);   // reverse: re-formation of the step

                // ---- loss and its seeds at the save points inside (tn, tnew]
                // Straight-line per point (round 5; a fifth of the kernel's time in round 4's phase profile, profiles/r04l, mostly selects and
                // per-species branches): the loss kind is ONE wave-uniform branch around the whole phase; an unobserved (or padding) species is
                // a zero weight, no clamp an infinite clamp; the end-of-step point needs no select (c1 = 0, c2 = 1 give h k2 + u_n bit for bit);
                // the mask is "the clamp changed nothing".  Same values as the branchy form it replaces (parity + cross-kernel tests).
                double A_[H], B1[H], B2[H], k2[H];
#pragma unroll
                for (int i = 0; i < H; ++i) { A_[i] = 0.0; B1[i] = 0.0; B2[i] = 0.0; k2[i] = k1[i] + dk[i]; }
                const double inv_h = frcp(h);
                auto in_step = [&]() -> bool { return ts_cur > tn; };
                auto seed_point = [&](const double (&dobs)[H], auto lk_) {
                    constexpr bool LK0 = decltype(lk_)::value;
                    const double ts = ts_cur;
                    CRNN_CHK(jsave - 1 >= jlo && jsave - 1 < nsave, 28);
                    ts_cur = ts_nxt;
                    ts_nxt = ts_lds[jsave - 3];
                    const bool at_end = (ts == tnew);
                    const double Th = at_end ? 1.0 : (ts - tn) * inv_h;
                    const double c1 = Th * (1.0 - Th) * inv12d;        // at the end of the step: 1 * 0 * inv12d = 0 exactly, no select
                    const double c2 = at_end ? 1.0 : Th * (Th - 2.0 * d_) * inv12d;
                    const double hc1 = h * c1, hc2 = h * c2;
#pragma unroll
                    for (int i = 0; i < H; ++i) {
                        const double v = fma(h, fma(c1, k1[i], c2 * k2[i]), un[i]);
                        const double vc = fmin(fmax(v, -ubc), ubc);      // v is finite here (accepted steps only): = clampv
                        const double rr = (dobs[i] - vc) * iyz[i];
                        double w;
                        if constexpr (LK0) { loss_sum += fabs(rr); w = signbit(rr) ? iyz[i] : -iyz[i]; }
                        else { loss_sum = fma(rr, rr, loss_sum); w = (-2.0 * rr) * iyz[i]; }
                        w = (vc == v) ? w : 0.0;
                        A_[i] += w;
                        B1[i] = fma(w, hc1, B1[i]);
                        B2[i] = fma(w, hc2, B2[i]);
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
                                while (in_step()) {
                                    double dD[H];
                                    load_row(jsave - 1, dD);
                                    seed_point(dD, lk_);
                                }
                            }
                        }
                    }
                };
                if (lk0) seeds(std::true_type{}); else seeds(std::false_type{});

                ADJ2_T(10);   // reverse: loss + seeds
                // ---- adjoint of the step (ros23_adj_kernel.hpp, same formulas; sums over species cross the pair once)
                double kb1[H], v[H], ub[H];
#pragma unroll
                for (int i = 0; i < H; ++i) { v[i] = fma(h, lam[i], B2[i]); ub[i] = lam[i] + A_[i]; }
#pragma unroll
                for (int i = 0; i < H; ++i) kb1[i] = B1[i] + v[i];
                solve_Tr(gg0, gr0, v);                 // v = W^-T kb2
#pragma unroll
                for (int i = 0; i < H; ++i) kb1[i] -= v[i];
                double av[NR];
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    double a = 0.0;
#pragma unroll
                    for (int i = 0; i < H; ++i) a = fma(v[i], WO(i, j), a);
                    av[j] = pair_sum(a);
                }
                double rho1[NR];   // av_j r_j(u_mid)
                {
                    double um[H];
#pragma unroll
                    for (int i = 0; i < H; ++i) um[i] = 0.0;
#pragma unroll
                    for (int j = 0; j < NR; ++j) {
                        const double rho = av[j] * r1[j];
                        rho1[j] = rho;
#pragma unroll
                        for (int i = 0; i < H; ++i) um[i] = fma(rho, WI(i, j), um[i]);
                    }
#pragma unroll
                    for (int i = 0; i < H; ++i) {
                        const double mm = um[i] * g1[i];
                        ub[i] += mm;
                        kb1[i] = fma(0.5 * h, mm, kb1[i]);
                    }
                }
                solve_Tr(gg0, gr0, kb1);               // kb1 = w = W^-T kb1
                ADJ2_T(11);   // reverse: two transposed solves + u_mid terms
                {
                    double s1[H], s2[H];
#pragma unroll
                    for (int i = 0; i < H; ++i) { s1[i] = 0.0; s2[i] = 0.0; }
#pragma unroll
                    for (int j = 0; j < NR; ++j) {
                        double aw = 0.0, q1 = 0.0, qd = 0.0;
#pragma unroll
                        for (int i = 0; i < H; ++i) {
                            aw = fma(kb1[i], WO(i, j), aw);
                            const double wg = WI(i, j) * gg0[i];
                            q1 = fma(wg, k1[i], q1);
                            qd = fma(wg, dk[i], qd);
                        }
                        aw = pair_sum(aw);
                        q1 = pair_sum(q1);
                        qd = pair_sum(qd);
                        const double c1j = fma(gam, q1, 1.0), czd = gam * qd;
                        const double pv = av[j] * gr0[j];        // gam a^v_j r_j
                        const double pw = aw * rr0[j];           // a^w_j r_j
                        const double gpw = gam * pw;
                        const double beta = fma(pw, c1j, pv * qd);
                        wbb[j] += beta + rho1[j];
                        const double ca = fma(rr0[j], czd, r1[j]), cb = rr0[j] * c1j;
#pragma unroll
                        for (int i = 0; i < H; ++i) {
                            const double mm = fma(pv, dk[i], gpw * k1[i]);
                            AWI_ADD(i, j, fma(rho1[j], x1[i], fma(beta, x0[i], gg0[i] * mm)));
                            const double wij = WI(i, j);
                            s1[i] = fma(beta, wij, s1[i]);
                            s2[i] = fma(wij, mm, s2[i]);
                            AWO_ADD(i, j, fma(v[i], ca, kb1[i] * cb));
                        }
                    }
#pragma unroll
                    for (int i = 0; i < H; ++i) {
                        const double g = gg0[i];
                        lam[i] = ub[i] + g * (s1[i] - g * s2[i]);   // g' = -g^2 inside the window, 0 outside
                    }
                }
                tnew = tn;
                --s;
                ADJ2_T(12);   // reverse: contraction into the gradient accumulators + lambda
            }
        }

        // ---- outputs
        if (start_saved && n_saved >= 1) {  // the saved initial point: a loss term without gradient
#pragma unroll
            for (int i = 0; i < H; ++i) {
                if (dro[i] >= 0) {
                    double v = prm.u0[(size_t)(m * H + i) * prm.B + b];
                    if (prm.clamp_pred) v = clampv(v, -kc->ub, kc->ub);
                    const double rr = (drows[doff[i]] - v) * IYS(i);
                    loss_sum += (prm.loss_kind == 0) ? fabs(rr) : rr * rr;
                }
            }
        }
        const double loss_tot = pair_sum(loss_sum);
        const double denom = (double)prm.n_obs * (double)n_saved;
        if (valid && m == 0) {
            prm.loss[b] = loss_tot * (n_saved > 0 ? 1.0 / denom : 0.0);
            prm.retcode[b] = rc;
            prm.n_saved[b] = n_saved;
            prm.n_accept[b] = nacc;
            prm.n_reject[b] = nrej;
        }
        // ---- sums over the 32 trajectories of this batch: every pair parks its scaled accumulators (each lane the rows it
        //      owns) and the five scalars in LDS, then lane k of the wavefront adds up column k over the 32 pairs in pair order.
        //      The result depends on the batch only, not on which wavefront processed it.
        {
            const double scale_ = (valid && n_saved > 0) ? 1.0 / denom : 0.0;
            double *const st = stage_lds + gib;      // column k of this pair: st[k * GPB]
#pragma unroll
            for (int i = 0; i < H; ++i) {
                if (own[i]) {
#pragma unroll
                    for (int j = 0; j < NR; ++j) {
                        st[L_::wi(m * H + i, j) * GPB] = awi[i][j] * scale_;
                        st[L_::wo(m * H + i, j) * GPB] = awo[i][j] * scale_;
                    }
                }
            }
            if (m == 0) {
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    st[L_::wb(j) * GPB] = wbb[j] * scale_;
                    if (HAS_T) st[L_::wi(NS, j) * GPB] = (wbb[j] * xT) * scale_;
                }
                st[(NTH + 0) * GPB] = valid ? loss_tot * scale_ : 0.0;
                st[(NTH + 1) * GPB] = (valid && rc == 0) ? 1.0 : 0.0;
                st[(NTH + 2) * GPB] = valid ? (double)nacc : 0.0;
                st[(NTH + 3) * GPB] = valid ? (double)nrej : 0.0;
                st[(NTH + 4) * GPB] = valid ? 1.0 : 0.0;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            CRNN_CHK((wave_base >> 5) < ((prm.count + 31) >> 5), 24);
            double *prow = adj.batch_partials + (size_t)(wave_base >> 5) * (NTH + kExtra);
            const int g0w = (tid & ~63) >> 1;        // first pair of this wavefront within the block
            if (lane < NTH + kExtra) {
                const double *src = stage_lds + lane * GPB + g0w;
                double a = 0.0;
                for (int k = 0; k < 32; ++k) a += src[k];
                prow[lane] = a;
            }
            __builtin_amdgcn_wave_barrier();
        }
        ADJ2_T(13);   // outputs + batch sums (and the next batch's start)
    }
#ifdef CRNN_ADJ2_PROF
    if (blockIdx.x == 0 && tid == 0)
        for (int k = 0; k < 14; ++k) g_adj2_prof[k] = prof_acc[k];
#endif
}

#undef WI
#undef WO
#undef ATL
#undef RTL
#undef IYS
#undef AWI_ADD
#undef AWO_ADD
}  // namespace crnn

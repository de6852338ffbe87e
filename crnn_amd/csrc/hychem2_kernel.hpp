// crnn_amd/csrc/hychem2_kernel.hpp -- gfx950 (MI355X): the HyChem pyrolysis CRNN (HyChem/crnn_pyrolysis_mass.jl) with TWO LANES
// PER TRAJECTORY (round 3).  Same mathematics, tape and outputs as hychem_kernel.hpp (read that header first: the RHS with its
// density coupling, the non-autonomous Rosenbrock23 step, the adjoint formulas); what changes is the mapping of a trajectory
// onto lanes, where the step's long-lived values wait (an LDS frame per lane instead of registers that spill), and how the
// gradient is accumulated (on chip, by the matrix unit, instead of 210 HBM accumulators per trajectory).
//
// hychem_kernel gives a trajectory one lane and parks W's 81 factors in LDS: 1.1 KB per trajectory, 128 trajectories per CU
// on two of its four SIMDs, ~25 000 instructions per step pair on the critical path of every wavefront.  Here an adjacent
// lane pair owns the trajectory and EVERYTHING of length ns is distributed over the pair, species 2 i + m in slot i of lane m
// (cyclic: the trailing rows of the LU stay balanced; lane 1's fifth slot is padding and carries log T through the logarithm):
//   * state, stages, adjoints (u, k1, dk, lambda, the seeds ...): five values per lane;
//   * W = I - gam J: each lane builds and keeps ITS FIVE ROWS IN REGISTERS (45 doubles), through the factorisation and the
//     solves -- no LDS for the matrix, so the block is not LDS-bound any more and all four SIMDs hold a wavefront;
//   * LU with partial pivoting, row-distributed: the pivot row reaches the other lane by DPP (quad_perm [1,0,3,2]), each lane
//     eliminates its own rows; the pivot search is a local scan + one exchange, a row swap (rare) a chain of selects;
//   * W x = b in axpy form (x_k broadcast, each lane updates its rows: the one-lane kernel's operation order, same bits);
//     W^T x = b in dot form (each lane's partial dot product over ITS rows, summed over the pair);
//   * contractions over the species: partial sums + pair_sum; the ten rates are replicated (five exponentials per lane,
//     exchanged), as are the step-size controller and the scalars.
// What remains replicated is the controller arithmetic and the rates.  The pair's 64-lane wavefront takes 32 trajectories
// (a "batch") from the queue.
//   * The lane's LDS frame (65 doubles, lane-major): with 512 registers a lane cannot hold a reverse step's live set, and what
//     the allocator spills goes to scratch MEMORY -- a wavefront alone on its SIMD waits out every reload (700 bytes of scratch
//     cost this kernel 40 % of its time).  So the values a step needs again much later are parked explicitly: rates, x, Y, k1,
//     k2 - k1 and scalars of the two points, the FSAL point in the forward sweep, the reactions' gradient factors.
//   * Gradient: per step and trajectory, w_in / w_b / w_out receive six outer products (species-or-feature factor) x (reaction
//     factor)^T.  Summed over the batch's trajectories that is a [12 x 32] . [32 x 10] contraction over the LANE axis per term --
//     v_mfma_f64_16x16x4_f64 (the one FP64 matrix shape; operands staged through LDS, see the MFMA stage).  The two 16 x 16
//     tiles live in AGPRs for the batch and are written once as ONE ROW of the partial-sum table per batch: run-to-run
//     identical (rows belong to batches, not to wavefronts), no atomics, no accumulator traffic to HBM.  Each trajectory's
//     weight 1 / (n_obs n_saved) rides on its loss seeds (the adjoint is linear in them).
#pragma once
#include "hychem_kernel.hpp"
#include "ros23_adj2_kernel.hpp"   // pair_sum, pair_and

namespace crnn {

typedef double hy_v4d __attribute__((ext_vector_type(4)));   // one lane's share of a 16 x 16 FP64 MFMA tile

__device__ __forceinline__ double pair_other(double a) {   // the other lane of the pair's value
    const int lo = __double2loint(a), hi = __double2hiint(a);
    const int plo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true);
    const int phi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true);
    return __hiloint2double(phi, plo);
}
// a wave-uniform double that came out of LDS (so in VGPRs) -> SGPRs: kernel-invariant scalars held in VGPRs for the whole kernel
// are what the allocator spills first
__device__ __forceinline__ double to_sgpr(double a) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(a)), __builtin_amdgcn_readfirstlane(__double2loint(a)));
}
__device__ __forceinline__ int pair_other_i(int a) { return __builtin_amdgcn_update_dpp(0, a, 0xB1, 0xF, 0xF, true); }

// the value held by the pair's lane `owner_odd`, in both lanes: one DPP move per half, quad_perm [1,1,3,3] / [0,0,2,2]
// (owner_odd is a constant wherever this is called from an unrolled loop; the other branch folds away)
__device__ __forceinline__ double pair_pick(const double mine, const bool owner_odd, const bool) {
    const int lo = __double2loint(mine), hi = __double2hiint(mine);
    if (owner_odd)
        return __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0xF5, 0xF, 0xF, false), __builtin_amdgcn_update_dpp(0, lo, 0xF5, 0xF, 0xF, true));
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0xA0, 0xF, 0xF, false), __builtin_amdgcn_update_dpp(0, lo, 0xA0, 0xF, 0xF, true));
}


// species-distributed (lane m holds species 2 i + m in own[i]) -> full-length, replicated: one broadcast per element
template <int NS, int H>
__device__ __forceinline__ void pair_gather(const double (&own)[H], const bool m1, double (&full)[NS]) {
#pragma unroll
    for (int c = 0; c < NS; ++c) full[c] = pair_pick(own[c >> 1], (c & 1) != 0, m1);
}
// full-length, replicated -> this lane's species
template <int NS, int H>
__device__ __forceinline__ void pair_own(const double (&full)[NS], const bool m1, double (&own)[H]) {
#pragma unroll
    for (int i = 0; i < H; ++i) own[i] = m1 ? (2 * i + 1 < NS ? full[2 * i + 1 < NS ? 2 * i + 1 : 0] : 0.0) : full[2 * i];
}

// (flog_ctl / fexp_ctl / sconst -- log and exp with their constants formed in SGPRs where they are used -- live in ros23_kernel.hpp)

template <int NS, int NR>
struct HyPoint2 {
    static constexpr int H = (NS + 1) / 2;
    double Yo[H];          // clamp(u) of this lane's species
    double xo[H];          // log clamp(C) of this lane's species
    double fo[H];          // right-hand side, this lane's species
    double xE, xL;         // -1/(R T), log T (replicated)
    double irho, iS;       // (the ten rates go to the lane's LDS frame: hy_point2's rf)
    unsigned cY, cC;       // bit c: u_c (C_c) inside its clamp window (all species, replicated)
};

// what a lane knows about its slots: species index (0 for the padding slot), whether the slot holds a species, 1/mw, gsc
template <int H>
struct HyLane {
    int ci[H];
    bool ow[H];
};
// 1/mw and gsc of the lane's slots come from the LDS-staged constants at each use (the padding slot reads species 0: every use
// of it is masked by ow or multiplies a zero) -- 20 VGPRs less than keeping them

// point evaluation on species-distributed u; the rates are left in the lane's LDS frame (rf[j * STRIDE], rf == nullptr: dropped)
template <int NS, int NR, int STRIDE>
__device__ __forceinline__ void hy_point2(const double *th, const KConst *kc, const double inv_R, const double (&uo)[(NS + 1) / 2],
                                          const double T, const double P, const bool m1, const HyLane<(NS + 1) / 2> &ln,
                                          HyPoint2<NS, NR> &pt, double *rf) {
    using L_ = LayH<NS, NR>;
    constexpr int H = (NS + 1) / 2;
    static_assert(NR == 2 * H, "the exponentials are split H + H");
    double S = 0.0;
    unsigned cY = 0, cC = 0;
#pragma unroll
    for (int i = 0; i < H; ++i) {
        const double c = fmin(fmax(uo[i], kc->lb), kc->ub);
        cY |= (ln.ow[i] && c == uo[i]) ? (1u << (2 * i)) : 0u;
        pt.Yo[i] = ln.ow[i] ? c : 0.0;
        S = fma(pt.Yo[i], kc->imw[ln.ci[i]], S);
    }
    S = pair_sum(S);
    const double RTS = kc->Ru * T * S;
    const double rho = P * frcp(RTS);
    pt.irho = RTS * frcp(P);
    pt.iS = frcp(S);
    {   // H logarithms per lane: its species; lane 1's padding slot takes T
        double a_[H], la_[H];
#pragma unroll
        for (int i = 0; i < H; ++i) {
            const double C = rho * (pt.Yo[i] * kc->imw[ln.ci[i]]) * 1e3;
            const double c = fmin(fmax(C, kc->lb), kc->ub);
            cC |= (ln.ow[i] && c == C) ? (1u << (2 * i)) : 0u;
            a_[i] = ln.ow[i] ? c : T;
        }
        flog_vec<H>(a_, la_);
#pragma unroll
        for (int i = 0; i < H; ++i) pt.xo[i] = la_[i];
        pt.xL = pair_pick(la_[H - 1], true, m1);
    }
    {
        const unsigned sh = m1 ? 1u : 0u;
        cY <<= sh; cC <<= sh;
        pt.cY = cY | (unsigned)pair_other_i((int)cY);
        pt.cC = cC | (unsigned)pair_other_i((int)cC);
    }
    pt.xE = inv_R * frcp(T);
    CRNN_SCHED_FENCE();
    double z[NR], xz[H];
#pragma unroll
    for (int i = 0; i < H; ++i) xz[i] = ln.ow[i] ? pt.xo[i] : 0.0;   // the padding slot (log T) drops out of the species' sum
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        double a = 0.0;
#pragma unroll
        for (int i = 0; i < H; ++i) a = fma(th[L_::wi(0, j) + ln.ci[i]], xz[i], a);
        z[j] = pair_sum(a) + fma(th[L_::wi(NS, j)], pt.xE, fma(th[L_::wi(NS + 1, j)], pt.xL, th[L_::wb(j)]));
    }
    CRNN_SCHED_FENCE();
    double r[NR];
    {   // H exponentials per lane, exchanged
        double a_[H], e_[H];
#pragma unroll
        for (int k = 0; k < H; ++k) a_[k] = m1 ? z[H + k] : z[k];
        fexp_vec<H>(a_, e_);
#pragma unroll
        for (int k = 0; k < H; ++k) { r[k] = pair_pick(e_[k], false, m1); r[H + k] = pair_pick(e_[k], true, m1); }
    }
    if (rf) {
#pragma unroll
        for (int j = 0; j < NR; ++j) rf[j * STRIDE] = r[j];
    }
    CRNN_SCHED_FENCE();
#pragma unroll
    for (int i = 0; i < H; ++i) {
        double a = 0.0;
#pragma unroll
        for (int j = 0; j < NR; ++j) a = fma(th[L_::wo(0, j) + ln.ci[i]], r[j], a);
        pt.fo[i] = ln.ow[i] ? a * kc->gsc[ln.ci[i]] * pt.irho : 0.0;
    }
}

// This lane's rows of W = I - gam J(u_n) (row of slot i in A[i][0 .. NS-1]) and its species of ft = df/dt at the point.
template <int NS, int NR, int STRIDE>
__device__ __forceinline__ void hy_jac_ft2(const double *th, const KConst *kc, const HyPoint2<NS, NR> &pt, const double *rf, const double gam,
                                           const double ld, const double xEd, const double xLd, const bool m1, const HyLane<(NS + 1) / 2> &ln,
                                           double (&A)[(NS + 1) / 2][NS], double (&fto)[(NS + 1) / 2]) {
    using L_ = LayH<NS, NR>;
    constexpr int H = (NS + 1) / 2;
    double gx[NS], sg[NS], Bj[NR], zd[NR];
    {
        double gxo[H], sgo[H];
#pragma unroll
        for (int i = 0; i < H; ++i) {
            const bool iy = ln.ow[i] && ((pt.cY >> ln.ci[i]) & 1u), ic = (pt.cC >> ln.ci[i]) & 1u;
            gxo[i] = (iy && ic) ? frcp(pt.Yo[i]) : 0.0;
            sgo[i] = iy ? kc->imw[ln.ci[i]] * pt.iS : 0.0;
        }
        pair_gather<NS, H>(gxo, m1, gx);
        pair_gather<NS, H>(sgo, m1, sg);
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        double b = 0.0;
#pragma unroll
        for (int i = 0; i < H; ++i) b += (ln.ow[i] && ((pt.cC >> ln.ci[i]) & 1u)) ? th[L_::wi(0, j) + ln.ci[i]] : 0.0;
        b = pair_sum(b);
        Bj[j] = b;
        zd[j] = fma(b, ld, fma(th[L_::wi(NS, j)], xEd, th[L_::wi(NS + 1, j)] * xLd));
    }
#pragma unroll
    for (int i = 0; i < H; ++i) {
        // theta re-read row by row (90 broadcast ds_reads): merged across the rows it would pin 180 VGPRs
        unsigned z_ = 0;
        asm volatile("" : "+s"(z_));
        const double *const thr = th + z_;
        const double Gi = ln.ow[i] ? kc->gsc[ln.ci[i]] * pt.irho : 0.0;
        double a[NR], tB = 0.0, tz = 0.0;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            a[j] = Gi * thr[L_::wo(0, j) + ln.ci[i]] * rf[j * STRIDE + z_];
            tB = fma(a[j], Bj[j], tB);
            tz = fma(a[j], zd[j], tz);
        }
        fto[i] = ln.ow[i] ? fma(-pt.fo[i], ld, tz) : 0.0;
#pragma unroll
        for (int c = 0; c < NS; ++c) {
            double s_ = 0.0;
#pragma unroll
            for (int j = 0; j < NR; ++j) s_ = fma(a[j], thr[L_::wi(c, j)], s_);
            const double Jic = fma(gx[c], s_, -sg[c] * (tB - pt.fo[i]));
            const bool diag = (c == 2 * i) ? !m1 : ((c == 2 * i + 1) ? m1 : false);
            A[i][c] = (diag ? 1.0 : 0.0) - gam * Jic;
            if (c % 3 == 2) CRNN_SCHED_FENCE();   // the scheduler would issue the row's 90 theta reads up front (180 VGPRs)
        }
        // one row at a time: the row's entries are pinned here (volatile asm keeps its order with the next row's z_), or the
        // optimiser sinks all 450 multiply-adds below all 450 theta reads and spills the operands
        opaque(A[i]);
        CRNN_SCHED_FENCE();
    }
}

// P A = L U, rows distributed (slot i of lane m = row 2 i + m), in place in registers; ros23_kernel.hpp's lu_factor operation
// for operation (first maximum as pivot, l = a_ik / a_kk by reciprocal, fma(-l, a_kc, a_ic)), so the factors carry its bits.
template <int NS>
__device__ __forceinline__ bool lu2_factor(double (&A)[(NS + 1) / 2][NS], const bool m1, double (&dinv)[NS], unsigned long long &piv, bool &anyp) {
    constexpr int H = (NS + 1) / 2;
    const int mo = m1 ? 1 : 0;
    static_assert(NS <= 16, "pivot rows are packed four bits each");
    bool ok = true;
    anyp = false;
    piv = 0;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int ks = k >> 1;
        const bool kodd = (k & 1) != 0;
        double best = fabs(pair_pick(A[ks][k], kodd, m1));
        int p = k;
#pragma unroll
        for (int i = 0; i < H; ++i) {
            if (2 * i + 1 <= k) continue;                    // neither lane's row is below the pivot row
            if (2 * i >= NS) continue;
            bool act = (2 * i > k) ? true : m1;              // 2 i == k: lane 0's slot is the pivot row itself
            if (2 * i + 1 >= NS) act = act && !m1;           // lane 1's padding slot
            const double v = fabs(A[i][k]);
            if (act && v > best) { best = v; p = 2 * i + mo; }
        }
        {
            const double ob = pair_other(best);
            const int op = pair_other_i(p);
            if (ob > best || (ob == best && op < p)) { best = ob; p = op; }
        }
        piv |= (unsigned long long)(unsigned)p << (4 * k);
        const bool need = (p != k);
        anyp = anyp || need;
#ifndef HY2_NO_SWAP
        if (__builtin_amdgcn_ballot_w64(need) != 0) {        // rare: skipped by the whole wave when no pair swaps
            if (need) {
#pragma unroll
                for (int c = 0; c < NS; ++c) {
                    const double rk = pair_pick(A[ks][c], kodd, m1);
                    double mine_p = 0.0;
#pragma unroll
                    for (int i = 0; i < H; ++i) mine_p = (2 * i + mo == p) ? A[i][c] : mine_p;
                    const double oth_p = pair_other(mine_p);
                    const double rp = ((p & 1) == mo) ? mine_p : oth_p;
                    A[ks][c] = (m1 == kodd) ? rp : A[ks][c];
#pragma unroll
                    for (int i = 0; i < H; ++i) A[i][c] = (2 * i + mo == p) ? rk : A[i][c];
                }
            }
        }
#endif
        double rowk[NS];
#pragma unroll
        for (int c = k; c < NS; ++c) rowk[c] = pair_pick(A[ks][c], kodd, m1);
        ok = ok && (rowk[k] != 0.0);
        const double inv = frcp(rowk[k]);
        dinv[k] = inv;
#pragma unroll
        for (int i = 0; i < H; ++i) {
            if (2 * i + 1 <= k) continue;
            if (2 * i >= NS) continue;
            const double l = A[i][k] * inv;
            if (2 * i > k) {
                A[i][k] = l;
#pragma unroll
                for (int c = k + 1; c < NS; ++c) A[i][c] = fma(-l, rowk[c], A[i][c]);
            } else if (2 * i + 1 < NS && m1) {               // 2 i == k: only lane 1's row (k + 1) is below the pivot row
                A[i][k] = l;
#pragma unroll
                for (int c = k + 1; c < NS; ++c) A[i][c] = fma(-l, rowk[c], A[i][c]);
            }
        }
        CRNN_SCHED_FENCE();
    }
    return ok;
}

// b <- P b (fwd) or P^T b (!fwd) on a distributed vector: rare (a pivoting wave), done on a gathered copy
template <int NS>
__device__ __forceinline__ void lu2_permute(const unsigned long long piv, const bool fwd, const bool m1, double (&b)[(NS + 1) / 2]) {
    constexpr int H = (NS + 1) / 2;
    double f[NS];
    pair_gather<NS, H>(b, m1, f);
    if (fwd) {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int p = (int)((piv >> (4 * k)) & 15u);
#pragma unroll
            for (int i = k + 1; i < NS; ++i) {
                const bool sw = (p == i);
                const double bk = f[k], bi = f[i];
                f[k] = sw ? bi : bk;
                f[i] = sw ? bk : bi;
            }
        }
    } else {
#pragma unroll
        for (int k = NS - 1; k >= 0; --k) {
            const int p = (int)((piv >> (4 * k)) & 15u);
#pragma unroll
            for (int i = k + 1; i < NS; ++i) {
                const bool sw = (p == i);
                const double bk = f[k], bi = f[i];
                f[k] = sw ? bi : bk;
                f[i] = sw ? bk : bi;
            }
        }
    }
    pair_own<NS, H>(f, m1, b);
}

// W x = b, axpy form (hychem_kernel.hpp's lu_solve_lds operation for operation)
template <int NS>
__device__ __forceinline__ void lu2_solve(const double (&A)[(NS + 1) / 2][NS], const double (&dinv)[NS], const unsigned long long piv,
                                          const bool wave_pivots, const bool m1, double (&b)[(NS + 1) / 2]) {
    constexpr int H = (NS + 1) / 2;
    if (wave_pivots) lu2_permute<NS>(piv, true, m1, b);
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const double a = pair_pick(b[k >> 1], (k & 1) != 0, m1);
#pragma unroll
        for (int i = 0; i < H; ++i) {
            if (2 * i + 1 <= k || 2 * i >= NS) continue;
            if (2 * i > k) b[i] = fma(-A[i][k], a, b[i]);
            else b[i] = m1 ? fma(-A[i][k], a, b[i]) : b[i];
        }
    }
#pragma unroll
    for (int k = NS - 1; k >= 0; --k) {
        const bool kodd = (k & 1) != 0;
        const double a = pair_pick(b[k >> 1], kodd, m1) * dinv[k];
        b[k >> 1] = (m1 == kodd) ? a : b[k >> 1];
#pragma unroll
        for (int i = 0; i < H; ++i) {
            if (2 * i >= k) continue;                        // neither lane's row is above row k
            if (2 * i + 1 < k) b[i] = fma(-A[i][k], a, b[i]);
            else b[i] = m1 ? b[i] : fma(-A[i][k], a, b[i]);  // 2 i + 1 == k: lane 1's slot is row k itself
        }
    }
}

// W^T x = b:  x = P^T L^-T U^-T b, dot form -- each lane sums over ITS rows, the pair adds the two partial sums
template <int NS>
__device__ __forceinline__ void lu2_solve_T(const double (&A)[(NS + 1) / 2][NS], const double (&dinv)[NS], const unsigned long long piv,
                                            const bool wave_pivots, const bool m1, double (&b)[(NS + 1) / 2]) {
    constexpr int H = (NS + 1) / 2;
#pragma unroll
    for (int c = 0; c < NS; ++c) {                           // U^T y = b: y_c = (b_c - sum_{k < c} U_kc y_k) / U_cc
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < H; ++i) {
            if (2 * i >= c) continue;
            if (2 * i + 1 < c) s = fma(A[i][c], b[i], s);
            else s = m1 ? s : fma(A[i][c], b[i], s);         // 2 i + 1 == c: lane 1's slot is row c itself
        }
        if (c > 0) s = pair_sum(s);
        const double y = (b[c >> 1] - s) * dinv[c];
        b[c >> 1] = (m1 == ((c & 1) != 0)) ? y : b[c >> 1];
    }
#pragma unroll
    for (int c = NS - 2; c >= 0; --c) {                      // L^T x = y: x_c = y_c - sum_{k > c} L_kc x_k
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < H; ++i) {
            if (2 * i + 1 <= c || 2 * i >= NS) continue;
            if (2 * i > c) { if (2 * i + 1 < NS) s = fma(A[i][c], b[i], s); else s = m1 ? s : fma(A[i][c], b[i], s); }
            else s = m1 ? fma(A[i][c], b[i], s) : s;         // 2 i == c: lane 0's slot is row c itself
        }
        s = pair_sum(s);
        const double x = b[c >> 1] - s;
        b[c >> 1] = (m1 == ((c & 1) != 0)) ? x : b[c >> 1];
    }
    if (wave_pivots) lu2_permute<NS>(piv, false, m1, b);
}

template <int NS, int NR, bool GRAD, int BLOCK>
__global__ __launch_bounds__(BLOCK) void hychem2_kernel(const SolveParams prm, const double *__restrict__ theta, const HyParams hp) {
    using L_ = LayH<NS, NR>;
    constexpr int NTH = L_::NTH;
    constexpr int H = (NS + 1) / 2;
    constexpr int GPB = BLOCK / 2;
    constexpr int RECW = NS + 2;
    // (separate LDS objects on purpose: carved out of ONE array -- theta first, so that its address folds into ds_read's immediate
    //  offset -- the kernel is 8 % slower: every frame store may then alias every theta load)
    __shared__ double kc_lds[kNConst];
    __shared__ double ts_lds[kMaxSave];
    __shared__ double th_lds[NTH];
    // the lane's LDS frame (slot k at fr[k BLOCK]; nobody else touches the column): what a step needs again much later is
    // parked here instead of being carried in registers (a value the allocator spills goes to scratch memory, and a wavefront
    // that runs alone on its SIMD waits out every reload: 700 bytes of scratch cost this kernel 40 % of its time) -- the rates
    // of the FSAL point / of u_n (0-9) and of the new point / u_mid (10-19); reverse sweep: x, Y of the two points (20-39),
    // k1, k2 - k1 (40-49), their scalars (50-57)
    constexpr int NFR = 65;      // 58-64: the FSAL point's Y, irho, iS (forward sweep).  Lane-major, pitch 65 doubles: every slot is an
                                 // immediate offset from ONE address register (slot-major, 2 KB apart, half the slots were out of the
                                 // 64 KB offset range and each had its own address register -- spilled), and the odd pitch is conflict-free
    __shared__ double fr_lds[NFR * BLOCK];
    // Gradient accumulation (GRAD): the sum over the wavefront's trajectories of the step's outer products  x (features) y^T
    // (reactions) is a contraction over the LANE axis -- v_mfma_f64_16x16x4_f64 does it: the lanes stage x (12 rows: 9 species,
    // -1/(R T), log T, 1 for w_b) and y (10 columns) of one term in this buffer, [row][trajectory] with a row pitch of 33 so
    // that the 16 rows an operand load touches fall into different banks; eight MFMAs (four trajectories of K each) add the
    // term to the wavefront's 16 x 16 tile.  The tiles (w_in | w_b and w_out: 2 x 4 doubles per lane) live in registers for the
    // whole kernel and are written out once; no accumulator ever goes to HBM.
    constexpr int ST_P = 33, ST_X = 12 * ST_P, ST_W = ST_X + NR * ST_P;
    __shared__ double st_lds[GRAD ? (BLOCK / 64) * ST_W : 1];
    const int tid = threadIdx.x;
    double *const fr = fr_lds + tid * NFR;
#define FR(k_) fr[(k_)]
    // wave-level ordering of LDS traffic between the lanes of the wavefront (one lane's stores, another lane's loads)
#define HY2_LDS_SYNC()                                                   \
    do {                                                                 \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");           \
        __builtin_amdgcn_wave_barrier();                                 \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");           \
    } while (0)
    // hychem_kernel.hpp's HY_FRESH_* with the opaque zero in an SGPR: the pointer stays uniform, so when registers run out it is
    // re-formed by one v_mov from the SGPR instead of being reloaded from scratch (as a VGPR value it was the most-reloaded
    // spill slot of the kernel: ~25 reloads per step pair)
#define HY2_FRESH_THETA(ptr)                    \
    do {                                        \
        unsigned z_ = 0;                        \
        asm volatile("" : "+s"(z_));            \
        (ptr) = th_lds + z_;                    \
    } while (0)
#define HY2_FRESH_KC(ptr)                                            \
    do {                                                             \
        unsigned z_ = 0;                                             \
        asm volatile("" : "+s"(z_));                                 \
        (ptr) = reinterpret_cast<const KConst *>(kc_lds + z_);       \
    } while (0)
    const int lane = tid & 63;
    const bool m1 = (lane & 1) != 0;
    const int gib = tid >> 1, giw = lane >> 1;
    for (int idx = tid; idx < kNConst; idx += BLOCK) kc_lds[idx] = reinterpret_cast<const double *>(prm.kc)[idx];
    for (int idx = tid; idx < hp.n_save_total; idx += BLOCK) ts_lds[idx] = prm.tsave[idx];
    for (int idx = tid; idx < NTH; idx += BLOCK) th_lds[idx] = theta[idx];
    if (GRAD) {   // (the MFMA stage reads the frame of lanes that have not run a step yet: make what they find finite)
        for (int k = 0; k < NFR; ++k) fr_lds[threadIdx.x * NFR + k] = 0.0;
    }
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

    hy_v4d acc_wi = {0.0, 0.0, 0.0, 0.0}, acc_wo = {0.0, 0.0, 0.0, 0.0}, acc_wi2 = {0.0, 0.0, 0.0, 0.0}, acc_wo2 = {0.0, 0.0, 0.0, 0.0};
    double *const st = st_lds + (GRAD ? (tid >> 6) * ST_W : 0);
    const int st_row = lane & 15, st_k = lane >> 4;       // MFMA operand (row or column, k) of this lane
    const double d_ = 0.29289321881345248, c32 = 7.4142135623730950, inv12d = 2.4142135623730950;
    const int nsave = prm.n_save, Dfull = hp.n_save_total;
    const double tend = to_sgpr(ts_lds[nsave - 1]), ts0 = to_sgpr(ts_lds[0]), t0 = to_sgpr(kc->t0);
    const double dtmax = to_sgpr(tend - t0);
    const double lqinit = to_sgpr(flog(kc->qoldinit));
    const double inv_qmax = to_sgpr(1.0 / kc->qmax), inv_qmin = to_sgpr(1.0 / kc->qmin);   // (the quotients the one-lane kernel forms each step)
    const bool start_saved = (ts0 == t0);
    double *const tape = hp.tape + (size_t)((size_t)blockIdx.x * GPB + gib) * hp.tape_cap * RECW;
#ifdef HY_PROF
    unsigned long long prof_acc[16] = {0}, prof_last = __builtin_readcyclecounter();
#endif

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

        // ================================================================== forward sweep
        double u[H];
        HyPoint2<NS, NR> p0;     // FSAL point (u, t): only f stays in registers (f0), the rest is parked in the frame
        double f0[H];
        unsigned f0cY = 0, f0cC = 0;     // the FSAL point's clamp masks
        double t = t0, dt = 0.0, lqold = lqinit;
        int iter = 0, jsave = 0, nacc = 0, nrej = 0;
        int rc = valid ? -1 : 0;
#pragma unroll
        for (int i = 0; i < H; ++i) u[i] = ln.ow[i] ? prm.u0[(size_t)ln.ci[i] * prm.B + b] : 0.0;
        {
            double T, P, Td, Pd;
            tab(t0, T, P, Td, Pd);
            hy_point2<NS, NR, 1>(th, kc, hp.inv_R, u, T, P, m1, ln, p0, fr);
#pragma unroll
            for (int i = 0; i < H; ++i) FR(58 + i) = p0.Yo[i];
            FR(63) = p0.irho; FR(64) = p0.iS;
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
            const double dt1 = (dm <= 1e-15) ? fmax(1e-6, dt0 * 1e-3) : exp(-0.5 * (4.605170185988091368 + flog(dm)));
            dt = fmax(kc->dtmin, fmin(fmin(100.0 * dt0, dt1), dtmax));
#pragma unroll
            for (int i = 0; i < H; ++i) f0[i] = p0.fo[i];
        }
        if (start_saved) {
            if (valid && prm.pred) {
#pragma unroll
                for (int i = 0; i < H; ++i) {
                    if (ln.ow[i]) {
                        double v = u[i];
                        if (prm.clamp_pred) v = clampv(v, -kc->ub, kc->ub);
                        prm.pred[((size_t)0 * NS + ln.ci[i]) * prm.B + b] = v;
                    }
                }
            }
            jsave = 1;
        }

        double pf_loss = 0.0;    // primal launch (GRAD = false): the loss, accumulated at the save points of the forward sweep
        while (__builtin_amdgcn_ballot_w64(rc < 0) != 0) {
            if (rc < 0) {
                ++iter;
                bool last = false;
                if (jsave >= nsave) rc = 0;
                else if (iter > prm.maxiters) rc = 1;
                if (t + dt * (1.0 + 1e-13) >= tend) { dt = tend - t; last = true; }
                if (rc < 0 && (!(dt > kc->dtmin) || t + dt == t)) rc = 2;
                if (rc < 0) {
                    HY2_FRESH_THETA(th); HY2_FRESH_KC(kc);
                    const double gam = d_ * dt;
                    const double tnew = last ? tend : t + dt;
                    double T, P, Td, Pd;
                    tab(t, T, P, Td, Pd);
                    double A[H][NS], dinv[NS], ft[H];
                    unsigned long long piv;
                    bool anyp;
                    HY_T(0);
                    {
                        unsigned zf_ = 0;
                        asm volatile("" : "+v"(zf_));
                        const double *const fq = fr + zf_;
#pragma unroll
                        for (int i = 0; i < H; ++i) { p0.Yo[i] = fq[58 + i]; p0.fo[i] = f0[i]; }
                        p0.irho = fq[63]; p0.iS = fq[64];
                        p0.cY = f0cY; p0.cC = f0cC;
                    }
                    hy_jac_ft2<NS, NR, 1>(th, kc, p0, fr, gam, Pd * frcp(P) - Td * frcp(T), -hp.inv_R * Td * frcp(T * T), Td * frcp(T), m1, ln, A, ft);
                    CRNN_SCHED_FENCE();
                    HY_T(1);
                    const bool okf = lu2_factor<NS>(A, m1, dinv, piv, anyp);
                    const bool wp = __builtin_amdgcn_ballot_w64(anyp) != 0;
                    HY_T(2);
                    double k1[H], dk[H], unew[H], f1[H];
#pragma unroll
                    for (int i = 0; i < H; ++i) k1[i] = fma(gam, ft[i], f0[i]);
                    lu2_solve<NS>(A, dinv, piv, wp, m1, k1);
                    CRNN_SCHED_FENCE();
                    HY_T(3);
                    HY2_FRESH_THETA(th); HY2_FRESH_KC(kc);
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
                    HY_T(4);
                    lu2_solve<NS>(A, dinv, piv, wp, m1, dk);
                    HY_T(3);
#pragma unroll
                    for (int i = 0; i < H; ++i) unew[i] = fma(dt, k1[i] + dk[i], u[i]);
                    CRNN_SCHED_FENCE();
                    HyPoint2<NS, NR> p2;
                    HY2_FRESH_THETA(th); HY2_FRESH_KC(kc);
                    {
                        double T2, P2, a_, b_;
                        tab(tnew, T2, P2, a_, b_);
                        hy_point2<NS, NR, 1>(th, kc, hp.inv_R, unew, T2, P2, m1, ln, p2, fr + 10);
                    }
                    CRNN_SCHED_FENCE();
                    HY_T(4);
                    double k3[H];
#pragma unroll
                    for (int i = 0; i < H; ++i) {
                        const double k2i = k1[i] + dk[i];
                        k3[i] = fma(dt, ft[i], p2.fo[i] - c32 * (k2i - f1[i]) - 2.0 * (k1[i] - f0[i]));
                    }
                    lu2_solve<NS>(A, dinv, piv, wp, m1, k3);
                    HY_T(3);
                    double es = 0.0;
                    int fin = okf ? 1 : 0;
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
                    const bool finite = pair_and(fin) != 0;
                    HY_T(14);
                    if (!finite) rc = 3;
                    else {
                        const bool ee_zero = (es == 0.0);
                        const double lEE = 0.5 * flog_ctl(ee_zero ? 1.0 : es);
                        const double lq11 = kc->beta1 * lEE;
                        double q = ee_zero ? inv_qmax
                                           : fmax(inv_qmax, fmin(inv_qmin, fexp_ctl(lq11 - kc->beta2 * lqold) / kc->gamma));
                        if (es <= 1.0) {
                            if (GRAD && nacc >= hp.tape_cap) {
                                rc = 5;
                                if (!m1) atomicAdd(hp.overflow, 1u);
                            } else {
                                if (GRAD) {
                                    CRNN_CHK(nacc >= 0 && nacc < hp.tape_cap, 22);
                                    double *rec = tape + (size_t)nacc * RECW;
                                    if (!m1) { rec[0] = t; rec[1] = dt; }
#pragma unroll
                                    for (int i = 0; i < H; ++i)
                                        if (ln.ow[i]) rec[2 + ln.ci[i]] = u[i];
                                }
                                ++nacc;
                                while (jsave < nsave) {
                                    const double ts = ts_lds[jsave];
                                    if (!(ts <= tnew)) break;
                                    if (prm.pred || !GRAD) {
                                        const bool at_end = (ts == tnew);
                                        const double Th = at_end ? 1.0 : (ts - t) / dt;
                                        const double c1 = at_end ? 0.0 : Th * (1.0 - Th) * inv12d;
                                        const double c2 = at_end ? 1.0 : Th * (Th - 2.0 * d_) * inv12d;
                                        CRNN_CHK((int64_t)(jsave + 1) * prm.n_obs <= prm.row_stride && jsave < Dfull, 23);
                                        const double *prow = prm.data + (size_t)b * prm.row_stride + (size_t)jsave * prm.n_obs;
#pragma unroll
                                        for (int i = 0; i < H; ++i) {
                                            if (ln.ow[i]) {
                                                const double k2i = k1[i] + dk[i];
                                                double v = at_end ? unew[i] : fma(dt, fma(c1, k1[i], c2 * k2i), u[i]);
                                                if (prm.clamp_pred) v = clampv(v, -kc->ub, kc->ub);
                                                if (prm.pred) prm.pred[((size_t)jsave * NS + ln.ci[i]) * prm.B + b] = v;
                                                if (!GRAD) {     // primal launch: the loss term of this save point, here (no tape, no reverse sweep)
                                                    const int dr = (int)kc->drow[ln.ci[i]];
                                                    if (dr >= 0) {
                                                        const double rr = (prow[dr] - v) * kc->inv_yscale[ln.ci[i]];
                                                        pf_loss = (prm.loss_kind == 0) ? pf_loss + fabs(rr) : fma(rr, rr, pf_loss);
                                                    }
                                                }
                                            }
                                        }
                                    }
                                    ++jsave;
                                }
#pragma unroll
                                for (int i = 0; i < H; ++i) u[i] = unew[i];
#pragma unroll
                                for (int i = 0; i < H; ++i) { f0[i] = p2.fo[i]; FR(58 + i) = p2.Yo[i]; }
                                FR(63) = p2.irho; FR(64) = p2.iS; f0cY = p2.cY; f0cC = p2.cC;
                                {   // the new point's rates (slots 10-19) become the FSAL point's (0-9)
                                    double rr_[NR];
#pragma unroll
                                    for (int j = 0; j < NR; ++j) rr_[j] = FR(10 + j);
#pragma unroll
                                    for (int j = 0; j < NR; ++j) FR(j) = rr_[j];
                                }
                                t = tnew;
                                if (q >= kc->qsteady_min && q <= kc->qsteady_max) q = 1.0;
                                lqold = ee_zero ? lqinit : fmax(lEE, lqinit);
                                dt = fmin(dt / q, dtmax);
                                if (jsave >= nsave) rc = 0;
                            }
                        } else {
                            ++nrej;
                            dt = dt / fmin(inv_qmin, fexp_ctl(lq11) / kc->gamma);
                        }
                    }
                    HY_T(15);
                }
            }
        }

        // ================================================================== reverse sweep: loss (+ adjoint)
        const int n_saved = jsave;
        const int jlo = start_saved ? 1 : 0;
        double lam[H];
#pragma unroll
        for (int i = 0; i < H; ++i) lam[i] = 0.0;
        double loss_sum = GRAD ? 0.0 : pf_loss;   // this lane's species only; the pair's sum is formed at the end
        double tnew = t;
        int s = (valid && GRAD) ? nacc - 1 : -1;     // (primal launch: no reverse sweep)
        // the trajectory's weight in the batch gradient, 1 / (n_obs n_saved) (reduce_gacc_kernel's scale): carried by the loss seeds,
        // so every adjoint quantity -- linear in the seeds -- arrives at the wavefront's tiles already weighted
        const double gscale = n_saved > 0 ? 1.0 / ((double)prm.n_obs * (double)n_saved) : 0.0;
        const double *const drows = prm.data + (size_t)b * prm.row_stride;
        int doff[H];
        bool obs[H];
#pragma unroll
        for (int i = 0; i < H; ++i) { const int dr = ln.ow[i] ? (int)kc->drow[ln.ci[i]] : -1; obs[i] = dr >= 0; doff[i] = obs[i] ? dr : 0; }
        // tape records are fetched TWO steps ahead: record s - 2 is requested in step s and consumed one iteration later, at the
        // same point -- a load awaited in the step that requests it stalls the lane for a full memory latency per step
        double rt = 0.0, rdt = 0.0, ru[H], qt = 0.0, qdt = 0.0, qu[H];
        auto fetch_rec = [&](int idx, double &t_, double &dt_, double (&u_)[H]) {
            CRNN_CHK(idx < hp.tape_cap, 25);
            const double *rec = tape + (size_t)(idx > 0 ? idx : 0) * RECW;
            t_ = rec[0]; dt_ = rec[1];
#pragma unroll
            for (int i = 0; i < H; ++i) { const double v = rec[2 + ln.ci[i]]; u_[i] = ln.ow[i] ? v : 0.0; }
        };
        fetch_rec(s, rt, rdt, ru);
        fetch_rec(s - 1, qt, qdt, qu);

        while (__builtin_amdgcn_ballot_w64(s >= 0) != 0) {
            const bool act = (s >= 0);
            // what the step hands to the MFMA stage below (outside the divergent region: the matrix unit ignores EXEC)
            // (most of it waits in the lane's frame; registers carry the direction data, v~, w~ and the lane's half of one factor)
            double c2h[NR / 2], sxw[H], sxv[H], svt[H], swt[H], s_xEd = 0.0, s_xLd = 0.0;
            if (s >= 0) {
                const double tn = rt, h = rdt;
                double un[H];
#pragma unroll
                for (int i = 0; i < H; ++i) un[i] = ru[i];
                // ---- re-form the step
                HY2_FRESH_THETA(th); HY2_FRESH_KC(kc);
                const double gam = d_ * h;
                double T, P, Td, Pd;
                tab(tn, T, P, Td, Pd);
                const double ld = Pd * frcp(P) - Td * frcp(T), xEd = -hp.inv_R * Td * frcp(T * T), xLd = Td * frcp(T);
                HyPoint2<NS, NR> pn, pm;
                HY_T(5);
                hy_point2<NS, NR, 1>(th, kc, hp.inv_R, un, T, P, m1, ln, pn, fr);
                HY_T(6);
                if (GRAD) {
#pragma unroll
                    for (int i = 0; i < H; ++i) FR(20 + i) = pn.xo[i];
                    FR(50) = pn.xE; FR(51) = pn.xL;
                }
                opaque(pn.Yo); opaque(pn.fo); opaque(pn.irho); opaque(pn.iS);
                HY2_FRESH_THETA(th); HY2_FRESH_KC(kc);
                double A[H][NS], dinv[NS], ft[H];
                unsigned long long piv;
                bool anyp;
                double k1[H], dk[H];
                hy_jac_ft2<NS, NR, 1>(th, kc, pn, fr, gam, ld, xEd, xLd, m1, ln, A, ft);
#pragma unroll
                for (int i = 0; i < H; ++i) k1[i] = fma(gam, ft[i], pn.fo[i]);
                opaque(k1);
                if (GRAD) {
#pragma unroll
                    for (int i = 0; i < H; ++i) FR(25 + i) = pn.Yo[i];
                    FR(52) = pn.irho; FR(53) = pn.iS;
                }
                CRNN_SCHED_FENCE();
                HY_T(7);
                (void)lu2_factor<NS>(A, m1, dinv, piv, anyp);
                const bool wp = __builtin_amdgcn_ballot_w64(anyp) != 0;
                HY_T(8);
                lu2_solve<NS>(A, dinv, piv, wp, m1, k1);
                HY_T(9);
                HY2_FRESH_THETA(th); HY2_FRESH_KC(kc);
                {
                    double u1[H], T1, P1, a_, b_;
#pragma unroll
                    for (int i = 0; i < H; ++i) u1[i] = fma(0.5 * h, k1[i], un[i]);
                    tab(tn + 0.5 * h, T1, P1, a_, b_);
                    hy_point2<NS, NR, 1>(th, kc, hp.inv_R, u1, T1, P1, m1, ln, pm, fr + 10);
                    if (GRAD) {
#pragma unroll
                        for (int i = 0; i < H; ++i) { FR(30 + i) = pm.xo[i]; FR(35 + i) = pm.Yo[i]; }
                        FR(54) = pm.xE; FR(55) = pm.xL; FR(56) = pm.irho; FR(57) = pm.iS;
                    }
                    opaque(pm.fo);
                }
#pragma unroll
                for (int i = 0; i < H; ++i) dk[i] = pm.fo[i] - k1[i];
                HY_T(6);
                lu2_solve<NS>(A, dinv, piv, wp, m1, dk);
                CRNN_SCHED_FENCE();
                HY_T(9);

                // ---- loss and seeds at the save points inside (tn, tnew]: each lane its own species
                double A_[H], B1[H], B2[H];
#pragma unroll
                for (int i = 0; i < H; ++i) { A_[i] = 0.0; B1[i] = 0.0; B2[i] = 0.0; }
                while (jsave > jlo && ts_lds[jsave - 1] > tn) {
                    const double ts = ts_lds[jsave - 1];
                    const bool at_end = (ts == tnew);
                    const double Th = at_end ? 1.0 : (ts - tn) / h;
                    const double c1 = at_end ? 0.0 : Th * (1.0 - Th) * inv12d;
                    const double c2 = at_end ? 1.0 : Th * (Th - 2.0 * d_) * inv12d;
                    const double *row = drows + (size_t)(jsave - 1) * prm.n_obs;
#pragma unroll
                    for (int i = 0; i < H; ++i) {
                        if (obs[i]) {
                            const double k2i = k1[i] + dk[i];
                            double v = at_end ? fma(h, k2i, un[i]) : fma(h, fma(c1, k1[i], c2 * k2i), un[i]);
                            double mask = 1.0;
                            if (prm.clamp_pred) {
                                mask = (v > kc->ub || v < -kc->ub) ? 0.0 : 1.0;
                                v = clampv(v, -kc->ub, kc->ub);
                            }
                            const double iy = kc->inv_yscale[ln.ci[i]];
                            const double rr = (row[doff[i]] - v) * iy;
                            double w;
                            if (prm.loss_kind == 0) { loss_sum += fabs(rr); w = signbit(rr) ? 1.0 : -1.0; }
                            else { loss_sum = fma(rr, rr, loss_sum); w = -2.0 * rr; }
                            w *= mask * iy * (GRAD ? gscale : 1.0);
                            A_[i] += w;
                            B1[i] = fma(w, h * c1, B1[i]);
                            B2[i] = fma(w, h * c2, B2[i]);
                        }
                    }
                    --jsave;
                }
                if (GRAD) {
#pragma unroll
                    for (int i = 0; i < H; ++i) { FR(40 + i) = k1[i]; FR(45 + i) = dk[i]; }
                }
                opaque(qt); opaque(qdt); opaque(qu);          // record s - 1 (requested one iteration ago) has arrived
                rt = qt; rdt = qdt;
#pragma unroll
                for (int i = 0; i < H; ++i) ru[i] = qu[i];
                fetch_rec(s - 2, qt, qdt, qu);                 // in flight until the next iteration
                HY_T(10);
                if (GRAD) {
                    unsigned zf_ = 0;
                    asm volatile("" : "+v"(zf_));          // the frame is re-read: the parked values are NOT kept in registers too
                    const double *const fq = fr + zf_;
#define FQ(k_) fq[(k_)]
                    const unsigned mcY = pm.cY, mcC = pm.cC;
                    const unsigned ncY = pn.cY, ncC = pn.cC;
                    CRNN_SCHED_FENCE();
                    double kb1[H], v[H], ub[H];
#pragma unroll
                    for (int i = 0; i < H; ++i) { v[i] = fma(h, lam[i], B2[i]); ub[i] = lam[i] + A_[i]; kb1[i] = B1[i] + v[i]; }
                    lu2_solve_T<NS>(A, dinv, piv, wp, m1, v);
                    HY_T(11);
                    double vto[H];      // (gsc .* v) of this lane's species
#pragma unroll
                    for (int i = 0; i < H; ++i) { kb1[i] -= v[i]; vto[i] = v[i] * kc->gsc[ln.ci[i]]; }
                    opaque(vto); opaque(kb1); opaque(ub);
                    CRNN_SCHED_FENCE();
                    HY2_FRESH_THETA(th); HY2_FRESH_KC(kc);
                    // -------- point u_mid: adjoint of v.f
                    {
                        double P2o[H], Psis[NR], psi = 0.0;
#pragma unroll
                        for (int i = 0; i < H; ++i) P2o[i] = 0.0;
                        const double m_irho = FQ(56);
#pragma unroll
                        for (int j = 0; j < NR; ++j) {
                            CRNN_SCHED_FENCE();
                            HY2_FRESH_THETA(th);      // per reaction: merged, the 10 x 12 theta reads would be issued up front
                            const double *wo_ = th + L_::wo(0, j), *wi_ = th + L_::wi(0, j);
                            double At = 0.0;
#pragma unroll
                            for (int i = 0; i < H; ++i) At = fma(vto[i], wo_[ln.ci[i]], At);
                            At = pair_sum(At);
                            const double ir = m_irho * FQ(10 + j);
                            const double Psi = At * ir;
                            psi += Psi;
                            Psis[j] = Psi;       // its w_in terms Psi_j x_mid join the u_n point's in the MFMA stage
#pragma unroll
                            for (int i = 0; i < H; ++i) P2o[i] = fma(Psi, wi_[ln.ci[i]], P2o[i]);   // (padding slot: never read)
                        }
                        double scp = 0.0;
#pragma unroll
                        for (int i = 0; i < H; ++i) scp += (ln.ow[i] && ((mcC >> ln.ci[i]) & 1u)) ? P2o[i] : 0.0;
                        scp = pair_sum(scp);
                        const double m_iS = FQ(57);
#pragma unroll
                        for (int i = 0; i < H; ++i) {
                            const bool iy = ln.ow[i] && ((mcY >> ln.ci[i]) & 1u), ic = (mcC >> ln.ci[i]) & 1u;
                            double m_ = iy ? kc->imw[ln.ci[i]] * m_iS * (psi - scp) : 0.0;
                            if (iy && ic) m_ = fma(P2o[i], frcp(FQ(35 + i)), m_);
                            ub[i] += m_;
                            kb1[i] = fma(0.5 * h, m_, kb1[i]);
                        }
                        // Psi_j -> frame slots 58-64 (the forward sweep's, idle now), 57 (iS of u_mid, consumed above) and 35-36 (Y_mid, likewise)
                        static_assert(NR == 10, "frame slots of Psi");
#pragma unroll
                        for (int j = 0; j < NR; ++j) FR(j < 7 ? 58 + j : (j == 7 ? 57 : 27 + j)) = Psis[j];
                    }
                    CRNN_SCHED_FENCE();
                    HY_T(12);
                    lu2_solve_T<NS>(A, dinv, piv, wp, m1, kb1);     // kb1 = w
                    HY_T(11);
                    opaque(kb1); opaque(ub); opaque(vto);
                    CRNN_SCHED_FENCE();
                    HY2_FRESH_THETA(th); HY2_FRESH_KC(kc);
                    // -------- point u_n: adjoint of w.f + gam ( v.Df[(dk,0)] + w.Df[(k1,1)] )
                    {
                        double wto[H], k1p[H], dkp[H], Yn[H];
#pragma unroll
                        for (int i = 0; i < H; ++i) {
                            wto[i] = kb1[i] * kc->gsc[ln.ci[i]];
                            k1p[i] = FQ(40 + i); dkp[i] = FQ(45 + i); Yn[i] = FQ(25 + i);
                        }
                        // (x of the two points and their -1/(RT), log T stay in the frame: slots 20-24, 30-34, 50, 51, 54, 55 feed the MFMA stage)
                        const double n_irho = FQ(52), n_iS = FQ(53), m_irho = FQ(56);
                        // direction data (this lane's species)
                        double Spv = 0.0, Spw = 0.0, xpvo[H], xpwo[H];
#pragma unroll
                        for (int i = 0; i < H; ++i) {
                            const double sg = (ln.ow[i] && ((ncY >> ln.ci[i]) & 1u)) ? kc->imw[ln.ci[i]] * n_iS : 0.0;
                            Spv = fma(sg, dkp[i], Spv);
                            Spw = fma(sg, k1p[i], Spw);
                        }
                        Spv = pair_sum(Spv);
                        Spw = pair_sum(Spw);
                        const double lpv = -Spv, lpw = ld - Spw;
#pragma unroll
                        for (int i = 0; i < H; ++i) {
                            const bool iy = ln.ow[i] && ((ncY >> ln.ci[i]) & 1u), ic = ln.ow[i] && ((ncC >> ln.ci[i]) & 1u);
                            const double gy = iy ? frcp(Yn[i]) : 0.0;
                            xpvo[i] = ic ? fma(gy, dkp[i], lpv) : 0.0;
                            xpwo[i] = ic ? fma(gy, k1p[i], lpw) : 0.0;
                        }
                        double PEo[H], P2vo[H], P2wo[H], SE = 0.0, psiv = 0.0, psiw = 0.0;
#pragma unroll
                        for (int i = 0; i < H; ++i) { PEo[i] = 0.0; P2vo[i] = 0.0; P2wo[i] = 0.0; }
#pragma unroll
                        for (int j = 0; j < NR; ++j) {
                            CRNN_SCHED_FENCE();
                            HY2_FRESH_THETA(th);      // per reaction: merged, the 10 x 12 theta reads would be issued up front
                            const double *wo_ = th + L_::wo(0, j), *wi_ = th + L_::wi(0, j);
                            double Av = 0.0, Aw = 0.0, zv = 0.0, zw = 0.0;
#pragma unroll
                            for (int i = 0; i < H; ++i) {
                                const double wij = wi_[ln.ci[i]];   // (padding slot: multiplies zeros / feeds sums read under the mask)
                                Av = fma(vto[i], wo_[ln.ci[i]], Av);
                                Aw = fma(wto[i], wo_[ln.ci[i]], Aw);
                                zv = fma(wij, xpvo[i], zv);
                                zw = fma(wij, xpwo[i], zw);
                            }
                            Av = pair_sum(Av); Aw = pair_sum(Aw); zv = pair_sum(zv);
                            zw = pair_sum(zw) + fma(wi_[NS], xEd, wi_[NS + 1] * xLd);
                            const double ir = n_irho * FQ(j);
                            const double Pv = Av * ir, Pw = Aw * ir;
                            const double yv = zv - lpv, yw = zw - lpw;
                            const double cw = fma(gam, yw, 1.0), cv = gam * yv;
                            const double E = fma(Pw, cw, Pv * cv);
                            SE += E; psiv += Pv; psiw += Pw;
                            const double irm = m_irho * FQ(10 + j);   // the u_mid point's irho r_j
                            // the step's gradient terms, per reaction (they meet the species factors in the MFMA stage):
                            //   w_in[m][j] += E x_n[m] + gam Pw x'_w[m] + gam Pv x'_v[m] + Psi x_mid[m]      (m: species, -1/(RT), log T)
                            //   w_b[j]     += E + Psi                 (Av irm = Psi: the same v . w_out contraction, the u_mid point's irho r_j)
                            //   w_out[i][j] += vt[i] (ir cv + irm) + wt[i] (ir cw)
                            // parked per reaction in frame slots whose contents have been consumed (the rates: read above; Y_n, k1,
                            // k2 - k1, the scalars: in registers since the top of the block); c2: the lane's half in registers
                            FR(j) = E; FR(10 + j) = gam * Pw; FR(40 + j) = gam * Pv;
                            FR(j < 5 ? 25 + j : (j < 8 ? 32 + j : 44 + j)) = fma(ir, cv, irm);
                            c2h[j % (NR / 2)] = (m1 == (j >= NR / 2)) ? ir * cw : c2h[j % (NR / 2)];
#pragma unroll
                            for (int i = 0; i < H; ++i) {
                                const double wij = wi_[ln.ci[i]];
                                PEo[i] = fma(E, wij, PEo[i]);
                                P2vo[i] = fma(Pv, wij, P2vo[i]);
                                P2wo[i] = fma(Pw, wij, P2wo[i]);
                            }
                        }
#pragma unroll
                        for (int i = 0; i < H; ++i) { sxw[i] = xpwo[i]; sxv[i] = xpvo[i]; svt[i] = vto[i]; swt[i] = wto[i]; }
                        s_xEd = xEd; s_xLd = xLd;
                        double scE = 0.0, scv = 0.0, scw = 0.0;
#pragma unroll
                        for (int i = 0; i < H; ++i) {
                            const bool ic = ln.ow[i] && ((ncC >> ln.ci[i]) & 1u);
                            scE += ic ? PEo[i] : 0.0;
                            scv += ic ? P2vo[i] : 0.0;
                            scw += ic ? P2wo[i] : 0.0;
                        }
                        scE = pair_sum(scE); scv = pair_sum(scv); scw = pair_sum(scw);
                        const double brk = (SE - scE) + gam * fma(Spw, scw - psiw, Spv * (scv - psiv));
#pragma unroll
                        for (int i = 0; i < H; ++i) {
                            const bool iy = ln.ow[i] && ((ncY >> ln.ci[i]) & 1u), ic = (ncC >> ln.ci[i]) & 1u;
                            double m_ = iy ? kc->imw[ln.ci[i]] * n_iS * brk : 0.0;
                            if (iy && ic) {
                                const double gy = frcp(Yn[i]);
                                m_ = fma(gy, PEo[i] - gam * gy * fma(P2wo[i], k1p[i], P2vo[i] * dkp[i]), m_);
                            }
                            lam[i] = ub[i] + m_;
                        }
                    }
                    HY_T(13);
#undef FQ
                }
                tnew = tn;
                --s;
            }
            if (GRAD) {
                // ---- the wavefront adds the step's six terms to its tiles (uniform control flow; inactive pairs stage zeros)
                HY_T(13);
                // this lane's half of the reactions' factors: reactions 5 m .. 5 m + 4
                unsigned zs_ = 0;
                asm volatile("" : "+v"(zs_));
                const double *const fs = fr + zs_;
                // staging addresses are formed here, per iteration, from opaque copies: as loop invariants they are computed once
                // per trajectory, do not fit the register file and come back from scratch (one memory latency each)
                int g_ = giw, sr_ = st_row, sk_ = st_k;
                asm volatile("" : "+v"(g_));
                asm volatile("" : "+v"(sr_));
                asm volatile("" : "+v"(sk_));
                double *const stw = st + g_;
                auto stage_term = [&](const double (&xs)[H], const double xE_, const double xL_, const double x1_, const double (&yh)[NR / 2],
                                      const int nrows, hy_v4d &acc_a, hy_v4d &acc_b) {
                    HY2_LDS_SYNC();          // the previous term's operand loads are done
#pragma unroll
                    for (int i = 0; i < H; ++i)
                        if (ln.ow[i]) stw[ln.ci[i] * ST_P] = xs[i];
                    if (!m1 && nrows > NS) {
                        stw[NS * ST_P] = xE_;
                        stw[(NS + 1) * ST_P] = xL_;
                        stw[(NS + 2) * ST_P] = x1_;
                    }
#pragma unroll
                    for (int k = 0; k < NR / 2; ++k) stw[ST_X + ((m1 ? NR / 2 : 0) + k) * ST_P] = act ? yh[k] : 0.0;
                    HY2_LDS_SYNC();
                    const bool arow = sr_ < nrows, bcol = sr_ < NR;
                    const double *pa = st + (arow ? sr_ : 0) * ST_P + sk_;
                    const double *pb = st + ST_X + (bcol ? sr_ : 0) * ST_P + sk_;
#pragma unroll
                    for (int n = 0; n < 8; n += 2) {      // two independent accumulation chains
                        const double a0 = arow ? pa[4 * n] : 0.0, b0 = bcol ? pb[4 * n] : 0.0;
                        const double a1 = arow ? pa[4 * n + 4] : 0.0, b1 = bcol ? pb[4 * n + 4] : 0.0;
                        acc_a = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc_a, 0, 0, 0);
                        acc_b = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc_b, 0, 0, 0);
                    }
                };
                // (inactive pairs: their reactions' factors are staged as zeros; the species factors are made finite -- stale frame
                //  contents are, the frame is cleared at kernel start -- so that 0 x them adds nothing)
                double xs_[H], yh_[NR / 2];
                const int hoff = m1 ? NR / 2 : 0;
#pragma unroll
                for (int i = 0; i < H; ++i) xs_[i] = fs[20 + i];
#pragma unroll
                for (int k = 0; k < NR / 2; ++k) yh_[k] = fs[hoff + k];
                stage_term(xs_, fs[50], fs[51], 1.0, yh_, NS + 3, acc_wi, acc_wi2);          // E x_n
#pragma unroll
                for (int i = 0; i < H; ++i) xs_[i] = act ? sxw[i] : 0.0;
#pragma unroll
                for (int k = 0; k < NR / 2; ++k) yh_[k] = fs[10 + hoff + k];
                stage_term(xs_, act ? s_xEd : 0.0, act ? s_xLd : 0.0, 0.0, yh_, NS + 3, acc_wi, acc_wi2);     // gam Pw x'_w
#pragma unroll
                for (int i = 0; i < H; ++i) xs_[i] = act ? sxv[i] : 0.0;
#pragma unroll
                for (int k = 0; k < NR / 2; ++k) yh_[k] = fs[40 + hoff + k];
                stage_term(xs_, 0.0, 0.0, 0.0, yh_, NS + 3, acc_wi, acc_wi2);                                  // gam Pv x'_v
#pragma unroll
                for (int i = 0; i < H; ++i) xs_[i] = fs[30 + i];
#pragma unroll
                for (int k = 0; k < NR / 2; ++k) {       // Psi_j: slots 58-64, 57, 35, 36
                    const int j1 = NR / 2 + k;
                    yh_[k] = fs[m1 ? (j1 < 7 ? 58 + j1 : (j1 == 7 ? 57 : 27 + j1)) : 58 + k];
                }
                stage_term(xs_, fs[54], fs[55], 1.0, yh_, NS + 3, acc_wi, acc_wi2);           // Psi x_mid
#pragma unroll
                for (int i = 0; i < H; ++i) xs_[i] = act ? svt[i] : 0.0;
#pragma unroll
                for (int k = 0; k < NR / 2; ++k) {       // c1_j: slots 25-29 | 37, 38, 39, 52, 53
                    const int j1 = NR / 2 + k;
                    yh_[k] = fs[m1 ? (j1 < 8 ? 32 + j1 : 44 + j1) : 25 + k];
                }
                stage_term(xs_, 0.0, 0.0, 0.0, yh_, NS, acc_wo, acc_wo2);                                      // v~ (ir cv + irm)
#pragma unroll
                for (int i = 0; i < H; ++i) xs_[i] = act ? swt[i] : 0.0;
                stage_term(xs_, 0.0, 0.0, 0.0, c2h, NS, acc_wo, acc_wo2);                                      // w~ (ir cw)
                HY_T(5);
            }
        }

        {
            if (start_saved && n_saved >= 1) {
#pragma unroll
                for (int i = 0; i < H; ++i) {
                    if (obs[i]) {
                        double v = prm.u0[(size_t)ln.ci[i] * prm.B + b];
                        if (prm.clamp_pred) v = clampv(v, -kc->ub, kc->ub);
                        const double rr = (drows[doff[i]] - v) * kc->inv_yscale[ln.ci[i]];
                        loss_sum += (prm.loss_kind == 0) ? fabs(rr) : rr * rr;
                    }
                }
            }
            if (GRAD) {
                // the batch's share of the gradient: row (queue position / 32) of the partial-sum table (theta layout, the five extras
                // zero) -- rows belong to BATCHES, not to wavefronts, so the table does not depend on which wavefront took which
                // batch and the reduction over it is run-to-run identical.  Tile element (i, j) is in lane j + 16 (i % 4),
                // component i / 4 (tools/ubench/mfma_f64_layout.hip).
                double *const row = hp.gacc + (size_t)(wave_base >> 5) * (NTH + kExtra);
                const int j = lane & 15;
                if (j < NR) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 4 * r + (lane >> 4);
                        if (i < NS + 2) row[L_::wi(i, j)] = acc_wi[r] + acc_wi2[r];
                        else if (i == NS + 2) row[L_::wb(j)] = acc_wi[r] + acc_wi2[r];
                        if (i < NS) row[L_::wo(i, j)] = acc_wo[r] + acc_wo2[r];
                    }
                }
                if (lane < kExtra) row[NTH + lane] = 0.0;
#pragma unroll
                for (int r = 0; r < 4; ++r) { acc_wi[r] = 0.0; acc_wo[r] = 0.0; acc_wi2[r] = 0.0; acc_wo2[r] = 0.0; }
            }
            const double loss_tot = pair_sum(loss_sum);
            if (valid && !m1) {
                const double denom = (double)prm.n_obs * (double)n_saved;
                prm.loss[b] = n_saved > 0 ? loss_tot / denom : 0.0;
                prm.retcode[b] = rc;
                prm.n_saved[b] = n_saved;
                prm.n_accept[b] = nacc;
                prm.n_reject[b] = nrej;
            }
        }
    }
#undef HY2_LDS_SYNC
#undef HY2_FRESH_THETA
#undef HY2_FRESH_KC
#undef FR
#ifdef HY_PROF
    if (hp.prof && blockIdx.x == 0 && tid == 0)
        for (int k = 0; k < 16; ++k) hp.prof[k] = prof_acc[k];
#endif
}

// The pair kernel's partial-sum table has one row per batch of 32 trajectories (its MFMA tiles); this adds one row per 256 trajectories that
// carries only the five extras (loss sum, converged count, accepted, rejected, count) -- reduce_project_kernel sums all rows.
__global__ __launch_bounds__(256) void hy2_extras_kernel(double *__restrict__ rows, int nth, const double *__restrict__ loss,
                                                         const int32_t *__restrict__ retcode, const int32_t *__restrict__ n_accept,
                                                         const int32_t *__restrict__ n_reject, int64_t first, int64_t count) {
    __shared__ double ex[256];
    const int tid = threadIdx.x;
    const int64_t r = (int64_t)blockIdx.x * 256 + tid;
    double *out = rows + (size_t)blockIdx.x * (nth + kExtra);
    for (int m = tid; m < nth; m += 256) out[m] = 0.0;
    double e[kExtra] = {0.0, 0.0, 0.0, 0.0, 0.0};
    if (r < count) {
        const int64_t b = first + r;
        e[0] = loss[b];
        e[1] = (retcode[b] == 0) ? 1.0 : 0.0;
        e[2] = (double)n_accept[b];
        e[3] = (double)n_reject[b];
        e[4] = 1.0;
    }
    for (int k = 0; k < kExtra; ++k) {
        __syncthreads();
        ex[tid] = e[k];
        __syncthreads();
        for (int s_ = 128; s_ > 0; s_ >>= 1) {
            if (tid < s_) ex[tid] += ex[tid + s_];
            __syncthreads();
        }
        if (tid == 0) out[nth + k] = ex[0];
    }
}

}  // namespace crnn

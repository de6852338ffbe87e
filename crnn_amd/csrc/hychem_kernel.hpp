// crnn_amd/csrc/hychem_kernel.hpp -- gfx950 (MI355X): the HyChem pyrolysis CRNN (HyChem/crnn_pyrolysis_mass.jl).
//
//   crnn!(du,u,p,t) :121-131   Y = clamp(u, lb, 10); rho = P / (Ru T sum(Y/MW)); C = rho Y/MW 1e3;
//                              x = [log(clamp(C, lb, 10)); -1/(R T); log T]; wdot = w_out exp(w_in' x + w_b);
//                              du = wdot MW / rho .* dydt_scale;  T = itpT(t), P = itpP(t) piecewise linear on tsteps
//   predict_n_ode   :135-140   solve over [0, tsteps[sample]], saveat tsteps[1:sample]
//   loss_n_ode      :143-147   mae(pred ./ yscale, data ./ yscale)
//   ForwardDiff.gradient :201  211 parameters
//
// Stepper: non-autonomous Rosenbrock23 (k1 = W^-1 (f0 + gam df/dt), dense 9x9 W with pivoted LU in registers; df/dt
// analytic on the current table segment).  Gradient: discrete adjoint of the accepted steps (see ros23_adj_kernel.hpp)
// -- with 210 effective weights forward tangents would cost 211 primal solves per trajectory.  The exp-log-linear
// structure with the density coupling gives, for a direction (k, tau) in (u, t):
//     Y' = cY k,  S'/S = sum sigma k (sigma = cY/(MW S)),  l' = tau (P'/P - T'/T) - S'/S      (l = log rho)
//     x'_i = cC_i (l' + cY_i k_i / Y_i),  x'_E = -tau inv_R T'/T^2,  x'_L = tau T'/T,  z' = w_in' x'
//     Df[(k,tau)]_i = g_i/rho sum_j w_out[i,j] r_j z'_j - f_i l'
// and the adjoint of a.f and of a.Df[(k,tau)] w.r.t. (u, theta) in O(ns nr) (derivation in DESIGN.md section 2).
//
// MI355X mapping: one lane per trajectory (64 per wavefront from a global queue), theta (210 doubles) through
// wave-uniform scalar loads, per-trajectory T/P tables read through a cached segment cursor, per-lane step tape
// (t, dt, u) in HBM, the 210 gradient accumulators of a trajectory in HBM ([block of 64][m][lane], updated with
// fire-and-forget global_atomic_add_f64 -- coalesced, never read back by this kernel), reduced by reduce_gacc_kernel.
#pragma once
#include "ros23_adj_kernel.hpp"

// theta is 210 doubles.  As scalar loads (the case2 kernels' choice) it does not work here: hoisted out of the step
// loops it needs 420 SGPRs, and re-issued per phase the wave spends 45 % of its cycles in s_waitcnt on ~450 scalar loads
// per step (SQ counters, profiles/).  It is staged in LDS once per block and read with broadcast ds_read; at the top of
// each phase the pointer is re-derived with an opaque zero offset (empty asm) so that the loads are not hoisted out of
// the loops (they would pin VGPRs for the whole kernel).  The LDS-staged problem constants get the same treatment.
#define HY_FRESH_THETA(ptr)                     \
    do {                                        \
        unsigned z_ = 0;                        \
        asm volatile("" : "+v"(z_));            \
        (ptr) = th_lds + z_;                    \
    } while (0)
#define HY_FRESH_KC(ptr)                                             \
    do {                                                             \
        unsigned z_ = 0;                                             \
        asm volatile("" : "+v"(z_));                                 \
        (ptr) = reinterpret_cast<const KConst *>(kc_lds + z_);       \
    } while (0)

// timing ablations of the gradient-accumulator traffic (tools/kvariants.sh): 1 = plain stores, 2 = dropped
#ifndef HY_ABL
#define HY_ABL 0
#endif
#if HY_ABL == 0
#define HY_ACC(ptr, val) unsafeAtomicAdd((ptr), (val))
#elif HY_ABL == 1
#define HY_ACC(ptr, val) (*(ptr) = (val))
#else
#define HY_ACC(ptr, val) asm volatile("" ::"v"(val))
#endif

// phase timing (tools/kvariants.sh build prof="-DHY_PROF=1"): wave 0 of block 0 accumulates s_memtime deltas per phase
#ifdef HY_PROF
#define HY_T(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_readcyclecounter(); prof_acc[k] += now_ - prof_last; prof_last = now_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define HY_T(k) do { } while (0)
#endif

namespace crnn {

struct HyParams {
    const double *tabs;        // [B][2][Dfull]: T then P on the saveat grid, trajectory-major
    double *tape;              // [lanes][tape_cap][NS + 2]
    int32_t tape_cap;
    unsigned int *overflow;
    double *gacc;              // [ceil(count/64)][NTH][64], zeroed before the launch
    int32_t n_save_total;      // Dfull: length of the table / saveat grid
    double inv_R;
    unsigned long long *prof;  // HY_PROF builds: 16 phase totals in s_memtime ticks
    const int32_t *perm;       // position in the queue -> trajectory (relative to first); null = identity (sort_steps_kernel)
};

template <int NS, int NR>
struct LayH {
    static constexpr int NF = NS + 2;                 // feature rows: species, -1/(R T), log T
    static constexpr int NTH = NR * (NF + 1 + NS);
    __device__ __forceinline__ static constexpr int wi(int m, int j) { return m + NF * j; }
    __device__ __forceinline__ static constexpr int wb(int j) { return NF * NR + j; }
    __device__ __forceinline__ static constexpr int wo(int i, int j) { return (NF + 1) * NR + i + NS * j; }
};

// Make register values opaque to the IR optimiser at a phase boundary (costs no instruction): without it the
// unrolled contractions of adjacent phases are fused / re-associated into forms with hundreds of live values.
template <int N>
__device__ __forceinline__ void opaque(double (&a)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(a[i]));
}
__device__ __forceinline__ void opaque(double &a) { asm volatile("" : "+v"(a)); }

template <int NS, int NR>
struct HyPoint {
    double Y[NS], x[NS + 2], r[NR], f[NS];
    double irho, iS;
    unsigned cY, cC;     // bit i: u_i (C_i) inside its clamp window
};

// W's factors live in LDS (element (i,c) of a lane at As[(i*NS+c)*BLOCK]): 81 doubles per trajectory are the single
// largest item of the live set, and they are touched in bursts (factor, 3-4 solves) -- registers are kept for the
// vectors.  P A = L U with partial pivoting; rows are swapped in place.
template <int NS, int BLOCK>
__device__ __forceinline__ bool lu_factor_lds(double *As, double (&dinv)[NS], int (&piv)[NS], bool &anyp) {
    bool ok = true;
    anyp = false;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        int p = k;
        double best = fabs(As[(k * NS + k) * BLOCK]);
#pragma unroll
        for (int i = k + 1; i < NS; ++i) {
            const double v = fabs(As[(i * NS + k) * BLOCK]);
            if (v > best) { best = v; p = i; }
        }
        piv[k] = p;
        if (p != k) {
            anyp = true;
#pragma unroll
            for (int c = 0; c < NS; ++c) {
                const double a = As[(k * NS + c) * BLOCK], b = As[(p * NS + c) * BLOCK];
                As[(k * NS + c) * BLOCK] = b;
                As[(p * NS + c) * BLOCK] = a;
            }
        }
        double rowk[NS];
#pragma unroll
        for (int c = k; c < NS; ++c) rowk[c] = As[(k * NS + c) * BLOCK];
        ok = ok && (rowk[k] != 0.0);
        const double inv = frcp(rowk[k]);
        dinv[k] = inv;
#pragma unroll
        for (int i = k + 1; i < NS; ++i) {
            const double l = As[(i * NS + k) * BLOCK] * inv;
            As[(i * NS + k) * BLOCK] = l;
#pragma unroll
            for (int c = k + 1; c < NS; ++c) As[(i * NS + c) * BLOCK] = fma(-l, rowk[c], As[(i * NS + c) * BLOCK]);
        }
        CRNN_SCHED_FENCE();
    }
    return ok;
}

// factor in registers (independent updates pipeline in the VALU; an in-place LDS factorisation is a chain of
// read-modify-write latencies), then park the factors in LDS for the 3-4 solves of the step
template <int NS, int BLOCK>
__device__ __forceinline__ bool lu_factor_to_lds(double (&A)[NS][NS], double *As, double (&dinv)[NS], int (&piv)[NS], bool &anyp) {
    const bool ok = lu_factor<NS>(A, dinv, piv, anyp);
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int c = 0; c < NS; ++c) As[(i * NS + c) * BLOCK] = A[i][c];
    return ok;
}

template <int NS, int BLOCK>
__device__ __forceinline__ void lu_solve_lds(const double *As, const double (&dinv)[NS], const int (&piv)[NS],
                                             const bool wave_pivots, double (&b)[NS]) {
    if (wave_pivots) {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
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
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const double a = b[k];
#pragma unroll
        for (int i = k + 1; i < NS; ++i) b[i] = fma(-As[(i * NS + k) * BLOCK], a, b[i]);
    }
#pragma unroll
    for (int k = NS - 1; k >= 0; --k) {
        b[k] *= dinv[k];
        const double a = b[k];
#pragma unroll
        for (int i = 0; i < k; ++i) b[i] = fma(-As[(i * NS + k) * BLOCK], a, b[i]);
    }
}

// A^T x = b:  x = P^T L^-T U^-T b
template <int NS, int BLOCK>
__device__ __forceinline__ void lu_solve_T_lds(const double *As, const double (&dinv)[NS], const int (&piv)[NS],
                                               const bool wave_pivots, double (&b)[NS]) {
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        b[k] *= dinv[k];
        const double a = b[k];
#pragma unroll
        for (int i = k + 1; i < NS; ++i) b[i] = fma(-As[(k * NS + i) * BLOCK], a, b[i]);
    }
#pragma unroll
    for (int k = NS - 1; k >= 0; --k) {
        const double a = b[k];
#pragma unroll
        for (int i = 0; i < k; ++i) b[i] = fma(-As[(k * NS + i) * BLOCK], a, b[i]);
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

// point evaluation: features, rates, f
template <int NS, int NR>
__device__ __forceinline__ void hy_point(const double *th, const KConst *kc, const double inv_R,
                                         const double (&u)[NS], const double T, const double P, HyPoint<NS, NR> &pt) {
    using L_ = LayH<NS, NR>;
    double S = 0.0;
    unsigned cY = 0, cC = 0;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const double c = fmin(fmax(u[i], kc->lb), kc->ub);
        cY |= (c == u[i]) ? (1u << i) : 0u;
        pt.Y[i] = c;
        S = fma(c, kc->imw[i], S);
    }
    const double RTS = kc->Ru * T * S;
    const double rho = P * frcp(RTS);
    pt.irho = RTS * frcp(P);
    pt.iS = frcp(S);
    double cl[NS + 1], lg[NS + 1];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const double C = rho * (pt.Y[i] * kc->imw[i]) * 1e3;
        const double c = fmin(fmax(C, kc->lb), kc->ub);
        cC |= (c == C) ? (1u << i) : 0u;
        cl[i] = c;
    }
    cl[NS] = T;
    {   // two half-width batches: the element-innermost log keeps ~6 temporaries per element alive
        constexpr int H = (NS + 1) / 2;
        double a_[H], la_[H], b_[NS + 1 - H], lb_[NS + 1 - H];
#pragma unroll
        for (int i = 0; i < H; ++i) a_[i] = cl[i];
#pragma unroll
        for (int i = H; i < NS + 1; ++i) b_[i - H] = cl[i];
        flog_vec<H>(a_, la_);
        CRNN_SCHED_FENCE();
        flog_vec<NS + 1 - H>(b_, lb_);
#pragma unroll
        for (int i = 0; i < H; ++i) lg[i] = la_[i];
#pragma unroll
        for (int i = H; i < NS + 1; ++i) lg[i] = lb_[i - H];
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) pt.x[i] = lg[i];
    pt.x[NS] = inv_R * frcp(T);
    pt.x[NS + 1] = lg[NS];
    pt.cY = cY;
    pt.cC = cC;
    CRNN_SCHED_FENCE();
    double z[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        double zz = th[L_::wb(j)];
#pragma unroll
        for (int m = 0; m < NS + 2; ++m) zz = fma(th[L_::wi(m, j)], pt.x[m], zz);
        z[j] = zz;
        if (j & 1) CRNN_SCHED_FENCE();   // at most two columns of w_in (44 SGPRs) in flight
    }
    CRNN_SCHED_FENCE();
    {
        constexpr int H = NR / 2;
        double a_[H], ea_[H], b_[NR - H], eb_[NR - H];
#pragma unroll
        for (int j = 0; j < H; ++j) a_[j] = z[j];
#pragma unroll
        for (int j = H; j < NR; ++j) b_[j - H] = z[j];
        fexp_vec<H>(a_, ea_);
        CRNN_SCHED_FENCE();
        fexp_vec<NR - H>(b_, eb_);
#pragma unroll
        for (int j = 0; j < H; ++j) pt.r[j] = ea_[j];
#pragma unroll
        for (int j = H; j < NR; ++j) pt.r[j] = eb_[j - H];
    }
    CRNN_SCHED_FENCE();
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        double a = 0.0;
#pragma unroll
        for (int j = 0; j < NR; ++j) a = fma(th[L_::wo(i, j)], pt.r[j], a);
        pt.f[i] = a * kc->gsc[i] * pt.irho;
        if (i & 1) CRNN_SCHED_FENCE();   // at most two rows of w_out in flight
    }
}

// W = I - gam J(u_n) (dense, in registers for the factorisation) and ft = df/dt at the point
template <int NS, int NR, int BLOCK>
__device__ __forceinline__ void hy_jac_ft(const double *th, const KConst *kc, const HyPoint<NS, NR> &pt,
                                          const double gam, const double ld, const double xEd, const double xLd,
                                          double (&A)[NS][NS], double (&ft)[NS]) {
    using L_ = LayH<NS, NR>;
    double gx[NS], sg[NS], Bj[NR], zd[NR];
#pragma unroll
    for (int c = 0; c < NS; ++c) {
        const bool iy = (pt.cY >> c) & 1u, ic = (pt.cC >> c) & 1u;
        gx[c] = (iy && ic) ? frcp(pt.Y[c]) : 0.0;
        sg[c] = iy ? kc->imw[c] * pt.iS : 0.0;
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        double b = 0.0;
#pragma unroll
        for (int m = 0; m < NS; ++m) b += ((pt.cC >> m) & 1u) ? th[L_::wi(m, j)] : 0.0;
        Bj[j] = b;
        zd[j] = fma(b, ld, fma(th[L_::wi(NS, j)], xEd, th[L_::wi(NS + 1, j)] * xLd));
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const double Gi = kc->gsc[i] * pt.irho;
        double a[NR], tB = 0.0, tz = 0.0;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            a[j] = Gi * th[L_::wo(i, j)] * pt.r[j];
            tB = fma(a[j], Bj[j], tB);
            tz = fma(a[j], zd[j], tz);
        }
        ft[i] = fma(-pt.f[i], ld, tz);
#pragma unroll
        for (int c = 0; c < NS; ++c) {
            double s_ = 0.0;
#pragma unroll
            for (int j = 0; j < NR; ++j) s_ = fma(a[j], th[L_::wi(c, j)], s_);
            const double Jic = fma(gx[c], s_, -sg[c] * (tB - pt.f[i]));
            A[i][c] = ((i == c) ? 1.0 : 0.0) - gam * Jic;
        }
        CRNN_SCHED_FENCE();   // one row at a time: a[], its sums and nine outputs
    }
}

template <int NS, int NR, bool GRAD, int BLOCK>
__global__ __launch_bounds__(BLOCK) void hychem_kernel(const SolveParams prm, const double *__restrict__ theta,
                                                       const HyParams hp) {
    using L_ = LayH<NS, NR>;
    constexpr int NTH = L_::NTH;
    constexpr int RECW = NS + 2;
    constexpr int NPARK = (NS + 2) + NR + NS + 2 + 2 * NS + NR;   // x, r, Y, irho, iS of the u_n point; k1; k2 - k1; r at u_mid
    constexpr int PK_R1 = (NS + 2) + NR + NS + 2 + 2 * NS;
    __shared__ double kc_lds[kNConst];
    __shared__ double ts_lds[kMaxSave];
    __shared__ double th_lds[NTH];
    __shared__ double A_lds[NS * NS * BLOCK];
    __shared__ double park_lds[GRAD ? NPARK * BLOCK : 1];
    const int tid = threadIdx.x;
    double *const As = A_lds + tid;
    double *const park = park_lds + (GRAD ? tid : 0);
    for (int idx = tid; idx < kNConst; idx += BLOCK) kc_lds[idx] = reinterpret_cast<const double *>(prm.kc)[idx];
    for (int idx = tid; idx < hp.n_save_total; idx += BLOCK) ts_lds[idx] = prm.tsave[idx];
    for (int idx = tid; idx < NTH; idx += BLOCK) th_lds[idx] = theta[idx];
    __syncthreads();
    const KConst *kc = reinterpret_cast<const KConst *>(kc_lds);
    const double *th = th_lds;

    const double d_ = 0.29289321881345248, c32 = 7.4142135623730950, inv12d = 2.4142135623730950;
    const int nsave = prm.n_save, Dfull = hp.n_save_total;
    const double tend = ts_lds[nsave - 1], ts0 = ts_lds[0], t0 = kc->t0;
    const double dtmax = tend - t0;
    const double lqinit = flog(kc->qoldinit);
    const bool start_saved = (ts0 == t0);
    const int lane = tid & 63;
#ifdef HY_PROF
    unsigned long long prof_acc[16] = {0}, prof_last = __builtin_readcyclecounter();
#endif
    double *const tape = hp.tape + (size_t)((size_t)blockIdx.x * BLOCK + tid) * hp.tape_cap * RECW;

    while (true) {
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(prm.queue, 64ULL);
        const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)base);
        const unsigned bhi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
        const int64_t wave_base = (int64_t)(((unsigned long long)bhi << 32) | blo);
        if (wave_base >= prm.count) break;
        const int64_t traj = wave_base + lane;
        const bool valid = traj < prm.count;
        const int64_t b = prm.first + (valid ? (hp.perm ? (int64_t)hp.perm[traj] : traj) : 0);
        const double *const tabT = hp.tabs + (size_t)b * 2 * Dfull;
        const double *const tabP = tabT + Dfull;

        // ---- table cursor: segment [ts[seg], ts[seg+1]] with its end values cached
        int seg = -1;
        double Ta = 0, Tb = 0, Pa = 0, Pb = 0, tsa = 0, idts = 0;
        auto tab = [&](const double tq, double &T, double &P, double &Td, double &Pd) {
            int sg = seg < 0 ? 0 : seg;
            while (sg + 1 < Dfull - 1 && ts_lds[sg + 1] <= tq) ++sg;
            while (sg > 0 && ts_lds[sg] > tq) --sg;
            if (sg != seg) {
                seg = sg;
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
        double u[NS];
        HyPoint<NS, NR> p0;     // FSAL point (u, t)
        double t = t0, dt = 0.0, lqold = lqinit;
        int iter = 0, jsave = 0, nacc = 0, nrej = 0;
        int rc = valid ? -1 : 0;
#pragma unroll
        for (int i = 0; i < NS; ++i) u[i] = prm.u0[(size_t)i * prm.B + b];
        {
            double T, P, Td, Pd;
            tab(t0, T, P, Td, Pd);
            hy_point<NS, NR>(th, kc, hp.inv_R, u, T, P, p0);
            // Hairer initial step (order 2)
            double d0 = 0.0, d1 = 0.0, sk[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                sk[i] = frcp(fma(fabs(u[i]), kc->rtol[i], kc->atol[i]));
                const double a = u[i] * sk[i], c = p0.f[i] * sk[i];
                d0 = fma(a, a, d0);
                d1 = fma(c, c, d1);
            }
            d0 = sqrt(d0 * (1.0 / NS));
            d1 = sqrt(d1 * (1.0 / NS));
            double dt0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * (d0 / d1);
            dt0 = fmin(dt0, dtmax);
            double u1[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) u1[i] = fma(dt0, p0.f[i], u[i]);
            HyPoint<NS, NR> p1;
            tab(t0 + dt0, T, P, Td, Pd);
            hy_point<NS, NR>(th, kc, hp.inv_R, u1, T, P, p1);
            double d2 = 0.0;
#pragma unroll
            for (int i = 0; i < NS; ++i) { const double e = (p1.f[i] - p0.f[i]) * sk[i]; d2 = fma(e, e, d2); }
            d2 = sqrt(d2 * (1.0 / NS)) / dt0;
            const double dm = fmax(d1, d2);
            const double dt1 = (dm <= 1e-15) ? fmax(1e-6, dt0 * 1e-3) : exp(-0.5 * (4.605170185988091368 + flog(dm)));
            dt = fmax(kc->dtmin, fmin(fmin(100.0 * dt0, dt1), dtmax));
        }
        if (start_saved) {
            if (valid && prm.pred) {
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    double v = u[i];
                    if (prm.clamp_pred) v = clampv(v, -kc->ub, kc->ub);
                    prm.pred[((size_t)0 * NS + i) * prm.B + b] = v;
                }
            }
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
                    HY_T(0);
                    HY_FRESH_THETA(th); HY_FRESH_KC(kc);
                    const double gam = d_ * dt;
                    const double tnew = last ? tend : t + dt;
                    double T, P, Td, Pd;
                    tab(t, T, P, Td, Pd);
                    double dinv[NS], ft[NS];
                    int piv[NS];
                    bool anyp, okf;
                    {
                        double A[NS][NS];
                        hy_jac_ft<NS, NR, BLOCK>(th, kc, p0, gam, Pd * frcp(P) - Td * frcp(T), -hp.inv_R * Td * frcp(T * T), Td * frcp(T), A, ft);
                        CRNN_SCHED_FENCE();
                        HY_T(1);
                        okf = lu_factor_to_lds<NS, BLOCK>(A, As, dinv, piv, anyp);
                    }
                    HY_T(2);
                    const bool wp = __builtin_amdgcn_ballot_w64(anyp) != 0;
                    double k1[NS], dk[NS], unew[NS], f1[NS];
#pragma unroll
                    for (int i = 0; i < NS; ++i) k1[i] = fma(gam, ft[i], p0.f[i]);
                    lu_solve_lds<NS, BLOCK>(As, dinv, piv, wp, k1);
                    CRNN_SCHED_FENCE();
                    HY_T(3);
                    HY_FRESH_THETA(th); HY_FRESH_KC(kc);
                    {
                        double u1[NS];
#pragma unroll
                        for (int i = 0; i < NS; ++i) u1[i] = fma(0.5 * dt, k1[i], u[i]);
                        HyPoint<NS, NR> p1;
                        double T1, P1, a_, b_;
                        tab(t + 0.5 * dt, T1, P1, a_, b_);
                        hy_point<NS, NR>(th, kc, hp.inv_R, u1, T1, P1, p1);
#pragma unroll
                        for (int i = 0; i < NS; ++i) f1[i] = p1.f[i];
                        opaque(f1);
                    }
                    HY_T(4);
#pragma unroll
                    for (int i = 0; i < NS; ++i) dk[i] = f1[i] - k1[i];
                    lu_solve_lds<NS, BLOCK>(As, dinv, piv, wp, dk);
#pragma unroll
                    for (int i = 0; i < NS; ++i) unew[i] = fma(dt, k1[i] + dk[i], u[i]);
                    CRNN_SCHED_FENCE();
                    HY_T(5);
                    HyPoint<NS, NR> p2;
                    HY_FRESH_THETA(th); HY_FRESH_KC(kc);
                    {
                        double T2, P2, a_, b_;
                        tab(tnew, T2, P2, a_, b_);
                        hy_point<NS, NR>(th, kc, hp.inv_R, unew, T2, P2, p2);
                    }
                    CRNN_SCHED_FENCE();
                    HY_T(6);
                    double k3[NS];
#pragma unroll
                    for (int i = 0; i < NS; ++i) {
                        const double k2i = k1[i] + dk[i];
                        k3[i] = fma(dt, ft[i], p2.f[i] - c32 * (k2i - f1[i]) - 2.0 * (k1[i] - p0.f[i]));
                    }
                    lu_solve_lds<NS, BLOCK>(As, dinv, piv, wp, k3);
                    double es = 0.0;
                    bool finite = okf;
#pragma unroll
                    for (int i = 0; i < NS; ++i) {
                        const double k2i = k1[i] + dk[i];
                        const double ev = dt * (1.0 / 6.0) * (k1[i] - 2.0 * k2i + k3[i]);
                        const double m = fmax(fabs(u[i]), fabs(unew[i]));
                        const double e = ev * frcp(fma(kc->rtol[i], m, kc->atol[i]));
                        es = fma(e, e, es);
                        finite = finite && isfinite(unew[i]) && isfinite(ev);
                    }
                    es = es * (1.0 / NS);
                    if (!finite) rc = 3;
                    else {
                        const bool ee_zero = (es == 0.0);
                        const double lEE = 0.5 * flog(ee_zero ? 1.0 : es);
                        const double lq11 = kc->beta1 * lEE;
                        double q = ee_zero ? 1.0 / kc->qmax
                                           : fmax(1.0 / kc->qmax, fmin(1.0 / kc->qmin, exp(lq11 - kc->beta2 * lqold) / kc->gamma));
                        if (es <= 1.0) {
                            if (nacc >= hp.tape_cap) {
                                rc = 5;
                                atomicAdd(hp.overflow, 1u);
                            } else {
                                double *rec = tape + (size_t)nacc * RECW;
                                rec[0] = t;
                                rec[1] = dt;
#pragma unroll
                                for (int i = 0; i < NS; ++i) rec[2 + i] = u[i];
                                ++nacc;
                                while (jsave < nsave) {
                                    const double ts = ts_lds[jsave];
                                    if (!(ts <= tnew)) break;
                                    if (prm.pred) {
                                        const bool at_end = (ts == tnew);
                                        const double Th = at_end ? 1.0 : (ts - t) / dt;
                                        const double c1 = at_end ? 0.0 : Th * (1.0 - Th) * inv12d;
                                        const double c2 = at_end ? 1.0 : Th * (Th - 2.0 * d_) * inv12d;
#pragma unroll
                                        for (int i = 0; i < NS; ++i) {
                                            const double k2i = k1[i] + dk[i];
                                            double v = at_end ? unew[i] : fma(dt, fma(c1, k1[i], c2 * k2i), u[i]);
                                            if (prm.clamp_pred) v = clampv(v, -kc->ub, kc->ub);
                                            prm.pred[((size_t)jsave * NS + i) * prm.B + b] = v;
                                        }
                                    }
                                    ++jsave;
                                }
#pragma unroll
                                for (int i = 0; i < NS; ++i) u[i] = unew[i];
                                p0 = p2;
                                t = tnew;
                                if (q >= kc->qsteady_min && q <= kc->qsteady_max) q = 1.0;
                                lqold = ee_zero ? lqinit : fmax(lEE, lqinit);
                                dt = fmin(dt / q, dtmax);
                                if (jsave >= nsave) rc = 0;
                            }
                        } else {
                            ++nrej;
                            dt = dt / fmin(1.0 / kc->qmin, exp(lq11) / kc->gamma);
                        }
                    }
                }
            }
        }

        // ================================================================== reverse sweep: loss (+ adjoint)
        const int n_saved = jsave;
        const int jlo = start_saved ? 1 : 0;
        double lam[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) lam[i] = 0.0;
        double loss_sum = 0.0;
        double tnew = t;
        int s = valid ? nacc - 1 : -1;
        double *const gacc = hp.gacc + (size_t)(wave_base >> 6) * NTH * 64 + lane;   // accumulator m at gacc[m * 64]
#define HY_ADD(m, val) HY_ACC(&gacc[(size_t)(m) * 64], (val))
        const double *const drows = prm.data + (size_t)b * prm.row_stride;
        int doff[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) { const int dr = (int)kc->drow[i]; doff[i] = dr >= 0 ? dr : 0; }
        auto load_row = [&](int j, double (&d)[NS]) {
            const double *row = drows + (size_t)(j > 0 ? j : 0) * prm.n_obs;
#pragma unroll
            for (int i = 0; i < NS; ++i) d[i] = row[doff[i]];
        };
        double rt = 0.0, rdt = 0.0, ru[NS];
        {
            const double *rec = tape + (size_t)(s > 0 ? s : 0) * RECW;
            rt = rec[0]; rdt = rec[1];
#pragma unroll
            for (int i = 0; i < NS; ++i) ru[i] = rec[2 + i];
        }

        while (__builtin_amdgcn_ballot_w64(s >= 0) != 0) {
            if (s >= 0) {
                const double tn = rt, h = rdt;
                double un[NS];
#pragma unroll
                for (int i = 0; i < NS; ++i) un[i] = ru[i];
                // ---- re-form the step
                HY_T(8);
                HY_FRESH_THETA(th); HY_FRESH_KC(kc);
                const double gam = d_ * h;
                double T, P, Td, Pd;
                tab(tn, T, P, Td, Pd);
                const double ld = Pd * frcp(P) - Td * frcp(T), xEd = -hp.inv_R * Td * frcp(T * T), xLd = Td * frcp(T);
                HyPoint<NS, NR> pn, pm;
                hy_point<NS, NR>(th, kc, hp.inv_R, un, T, P, pn);
                // opaque to the optimiser from here: otherwise the Jacobian build is fused into the point evaluation
                // (products of rates and weights formed early, ~200 values live)
                opaque(pn.r); opaque(pn.Y); opaque(pn.f); opaque(pn.x); opaque(pn.irho); opaque(pn.iS);
                HY_FRESH_THETA(th); HY_FRESH_KC(kc);
                double dinv[NS], ft[NS];
                int piv[NS];
                bool anyp;
                unsigned ncY = 0, ncC = 0;
                double k1[NS], dk[NS];
                {
                    double A[NS][NS];
                    hy_jac_ft<NS, NR, BLOCK>(th, kc, pn, gam, ld, xEd, xLd, A, ft);
#pragma unroll
                    for (int i = 0; i < NS; ++i) k1[i] = fma(gam, ft[i], pn.f[i]);
                    ncY = pn.cY; ncC = pn.cC;
                    if (GRAD) {   // park the u_n point: it is needed again only by the last phase of the step
#pragma unroll
                        for (int m = 0; m < NS + 2; ++m) park[m * BLOCK] = pn.x[m];
#pragma unroll
                        for (int j = 0; j < NR; ++j) park[(NS + 2 + j) * BLOCK] = pn.r[j];
#pragma unroll
                        for (int i = 0; i < NS; ++i) park[(NS + 2 + NR + i) * BLOCK] = pn.Y[i];
                        park[(2 * NS + 2 + NR) * BLOCK] = pn.irho;
                        park[(2 * NS + 3 + NR) * BLOCK] = pn.iS;
                    }
                    opaque(k1);
                    CRNN_SCHED_FENCE();
                    HY_T(9);
                    (void)lu_factor_to_lds<NS, BLOCK>(A, As, dinv, piv, anyp);
                }
                HY_T(10);
                const bool wp = __builtin_amdgcn_ballot_w64(anyp) != 0;
                lu_solve_lds<NS, BLOCK>(As, dinv, piv, wp, k1);
                HY_FRESH_THETA(th); HY_FRESH_KC(kc);
                {
                    double u1[NS], T1, P1, a_, b_;
#pragma unroll
                    for (int i = 0; i < NS; ++i) u1[i] = fma(0.5 * h, k1[i], un[i]);
                    tab(tn + 0.5 * h, T1, P1, a_, b_);
                    hy_point<NS, NR>(th, kc, hp.inv_R, u1, T1, P1, pm);
                    opaque(pm.r); opaque(pm.Y); opaque(pm.f); opaque(pm.x); opaque(pm.irho); opaque(pm.iS);
                }
#pragma unroll
                for (int i = 0; i < NS; ++i) dk[i] = pm.f[i] - k1[i];
                lu_solve_lds<NS, BLOCK>(As, dinv, piv, wp, dk);
                CRNN_SCHED_FENCE();
                HY_T(11);

                // ---- loss and seeds at the save points inside (tn, tnew]
                double A_[NS], B1[NS], B2[NS];
#pragma unroll
                for (int i = 0; i < NS; ++i) { A_[i] = 0.0; B1[i] = 0.0; B2[i] = 0.0; }
                auto in_step = [&]() -> bool { return jsave > jlo && ts_lds[jsave - 1] > tn; };
                auto seed_point = [&](const double (&dobs)[NS]) {
                    const double ts = ts_lds[jsave - 1];
                    const bool at_end = (ts == tnew);
                    const double Th = at_end ? 1.0 : (ts - tn) / h;
                    const double c1 = at_end ? 0.0 : Th * (1.0 - Th) * inv12d;
                    const double c2 = at_end ? 1.0 : Th * (Th - 2.0 * d_) * inv12d;
#pragma unroll
                    for (int i = 0; i < NS; ++i) {
                        const int dr = (int)kc->drow[i];
                        if (dr >= 0) {
                            const double k2i = k1[i] + dk[i];
                            double v = at_end ? fma(h, k2i, un[i]) : fma(h, fma(c1, k1[i], c2 * k2i), un[i]);
                            double mask = 1.0;
                            if (prm.clamp_pred) {
                                mask = (v > kc->ub || v < -kc->ub) ? 0.0 : 1.0;
                                v = clampv(v, -kc->ub, kc->ub);
                            }
                            const double iy = kc->inv_yscale[i];
                            const double rr = (dobs[i] - v) * iy;
                            double w;
                            if (prm.loss_kind == 0) { loss_sum += fabs(rr); w = signbit(rr) ? 1.0 : -1.0; }
                            else { loss_sum = fma(rr, rr, loss_sum); w = -2.0 * rr; }
                            w *= mask * iy;
                            A_[i] += w;
                            B1[i] = fma(w, h * c1, B1[i]);
                            B2[i] = fma(w, h * c2, B2[i]);
                        }
                    }
                    --jsave;
                };
                while (in_step()) {
                    double dD[NS];
                    load_row(jsave - 1, dD);
                    seed_point(dD);
                }

                HY_T(12);
                {   // Next tape record: fetched AND awaited here, before this step's ~420 accumulator atomics are issued.
                    // gfx9 has one vmcnt for loads, stores and atomics: a load waited for after the atomics would drain
                    // them first (measured: 30-60 % of the reverse sweep sat in that wait).
                    const double *rec = tape + (size_t)(s > 0 ? s - 1 : 0) * RECW;
                    rt = rec[0]; rdt = rec[1];
#pragma unroll
                    for (int i = 0; i < NS; ++i) ru[i] = rec[2 + i];
                    opaque(rt); opaque(rdt); opaque(ru);
                }
                if (GRAD) {
#pragma unroll
                    for (int i = 0; i < NS; ++i) { park[(2 * NS + 4 + NR + i) * BLOCK] = k1[i]; park[(3 * NS + 4 + NR + i) * BLOCK] = dk[i]; }
#pragma unroll
                    for (int j = 0; j < NR; ++j) park[(PK_R1 + j) * BLOCK] = pm.r[j];
                    CRNN_SCHED_FENCE();
                    double kb1[NS], v[NS], ub[NS];
#pragma unroll
                    for (int i = 0; i < NS; ++i) { v[i] = fma(h, lam[i], B2[i]); ub[i] = lam[i] + A_[i]; kb1[i] = B1[i] + v[i]; }
                    lu_solve_T_lds<NS, BLOCK>(As, dinv, piv, wp, v);
#pragma unroll
                    for (int i = 0; i < NS; ++i) kb1[i] -= v[i];
                    double vt[NS];
#pragma unroll
                    for (int i = 0; i < NS; ++i) vt[i] = v[i] * kc->gsc[i];
                    opaque(vt); opaque(kb1); opaque(ub);
                    CRNN_SCHED_FENCE();
                    HY_FRESH_THETA(th); HY_FRESH_KC(kc);
                    // -------- point u_mid: adjoint of v.f   (a rolled loop over the reactions: 20 weights in flight)
                    double irho_mid = pm.irho;
                    opaque(irho_mid);
                    {
                        double P2[NS], psi = 0.0;
#pragma unroll
                        for (int m = 0; m < NS; ++m) P2[m] = 0.0;
#pragma unroll 1
                        for (int j = 0; j < NR; ++j) {
                            const double *wo_ = th + L_::wo(0, j), *wi_ = th + L_::wi(0, j);
                            double At = 0.0;
#pragma unroll
                            for (int i = 0; i < NS; ++i) At = fma(vt[i], wo_[i], At);
                            const double ir = pm.irho * park[(PK_R1 + j) * BLOCK];
                            const double Psi = At * ir;
                            psi += Psi;
                            double *gj = gacc + (size_t)L_::wi(0, j) * 64;
                            // (the w_b and w_out terms of this point, Psi_j and vt_i ir_j, are added together with those of
                            //  the point u_n below -- 100 fewer accumulator atomics per step; they need irho_mid and the parked r1)
#pragma unroll
                            for (int m = 0; m < NS + 2; ++m) HY_ACC(gj + (size_t)m * 64, Psi * pm.x[m]);
#pragma unroll
                            for (int m = 0; m < NS; ++m) P2[m] = fma(Psi, wi_[m], P2[m]);
                        }
                        double scp = 0.0;
#pragma unroll
                        for (int i = 0; i < NS; ++i) scp += ((pm.cC >> i) & 1u) ? P2[i] : 0.0;
#pragma unroll
                        for (int c = 0; c < NS; ++c) {
                            const bool iy = (pm.cY >> c) & 1u, ic = (pm.cC >> c) & 1u;
                            double m_ = iy ? kc->imw[c] * pm.iS * (psi - scp) : 0.0;
                            if (iy && ic) m_ = fma(P2[c], frcp(pm.Y[c]), m_);
                            ub[c] += m_;
                            kb1[c] = fma(0.5 * h, m_, kb1[c]);
                        }
                    }
                    CRNN_SCHED_FENCE();
                    HY_T(13);
                    lu_solve_T_lds<NS, BLOCK>(As, dinv, piv, wp, kb1);     // kb1 = w
                    opaque(kb1); opaque(ub); opaque(vt);
                    CRNN_SCHED_FENCE();
                    HY_FRESH_THETA(th); HY_FRESH_KC(kc);
                    // -------- point u_n: adjoint of w.f + gam ( v.Df[(dk,0)] + w.Df[(k1,1)] )
                    {
                        // reload the parked u_n point and the stages (through a laundered pointer: otherwise the
                        // compiler forwards the stored values and keeps them in registers after all)
                        unsigned zp_ = 0;
                        asm volatile("" : "+v"(zp_));
                        const double *pk = park + zp_;
                        HyPoint<NS, NR> pn;
                        double k1[NS], dk[NS];
#pragma unroll
                        for (int m = 0; m < NS + 2; ++m) pn.x[m] = pk[m * BLOCK];
#pragma unroll
                        for (int i = 0; i < NS; ++i) pn.Y[i] = pk[(NS + 2 + NR + i) * BLOCK];
                        pn.irho = pk[(2 * NS + 2 + NR) * BLOCK];
                        pn.iS = pk[(2 * NS + 3 + NR) * BLOCK];
                        pn.cY = ncY; pn.cC = ncC;
#pragma unroll
                        for (int i = 0; i < NS; ++i) { k1[i] = pk[(2 * NS + 4 + NR + i) * BLOCK]; dk[i] = pk[(3 * NS + 4 + NR + i) * BLOCK]; }
                        double wt[NS];
#pragma unroll
                        for (int i = 0; i < NS; ++i) wt[i] = kb1[i] * kc->gsc[i];
                        // direction data
                        double Spv = 0.0, Spw = 0.0, xpv[NS + 2], xpw[NS + 2];
#pragma unroll
                        for (int i = 0; i < NS; ++i) {
                            const double sg = ((pn.cY >> i) & 1u) ? kc->imw[i] * pn.iS : 0.0;
                            Spv = fma(sg, dk[i], Spv);
                            Spw = fma(sg, k1[i], Spw);
                        }
                        const double lpv = -Spv, lpw = ld - Spw;
#pragma unroll
                        for (int i = 0; i < NS; ++i) {
                            const bool iy = (pn.cY >> i) & 1u, ic = (pn.cC >> i) & 1u;
                            const double gy = iy ? frcp(pn.Y[i]) : 0.0;
                            xpv[i] = ic ? fma(gy, dk[i], lpv) : 0.0;
                            xpw[i] = ic ? fma(gy, k1[i], lpw) : 0.0;
                        }
                        xpv[NS] = 0.0; xpv[NS + 1] = 0.0;
                        xpw[NS] = xEd; xpw[NS + 1] = xLd;
                        double PE[NS], P2v[NS], P2w[NS], SE = 0.0, psiv = 0.0, psiw = 0.0;
#pragma unroll
                        for (int m = 0; m < NS; ++m) { PE[m] = 0.0; P2v[m] = 0.0; P2w[m] = 0.0; }
#pragma unroll 1
                        for (int j = 0; j < NR; ++j) {
                            const double *wo_ = th + L_::wo(0, j), *wi_ = th + L_::wi(0, j);
                            double Av = 0.0, Aw = 0.0, zv = 0.0, zw = 0.0;
#pragma unroll
                            for (int i = 0; i < NS; ++i) {
                                Av = fma(vt[i], wo_[i], Av);
                                Aw = fma(wt[i], wo_[i], Aw);
                            }
#pragma unroll
                            for (int m = 0; m < NS + 2; ++m) {
                                zv = fma(wi_[m], xpv[m], zv);
                                zw = fma(wi_[m], xpw[m], zw);
                            }
                            const double ir = pn.irho * pk[(NS + 2 + j) * BLOCK];
                            const double Pv = Av * ir, Pw = Aw * ir;
                            const double yv = zv - lpv, yw = zw - lpw;
                            const double cw = fma(gam, yw, 1.0), cv = gam * yv;
                            const double E = fma(Pw, cw, Pv * cv);
                            SE += E; psiv += Pv; psiw += Pw;
                            const double irm = irho_mid * pk[(PK_R1 + j) * BLOCK];   // the u_mid point's irho r_j
                            HY_ACC(gacc + (size_t)L_::wb(j) * 64, fma(Av, irm, E));
                            const double gPv = gam * Pv, gPw = gam * Pw;
                            double *gj = gacc + (size_t)L_::wi(0, j) * 64;
#pragma unroll
                            for (int m = 0; m < NS + 2; ++m) HY_ACC(gj + (size_t)m * 64, fma(E, pn.x[m], fma(gPw, xpw[m], gPv * xpv[m])));
                            double *go = gacc + (size_t)L_::wo(0, j) * 64;
#pragma unroll
                            for (int i = 0; i < NS; ++i) HY_ACC(go + (size_t)i * 64, fma(vt[i], fma(ir, cv, irm), wt[i] * (ir * cw)));
#pragma unroll
                            for (int m = 0; m < NS; ++m) {
                                PE[m] = fma(E, wi_[m], PE[m]);
                                P2v[m] = fma(Pv, wi_[m], P2v[m]);
                                P2w[m] = fma(Pw, wi_[m], P2w[m]);
                            }
                        }
                        double scE = 0.0, scv = 0.0, scw = 0.0;
#pragma unroll
                        for (int i = 0; i < NS; ++i) {
                            const bool ic = (pn.cC >> i) & 1u;
                            scE += ic ? PE[i] : 0.0;
                            scv += ic ? P2v[i] : 0.0;
                            scw += ic ? P2w[i] : 0.0;
                        }
                        const double brk = (SE - scE) + gam * fma(Spw, scw - psiw, Spv * (scv - psiv));
#pragma unroll
                        for (int c = 0; c < NS; ++c) {
                            const bool iy = (pn.cY >> c) & 1u, ic = (pn.cC >> c) & 1u;
                            double m_ = iy ? kc->imw[c] * pn.iS * brk : 0.0;
                            if (iy && ic) {
                                const double gy = frcp(pn.Y[c]);
                                m_ = fma(gy, PE[c] - gam * gy * fma(P2w[c], k1[c], P2v[c] * dk[c]), m_);
                            }
                            lam[c] = ub[c] + m_;
                        }
                    }
                }
                HY_T(14);
                tnew = tn;
                --s;
            }
        }
#undef HY_ADD

        if (valid) {
            if (start_saved && n_saved >= 1) {
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
            prm.loss[b] = n_saved > 0 ? loss_sum / denom : 0.0;
            prm.retcode[b] = rc;
            prm.n_saved[b] = n_saved;
            prm.n_accept[b] = nacc;
            prm.n_reject[b] = nrej;
        }
    }
#ifdef HY_PROF
    if (hp.prof && blockIdx.x == 0 && tid == 0)
        for (int k = 0; k < 16; ++k) hp.prof[k] = prof_acc[k];
#endif
}

// Ensemble reduction of the HBM gradient accumulators gacc[blk64][m][lane] (row r = blk64*64 + lane), each row scaled by
// its 1/(n_obs n_saved); same partials layout and fixed summation order as reduce_traj_kernel (256 rows per block).
__global__ __launch_bounds__(256) void reduce_gacc_kernel(const double *__restrict__ gacc, int nth, int n_obs,
                                                          const double *__restrict__ loss, const int32_t *__restrict__ retcode,
                                                          const int32_t *__restrict__ n_saved,
                                                          const int32_t *__restrict__ n_accept,
                                                          const int32_t *__restrict__ n_reject, int64_t first, int64_t count,
                                                          const int32_t *__restrict__ perm, double *__restrict__ partials) {
    __shared__ double sh[4][256];
    __shared__ double ex[256];
    const int npart = nth + kExtra;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t r = (int64_t)blockIdx.x * 256 + tid;            // this thread's trajectory row
    double *out = partials + (size_t)blockIdx.x * npart;
    double scale = 0.0;
    if (r < count) {   // accumulator row r belongs to the trajectory queued at position r
        const int ns_ = n_saved[first + (perm ? (int64_t)perm[r] : r)];
        scale = ns_ > 0 ? 1.0 / ((double)n_obs * (double)ns_) : 0.0;
    }
    const double *g = gacc + (size_t)(r >> 6) * nth * 64 + lane;
    for (int m = 0; m < nth; ++m) {
        double v = (r < count) ? g[(size_t)m * 64] * scale : 0.0;
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);   // fixed-order tree within the wave
        if (lane == 0) sh[w][m] = v;
    }
    __syncthreads();
    if (tid < nth) out[tid] = (sh[0][tid] + sh[1][tid]) + (sh[2][tid] + sh[3][tid]);
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

// crnn_amd/csrc/ros23_kernel.hpp -- gfx950 (MI355X) device code of the CRNN hot path.
//
// One kernel integrates an ensemble of initial conditions of the CRNN ODE
//     du/dt = scale .* ( w_out * exp( w_in' * [log(clamp(u,lb,ub)); inv_R/T] + w_b ) )
// (reference: case2/case2.jl:114-118, case1/case1.jl:80-83,
// robertson/rober_crnn.jl:113-116) with the adaptive Rosenbrock23 method the
// reference asks OrdinaryDiffEq for (case2/case2.jl:26, rober_crnn.jl:33),
// evaluates loss_neuralode (case2/case2.jl:132-137) on the fly at the saveat
// points and pushes forward tangents through every accepted step -- the
// arithmetic ForwardDiff.gradient (case2/case2.jl:195) performs on the solver.
//
// MI355X mapping:
//   * a *lane group* of L = ceil(P/C) lanes owns one trajectory; each lane owns
//     C tangent columns (directions in parameter space); 64/L groups share a
//     wavefront.  The small primal step is computed redundantly by the L lanes
//     (no cross-lane traffic in the stepper);
//   * the step is split in two register phases.  PRIMAL: W = I - gam*J is built,
//     LU-factored with partial pivoting and back-substituted in VGPRs, the three
//     stages, the error estimate, the PI controller and the saveat/loss code run;
//     everything the tangents need is published once per group in an LDS *step
//     record* (lane 0 of the group writes, all L lanes read = broadcast).
//     TANGENT: per column, operands stream from the record / the LDS-staged
//     d theta/d p matrix, the column itself lives in a per-lane LDS slot; only
//     the LU factors stay in registers across the phase.  This keeps the kernel
//     free of scratch (the first version spilled 1 KB/lane and moved 80x the
//     algorithmic HBM bytes);
//   * groups are persistent: when a trajectory finishes the group loads the
//     next one, so step-count divergence never idles the stepper;
//   * theta is read through wave-uniform scalar loads (SGPR operands);
//   * all HBM traffic is the compulsory u0 / data / loss stream, IC-fastest so
//     consecutive groups touch consecutive addresses;
//   * gradients are reduced deterministically: lane -> LDS -> per-block partial
//     -> fixed-order second kernel (bitwise reproducible for a given launch
//     geometry: replicated optimiser states stay identical across ranks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Scheduling fence between the phases of a tangent column: stops the pre-RA scheduler from hoisting the LDS loads of
// later phases to the top of the column (which inflates the live register set and forces VGPR<->AGPR traffic).
#ifndef CRNN_SCHED_FENCE
#define CRNN_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

namespace crnn {
// Bounds-checked build (-DCRNN_BOUNDS_CHECK; tools/crossbuild.py variant "O3chk"): indexed global-memory accesses -- tape records,
// observed rows, tables, u0 / pred / per-trajectory outputs, queue permutations, partial rows -- are checked against their extents
// on the lanes that perform them; violations are counted in g_bounds[0], the site code of the first one is kept in g_bounds[1]
// (crnn_debug_bounds reads the pair).  Release builds compile the checks away.
#ifdef CRNN_BOUNDS_CHECK
__device__ unsigned int g_bounds[2];
#define CRNN_CHK(cond, code) do { if (!(cond)) { if (atomicAdd(&g_bounds[0], 1u) == 0u) g_bounds[1] = (unsigned)(code); } } while (0)
#else
#define CRNN_CHK(cond, code) ((void)0)
#endif
}  // namespace crnn

namespace crnn {

constexpr int kMaxN = 12;
constexpr int kMaxSave = 256;  // max saveat points (LDS-staged)
constexpr int kExtra = 5;  // loss_sum, n_ok, n_accept, n_reject, n_traj
// The reduced vector every gradient path leaves in Ctx::d_red -- and the ranks all-reduce -- has ONE layout:
//   [ grad_sum(P) | n_overflow | loss_sum, n_ok, n_accept, n_reject, n_traj ]      (P + kTail doubles)
// n_overflow = trajectories that outran the adjoint tape in this launch (0 on the forward-tangent paths).
constexpr int kTail = kExtra + 1;

// Problem constants: one copy in device memory, staged to LDS by every block.
struct KConst {
    double lb, ub, inv_R, t0;
    double gamma, qmin, qmax, beta1, beta2, qsteady_min, qsteady_max, qoldinit, dtmin;
    double atol[kMaxN], rtol[kMaxN], scale[kMaxN], inv_yscale[kMaxN];
    double drow[kMaxN];  // species -> row of data (as double), -1 if unobserved
    double mw[kMaxN];    // HyChem: molar masses, their reciprocals, mw .* dydt_scale; Ru: gas constant [J/(kmol K)]
    double imw[kMaxN];
    double gsc[kMaxN];
    double Ru;
};
constexpr int kNConst = sizeof(KConst) / sizeof(double);

struct SolveParams {
    const double *u0;      // [n][B]
    const double *data;    // trajectory-major copy [B][n_save_total][n_obs] (transposed once at upload)
    int64_t row_stride;    // n_save_total * n_obs
    const double *tsave;   // [n_save_total]
    double *pred;          // [n_save_total][n][B] or null
    double *loss;          // [B]
    int32_t *retcode;      // [B]
    int32_t *n_saved;      // [B]
    int32_t *n_accept;     // [B]
    int32_t *n_reject;     // [B]
    double *gtraj;         // [count][L*C] per-trajectory gradient rows (row = trajectory - first)
    const KConst *kc;
    unsigned long long *queue;  // work queue head (zeroed before the launch): next unassigned trajectory - #groups
    int64_t B, first, count;
    int32_t n_save;        // active save points
    int32_t P;             // tangent directions
    int32_t maxiters, clamp_pred, loss_kind, n_obs;
    int32_t norm_cols;     // dual-norm kernels: 0 = divide the squared norm by n; N > 0 = by n (1 + N), N partials per Dual
    const int32_t *perm;   // dual-norm kernels: position in the queue -> trajectory (relative to first); null = index order
    int32_t n_chunks;      // ros23_sens_kernel: > 1 = all ForwardDiff chunks of the P directions in this launch, chunk_size partials each
    int32_t chunk_size;
};

// ---------------------------------------------------------------------------
// arithmetic helpers
// ---------------------------------------------------------------------------
// Julia's clamp(x, lo, hi) = ifelse(x > hi, hi, ifelse(x < lo, lo, x)): a NaN passes through (the trajectory then
// fails with retcode Unstable, as in the reference) -- fmin/fmax would silently replace it by a bound.
__device__ __forceinline__ double clampv(double v, double lo, double hi) { return v > hi ? hi : (v < lo ? lo : v); }

// 1/a: hardware seed + two Newton steps (<= 1-2 ulp; no denormal / inf special cases needed here)
__device__ __forceinline__ double frcp(double a) {
    double x = __builtin_amdgcn_rcp(a);
    double e = fma(-a, x, 1.0);
    x = fma(x, e, x);
    e = fma(-a, x, 1.0);
    x = fma(x, e, x);
    return x;
}

// 1/a to ~2e-15 (v_rcp_f64 delivers 4.6e-8, one Newton step 2.2e-15, two 1.1e-16: tools/ubench/rcp_acc.hip).  Enough where the
// quotient it feeds is corrected afterwards, as in flog / flog_vec.
__device__ __forceinline__ double frcp1(double a) {
    double x = __builtin_amdgcn_rcp(a);
    const double e = fma(-a, x, 1.0);
    return fma(x, e, x);
}

// log(x) for finite x > 0: fdlibm e_log.c scheme (< 1 ulp), ~35 instructions instead of
// ocml's ~100-instruction double-double evaluation.  x is always a clamped concentration
// (lb <= x <= ub) or a positive error norm here, so no zero / negative / inf / nan handling.
__device__ __forceinline__ double flog(double x) {
    double m = __builtin_amdgcn_frexp_mant(x);      // [0.5, 1)
    int k = __builtin_amdgcn_frexp_exp(x);
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;                             // [sqrt(1/2), sqrt 2)
    k = lo ? k - 1 : k;
    const double f = m - 1.0;
    const double r = frcp1(2.0 + f);
    double s = f * r;
    s = fma(fma(-(2.0 + f), s, f), r, s);           // one correction: s = f/(2+f) to < 1 ulp (error of r squared)
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01),
                                     2.857142874366239149e-01), 6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    return fma(dk, 6.93147180369123816490e-01, -((hfsq - fma(s, hfsq + R, dk * 1.90821492927058770002e-10)) - f));
}

// A double constant formed in an SGPR pair WHERE IT IS USED (two s_mov_b32 with literals) instead of once per kernel: the addend of a
// v_fmac chain has to be a register, the compiler materialises such constants ahead of the loops, cannot keep them in a full
// register file and reloads them from SCRATCH at every use (a memory latency per constant in a kernel that runs one wavefront per
// SIMD).  The moves are the asm itself: a constant merely passed THROUGH an SGPR-constrained asm is still hoisted and spilled
// (cathode kernels, round 5).  (Under the test suite's SIMT emulation the asm is dropped and the initialisers stay.)
template <unsigned LO, unsigned HI>
__device__ __forceinline__ double sconst_bits() {
    unsigned lo = LO, hi = HI;
    asm volatile("s_mov_b32 %0, %2\n\ts_mov_b32 %1, %3" : "=s"(lo), "=s"(hi) : "n"(LO), "n"(HI));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | (unsigned long long)lo);
}
#define CRNN_SCONST(x) (::crnn::sconst_bits<(unsigned)__builtin_bit_cast(unsigned long long, (double)(x)), \
                                            (unsigned)(__builtin_bit_cast(unsigned long long, (double)(x)) >> 32)>())
// flog, bit for bit, for kernels whose register file is full (first: the step-size controller of hychem2_kernel).  The five constants that START a multiply-add chain are
// addends of v_fmac, i.e. live in VGPRs; the compiler materialises them once per kernel, cannot keep ten registers for a function
// that runs once per step, and reloads them from SCRATCH at each call (five memory latencies per step).  Passed through an
// SGPR-constrained asm they are formed where they are used (two s_mov each).
__device__ __forceinline__ double flog_ctl(double x) {
    double m = __builtin_amdgcn_frexp_mant(x);      // [0.5, 1)
    int k = __builtin_amdgcn_frexp_exp(x);
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;                             // [sqrt(1/2), sqrt 2)
    k = lo ? k - 1 : k;
    const double f = m - 1.0;
    const double r = frcp1(2.0 + f);
    double s = f * r;
    s = fma(fma(-(2.0 + f), s, f), r, s);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, CRNN_SCONST(2.222219843214978396e-01)), CRNN_SCONST(3.999999999940941908e-01));
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, CRNN_SCONST(1.818357216161805012e-01)),
                                     CRNN_SCONST(2.857142874366239149e-01)), CRNN_SCONST(6.666666666666735130e-01));
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    return fma(dk, 6.93147180369123816490e-01, -((hfsq - fma(s, hfsq + R, dk * 1.90821492927058770002e-10)) - f));
}

// exp for the step-size controller: __ocml_exp_f64's arithmetic, operation for operation (the same bits as the exp() call it
// replaces -- the constants are the device library's, read off the ISA), with the same cure as flog_ctl: the ten coefficients that
// are ADDENDS of the multiply-add chain were materialised once per kernel in fourteen AGPRs and two scratch slots (two memory
// latencies per step, in the controller's dependent chain); formed in SGPRs where they are used they cost two s_mov each.
__device__ __forceinline__ double fexp_ctl(double x) {
    const double dn = __builtin_rint(x * 0x1.71547652b82fep+0);
    double t = fma(-dn, 0x1.62e42fefa39efp-1, x);
    t = fma(-dn, 0x1.abc9e3b39803fp-56, t);
    double p = fma(t, 0x1.ade156a5dcb37p-26, CRNN_SCONST(0x1.28af3fca7ab0cp-22));
    p = fma(t, p, CRNN_SCONST(0x1.71dee623fde64p-19));
    p = fma(t, p, CRNN_SCONST(0x1.a01997c89e6b0p-16));
    p = fma(t, p, CRNN_SCONST(0x1.a01a014761f6ep-13));
    p = fma(t, p, CRNN_SCONST(0x1.6c16c1852b7b0p-10));
    p = fma(t, p, CRNN_SCONST(0x1.1111111122322p-7));
    p = fma(t, p, CRNN_SCONST(0x1.55555555502a1p-5));
    p = fma(t, p, CRNN_SCONST(0x1.5555555555511p-3));
    p = fma(t, p, CRNN_SCONST(0x1.000000000000bp-1));
    p = fma(t, p, 1.0);
    p = fma(t, p, 1.0);
    double z = __builtin_amdgcn_ldexp(p, (int)dn);
    z = x > 1024.0 ? __builtin_inf() : z;
    return x < -1075.0 ? 0.0 : z;
}


template <int NS>
__device__ __forceinline__ bool lu_factor(double (&A)[NS][NS], double (&dinv)[NS], int (&piv)[NS], bool &anyp) {
    bool ok = true;
    anyp = false;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        int p = k;
        double best = fabs(A[k][k]);
#pragma unroll
        for (int i = k + 1; i < NS; ++i) {
            double v = fabs(A[i][k]);
            if (v > best) { best = v; p = i; }
        }
        piv[k] = p;
        anyp = anyp || (p != k);
        if (p != k) {  // skipped by the whole wave when no lane needs a swap
            // value-level selects (no control flow, no addressable temporaries): keeps A in VGPRs
#pragma unroll
            for (int i = k + 1; i < NS; ++i) {
                const bool sw = (p == i);
#pragma unroll
                for (int c = 0; c < NS; ++c) {
                    const double ak = A[k][c], ai = A[i][c];
                    A[k][c] = sw ? ai : ak;
                    A[i][c] = sw ? ak : ai;
                }
            }
        }
        double d = A[k][k];
        ok = ok && (d != 0.0);
        double inv = frcp(d);
        dinv[k] = inv;
#pragma unroll
        for (int i = k + 1; i < NS; ++i) A[i][k] *= inv;
#pragma unroll
        for (int c = k + 1; c < NS; ++c) {
            double a = A[k][c];
#pragma unroll
            for (int i = k + 1; i < NS; ++i) A[i][c] = fma(-A[i][k], a, A[i][c]);
        }
    }
    return ok;
}

template <int NS>
__device__ __forceinline__ void lu_solve(const double (&A)[NS][NS], const double (&dinv)[NS], const int (&piv)[NS],
                                         const bool wave_pivots, double (&b)[NS]) {
    // Row interchanges are rare for W = I - gam*J; the whole block is skipped by a scalar branch
    // unless some lane of the wavefront pivoted in this step (wave_pivots is wave-uniform).
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
        double a = b[k];
#pragma unroll
        for (int i = k + 1; i < NS; ++i) b[i] = fma(-A[i][k], a, b[i]);
    }
#pragma unroll
    for (int k = NS - 1; k >= 0; --k) {
        b[k] *= dinv[k];
        double a = b[k];
#pragma unroll
        for (int i = 0; i < k; ++i) b[i] = fma(-A[i][k], a, b[i]);
    }
}

template <int NS, int NR, bool HAS_T>
struct Lay {
    static constexpr int N = NS + (HAS_T ? 1 : 0);
    static constexpr int NTH = NR * (N + 1 + NS);
    static constexpr int NTHP = NTH | 1;  // odd LDS pitch
    __device__ __forceinline__ static constexpr int wi(int c, int j) { return c + N * j; }
    __device__ __forceinline__ static constexpr int wb(int j) { return N * NR + j; }
    __device__ __forceinline__ static constexpr int wo(int i, int j) { return (N + 1) * NR + i + NS * j; }
};

// step-record field offsets (in doubles, per group; record laid out [field][group])
template <int NS, int NR>
struct Rec {
    static constexpr int X0 = 0;               // features at u_n   (two ping-pong point areas: PA then PB)
    static constexpr int G0 = X0 + NS;
    static constexpr int R0 = G0 + NS;
    static constexpr int UP = R0 + NR;         // u at the point
    static constexpr int FP = UP + NS;         // f at the point
    static constexpr int PT = FP + NS;         // size of one point area
    static constexpr int PB = PT;              // second point area
    static constexpr int K1 = 2 * PT;
    static constexpr int DK = K1 + NS;
    static constexpr int X1 = DK + NS;
    static constexpr int G1 = X1 + NS;
    static constexpr int R1 = G1 + NS;
    static constexpr int C1J = R1 + NR;
    static constexpr int CZD = C1J + NR;
    static constexpr int GR0 = CZD + NR;
    static constexpr int AA = GR0 + NR;
    static constexpr int B1 = AA + NS;
    static constexpr int B2 = B1 + NS;
    static constexpr int DT = B2 + NS;
    static constexpr int NREC = DT + 1;
};

// x = log(clamp(u)), g = dx/du (0 outside the closed window, as ForwardDiff's clamp).
// Written species-innermost: every polynomial coefficient of the logarithm (a 64-bit literal = two s_mov) is
// materialised once and serves all NS species; at one wavefront per SIMD every instruction, scalar ones included,
// costs an issue slot.  c = min(max(u, lb), ub): a NaN state does not reach here unnoticed -- the stepper rejects
// non-finite stage results (retcode Unstable) before they are used.
// x_i = log(c_i) for N finite positive arguments at once (fdlibm e_log.c scheme, < 1 ulp), element-innermost.
template <int N>
__device__ __forceinline__ void flog_vec(const double (&c)[N], double (&x)[N]) {
    double m[N], f[N], r[N], s[N], z[N], w[N], t1[N], t2[N], dk[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double mm = __builtin_amdgcn_frexp_mant(c[i]);   // [0.5, 1)
        int k = __builtin_amdgcn_frexp_exp(c[i]);
        const bool lo = mm < 0.70710678118654752440;
        mm = lo ? mm + mm : mm;                          // [sqrt(1/2), sqrt 2)
        k = lo ? k - 1 : k;
        m[i] = mm;
        dk[i] = (double)k;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) { f[i] = m[i] - 1.0; r[i] = frcp1(2.0 + f[i]); }
#pragma unroll
    for (int i = 0; i < N; ++i) { s[i] = f[i] * r[i]; s[i] = fma(fma(-(2.0 + f[i]), s[i], f[i]), r[i], s[i]); }
#pragma unroll
    for (int i = 0; i < N; ++i) { z[i] = s[i] * s[i]; w[i] = z[i] * z[i]; }
#pragma unroll
    for (int i = 0; i < N; ++i) t1[i] = fma(w[i], 1.531383769920937332e-01, 2.222219843214978396e-01);
#pragma unroll
    for (int i = 0; i < N; ++i) t1[i] = fma(w[i], t1[i], 3.999999999940941908e-01);
#pragma unroll
    for (int i = 0; i < N; ++i) t2[i] = fma(w[i], 1.479819860511658591e-01, 1.818357216161805012e-01);
#pragma unroll
    for (int i = 0; i < N; ++i) t2[i] = fma(w[i], t2[i], 2.857142874366239149e-01);
#pragma unroll
    for (int i = 0; i < N; ++i) t2[i] = fma(w[i], t2[i], 6.666666666666735130e-01);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double R = fma(z[i], t2[i], w[i] * t1[i]);
        const double hfsq = 0.5 * f[i] * f[i];
        x[i] = fma(dk[i], 6.93147180369123816490e-01, -((hfsq - fma(s[i], hfsq + R, dk[i] * 1.90821492927058770002e-10)) - f[i]));
    }
}

template <int NS>
__device__ __forceinline__ void features(const double (&u)[NS], double lb, double ub, double (&x)[NS], double (&g)[NS]) {
    double c[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        c[i] = fmin(fmax(u[i], lb), ub);
        g[i] = (c[i] == u[i]) ? frcp1(c[i]) : 0.0;   // 2e-15: W is a W-method matrix and both sweeps use the same g
    }
    flog_vec<NS>(c, x);
}

// e^z for NR arguments at once, reaction-innermost so that each constant is materialised once (see features()).
// z = k ln2 + r, |r| <= ln2/2;  e^r by its Taylor polynomial of degree 13 (truncation 4e-18 relative);  2^k by ldexp,
// which saturates to inf / 0 exactly where exp overflows / underflows; NaN propagates.  < 1 ulp typical, 2 ulp worst.
template <int NR>
__device__ __forceinline__ void fexp_vec(const double (&z)[NR], double (&e)[NR]) {
    double kd[NR], r[NR], p[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) kd[j] = __builtin_rint(z[j] * 1.44269504088896338700e+00);
#pragma unroll
    for (int j = 0; j < NR; ++j) r[j] = fma(kd[j], -6.93147180369123816490e-01, z[j]);
#pragma unroll
    for (int j = 0; j < NR; ++j) r[j] = fma(kd[j], -1.90821492927058770002e-10, r[j]);
#pragma unroll
    for (int j = 0; j < NR; ++j) p[j] = fma(r[j], 1.6059043836821613e-10, 2.08767569878681e-09);   // 1/13!, 1/12!
#pragma unroll
    for (int j = 0; j < NR; ++j) p[j] = fma(r[j], p[j], 2.505210838544172e-08);                    // 1/11!
#pragma unroll
    for (int j = 0; j < NR; ++j) p[j] = fma(r[j], p[j], 2.755731922398589e-07);                    // 1/10!
#pragma unroll
    for (int j = 0; j < NR; ++j) p[j] = fma(r[j], p[j], 2.7557319223985893e-06);                   // 1/9!
#pragma unroll
    for (int j = 0; j < NR; ++j) p[j] = fma(r[j], p[j], 2.48015873015873e-05);                     // 1/8!
#pragma unroll
    for (int j = 0; j < NR; ++j) p[j] = fma(r[j], p[j], 1.984126984126984e-04);                    // 1/7!
#pragma unroll
    for (int j = 0; j < NR; ++j) p[j] = fma(r[j], p[j], 1.388888888888889e-03);                    // 1/6!
#pragma unroll
    for (int j = 0; j < NR; ++j) p[j] = fma(r[j], p[j], 8.333333333333333e-03);                    // 1/5!
#pragma unroll
    for (int j = 0; j < NR; ++j) p[j] = fma(r[j], p[j], 4.1666666666666664e-02);                   // 1/4!
#pragma unroll
    for (int j = 0; j < NR; ++j) p[j] = fma(r[j], p[j], 1.6666666666666666e-01);                   // 1/3!
#pragma unroll
    for (int j = 0; j < NR; ++j) p[j] = fma(r[j], p[j], 0.5);
#pragma unroll
    for (int j = 0; j < NR; ++j) { const double r2 = r[j] * r[j]; p[j] = fma(r2, p[j], r[j]) + 1.0; }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        // |z| beyond +-1100 would overflow the int conversion's meaning, not its saturation: clamp k, ldexp saturates
        const int k = (int)fmin(fmax(kd[j], -2200.0), 2200.0);
        e[j] = __builtin_amdgcn_ldexp(p[j], k);
    }
}

// fexp_vec with its eleven addend constants through CRNN_SCONST(): the same operations on the same values (the same bits), for kernels
// whose registers are full and whose vectors are short (hychem_sens2_kernel: one exponential per lane and point).
template <int NR>
__device__ __forceinline__ void fexp_vec_s(const double (&z)[NR], double (&e)[NR]) {
    double kd[NR], r[NR], p[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) kd[j] = __builtin_rint(z[j] * 1.44269504088896338700e+00);
#pragma unroll
    for (int j = 0; j < NR; ++j) r[j] = fma(kd[j], -6.93147180369123816490e-01, z[j]);
#pragma unroll
    for (int j = 0; j < NR; ++j) r[j] = fma(kd[j], -1.90821492927058770002e-10, r[j]);
#pragma unroll
    for (int j = 0; j < NR; ++j) p[j] = fma(r[j], 1.6059043836821613e-10, CRNN_SCONST(2.08767569878681e-09));   // 1/13!, 1/12!
#pragma unroll
    for (int j = 0; j < NR; ++j) p[j] = fma(r[j], p[j], CRNN_SCONST(2.505210838544172e-08));                    // 1/11!
#pragma unroll
    for (int j = 0; j < NR; ++j) p[j] = fma(r[j], p[j], CRNN_SCONST(2.755731922398589e-07));                    // 1/10!
#pragma unroll
    for (int j = 0; j < NR; ++j) p[j] = fma(r[j], p[j], CRNN_SCONST(2.7557319223985893e-06));                   // 1/9!
#pragma unroll
    for (int j = 0; j < NR; ++j) p[j] = fma(r[j], p[j], CRNN_SCONST(2.48015873015873e-05));                     // 1/8!
#pragma unroll
    for (int j = 0; j < NR; ++j) p[j] = fma(r[j], p[j], CRNN_SCONST(1.984126984126984e-04));                    // 1/7!
#pragma unroll
    for (int j = 0; j < NR; ++j) p[j] = fma(r[j], p[j], CRNN_SCONST(1.388888888888889e-03));                    // 1/6!
#pragma unroll
    for (int j = 0; j < NR; ++j) p[j] = fma(r[j], p[j], CRNN_SCONST(8.333333333333333e-03));                    // 1/5!
#pragma unroll
    for (int j = 0; j < NR; ++j) p[j] = fma(r[j], p[j], CRNN_SCONST(4.1666666666666664e-02));                   // 1/4!
#pragma unroll
    for (int j = 0; j < NR; ++j) p[j] = fma(r[j], p[j], CRNN_SCONST(1.6666666666666666e-01));                   // 1/3!
#pragma unroll
    for (int j = 0; j < NR; ++j) p[j] = fma(r[j], p[j], CRNN_SCONST(0.5));
#pragma unroll
    for (int j = 0; j < NR; ++j) { const double r2 = r[j] * r[j]; p[j] = fma(r2, p[j], r[j]) + 1.0; }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        // |z| beyond +-1100 would overflow the int conversion's meaning, not its saturation: clamp k, ldexp saturates
        const int k = (int)fmin(fmax(kd[j], -2200.0), 2200.0);
        e[j] = __builtin_amdgcn_ldexp(p[j], k);
    }
}

// r_j = exp(bT_j + sum_i w_in[i,j] x_i),  bT_j = w_b[j] + w_in[T,j] * inv_R/T
template <int NS, int NR, bool HAS_T>
__device__ __forceinline__ void rates(const double *__restrict__ th, const double (&x)[NS], const double (&bT)[NR],
                                      double (&r)[NR]) {
    using L = Lay<NS, NR, HAS_T>;
    double z[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        double zz = bT[j];
#pragma unroll
        for (int i = 0; i < NS; ++i) zz = fma(th[L::wi(i, j)], x[i], zz);
        z[j] = zz;
    }
    fexp_vec<NR>(z, r);
}

template <int NS, int NR, bool HAS_T, bool USE_SCALE>
__device__ __forceinline__ void rhs_from_rates(const double *__restrict__ th, const double (&r)[NR], const double *sc_s,
                                               double (&f)[NS]) {
    using L = Lay<NS, NR, HAS_T>;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        double a = 0.0;
#pragma unroll
        for (int j = 0; j < NR; ++j) a = fma(th[L::wo(i, j)], r[j], a);
        f[i] = USE_SCALE ? a * sc_s[i] : a;
    }
}

// ---------------------------------------------------------------------------
// W = I - gam*J solvers.  J = A B^T with A[i][j] = sc_i w_out[i,j] r_j (ns x nr), B^T[j][c] = w_in[c,j] g_c (nr x ns).
//   DenseLU   : builds W (ns x ns), LU with partial pivoting, triangular solves        (ns <= nr: robertson)
//   Woodbury  : W^-1 = I + gam A (I_nr - gam B^T A)^-1 B^T; only the nr x nr matrix M is factored (pivoted LU);
//               a solve costs the same ~57 operations but the factorisation is 4x cheaper and the state that
//               must survive the tangent phase shrinks from 48 to 15 doubles per lane       (nr < ns: case1, case2)
// Both are algebraically the reference's `W \ b`; they differ from it (and from each other) only by rounding.
// ---------------------------------------------------------------------------
template <int NS, int NR, bool HAS_T, bool USE_SCALE>
struct DenseLU {
    using L_ = Lay<NS, NR, HAS_T>;
    double A[NS][NS], dinv[NS];
    int piv[NS];
    bool wave_pivots;
    __device__ __forceinline__ bool factor(const double *__restrict__ th, const double (&g)[NS], const double (&r)[NR],
                                           const double gam, const double *sc_s) {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            double a[NR];
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                a[j] = th[L_::wo(i, j)] * r[j];
                if (USE_SCALE) a[j] *= sc_s[i];
            }
#pragma unroll
            for (int c = 0; c < NS; ++c) {
                double s_ = 0.0;
#pragma unroll
                for (int j = 0; j < NR; ++j) s_ = fma(a[j], th[L_::wi(c, j)], s_);
                A[i][c] = ((i == c) ? 1.0 : 0.0) - gam * (s_ * g[c]);
            }
        }
        bool anyp;
        const bool ok = lu_factor<NS>(A, dinv, piv, anyp);
        wave_pivots = __builtin_amdgcn_ballot_w64(anyp) != 0;
        return ok;
    }
    // Rosenbrock23(autodiff = false) (case2/case2.jl:26, robertson/rober_crnn_lm.jl:34): W = I - gam J with J filled by
    // FiniteDiff.finite_difference_jacobian!(J, f, u, Val(:forward)): column c = (f(u + eps_c e_c) - f(u)) / eps_c,
    // eps_c = max(sqrt(eps) |u_c|, sqrt(eps)) [UNVERIFIED-DEP: FiniteDiff.jl's default step, restated].
    // rhs(up, fp) evaluates the right-hand side; f0 = f(u) is the value the stepper holds.  A temperature state needs no column:
    // its row of J is zero and every right-hand side W meets has a zero temperature component, so that block never acts.
    template <class F>
    __device__ __forceinline__ bool factor_fd(const double (&u)[NS], const double (&f0)[NS], const double gam, F &&rhs) {
        const double rel = 1.4901161193847656e-08;   // sqrt(2^-52)
#pragma unroll
        for (int c = 0; c < NS; ++c) {
            double up[NS], fp[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) up[i] = u[i];
            const double eps = fmax(rel * fabs(u[c]), rel);
            up[c] = u[c] + eps;
            rhs(up, fp);
#pragma unroll
            for (int i = 0; i < NS; ++i) A[i][c] = ((i == c) ? 1.0 : 0.0) - gam * ((fp[i] - f0[i]) / eps);
        }
        bool anyp;
        const bool ok = lu_factor<NS>(A, dinv, piv, anyp);
        wave_pivots = __builtin_amdgcn_ballot_w64(anyp) != 0;
        return ok;
    }
    // g, gr (= gam * r) are unused here; the signature is shared with Woodbury
    __device__ __forceinline__ void solve(const double *__restrict__, const double (&)[NS], const double (&)[NR],
                                          const double *, double (&b)[NS]) const {
        lu_solve<NS>(A, dinv, piv, wave_pivots, b);
    }
};

template <int NS, int NR, bool HAS_T, bool USE_SCALE>
struct Woodbury {
    using L_ = Lay<NS, NR, HAS_T>;
    double M[NR][NR], dinv[NR];
    int piv[NR];
    bool wave_pivots;
    __device__ __forceinline__ bool factor(const double *__restrict__ th, const double (&g)[NS], const double (&r)[NR],
                                           const double gam, const double *sc_s) {
        // M[j][l] = delta_jl - gam r_l sum_c w_in[c,j] g_c sc_c w_out[c,l]
        double gsc[NS];
#pragma unroll
        for (int c = 0; c < NS; ++c) gsc[c] = USE_SCALE ? g[c] * sc_s[c] : g[c];
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            double t[NS];
#pragma unroll
            for (int c = 0; c < NS; ++c) t[c] = th[L_::wi(c, j)] * gsc[c];
#pragma unroll
            for (int l = 0; l < NR; ++l) {
                double s_ = 0.0;
#pragma unroll
                for (int c = 0; c < NS; ++c) s_ = fma(t[c], th[L_::wo(c, l)], s_);
                M[j][l] = ((j == l) ? 1.0 : 0.0) - (gam * r[l]) * s_;
            }
        }
        bool anyp;
        const bool ok = lu_factor<NR>(M, dinv, piv, anyp);
        wave_pivots = __builtin_amdgcn_ballot_w64(anyp) != 0;
        return ok;
    }
    // b <- W^-1 b = b + sc .* ( w_out (gr .* M^-1 (w_in^T (g .* b))) ),  gr = gam * r
    __device__ __forceinline__ void solve(const double *__restrict__ th, const double (&g)[NS], const double (&gr)[NR],
                                          const double *sc_s, double (&b)[NS]) const {
        double y[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) y[j] = 0.0;
#pragma unroll
        for (int c = 0; c < NS; ++c) {
            const double t = g[c] * b[c];
#pragma unroll
            for (int j = 0; j < NR; ++j) y[j] = fma(th[L_::wi(c, j)], t, y[j]);
        }
        lu_solve<NR>(M, dinv, piv, wave_pivots, y);
#pragma unroll
        for (int j = 0; j < NR; ++j) y[j] *= gr[j];
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            double a = 0.0;
#pragma unroll
            for (int j = 0; j < NR; ++j) a = fma(th[L_::wo(i, j)], y[j], a);
            b[i] = USE_SCALE ? fma(a, sc_s[i], b[i]) : b[i] + a;
        }
    }
};

template <bool WB, int NS, int NR, bool HAS_T, bool USE_SCALE>
struct SolverSel { using type = DenseLU<NS, NR, HAS_T, USE_SCALE>; };
template <int NS, int NR, bool HAS_T, bool USE_SCALE>
struct SolverSel<true, NS, NR, HAS_T, USE_SCALE> { using type = Woodbury<NS, NR, HAS_T, USE_SCALE>; };

// ---------------------------------------------------------------------------
// the fused solve + loss + tangent kernel
//   C = tangent columns per lane, L = lanes per trajectory (C = 0, L = 1: primal only)
// Per-trajectory outputs (loss, retcode, n_saved, step counts, gradient row); the
// ensemble sums are formed by reduce_traj_kernel in a fixed order, so results do not
// depend on the launch geometry of this kernel.
// ---------------------------------------------------------------------------
template <int NS, int NR, bool HAS_T, bool USE_SCALE, int C, int L, int BLOCK>
__global__ __launch_bounds__(BLOCK) void ros23_kernel(const SolveParams prm, const double *__restrict__ theta,
                                                      const double *__restrict__ dtheta) {
    using L_ = Lay<NS, NR, HAS_T>;
    using R_ = Rec<NS, NR>;
    constexpr int N = L_::N;
    constexpr int NTH = L_::NTH;
    constexpr int NTHP = L_::NTHP;
    constexpr int CC = (C > 0) ? C : 1;
    constexpr int NREC = R_::NREC;
    constexpr int WAVES = BLOCK / 64;
    constexpr int GPW = 64 / L;          // groups (trajectories) per wavefront
    constexpr int PPAD = L * CC;         // padded number of tangent columns
    static_assert(C > 0 || L == 1, "primal-only variant uses one lane per trajectory");
    static_assert(NREC == 2 * (4 * NS + NR) + 4 * NS + NR + 3 * NR + 3 * NS + 1, "record layout");

    // distinct LDS objects (no aliasing between them)
    __shared__ double kc_lds[kNConst];
    __shared__ double ts_lds[kMaxSave];                           // saveat times
    __shared__ double dth_lds[C > 0 ? PPAD * NTHP : 1];          // d theta / d p, zero padded, odd pitch
    __shared__ double S_lds[C > 0 ? WAVES * C * NS * 64 : 1];    // tangent columns, one slot per lane
    __shared__ double rec_lds[C > 0 ? WAVES * NREC * GPW : 1];   // step records, one per group

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int grp = lane / L;
    const int chunk = lane - grp * L;
    const bool lane_active = grp < GPW;
    const bool lead = lane_active && chunk == 0;

    double *S_s = S_lds + (C > 0 ? wave * C * NS * 64 + lane : 0);            // (qc, i) at S_s[(qc*NS+i)*64]
    double *rec = rec_lds + (C > 0 ? wave * NREC * GPW + (lane_active ? grp : 0) : 0);  // field f at rec[f*GPW]

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

    // Work distribution: the first #groups trajectories are assigned statically; afterwards a group takes the next
    // unassigned trajectory from a global queue (one atomic per trajectory, issued a whole trajectory ahead of use so
    // its latency is hidden).  Per-trajectory results do not depend on who computes them, so this stays deterministic.
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
    const double d_ = 0.29289321881345248;    // 1/(2+sqrt 2)
    const double c32 = 7.4142135623730950;    // 6+sqrt 2
    const double inv12d = 2.4142135623730950; // 1/(1-2d)
    const int nsave = prm.n_save;
    const double tend = prm.tsave[nsave - 1];
    const double ts0 = prm.tsave[0];
    const double t0 = kc->t0;
    const double dtmax = tend - t0;
    const double lqinit = flog(kc->qoldinit);

    // per-trajectory state carried in registers between steps
    double u[NS], f0[NS], g0[NS], r0[NR], bT[NR];
    double gtr[CC];
    double dA[NS], dB[NS];   // observed data of save points jsave / jsave+1, prefetched (HBM latency hidden)
    double xT = 0.0;
    double t = 0.0, dt = 0.0, lqold = 0.0, loss_sum = 0.0;
    int iter = 0, jsave = 0, par = 0, nacc = 0, nrej = 0;
    int64_t b = 0;
    bool need_init = true;

    // data row of save point j for this trajectory (unobserved species read row 0 and are ignored)
    auto load_row = [&](int j, double (&d)[NS]) {
        const int jj = j < nsave ? j : nsave - 1;
        const double *row = prm.data + (size_t)b * prm.row_stride + (size_t)jj * prm.n_obs;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int dr = (int)kc->drow[i];
            d[i] = row[dr >= 0 ? dr : 0];
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
            double x0[NS];
            features<NS>(u, kc->lb, kc->ub, x0, g0);
            rates<NS, NR, HAS_T>(th, x0, bT, r0);
            rhs_from_rates<NS, NR, HAS_T, USE_SCALE>(th, r0, kc->scale, f0);
            par = 0;
            if (C > 0 && lead) {
#pragma unroll
                for (int i = 0; i < NS; ++i) { rec[(R_::X0 + i) * GPW] = x0[i]; rec[(R_::G0 + i) * GPW] = g0[i]; }
#pragma unroll
                for (int j = 0; j < NR; ++j) rec[(R_::R0 + j) * GPW] = r0[j];
            }
            // Hairer initial step (OrdinaryDiffEq ode_determine_initdt, order 2)
            {
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
                // 10^(-(2 + log10 dm)/2) = exp(-(ln 100 + ln dm)/2)
                double dt1 = (dm <= 1e-15) ? fmax(1e-6, dt0 * 1e-3) : exp(-0.5 * (4.605170185988091368 + flog(dm)));
                dt = fmax(kc->dtmin, fmin(fmin(100.0 * dt0, dt1), dtmax));
            }
            t = t0;
            lqold = lqinit;
            iter = 0;
            jsave = 0;
            nacc = 0;
            nrej = 0;
            loss_sum = 0.0;
#pragma unroll
            for (int q = 0; q < CC; ++q) gtr[q] = 0.0;
            if (C > 0) {
#pragma unroll
                for (int q = 0; q < C * NS; ++q) S_s[q * 64] = 0.0;
            }
            // save_start: saveat contains tspan[1]  (case2: tsteps[1] = 0)
            if (ts0 == t0) {
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

        // ==================================================================
        // PRIMAL phase: one Rosenbrock23 attempt
        // ==================================================================
        int rc = -1;  // -1: keep going; >= 0: trajectory finished with this retcode
        ++iter;
        bool last = false;
        if (jsave >= nsave) rc = 0;            // horizon = tspan[1]: nothing to integrate
        else if (iter > prm.maxiters) rc = 1;
        if (t + dt * (1.0 + 1e-13) >= tend) { dt = tend - t; last = true; }
        if (rc < 0 && (!(dt > kc->dtmin) || t + dt == t)) rc = 2;

        bool accept = false;
        double q = 1.0, lq11 = 0.0, lEE = 0.0;
        bool ee_zero = false;
        typename SolverSel<(NR < NS), NS, NR, HAS_T, USE_SCALE>::type W;
        W.wave_pivots = false;
        const double gam = d_ * dt;
        const int pcur = par ? R_::PB : 0, pnxt = par ? 0 : R_::PB;
        double gr0[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) gr0[j] = gam * r0[j];
        if (rc < 0) {
            double k1[NS], dk[NS], unew[NS], f2[NS];
            const bool okf = W.factor(th, g0, r0, gam, kc->scale);
            // stage 1
#pragma unroll
            for (int i = 0; i < NS; ++i) k1[i] = f0[i];
            W.solve(th, g0, gr0, kc->scale, k1);
            double f1[NS];
            {
                double u1[NS], x1[NS], g1[NS], r1[NR];
#pragma unroll
                for (int i = 0; i < NS; ++i) u1[i] = fma(0.5 * dt, k1[i], u[i]);
                features<NS>(u1, kc->lb, kc->ub, x1, g1);
                rates<NS, NR, HAS_T>(th, x1, bT, r1);
                rhs_from_rates<NS, NR, HAS_T, USE_SCALE>(th, r1, kc->scale, f1);
                if (C > 0 && lead) {
#pragma unroll
                    for (int i = 0; i < NS; ++i) { rec[(R_::X1 + i) * GPW] = x1[i]; rec[(R_::G1 + i) * GPW] = g1[i]; }
#pragma unroll
                    for (int j = 0; j < NR; ++j) rec[(R_::R1 + j) * GPW] = r1[j];
                }
            }
            // stage 2
#pragma unroll
            for (int i = 0; i < NS; ++i) dk[i] = f1[i] - k1[i];
            W.solve(th, g0, gr0, kc->scale, dk);
#pragma unroll
            for (int i = 0; i < NS; ++i) unew[i] = fma(dt, k1[i] + dk[i], u[i]);
            double g2[NS], r2[NR];
            {
                double x2[NS];
                features<NS>(unew, kc->lb, kc->ub, x2, g2);
                rates<NS, NR, HAS_T>(th, x2, bT, r2);
                rhs_from_rates<NS, NR, HAS_T, USE_SCALE>(th, r2, kc->scale, f2);
                if (C > 0 && lead) {  // next point area (becomes point 0 when the step is accepted)
#pragma unroll
                    for (int i = 0; i < NS; ++i) {
                        rec[(pnxt + R_::X0 + i) * GPW] = x2[i];
                        rec[(pnxt + R_::G0 + i) * GPW] = g2[i];
                        rec[(pnxt + R_::UP + i) * GPW] = unew[i];
                        rec[(pnxt + R_::FP + i) * GPW] = f2[i];
                    }
#pragma unroll
                    for (int j = 0; j < NR; ++j) rec[(pnxt + R_::R0 + j) * GPW] = r2[j];
                }
            }
            // stage 3 + error estimate
            double k3[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                double k2i = k1[i] + dk[i];
                k3[i] = f2[i] - c32 * (k2i - f1[i]) - 2.0 * (k1[i] - f0[i]);
            }
            W.solve(th, g0, gr0, kc->scale, k3);
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
            es = es * (1.0 / N);  // EEst^2; the constant T state contributes a zero residual
            if (!finite) rc = 3;
            else {
                // PI controller (OrdinaryDiffEq PIController), in log space:
                //   q = EEst^beta1 / qold^beta2 / gamma, clipped to [1/qmax, 1/qmin]
                ee_zero = (es == 0.0);
                lEE = 0.5 * flog(ee_zero ? 1.0 : es);
                lq11 = kc->beta1 * lEE;
                q = ee_zero ? 1.0 / kc->qmax
                            : fmax(1.0 / kc->qmax, fmin(1.0 / kc->qmin, exp(lq11 - kc->beta2 * lqold) / kc->gamma));
                accept = (es <= 1.0);
            }

            if (rc < 0 && accept) {
                const double tnew = last ? tend : t + dt;
                // ---- saveat points inside (t, tnew]: dense output
                //      u(t + Th dt) = u + dt (c1 k1 + c2 k2),
                //      c1 = Th (1-Th)/(1-2d),  c2 = Th (Th-2d)/(1-2d)
                // loss-gradient seeds folded into three vectors:
                //      g_k += A.s_k + B1.k1'_k + B2.k2'_k
                double A_[NS], B1[NS], B2[NS];
#pragma unroll
                for (int i = 0; i < NS; ++i) { A_[i] = 0.0; B1[i] = 0.0; B2[i] = 0.0; }
                while (jsave < nsave) {
                    const double ts = ts_lds[jsave];
                    if (!(ts <= tnew)) break;
                    const bool at_end = (ts == tnew);
                    const double Th = at_end ? 1.0 : (ts - t) / dt;
                    const double c1 = at_end ? 0.0 : Th * (1.0 - Th) * inv12d;
                    const double c2 = at_end ? 1.0 : Th * (Th - 2.0 * d_) * inv12d;
#pragma unroll
                    for (int i = 0; i < NS; ++i) {
                        double k2i = k1[i] + dk[i];
                        double v = at_end ? unew[i] : fma(dt, fma(c1, k1[i], c2 * k2i), u[i]);
                        double mask = 1.0;
                        if (prm.clamp_pred) {
                            mask = (v > kc->ub || v < -kc->ub) ? 0.0 : 1.0;
                            v = clampv(v, -kc->ub, kc->ub);
                        }
                        if (prm.pred && chunk == 0) prm.pred[((size_t)jsave * N + i) * prm.B + b] = v;
                        int dr = (int)kc->drow[i];
                        if (dr >= 0) {
                            double iy = kc->inv_yscale[i];
                            double rr = (dA[i] - v) * iy;
                            double w;
                            if (prm.loss_kind == 0) { loss_sum += fabs(rr); w = signbit(rr) ? 1.0 : -1.0; }
                            else { loss_sum = fma(rr, rr, loss_sum); w = -2.0 * rr; }
                            w *= mask * iy;
                            A_[i] += w;
                            B1[i] = fma(w, dt * c1, B1[i]);
                            B2[i] = fma(w, dt * c2, B2[i]);
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
                if (C > 0) {
                    // publish the step record for the tangent phase
                    double c1j[NR], czd[NR];
#pragma unroll
                    for (int j = 0; j < NR; ++j) {
                        double z1 = 0.0, zd = 0.0;
#pragma unroll
                        for (int c = 0; c < NS; ++c) {
                            double wg = th[L_::wi(c, j)] * g0[c];
                            z1 = fma(wg, k1[c], z1);
                            zd = fma(wg, dk[c], zd);
                        }
                        c1j[j] = fma(gam, z1, 1.0);  // 1 + gam * z_j(k1)
                        czd[j] = gam * zd;           // gam * z_j(k2-k1)
                    }
                    if (lead) {
#pragma unroll
                        for (int i = 0; i < NS; ++i) {
                            rec[(R_::K1 + i) * GPW] = k1[i];
                            rec[(R_::DK + i) * GPW] = dk[i];
                            rec[(R_::AA + i) * GPW] = A_[i];
                            rec[(R_::B1 + i) * GPW] = B1[i];
                            rec[(R_::B2 + i) * GPW] = B2[i];
                        }
#pragma unroll
                        for (int j = 0; j < NR; ++j) {
                            rec[(R_::C1J + j) * GPW] = c1j[j];
                            rec[(R_::CZD + j) * GPW] = czd[j];
                            rec[(R_::GR0 + j) * GPW] = gr0[j];
                        }
                    }
                }
                if (C == 0) {  // primal-only variant: the FSAL point stays in registers
#pragma unroll
                    for (int i = 0; i < NS; ++i) { u[i] = unew[i]; f0[i] = f2[i]; g0[i] = g2[i]; }
#pragma unroll
                    for (int j = 0; j < NR; ++j) r0[j] = r2[j];
                }
                t = tnew;
            }
        }
        __builtin_amdgcn_wave_barrier();

        if (rc < 0 && accept) {
            ++nacc;
            // ==============================================================
            // TANGENT phase: forward tangents of the accepted step, C columns per lane.
            // Operands stream from the group's step record; only the factored W stays in registers.
            // ==============================================================
#ifdef CRNN_DBG_SKIP_TANGENT
            if (false) {
#else
            if (C > 0) {
#endif
                const double hdt = 0.5 * dt;
#pragma unroll 1
                for (int qc = 0; qc < C; ++qc) {
                    const double *dcol = dth_lds + (chunk * C + qc) * NTHP;
                    double *Sq = S_s + qc * NS * 64;
                    // ---- pass 1 (species-major): e0_j, theta-direct part of e1_j, z'_j(k1), z'_j(dk) ----
                    double e0[NR], e1d[NR], zp1[NR], zpd[NR];
                    double gq[NS], grq[NR];   // g at u_n and gam*r at u_n: operands of the W solves
#pragma unroll
                    for (int j = 0; j < NR; ++j) {
                        double e = dcol[L_::wb(j)];
                        if (HAS_T) e = fma(dcol[L_::wi(NS, j)], xT, e);
                        e0[j] = e; e1d[j] = e; zp1[j] = 0.0; zpd[j] = 0.0;
                    }
#pragma unroll
                    for (int c = 0; c < NS; ++c) {
                        const double sc_ = Sq[c * 64];
                        const double g = rec[(pcur + R_::G0 + c) * GPW];
                        gq[c] = g;
                        const double x0c = rec[(pcur + R_::X0 + c) * GPW];
                        const double x1c = rec[(R_::X1 + c) * GPW];
                        const double k1c = rec[(R_::K1 + c) * GPW];
                        const double dkc = rec[(R_::DK + c) * GPW];
                        const double gsv = g * sc_;
                        const double hsv = -g * gsv;   // g' = -g^2 s inside the window (g = 1/u), 0 outside
#pragma unroll
                        for (int j = 0; j < NR; ++j) {
                            const double dwi = dcol[L_::wi(c, j)];
                            const double wi = th[L_::wi(c, j)];
                            e0[j] = fma(dwi, x0c, e0[j]);
                            e0[j] = fma(wi, gsv, e0[j]);
                            e1d[j] = fma(dwi, x1c, e1d[j]);
                            const double m = fma(dwi, g, wi * hsv);
                            zp1[j] = fma(m, k1c, zp1[j]);
                            zpd[j] = fma(m, dkc, zpd[j]);
                        }
                    }
                    CRNN_SCHED_FENCE();
                    // ---- pass 2 (reaction-major): rhs1 = f0' + gam J' k1, w2 = gam J' dk, f1d = dw_out r1 ----
                    double rhs1[NS], w2[NS], f1d[NS];
#pragma unroll
                    for (int i = 0; i < NS; ++i) { rhs1[i] = 0.0; w2[i] = 0.0; f1d[i] = 0.0; }
#pragma unroll
                    for (int j = 0; j < NR; ++j) {
                        const double r0j = rec[(pcur + R_::R0 + j) * GPW], r1j = rec[(R_::R1 + j) * GPW];
                        const double gr = rec[(R_::GR0 + j) * GPW];
                        grq[j] = gr;
                        const double c1 = rec[(R_::C1J + j) * GPW], cz = rec[(R_::CZD + j) * GPW];
                        const double y1 = gr * zp1[j], yd = gr * zpd[j];
#pragma unroll
                        for (int i = 0; i < NS; ++i) {
                            const double wo = th[L_::wo(i, j)];
                            const double dwo = dcol[L_::wo(i, j)];
                            const double H = fma(wo, e0[j], dwo) * r0j;
                            rhs1[i] = fma(H, c1, rhs1[i]);
                            rhs1[i] = fma(wo, y1, rhs1[i]);
                            w2[i] = fma(H, cz, w2[i]);
                            w2[i] = fma(wo, yd, w2[i]);
                            f1d[i] = fma(dwo, r1j, f1d[i]);
                        }
                    }
                    if (USE_SCALE) {
#pragma unroll
                        for (int i = 0; i < NS; ++i) { double sc = kc->scale[i]; rhs1[i] *= sc; w2[i] *= sc; }
                    }
                    CRNN_SCHED_FENCE();
                    // W k1' = f0' + gam (J' k1)
                    W.solve(th, gq, grq, kc->scale, rhs1);  // rhs1 now holds k1'
                    CRNN_SCHED_FENCE();
                    // f1' at u1 with s1 = s + dt/2 k1'
                    double gs1[NS];
#pragma unroll
                    for (int c = 0; c < NS; ++c) gs1[c] = rec[(R_::G1 + c) * GPW] * fma(hdt, rhs1[c], Sq[c * 64]);
                    double rhs2[NS];
#pragma unroll
                    for (int i = 0; i < NS; ++i) rhs2[i] = f1d[i];
#pragma unroll
                    for (int j = 0; j < NR; ++j) {
                        double e = e1d[j];
#pragma unroll
                        for (int c = 0; c < NS; ++c) e = fma(th[L_::wi(c, j)], gs1[c], e);
                        const double er = e * rec[(R_::R1 + j) * GPW];
#pragma unroll
                        for (int i = 0; i < NS; ++i) rhs2[i] = fma(th[L_::wo(i, j)], er, rhs2[i]);
                    }
                    CRNN_SCHED_FENCE();
                    // W (k2-k1)' = f1' - k1' + gam J'(k2-k1)
#pragma unroll
                    for (int i = 0; i < NS; ++i) rhs2[i] = (USE_SCALE ? rhs2[i] * kc->scale[i] : rhs2[i]) - rhs1[i] + w2[i];
                    W.solve(th, gq, grq, kc->scale, rhs2);
                    double acc = 0.0;
#pragma unroll
                    for (int i = 0; i < NS; ++i) {
                        const double si = Sq[i * 64];
                        const double k2p = rhs1[i] + rhs2[i];
                        acc = fma(rec[(R_::AA + i) * GPW], si, acc);
                        acc = fma(rec[(R_::B1 + i) * GPW], rhs1[i], acc);
                        acc = fma(rec[(R_::B2 + i) * GPW], k2p, acc);
                        Sq[i * 64] = fma(dt, k2p, si);
                    }
                    gtr[qc] += acc;
                }
            }
            // ---- advance: reload the FSAL point from the record (it was parked there) ----
            par ^= 1;
            if (C > 0) {
                const int pn = par ? R_::PB : 0;
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    u[i] = rec[(pn + R_::UP + i) * GPW];
                    f0[i] = rec[(pn + R_::FP + i) * GPW];
                    g0[i] = rec[(pn + R_::G0 + i) * GPW];
                }
#pragma unroll
                for (int j = 0; j < NR; ++j) r0[j] = rec[(pn + R_::R0 + j) * GPW];
            }
            // step_accept_controller
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
            // mae/mse over the saved prefix (rober_crnn.jl:141 data[:, 1:size(pred)[2]])
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

// ---------------------------------------------------------------------------
// Deterministic ensemble reduction of the per-trajectory outputs.
//   partials[blk][0 .. ppad)          sum over the block's trajectory range of gtraj rows
//   partials[blk][ppad .. ppad+5)     loss_sum, n_ok, n_accept, n_reject, n_traj
// Each block owns a contiguous range of trajectories; all sums run in a fixed order.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void reduce_traj_kernel(const double *__restrict__ gtraj, int ppad,
                                                          const double *__restrict__ loss,
                                                          const int32_t *__restrict__ retcode,
                                                          const int32_t *__restrict__ n_accept,
                                                          const int32_t *__restrict__ n_reject, int64_t first,
                                                          int64_t count, int rows_per_block,
                                                          double *__restrict__ partials) {
    __shared__ double sh[256];
    const int npart = ppad + kExtra;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = (r0 + rows_per_block < count) ? r0 + rows_per_block : count;
    double *out = partials + (size_t)blockIdx.x * npart;
    const int tid = threadIdx.x;
    if (ppad > 0) {
        const int rl = 256 / ppad;            // row lanes per block
        const int col = tid % ppad, rlane = tid / ppad;
        double a = 0.0;
        if (rlane < rl)
            for (int64_t r = r0 + rlane; r < r1; r += rl) a += gtraj[(size_t)r * ppad + col];
        sh[tid] = a;
        __syncthreads();
        if (tid < ppad) {
            double s = 0.0;
            for (int k = 0; k < rl; ++k) s += sh[k * ppad + tid];
            out[tid] = s;
        }
        __syncthreads();
    }
    double e[kExtra] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int64_t r = r0 + tid; r < r1; r += 256) {
        const int64_t b = first + r;
        e[0] += loss[b];
        e[1] += (retcode[b] == 0) ? 1.0 : 0.0;
        e[2] += (double)n_accept[b];
        e[3] += (double)n_reject[b];
        e[4] += 1.0;
    }
    for (int k = 0; k < kExtra; ++k) {
        sh[tid] = e[k];
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) sh[tid] += sh[tid + s];
            __syncthreads();
        }
        if (tid == 0) out[ppad + k] = sh[0];
        __syncthreads();
    }
}

// One-time layout change at upload: IC-fastest data[(j*n_obs+i)*B + b] -> trajectory-major dst[b*rows + r].
// Every 128-byte line of the transposed copy belongs to one trajectory, so the solver fetches each line once
// no matter how far the persistent lane groups drift apart in time.
__global__ __launch_bounds__(256) void transpose_data_kernel(const double *__restrict__ src, double *__restrict__ dst,
                                                             int64_t B, int rows) {
    __shared__ double tile[32][33];
    const int64_t b0 = (int64_t)blockIdx.x * 32;
    const int r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        int r = r0 + k;
        int64_t b = b0 + tx;
        if (r < rows && b < B) tile[k][tx] = src[(size_t)r * B + b];
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        int64_t b = b0 + k;
        int r = r0 + tx;
        if (r < rows && b < B) dst[(size_t)b * rows + r] = tile[tx][k];
    }
}

// Fixed-order reduction of the per-block partials: sum_blk partials[blk][k], written in the common layout
// [grad(P) | n_overflow = 0 | extras]: of the ppad = L*C gradient columns only the first P are real directions.
// One block per column, 256 threads: strided serial sums then an LDS tree.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const double *__restrict__ partials, int nblk, int ppad, int P,
                                                              double *__restrict__ out) {
    __shared__ double sh[256];
    const int k = blockIdx.x;
    const int npart = ppad + kExtra;
    if (k >= P && k < ppad) return;              // padding column (block-uniform)
    const int dst = k < P ? k : P + 1 + (k - ppad);
    if (k == ppad && threadIdx.x == 0) out[P] = 0.0;
    double a = 0.0;
    for (int bI = threadIdx.x; bI < nblk; bI += 256) a += partials[(size_t)bI * npart + k];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[dst] = sh[0];
}

}  // namespace crnn

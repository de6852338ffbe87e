// crnn_amd/csrc/hychem2_kernel.hpp -- gfx950 (MI355X): the HyChem pyrolysis CRNN (HyChem/crnn_pyrolysis_mass.jl) with TWO LANES
// PER TRAJECTORY (round 3).  Same mathematics, tape, accumulators and outputs as hychem_kernel.hpp (read that header first:
// the RHS with its density coupling, the non-autonomous Rosenbrock23 step, the adjoint formulas); what changes is the mapping.
//
// hychem_kernel gives a trajectory one lane and needs 1.1 KB of LDS for it (W's factors, the parked u_n point): 128
// trajectories per CU, two wavefronts on four SIMDs, and a launch that lasts as long as its longest wavefront's chain of
// ~25 000 instructions per step pair (one GPU's share of BASELINE config 4, 32 768 trajectories, is exactly one generation
// of such wavefronts).  Here an adjacent lane pair owns the trajectory and its LDS slot:
//   * replicated in both lanes (identical instructions on identical operands, so identical bits: the pair never diverges):
//     the state vectors u, k1, dk, lambda ..., the clamped mass fractions, density, the rates r, the step-size controller,
//     the LU factorisation (both lanes factor the same 9 x 9 matrix and write the same factors to the pair's LDS slot) and
//     the triangular solves -- together ~10 % of a step;
//   * split over the pair, lane m owning species 5m .. 5m+4 (lane 1: four species, its fifth slot takes log T): the
//     logarithms, the exponentials (five each, exchanged), the rows of the Jacobian (each lane builds its rows of W straight
//     into the pair's LDS slot), every contraction over the species and the gradient accumulators of the species' rows of
//     w_in / w_out (global atomics as before, half as many per lane);
//   * what crosses the pair: sums over the species (one DPP step: a + quad_perm[1,0,3,2](a), the same bits in both lanes)
//     and "gathers" of a species-distributed vector into the replicated full-length one (one DPP move per element).
// The pair's 64-lane wavefront takes 32 trajectories from the queue; a 256-lane block (128 pairs) uses the same 146 KB of LDS
// as the one-lane kernel's 128-lane block, so all four SIMDs of a CU work.
#pragma once
#include "hychem_kernel.hpp"
#include "ros23_adj2_kernel.hpp"   // pair_sum

namespace crnn {

__device__ __forceinline__ double pair_other(double a) {   // the other lane of the pair's value
    const int lo = __double2loint(a), hi = __double2hiint(a);
    const int plo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, false);
    const int phi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, false);
    return __hiloint2double(phi, plo);
}

// species-distributed (lane m holds species H m + i in own[i]) -> full-length, replicated
template <int NS, int H>
__device__ __forceinline__ void pair_gather(const double (&own)[H], const bool m1, double (&full)[NS]) {
    double oth[H];
#pragma unroll
    for (int i = 0; i < H; ++i) oth[i] = pair_other(own[i]);
#pragma unroll
    for (int c = 0; c < NS; ++c) {
        const int i = c < H ? c : c - H;
        full[c] = (c < H) ? (m1 ? oth[i] : own[i]) : (m1 ? own[i] : oth[i]);
    }
}
// full-length, replicated -> this lane's species
template <int NS, int H>
__device__ __forceinline__ void pair_own(const double (&full)[NS], const bool m1, double (&own)[H]) {
#pragma unroll
    for (int i = 0; i < H; ++i) own[i] = m1 ? (H + i < NS ? full[H + i < NS ? H + i : 0] : 0.0) : full[i];
}

template <int NS, int NR>
struct HyPoint2 {
    static constexpr int H = (NS + 1) / 2;
    double Y[NS];          // clamp(u) (replicated)
    double xo[H];          // log clamp(C) of this lane's species
    double xE, xL;         // -1/(R T), log T (replicated)
    double r[NR];          // rates (replicated)
    double f[NS], fo[H];   // right-hand side: replicated, and this lane's species
    double irho, iS;
    unsigned cY, cC;       // bit i: u_i (C_i) inside its clamp window (replicated)
};

// point evaluation.  ci[i] = species index of this lane's slot i (0 for the padding slot), ow[i] = slot holds a species.
template <int NS, int NR>
__device__ __forceinline__ void hy_point2(const double *th, const KConst *kc, const double inv_R, const double (&u)[NS], const double T,
                                          const double P, const bool m1, const int (&ci)[(NS + 1) / 2], const bool (&ow)[(NS + 1) / 2],
                                          HyPoint2<NS, NR> &pt) {
    using L_ = LayH<NS, NR>;
    constexpr int H = (NS + 1) / 2;
    static_assert(NR == 2 * H, "the exponentials are split H + H");
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
    double cl[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const double C = rho * (pt.Y[i] * kc->imw[i]) * 1e3;
        const double c = fmin(fmax(C, kc->lb), kc->ub);
        cC |= (c == C) ? (1u << i) : 0u;
        cl[i] = c;
    }
    pt.cY = cY;
    pt.cC = cC;
    {   // H logarithms per lane: lane 0 species 0 .. H-1, lane 1 species H .. NS-1 and T
        double a_[H], la_[H];
#pragma unroll
        for (int i = 0; i < H; ++i) a_[i] = m1 ? (H + i < NS ? cl[H + i < NS ? H + i : 0] : T) : cl[i];
        flog_vec<H>(a_, la_);
#pragma unroll
        for (int i = 0; i < H; ++i) pt.xo[i] = la_[i];
        const double lt_other = pair_other(la_[H - 1]);
        pt.xL = m1 ? la_[H - 1] : lt_other;
    }
    pt.xE = inv_R * frcp(T);
    CRNN_SCHED_FENCE();
    double z[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        double a = 0.0;
#pragma unroll
        for (int i = 0; i < H; ++i) a = fma(ow[i] ? th[L_::wi(0, j) + ci[i]] : 0.0, pt.xo[i], a);
        z[j] = pair_sum(a) + fma(th[L_::wi(NS, j)], pt.xE, fma(th[L_::wi(NS + 1, j)], pt.xL, th[L_::wb(j)]));
    }
    CRNN_SCHED_FENCE();
    {   // H exponentials per lane, exchanged
        double a_[H], e_[H], o_[H];
#pragma unroll
        for (int k = 0; k < H; ++k) a_[k] = m1 ? z[H + k] : z[k];
        fexp_vec<H>(a_, e_);
#pragma unroll
        for (int k = 0; k < H; ++k) o_[k] = pair_other(e_[k]);
#pragma unroll
        for (int k = 0; k < H; ++k) { pt.r[k] = m1 ? o_[k] : e_[k]; pt.r[H + k] = m1 ? e_[k] : o_[k]; }
    }
    CRNN_SCHED_FENCE();
#pragma unroll
    for (int i = 0; i < H; ++i) {
        double a = 0.0;
#pragma unroll
        for (int j = 0; j < NR; ++j) a = fma(th[L_::wo(0, j) + ci[i]], pt.r[j], a);
        pt.fo[i] = ow[i] ? a * kc->gsc[ci[i]] * pt.irho : 0.0;
    }
    pair_gather<NS, H>(pt.fo, m1, pt.f);
}

// This lane's rows of W = I - gam J(u_n), written straight into the pair's LDS slot (element (i, c) at As[(i NS + c) GPB]),
// and ft = df/dt at the point (replicated).
template <int NS, int NR, int GPB>
__device__ __forceinline__ void hy_jac_ft2(const double *th, const KConst *kc, const HyPoint2<NS, NR> &pt, const double gam, const double ld,
                                           const double xEd, const double xLd, const bool m1, const int (&ci)[(NS + 1) / 2],
                                           const bool (&ow)[(NS + 1) / 2], double *As, double (&ft)[NS]) {
    using L_ = LayH<NS, NR>;
    constexpr int H = (NS + 1) / 2;
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
        for (int i = 0; i < H; ++i) b += (ow[i] && ((pt.cC >> ci[i]) & 1u)) ? th[L_::wi(0, j) + ci[i]] : 0.0;
        b = pair_sum(b);
        Bj[j] = b;
        zd[j] = fma(b, ld, fma(th[L_::wi(NS, j)], xEd, th[L_::wi(NS + 1, j)] * xLd));
    }
    double fto[H];
#pragma unroll
    for (int i = 0; i < H; ++i) {
        const double Gi = kc->gsc[ci[i]] * pt.irho;
        double a[NR], tB = 0.0, tz = 0.0;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            a[j] = Gi * th[L_::wo(0, j) + ci[i]] * pt.r[j];
            tB = fma(a[j], Bj[j], tB);
            tz = fma(a[j], zd[j], tz);
        }
        fto[i] = ow[i] ? fma(-pt.fo[i], ld, tz) : 0.0;
#pragma unroll
        for (int c = 0; c < NS; ++c) {
            double s_ = 0.0;
#pragma unroll
            for (int j = 0; j < NR; ++j) s_ = fma(a[j], th[L_::wi(c, j)], s_);
            const double Jic = fma(gx[c], s_, -sg[c] * (tB - pt.fo[i]));
            if (ow[i]) As[(size_t)(ci[i] * NS + c) * GPB] = ((ci[i] == c) ? 1.0 : 0.0) - gam * Jic;
        }
        CRNN_SCHED_FENCE();   // one row at a time
    }
    pair_gather<NS, H>(fto, m1, ft);
}

template <int NS, int NR, bool GRAD, int BLOCK>
__global__ __launch_bounds__(BLOCK) void hychem2_kernel(const SolveParams prm, const double *__restrict__ theta, const HyParams hp) {
    using L_ = LayH<NS, NR>;
    constexpr int NTH = L_::NTH;
    constexpr int H = (NS + 1) / 2;
    constexpr int GPB = BLOCK / 2;
    constexpr int RECW = NS + 2;
    constexpr int NPARK = (NS + 2) + NR + NS + 2 + 2 * NS + NR;   // x, r, Y, irho, iS of the u_n point; k1; k2 - k1; r at u_mid
    constexpr int PK_R1 = (NS + 2) + NR + NS + 2 + 2 * NS;
    __shared__ double kc_lds[kNConst];
    __shared__ double ts_lds[kMaxSave];
    __shared__ double th_lds[NTH];
    __shared__ double A_lds[NS * NS * GPB];
    __shared__ double park_lds[GRAD ? NPARK * GPB : 1];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const bool m1 = (lane & 1) != 0;
    const int gib = tid >> 1, giw = lane >> 1;
    double *const As = A_lds + gib;
    double *const park = park_lds + (GRAD ? gib : 0);
    for (int idx = tid; idx < kNConst; idx += BLOCK) kc_lds[idx] = reinterpret_cast<const double *>(prm.kc)[idx];
    for (int idx = tid; idx < hp.n_save_total; idx += BLOCK) ts_lds[idx] = prm.tsave[idx];
    for (int idx = tid; idx < NTH; idx += BLOCK) th_lds[idx] = theta[idx];
    __syncthreads();
    const KConst *kc = reinterpret_cast<const KConst *>(kc_lds);
    const double *th = th_lds;
    int ci[H];
    bool ow[H];
#pragma unroll
    for (int i = 0; i < H; ++i) { const int c = (m1 ? H : 0) + i; ow[i] = c < NS; ci[i] = ow[i] ? c : 0; }
    // wave-level ordering of the pair's LDS traffic (one lane's stores, the other lane's loads)
#define HY2_LDS_SYNC()                                                   \
    do {                                                                 \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");           \
        __builtin_amdgcn_wave_barrier();                                 \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");           \
    } while (0)

    const double d_ = 0.29289321881345248, c32 = 7.4142135623730950, inv12d = 2.4142135623730950;
    const int nsave = prm.n_save, Dfull = hp.n_save_total;
    const double tend = ts_lds[nsave - 1], ts0 = ts_lds[0], t0 = kc->t0;
    const double dtmax = tend - t0;
    const double lqinit = flog(kc->qoldinit);
    const bool start_saved = (ts0 == t0);
    double *const tape = hp.tape + (size_t)((size_t)blockIdx.x * GPB + gib) * hp.tape_cap * RECW;

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
                Ta = tabT[sg]; Tb = tabT[sg + 1]; Pa = tabP[sg]; Pb = tabP[sg + 1];
                tsa = ts_lds[sg];
                idts = frcp(ts_lds[sg + 1] - tsa);
            }
            Td = (Tb - Ta) * idts;
            Pd = (Pb - Pa) * idts;
            T = fma(tq - tsa, Td, Ta);
            P = fma(tq - tsa, Pd, Pa);
        };
        // factor the pair's W (both lanes: the same matrix, the same factors) and park the factors in the pair's LDS slot
        auto factor = [&](double (&dinv)[NS], int (&piv)[NS], bool &anyp) -> bool {
            HY2_LDS_SYNC();            // both lanes' rows are in LDS
            double A[NS][NS];
#pragma unroll
            for (int i = 0; i < NS; ++i)
#pragma unroll
                for (int c = 0; c < NS; ++c) A[i][c] = As[(i * NS + c) * GPB];
            HY2_LDS_SYNC();            // both lanes have read W before either overwrites it with factors
            const bool ok = lu_factor_to_lds<NS, GPB>(A, As, dinv, piv, anyp);
            HY2_LDS_SYNC();
            return ok;
        };

        // ================================================================== forward sweep
        double u[NS];
        HyPoint2<NS, NR> p0;     // FSAL point (u, t)
        double t = t0, dt = 0.0, lqold = lqinit;
        int iter = 0, jsave = 0, nacc = 0, nrej = 0;
        int rc = valid ? -1 : 0;
#pragma unroll
        for (int i = 0; i < NS; ++i) u[i] = prm.u0[(size_t)i * prm.B + b];
        {
            double T, P, Td, Pd;
            tab(t0, T, P, Td, Pd);
            hy_point2<NS, NR>(th, kc, hp.inv_R, u, T, P, m1, ci, ow, p0);
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
            HyPoint2<NS, NR> p1;
            tab(t0 + dt0, T, P, Td, Pd);
            hy_point2<NS, NR>(th, kc, hp.inv_R, u1, T, P, m1, ci, ow, p1);
            double d2 = 0.0;
#pragma unroll
            for (int i = 0; i < NS; ++i) { const double e = (p1.f[i] - p0.f[i]) * sk[i]; d2 = fma(e, e, d2); }
            d2 = sqrt(d2 * (1.0 / NS)) / dt0;
            const double dm = fmax(d1, d2);
            const double dt1 = (dm <= 1e-15) ? fmax(1e-6, dt0 * 1e-3) : exp(-0.5 * (4.605170185988091368 + flog(dm)));
            dt = fmax(kc->dtmin, fmin(fmin(100.0 * dt0, dt1), dtmax));
        }
        if (start_saved) {
            if (valid && prm.pred && !m1) {
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
                    HY_FRESH_THETA(th); HY_FRESH_KC(kc);
                    const double gam = d_ * dt;
                    const double tnew = last ? tend : t + dt;
                    double T, P, Td, Pd;
                    tab(t, T, P, Td, Pd);
                    double dinv[NS], ft[NS];
                    int piv[NS];
                    bool anyp;
                    hy_jac_ft2<NS, NR, GPB>(th, kc, p0, gam, Pd * frcp(P) - Td * frcp(T), -hp.inv_R * Td * frcp(T * T), Td * frcp(T), m1, ci, ow, As, ft);
                    CRNN_SCHED_FENCE();
                    const bool okf = factor(dinv, piv, anyp);
                    const bool wp = __builtin_amdgcn_ballot_w64(anyp) != 0;
                    double k1[NS], dk[NS], unew[NS], f1[NS];
#pragma unroll
                    for (int i = 0; i < NS; ++i) k1[i] = fma(gam, ft[i], p0.f[i]);
                    lu_solve_lds<NS, GPB>(As, dinv, piv, wp, k1);
                    CRNN_SCHED_FENCE();
                    HY_FRESH_THETA(th); HY_FRESH_KC(kc);
                    {
                        double u1[NS];
#pragma unroll
                        for (int i = 0; i < NS; ++i) u1[i] = fma(0.5 * dt, k1[i], u[i]);
                        HyPoint2<NS, NR> p1;
                        double T1, P1, a_, b_;
                        tab(t + 0.5 * dt, T1, P1, a_, b_);
                        hy_point2<NS, NR>(th, kc, hp.inv_R, u1, T1, P1, m1, ci, ow, p1);
#pragma unroll
                        for (int i = 0; i < NS; ++i) f1[i] = p1.f[i];
                        opaque(f1);
                    }
#pragma unroll
                    for (int i = 0; i < NS; ++i) dk[i] = f1[i] - k1[i];
                    lu_solve_lds<NS, GPB>(As, dinv, piv, wp, dk);
#pragma unroll
                    for (int i = 0; i < NS; ++i) unew[i] = fma(dt, k1[i] + dk[i], u[i]);
                    CRNN_SCHED_FENCE();
                    HyPoint2<NS, NR> p2;
                    HY_FRESH_THETA(th); HY_FRESH_KC(kc);
                    {
                        double T2, P2, a_, b_;
                        tab(tnew, T2, P2, a_, b_);
                        hy_point2<NS, NR>(th, kc, hp.inv_R, unew, T2, P2, m1, ci, ow, p2);
                    }
                    CRNN_SCHED_FENCE();
                    double k3[NS];
#pragma unroll
                    for (int i = 0; i < NS; ++i) {
                        const double k2i = k1[i] + dk[i];
                        k3[i] = fma(dt, ft[i], p2.f[i] - c32 * (k2i - f1[i]) - 2.0 * (k1[i] - p0.f[i]));
                    }
                    lu_solve_lds<NS, GPB>(As, dinv, piv, wp, k3);
                    double es = 0.0;
                    bool finite = okf;
#pragma unroll
                    for (int i = 0; i < NS; ++i) {
                        const double k2i = k1[i] + dk[i];
                        const double ev = dt * (1.0 / 6.0) * (k1[i] - 2.0 * k2i + k3[i]);
                        const double mx = fmax(fabs(u[i]), fabs(unew[i]));
                        const double e = ev * frcp(fma(kc->rtol[i], mx, kc->atol[i]));
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
                                if (!m1) atomicAdd(hp.overflow, 1u);
                            } else {
                                double *rec = tape + (size_t)nacc * RECW;
                                if (!m1) { rec[0] = t; rec[1] = dt; }
#pragma unroll
                                for (int i = 0; i < H; ++i)
                                    if (ow[i]) rec[2 + ci[i]] = m1 ? u[H + i < NS ? H + i : 0] : u[i];
                                ++nacc;
                                while (jsave < nsave) {
                                    const double ts = ts_lds[jsave];
                                    if (!(ts <= tnew)) break;
                                    if (prm.pred && !m1) {
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
        double loss_sum = 0.0;            // this lane's species only; the pair's sum is formed at the end
        double tnew = t;
        int s = valid ? nacc - 1 : -1;
        // accumulator m of queue position r = wave_base + giw: gacc[(r >> 6) NTH 64 + m 64 + (r & 63)] (reduce_gacc_kernel's layout)
        double *const gacc = hp.gacc + (size_t)((wave_base + giw) >> 6) * NTH * 64 + ((wave_base + giw) & 63);
        const double *const drows = prm.data + (size_t)b * prm.row_stride;
        int doff[H];
        bool obs[H];
#pragma unroll
        for (int i = 0; i < H; ++i) { const int dr = ow[i] ? (int)kc->drow[ci[i]] : -1; obs[i] = dr >= 0; doff[i] = obs[i] ? dr : 0; }
        double rt = 0.0, rdt = 0.0, ru[NS];
        auto load_rec = [&](int idx) {
            const double *rec = tape + (size_t)(idx > 0 ? idx : 0) * RECW;
            rt = rec[0]; rdt = rec[1];
#pragma unroll
            for (int i = 0; i < NS; ++i) ru[i] = rec[2 + i];
        };
        load_rec(s);

        while (__builtin_amdgcn_ballot_w64(s >= 0) != 0) {
            if (s >= 0) {
                const double tn = rt, h = rdt;
                double un[NS];
#pragma unroll
                for (int i = 0; i < NS; ++i) un[i] = ru[i];
                // ---- re-form the step
                HY_FRESH_THETA(th); HY_FRESH_KC(kc);
                const double gam = d_ * h;
                double T, P, Td, Pd;
                tab(tn, T, P, Td, Pd);
                const double ld = Pd * frcp(P) - Td * frcp(T), xEd = -hp.inv_R * Td * frcp(T * T), xLd = Td * frcp(T);
                HyPoint2<NS, NR> pn, pm;
                hy_point2<NS, NR>(th, kc, hp.inv_R, un, T, P, m1, ci, ow, pn);
                opaque(pn.r); opaque(pn.Y); opaque(pn.f); opaque(pn.fo); opaque(pn.xo); opaque(pn.irho); opaque(pn.iS);
                HY_FRESH_THETA(th); HY_FRESH_KC(kc);
                double dinv[NS], ft[NS];
                int piv[NS];
                bool anyp;
                double k1[NS], dk[NS];
                hy_jac_ft2<NS, NR, GPB>(th, kc, pn, gam, ld, xEd, xLd, m1, ci, ow, As, ft);
#pragma unroll
                for (int i = 0; i < NS; ++i) k1[i] = fma(gam, ft[i], pn.f[i]);
                opaque(k1);
                CRNN_SCHED_FENCE();
                (void)factor(dinv, piv, anyp);
                const bool wp = __builtin_amdgcn_ballot_w64(anyp) != 0;
                lu_solve_lds<NS, GPB>(As, dinv, piv, wp, k1);
                HY_FRESH_THETA(th); HY_FRESH_KC(kc);
                {
                    double u1[NS], T1, P1, a_, b_;
#pragma unroll
                    for (int i = 0; i < NS; ++i) u1[i] = fma(0.5 * h, k1[i], un[i]);
                    tab(tn + 0.5 * h, T1, P1, a_, b_);
                    hy_point2<NS, NR>(th, kc, hp.inv_R, u1, T1, P1, m1, ci, ow, pm);
                    opaque(pm.r); opaque(pm.Y); opaque(pm.f); opaque(pm.xo); opaque(pm.irho); opaque(pm.iS);
                }
#pragma unroll
                for (int i = 0; i < NS; ++i) dk[i] = pm.f[i] - k1[i];
                lu_solve_lds<NS, GPB>(As, dinv, piv, wp, dk);
                CRNN_SCHED_FENCE();

                // ---- loss and seeds at the save points inside (tn, tnew]: each lane its own species
                double k1o[H], dko[H], uno[H];
                pair_own<NS, H>(k1, m1, k1o);
                pair_own<NS, H>(dk, m1, dko);
                pair_own<NS, H>(un, m1, uno);
                double Ao[H], B1o[H], B2o[H];
#pragma unroll
                for (int i = 0; i < H; ++i) { Ao[i] = 0.0; B1o[i] = 0.0; B2o[i] = 0.0; }
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
                            const double k2i = k1o[i] + dko[i];
                            double v = at_end ? fma(h, k2i, uno[i]) : fma(h, fma(c1, k1o[i], c2 * k2i), uno[i]);
                            double mask = 1.0;
                            if (prm.clamp_pred) {
                                mask = (v > kc->ub || v < -kc->ub) ? 0.0 : 1.0;
                                v = clampv(v, -kc->ub, kc->ub);
                            }
                            const double iy = kc->inv_yscale[ci[i]];
                            const double rr = (row[doff[i]] - v) * iy;
                            double w;
                            if (prm.loss_kind == 0) { loss_sum += fabs(rr); w = signbit(rr) ? 1.0 : -1.0; }
                            else { loss_sum = fma(rr, rr, loss_sum); w = -2.0 * rr; }
                            w *= mask * iy;
                            Ao[i] += w;
                            B1o[i] = fma(w, h * c1, B1o[i]);
                            B2o[i] = fma(w, h * c2, B2o[i]);
                        }
                    }
                    --jsave;
                }
                load_rec(s - 1);     // next tape record, fetched and awaited before this step's accumulator atomics are issued
                opaque(rt); opaque(rdt); opaque(ru);
                if (GRAD) {
                    double A_[NS], B1[NS], B2[NS];
                    pair_gather<NS, H>(Ao, m1, A_);
                    pair_gather<NS, H>(B1o, m1, B1);
                    pair_gather<NS, H>(B2o, m1, B2);
                    // park what the last phase of the step needs again (the pair shares the slot: each lane its species' x, lane 0 the rest)
#pragma unroll
                    for (int i = 0; i < H; ++i)
                        if (ow[i]) park[ci[i] * GPB] = pn.xo[i];
                    if (!m1) {
                        park[NS * GPB] = pn.xE; park[(NS + 1) * GPB] = pn.xL;
#pragma unroll
                        for (int j = 0; j < NR; ++j) { park[(NS + 2 + j) * GPB] = pn.r[j]; park[(PK_R1 + j) * GPB] = pm.r[j]; }
#pragma unroll
                        for (int i = 0; i < NS; ++i) {
                            park[(NS + 2 + NR + i) * GPB] = pn.Y[i];
                            park[(2 * NS + 4 + NR + i) * GPB] = k1[i];
                            park[(3 * NS + 4 + NR + i) * GPB] = dk[i];
                        }
                        park[(2 * NS + 2 + NR) * GPB] = pn.irho;
                        park[(2 * NS + 3 + NR) * GPB] = pn.iS;
                    }
                    const unsigned ncY = pn.cY, ncC = pn.cC;
                    CRNN_SCHED_FENCE();
                    double kb1[NS], v[NS], ub[NS];
#pragma unroll
                    for (int i = 0; i < NS; ++i) { v[i] = fma(h, lam[i], B2[i]); ub[i] = lam[i] + A_[i]; kb1[i] = B1[i] + v[i]; }
                    lu_solve_T_lds<NS, GPB>(As, dinv, piv, wp, v);
#pragma unroll
                    for (int i = 0; i < NS; ++i) kb1[i] -= v[i];
                    double vto[H];      // (gsc .* v) of this lane's species
                    {
                        double vo[H];
                        pair_own<NS, H>(v, m1, vo);
#pragma unroll
                        for (int i = 0; i < H; ++i) vto[i] = ow[i] ? vo[i] * kc->gsc[ci[i]] : 0.0;
                    }
                    opaque(vto); opaque(kb1); opaque(ub);
                    CRNN_SCHED_FENCE();
                    HY_FRESH_THETA(th); HY_FRESH_KC(kc);
                    // -------- point u_mid: adjoint of v.f
                    double irho_mid = pm.irho;
                    opaque(irho_mid);
                    {
                        double P2o[H], psi = 0.0;
#pragma unroll
                        for (int i = 0; i < H; ++i) P2o[i] = 0.0;
#pragma unroll 1
                        for (int j = 0; j < NR; ++j) {
                            const double *wo_ = th + L_::wo(0, j), *wi_ = th + L_::wi(0, j);
                            double At = 0.0;
#pragma unroll
                            for (int i = 0; i < H; ++i) At = fma(vto[i], wo_[ci[i]], At);
                            At = pair_sum(At);
                            const double ir = pm.irho * pm.r[j];
                            const double Psi = At * ir;
                            psi += Psi;
                            double *gj = gacc + (size_t)L_::wi(0, j) * 64;
#pragma unroll
                            for (int i = 0; i < H; ++i) {
                                if (ow[i]) HY_ACC(gj + (size_t)ci[i] * 64, Psi * pm.xo[i]);
                                P2o[i] = fma(Psi, ow[i] ? wi_[ci[i]] : 0.0, P2o[i]);
                            }
                            if (!m1) { HY_ACC(gj + (size_t)NS * 64, Psi * pm.xE); HY_ACC(gj + (size_t)(NS + 1) * 64, Psi * pm.xL); }
                        }
                        double scp = 0.0;
#pragma unroll
                        for (int i = 0; i < H; ++i) scp += (ow[i] && ((pm.cC >> ci[i]) & 1u)) ? P2o[i] : 0.0;
                        scp = pair_sum(scp);
                        double mo[H], mf[NS];
#pragma unroll
                        for (int i = 0; i < H; ++i) {
                            const bool iy = ow[i] && ((pm.cY >> ci[i]) & 1u), ic = (pm.cC >> ci[i]) & 1u;
                            double m_ = iy ? kc->imw[ci[i]] * pm.iS * (psi - scp) : 0.0;
                            if (iy && ic) m_ = fma(P2o[i], frcp(m1 ? pm.Y[H + i < NS ? H + i : 0] : pm.Y[i]), m_);
                            mo[i] = m_;
                        }
                        pair_gather<NS, H>(mo, m1, mf);
#pragma unroll
                        for (int c = 0; c < NS; ++c) { ub[c] += mf[c]; kb1[c] = fma(0.5 * h, mf[c], kb1[c]); }
                    }
                    CRNN_SCHED_FENCE();
                    lu_solve_T_lds<NS, GPB>(As, dinv, piv, wp, kb1);     // kb1 = w
                    opaque(kb1); opaque(ub); opaque(vto);
                    CRNN_SCHED_FENCE();
                    HY_FRESH_THETA(th); HY_FRESH_KC(kc);
                    // -------- point u_n: adjoint of w.f + gam ( v.Df[(dk,0)] + w.Df[(k1,1)] )
                    {
                        HY2_LDS_SYNC();        // the pair's parked values (each lane wrote a part)
                        unsigned zp_ = 0;
                        asm volatile("" : "+v"(zp_));
                        const double *pk = park + zp_;
                        double xno[H], Yo[H], k1p[H], dkp[H], wto[H];
#pragma unroll
                        for (int i = 0; i < H; ++i) {
                            xno[i] = pk[ci[i] * GPB];
                            Yo[i] = pk[(NS + 2 + NR + ci[i]) * GPB];
                            k1p[i] = pk[(2 * NS + 4 + NR + ci[i]) * GPB];
                            dkp[i] = pk[(3 * NS + 4 + NR + ci[i]) * GPB];
                        }
                        const double xnE = pk[NS * GPB], xnL = pk[(NS + 1) * GPB];
                        const double n_irho = pk[(2 * NS + 2 + NR) * GPB], n_iS = pk[(2 * NS + 3 + NR) * GPB];
                        {
                            double wo2[H];
                            pair_own<NS, H>(kb1, m1, wo2);
#pragma unroll
                            for (int i = 0; i < H; ++i) wto[i] = ow[i] ? wo2[i] * kc->gsc[ci[i]] : 0.0;
                        }
                        // direction data (this lane's species)
                        double Spv = 0.0, Spw = 0.0, xpvo[H], xpwo[H];
#pragma unroll
                        for (int i = 0; i < H; ++i) {
                            const double sg = (ow[i] && ((ncY >> ci[i]) & 1u)) ? kc->imw[ci[i]] * n_iS : 0.0;
                            Spv = fma(sg, dkp[i], Spv);
                            Spw = fma(sg, k1p[i], Spw);
                        }
                        Spv = pair_sum(Spv);
                        Spw = pair_sum(Spw);
                        const double lpv = -Spv, lpw = ld - Spw;
#pragma unroll
                        for (int i = 0; i < H; ++i) {
                            const bool iy = ow[i] && ((ncY >> ci[i]) & 1u), ic = ow[i] && ((ncC >> ci[i]) & 1u);
                            const double gy = iy ? frcp(Yo[i]) : 0.0;
                            xpvo[i] = ic ? fma(gy, dkp[i], lpv) : 0.0;
                            xpwo[i] = ic ? fma(gy, k1p[i], lpw) : 0.0;
                        }
                        double PEo[H], P2vo[H], P2wo[H], SE = 0.0, psiv = 0.0, psiw = 0.0;
#pragma unroll
                        for (int i = 0; i < H; ++i) { PEo[i] = 0.0; P2vo[i] = 0.0; P2wo[i] = 0.0; }
#pragma unroll 1
                        for (int j = 0; j < NR; ++j) {
                            const double *wo_ = th + L_::wo(0, j), *wi_ = th + L_::wi(0, j);
                            double Av = 0.0, Aw = 0.0, zv = 0.0, zw = 0.0;
#pragma unroll
                            for (int i = 0; i < H; ++i) {
                                const double wij = ow[i] ? wi_[ci[i]] : 0.0;
                                Av = fma(vto[i], wo_[ci[i]], Av);
                                Aw = fma(wto[i], wo_[ci[i]], Aw);
                                zv = fma(wij, xpvo[i], zv);
                                zw = fma(wij, xpwo[i], zw);
                            }
                            Av = pair_sum(Av); Aw = pair_sum(Aw); zv = pair_sum(zv);
                            zw = pair_sum(zw) + fma(wi_[NS], xEd, wi_[NS + 1] * xLd);
                            const double ir = n_irho * pk[(NS + 2 + j) * GPB];
                            const double Pv = Av * ir, Pw = Aw * ir;
                            const double yv = zv - lpv, yw = zw - lpw;
                            const double cw = fma(gam, yw, 1.0), cv = gam * yv;
                            const double E = fma(Pw, cw, Pv * cv);
                            SE += E; psiv += Pv; psiw += Pw;
                            const double irm = irho_mid * pk[(PK_R1 + j) * GPB];   // the u_mid point's irho r_j
                            if (!m1) HY_ACC(gacc + (size_t)L_::wb(j) * 64, fma(Av, irm, E));
                            const double gPv = gam * Pv, gPw = gam * Pw;
                            double *gj = gacc + (size_t)L_::wi(0, j) * 64;
                            double *go = gacc + (size_t)L_::wo(0, j) * 64;
#pragma unroll
                            for (int i = 0; i < H; ++i) {
                                if (ow[i]) {
                                    HY_ACC(gj + (size_t)ci[i] * 64, fma(E, xno[i], fma(gPw, xpwo[i], gPv * xpvo[i])));
                                    HY_ACC(go + (size_t)ci[i] * 64, fma(vto[i], fma(ir, cv, irm), wto[i] * (ir * cw)));
                                }
                                const double wij = ow[i] ? wi_[ci[i]] : 0.0;
                                PEo[i] = fma(E, wij, PEo[i]);
                                P2vo[i] = fma(Pv, wij, P2vo[i]);
                                P2wo[i] = fma(Pw, wij, P2wo[i]);
                            }
                            if (!m1) {
                                HY_ACC(gj + (size_t)NS * 64, fma(E, xnE, gPw * xEd));
                                HY_ACC(gj + (size_t)(NS + 1) * 64, fma(E, xnL, gPw * xLd));
                            }
                        }
                        double scE = 0.0, scv = 0.0, scw = 0.0;
#pragma unroll
                        for (int i = 0; i < H; ++i) {
                            const bool ic = ow[i] && ((ncC >> ci[i]) & 1u);
                            scE += ic ? PEo[i] : 0.0;
                            scv += ic ? P2vo[i] : 0.0;
                            scw += ic ? P2wo[i] : 0.0;
                        }
                        scE = pair_sum(scE); scv = pair_sum(scv); scw = pair_sum(scw);
                        const double brk = (SE - scE) + gam * fma(Spw, scw - psiw, Spv * (scv - psiv));
                        double mo[H], mf[NS];
#pragma unroll
                        for (int i = 0; i < H; ++i) {
                            const bool iy = ow[i] && ((ncY >> ci[i]) & 1u), ic = (ncC >> ci[i]) & 1u;
                            double m_ = iy ? kc->imw[ci[i]] * n_iS * brk : 0.0;
                            if (iy && ic) {
                                const double gy = frcp(Yo[i]);
                                m_ = fma(gy, PEo[i] - gam * gy * fma(P2wo[i], k1p[i], P2vo[i] * dkp[i]), m_);
                            }
                            mo[i] = m_;
                        }
                        pair_gather<NS, H>(mo, m1, mf);
#pragma unroll
                        for (int c = 0; c < NS; ++c) lam[c] = ub[c] + mf[c];
                        HY2_LDS_SYNC();        // the parked values are consumed before the next step parks again
                    }
                }
                tnew = tn;
                --s;
            }
        }

        {
            if (start_saved && n_saved >= 1) {
#pragma unroll
                for (int i = 0; i < H; ++i) {
                    if (obs[i]) {
                        double v = prm.u0[(size_t)ci[i] * prm.B + b];
                        if (prm.clamp_pred) v = clampv(v, -kc->ub, kc->ub);
                        const double rr = (drows[doff[i]] - v) * kc->inv_yscale[ci[i]];
                        loss_sum += (prm.loss_kind == 0) ? fabs(rr) : rr * rr;
                    }
                }
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
}

}  // namespace crnn

// crnn_amd/csrc/hychem_sens_kernel.hpp -- gfx950 (MI355X): the HyChem gradient as the reference evaluates it.
//
// Reference: HyChem/crnn_pyrolysis_mass.jl:201  grad = ForwardDiff.gradient(x -> loss_n_ode(x, sample), p): Duals through the adaptive
// solve.  ForwardDiff works through the 211 parameters in chunks of 12 (pickchunksize(211) = 12: seventeen chunks of 12 and one of
// 7 with five zero partials), EVERY CHUNK IS ITS OWN ADAPTIVE SOLVE, and DiffEqBase's norm of a Dual-valued state weighs the
// partials with the value -- the accept / reject decisions and step sizes of a chunk see that chunk's tangents
// (ros23_sens_kernel.hpp has the norm; [UNVERIFIED-DEP] like it).  crnn_config.errnorm_sens = 1 / 2 on the HyChem preset selects
// this kernel for gradient calls; it is also the only forward-tangent path of the HyChem model (the default gradient is the
// discrete adjoint, hychem2_kernel.hpp: one forward and one reverse sweep instead of 18 x (1 + 12) forward solves).
//
// This is the GENERAL kernel of that mode: any twelve directions d theta (dense rows).  The rows of p2vec's Jacobian -- every gradient
// call of the training loop -- touch one reaction each and run through hychem_sens2_kernel.hpp instead (sparse directions; crnn_capi.hip
// checks the rows and picks); what arrives here are a caller's own directions through crnn_solve.
//
// Mapping: a GROUP OF 12 LANES per trajectory, one tangent column per lane (five groups per wavefront, four lanes idle); every lane of a
// group carries the primal redundantly -- hychem_kernel.hpp's point evaluation, analytic Jacobian and pivoted LU, the same operations in
// the same order in all twelve lanes, so the group never diverges; ONE copy of W's factors per trajectory in LDS (the lanes store
// identical values to one address, reads are broadcasts: 31 KB per block of 128, two blocks per CU) -- and ITS column through EVERY
// ATTEMPT (the decision needs the tangents), including the third stage's
//     W k3' = f2' - c32 (k2' - f1') - 2 (k1' - f0') + dt ft' + gam J' k3
// that only the error estimate uses.  The column's tangents are hychem_tan.hpp's closed forms on the primal evaluations the attempt
// already holds (no logarithm or exponential taken twice; one point for the three mixed derivatives of a step), pinned on the host
// against the complex step (tests/test_hychem.py); only the two evaluations of the initial step size go through the small dual-number
// type below (hy_f<T> over first-order duals).  The group sums the lanes' contributions to the norm by ds_bpermute in lane order, takes
// the decision, and only then commits the attempt: the new tangent column, the column's gradient increments at the save points inside
// the step.  (Round 4 shipped this kernel with nested duals for everything and a copy of W per lane -- 5.2 KB of scratch, 108 KB of LDS;
// round 5 ran both variants through the test suite's SIMT emulator against the CPU restatement, chunk for chunk, and kept this one: 3.5 KB, 31 KB.)
#pragma once
#include "hychem_kernel.hpp"
#include "hychem_tan.hpp"

namespace crnn {

// ---- dual numbers: value + one partial; nests (Du<Du<double>>: value, eps, del, eps del)
template <class S>
struct Du {
    S v, d;
    __device__ __forceinline__ Du() {}
    __device__ __forceinline__ Du(double c) : v(c), d(0.0) {}
    __device__ __forceinline__ Du(const S &v_, const S &d_) : v(v_), d(d_) {}
};
__device__ __forceinline__ double du_val(double a) { return a; }
template <class S>
__device__ __forceinline__ double du_val(const Du<S> &a) { return du_val(a.v); }
template <class S>
__device__ __forceinline__ Du<S> operator+(const Du<S> &a, const Du<S> &b) { return Du<S>(a.v + b.v, a.d + b.d); }
template <class S>
__device__ __forceinline__ Du<S> operator-(const Du<S> &a, const Du<S> &b) { return Du<S>(a.v - b.v, a.d - b.d); }
template <class S>
__device__ __forceinline__ Du<S> operator*(const Du<S> &a, const Du<S> &b) { return Du<S>(a.v * b.v, a.v * b.d + a.d * b.v); }
template <class S>
__device__ __forceinline__ Du<S> operator/(const Du<S> &a, const Du<S> &b) {
    const S q = a.v / b.v;
    return Du<S>(q, (a.d - q * b.d) / b.v);
}
__device__ __forceinline__ double du_log(double a) { return log(a); }
__device__ __forceinline__ double du_exp(double a) { return exp(a); }
template <class S>
__device__ __forceinline__ Du<S> du_log(const Du<S> &a) { return Du<S>(du_log(a.v), a.d / a.v); }
template <class S>
__device__ __forceinline__ Du<S> du_exp(const Du<S> &a) {
    const S e = du_exp(a.v);
    return Du<S>(e, a.d * e);
}
// Julia's clamp on a Dual: inside the closed window the number passes through, outside the bound (a constant) comes back
template <class T>
__device__ __forceinline__ T du_clamp(const T &a, const double lo, const double hi) {
    const double v = du_val(a);
    return v > hi ? T(hi) : (v < lo ? T(lo) : a);
}

// The HyChem right-hand side (crnn_pyrolysis_mass.jl:121-131) over T = double, Du<double> or Du<Du<double>>:
//   Y = clamp(u, lb, 10); rho = P / (Ru T sum Y/MW); C = rho Y / MW 1e3; x = [log clamp(C, lb, 10); -1/(R T); log T];
//   du = w_out exp(w_in' x + w_b) MW / rho dydt_scale.   thf(m): theta_m as a T.
template <int NS, int NR, class T, class TH>
__device__ __forceinline__ void hy_f(const TH &thf, const KConst *kc, const double inv_R, const T (&u)[NS], const T &Tt, const T &Pt, T (&f)[NS]) {
    using L_ = LayH<NS, NR>;
    T Y[NS], S(0.0);
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        Y[i] = du_clamp(u[i], kc->lb, kc->ub);
        S = S + Y[i] * T(kc->imw[i]);
    }
    const T rho = Pt / (T(kc->Ru) * Tt * S);
    T x[NS + 2];
#pragma unroll
    for (int i = 0; i < NS; ++i) x[i] = du_log(du_clamp(rho * (Y[i] * T(kc->imw[i])) * T(1e3), kc->lb, kc->ub));
    x[NS] = T(inv_R) / Tt;
    x[NS + 1] = du_log(Tt);
    T r[NR];
    for (int j = 0; j < NR; ++j) {      // (rolled: one column of w_in in flight)
        T z = thf(L_::wb(j));
#pragma unroll
        for (int m = 0; m < NS + 2; ++m) z = z + thf(L_::wi(m, j)) * x[m];
        r[j] = du_exp(z);
    }
    const T irho = T(1.0) / rho;
    for (int i = 0; i < NS; ++i) {
        T a(0.0);
#pragma unroll
        for (int j = 0; j < NR; ++j) a = a + thf(L_::wo(i, j)) * r[j];
        f[i] = a * T(kc->gsc[i]) * irho;
    }
}

struct HySensParams {
    const double *dth;     // [n_dir][NTH] the chunk's directions d theta / d p_k (rows), n_dir <= 12
    int32_t n_dir;         // real directions of this chunk (the others are zero partials)
    int32_t mode;          // 1: squared norm / length(u); 2: / totallength(u) = ns (1 + dual_partials)
    int32_t dual_partials; // partials per Dual (12)
    int32_t n_chunks;      // > 1: dth holds all n_total directions; block b works on chunk b % n_chunks (its 12 rows of dth), the
    int32_t n_total;       //      blocks of a chunk share its trajectories; gradient rows compact [count][n_total]; no per-trajectory
                           //      losses / statistics (the plain solve behind the chunks writes them).  One launch instead of 18: a
                           //      chunk launch of 1 024 trajectories is 192 wavefronts, a fifth of the chip, for one generation
};

template <int NS, int NR, int BLOCK>
__global__ __launch_bounds__(BLOCK) void hychem_sens_kernel(const SolveParams prm, const double *__restrict__ theta, const HyParams hp,
                                                            const HySensParams sp) {
    using L_ = LayH<NS, NR>;
    constexpr int NTH = L_::NTH;
    constexpr int C = 12;                       // lanes per trajectory = columns of a chunk
    constexpr int GPW = 64 / C;                 // groups per wavefront (5; lanes 60-63 idle)
    __shared__ double kc_lds[kNConst];
    __shared__ double ts_lds[kMaxSave];
    __shared__ double th_lds[NTH];
    __shared__ double dth_lds[C * NTH];
    constexpr int LUS = (BLOCK / 64) * GPW;     // W's factors of every trajectory: element e of group g at lu_lds[e * LUS + g]
    __shared__ double lu_lds[NS * NS * LUS];
    const int tid = threadIdx.x;
    for (int idx = tid; idx < kNConst; idx += BLOCK) kc_lds[idx] = reinterpret_cast<const double *>(prm.kc)[idx];
    for (int idx = tid; idx < hp.n_save_total; idx += BLOCK) ts_lds[idx] = prm.tsave[idx];
    for (int idx = tid; idx < NTH; idx += BLOCK) th_lds[idx] = theta[idx];
    const int nch = sp.n_chunks > 1 ? sp.n_chunks : 1;
    const int cid = nch > 1 ? (int)(blockIdx.x % nch) : 0;
    const int ndir = nch > 1 ? min(C, sp.n_total - cid * C) : sp.n_dir;       // real directions of this block's chunk
    const double *const dth_g = sp.dth + (size_t)cid * C * NTH;
    for (int idx = tid; idx < C * NTH; idx += BLOCK) dth_lds[idx] = (idx / NTH) < ndir ? dth_g[idx] : 0.0;
    __syncthreads();
    const KConst *kc = reinterpret_cast<const KConst *>(kc_lds);
    const double *th = th_lds;
    const int lane = tid & 63;
    const int grp = lane / C, col = lane - grp * C;
    const bool lane_on = grp < GPW;
    const int gbase = grp * C;                  // first lane of the group within the wavefront
    const double *const dthc = dth_lds + (lane_on ? col : 0) * NTH;
    double *const As = lu_lds + (tid >> 6) * GPW + (lane_on ? grp : 0);
    const int64_t groups_total = (int64_t)(gridDim.x / nch) * (BLOCK / 64) * GPW;      // groups working on this block's chunk
    int64_t traj = ((int64_t)(blockIdx.x / nch) * (BLOCK / 64) + (tid >> 6)) * GPW + grp;
    if (!lane_on) traj = prm.count;             // the idle lanes never start a trajectory

    const double d_ = 0.29289321881345248, c32 = 7.4142135623730950, inv12d = 2.4142135623730950;
    const int nsave = prm.n_save, Dfull = hp.n_save_total;
    const double tend = ts_lds[nsave - 1], ts0 = ts_lds[0], t0 = kc->t0;
    const double dtmax = tend - t0;
    const double lqinit = flog(kc->qoldinit);
    const bool start_saved = (ts0 == t0);
    const double inv_div = sp.mode == 2 ? 1.0 / ((double)NS * (1.0 + (double)sp.dual_partials)) : 1.0 / (double)NS;

    typedef Du<double> D1;
    auto th1 = [&](const int m) -> D1 { return D1(th[m], dthc[m]); };
    auto group_sum = [&](const double v) -> double {
        double a = 0.0;
#pragma unroll
        for (int q = 0; q < C; ++q) a += __shfl(v, gbase + q);
        return a;
    };

    while (traj < prm.count) {
        const int64_t b = prm.first + traj;
        CRNN_CHK(b >= 0 && b < prm.B, 26);
        const double *const tabT = hp.tabs + (size_t)b * 2 * Dfull;
        const double *const tabP = tabT + Dfull;
        auto tab = [&](const double tq, double &T, double &P, double &Td, double &Pd) {
            int sg = 0;
            while (sg + 1 < Dfull - 1 && ts_lds[sg + 1] <= tq) ++sg;
            const double idts = frcp(ts_lds[sg + 1] - ts_lds[sg]);
            Td = (tabT[sg + 1] - tabT[sg]) * idts;
            Pd = (tabP[sg + 1] - tabP[sg]) * idts;
            T = fma(tq - ts_lds[sg], Td, tabT[sg]);
            P = fma(tq - ts_lds[sg], Pd, tabP[sg]);
        };
        // first-order tangent f'(point; s) of this lane's column
        auto jvp = [&](const double (&uu)[NS], const double (&ss)[NS], const double Tq, const double Pq, double (&fp)[NS]) {
            D1 ud[NS], fd[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) ud[i] = D1(uu[i], ss[i]);
            hy_f<NS, NR, D1>(th1, kc, hp.inv_R, ud, D1(Tq), D1(Pq), fd);
#pragma unroll
            for (int i = 0; i < NS; ++i) fp[i] = fd[i].d;
        };
        double u[NS], s[NS], f0[NS], f0p[NS];
        HyPoint<NS, NR> p0;
        double t = t0, dt = 0.0, lqold = lqinit, loss_sum = 0.0, gsum = 0.0;
        int iter = 0, jsave = 0, nacc = 0, nrej = 0, rc = -1;
#pragma unroll
        for (int i = 0; i < NS; ++i) { u[i] = prm.u0[(size_t)i * prm.B + b]; s[i] = 0.0; }
        {
            double T, P, Td, Pd;
            tab(t0, T, P, Td, Pd);
            hy_point<NS, NR>(th, kc, hp.inv_R, u, T, P, p0);
#pragma unroll
            for (int i = 0; i < NS; ++i) f0[i] = p0.f[i];
            jvp(u, s, T, P, f0p);
            // Hairer's initial step with the dual-inclusive norms (ros23_sens_kernel.hpp: sens_init_dt)
            double sk[NS], d0 = 0.0, d1 = 0.0, d1p = 0.0;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                sk[i] = frcp(fma(fabs(u[i]), kc->rtol[i], kc->atol[i]));
                const double a = u[i] * sk[i], c = f0[i] * sk[i], e = f0p[i] * sk[i];
                d0 = fma(a, a, d0); d1 = fma(c, c, d1); d1p = fma(e, e, d1p);
            }
            d1 += group_sum(d1p);
            d0 = sqrt(d0 * inv_div); d1 = sqrt(d1 * inv_div);
            double dt0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * (d0 / d1);
            dt0 = fmin(dt0, dtmax);
            double u1[NS], s1[NS], f1p[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) { u1[i] = fma(dt0, f0[i], u[i]); s1[i] = dt0 * f0p[i]; }
            HyPoint<NS, NR> p1;
            tab(t0 + dt0, T, P, Td, Pd);
            hy_point<NS, NR>(th, kc, hp.inv_R, u1, T, P, p1);
            jvp(u1, s1, T, P, f1p);
            double d2 = 0.0, d2p = 0.0;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const double e = (p1.f[i] - f0[i]) * sk[i], ep = (f1p[i] - f0p[i]) * sk[i];
                d2 = fma(e, e, d2); d2p = fma(ep, ep, d2p);
            }
            d2 += group_sum(d2p);
            d2 = sqrt(d2 * inv_div) / dt0;
            const double dm = fmax(d1, d2);
            const double dt1 = (dm <= 1e-15) ? fmax(1e-6, dt0 * 1e-3) : exp(-0.5 * (4.605170185988091368 + flog(dm)));
            dt = fmax(kc->dtmin, fmin(fmin(100.0 * dt0, dt1), dtmax));
        }
        // a save point: prediction / loss term of the primal (written by the group's first lane), the column's gradient increment
        auto save_point = [&](const double (&v_)[NS], const double (&vp)[NS], const int j) {
            CRNN_CHK(j >= 0 && (int64_t)(j + 1) * prm.n_obs <= prm.row_stride, 27);
            const double *prow = prm.data + (size_t)b * prm.row_stride + (size_t)j * prm.n_obs;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                double v = v_[i], dv = vp[i];
                if (prm.clamp_pred) { const double cl = clampv(v, -kc->ub, kc->ub); dv = (cl == v) ? dv : 0.0; v = cl; }
                if (prm.pred && col == 0) prm.pred[((size_t)j * NS + i) * prm.B + b] = v;
                const int dr = (int)kc->drow[i];
                if (dr >= 0) {
                    const double rr = (prow[dr] - v) * kc->inv_yscale[i];
                    if (prm.loss_kind == 0) { loss_sum += fabs(rr); gsum = fma((signbit(rr) ? 1.0 : -1.0) * kc->inv_yscale[i], dv, gsum); }
                    else { loss_sum = fma(rr, rr, loss_sum); gsum = fma(-2.0 * rr * kc->inv_yscale[i], dv, gsum); }
                }
            }
        };
        if (start_saved) { save_point(u, s, 0); jsave = 1; }

        while (rc < 0) {
            ++iter;
            bool last = false;
            if (jsave >= nsave) { rc = 0; break; }
            if (iter > prm.maxiters) { rc = 1; break; }
            if (t + dt * (1.0 + 1e-13) >= tend) { dt = tend - t; last = true; }
            if (!(dt > kc->dtmin) || t + dt == t) { rc = 2; break; }
            const double gam = d_ * dt;
            const double tnew = last ? tend : t + dt;
            double T, P, Td, Pd;
            tab(t, T, P, Td, Pd);
            double A[NS][NS], dinv[NS], ft[NS];
            int piv[NS];
            bool anyp;
            hy_jac_ft<NS, NR, BLOCK>(th, kc, p0, gam, Pd * frcp(P) - Td * frcp(T), -hp.inv_R * Td * frcp(T * T), Td * frcp(T), A, ft);
            const bool okf = lu_factor_to_lds<NS, LUS>(A, As, dinv, piv, anyp);
            const bool wp = __builtin_amdgcn_ballot_w64(anyp) != 0;
            double k1[NS], dk[NS], k3[NS], u1[NS], unew[NS], f1[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) k1[i] = fma(gam, ft[i], f0[i]);
            lu_solve_lds<NS, LUS>(As, dinv, piv, wp, k1);
#pragma unroll
            for (int i = 0; i < NS; ++i) u1[i] = fma(0.5 * dt, k1[i], u[i]);
            HyPoint<NS, NR> p1, p2;
            double T1, P1, T2, P2, a_, b_;
            tab(t + 0.5 * dt, T1, P1, a_, b_);
            hy_point<NS, NR>(th, kc, hp.inv_R, u1, T1, P1, p1);
#pragma unroll
            for (int i = 0; i < NS; ++i) { f1[i] = p1.f[i]; dk[i] = f1[i] - k1[i]; }
            lu_solve_lds<NS, LUS>(As, dinv, piv, wp, dk);
#pragma unroll
            for (int i = 0; i < NS; ++i) unew[i] = fma(dt, k1[i] + dk[i], u[i]);
            tab(tnew, T2, P2, a_, b_);
            hy_point<NS, NR>(th, kc, hp.inv_R, unew, T2, P2, p2);
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const double k2i = k1[i] + dk[i];
                k3[i] = fma(dt, ft[i], p2.f[i] - c32 * (k2i - f1[i]) - 2.0 * (k1[i] - f0[i]));
            }
            lu_solve_lds<NS, LUS>(As, dinv, piv, wp, k3);
            // ---- this lane's column through the attempt
            double k1p[NS], k2p[NS], snew[NS], f2p[NS];
            {
                double mx[NS], s1[NS], f1p[NS], dkp[NS], k3p[NS];
                const HyTanConst tk{kc->lb, kc->ub, hp.inv_R, kc->Ru, kc->imw, kc->gsc};
                HyTanPt<NS, NR> pt;
                HyTanCol<NS, NR> cl;
                // a HyTanPt from the primal evaluation the attempt already holds (no logarithm or exponential is taken twice)
                auto from_point = [&](const HyPoint<NS, NR> &pp, HyTanPt<NS, NR> &pq) {
#pragma unroll
                    for (int i = 0; i < NS; ++i) {
                        pq.q[i] = ((pp.cY >> i) & 1u) ? frcp(pp.Y[i]) : 0.0;
                        pq.a[i] = ((pp.cC >> i) & 1u) ? 1.0 : 0.0;
                        pq.f[i] = pp.f[i];
                        pq.K[i] = kc->gsc[i] * pp.irho;
                    }
#pragma unroll
                    for (int m = 0; m < NS + 2; ++m) pq.x[m] = pp.x[m];
#pragma unroll
                    for (int j = 0; j < NR; ++j) { pq.r[j] = pp.r[j]; pq.zt[j] = 0.0; }
                    pq.iS = pp.iS;
                    pq.ld = pq.e1 = pq.e2 = 0.0;
                };
                from_point(p0, pt);
                hy_tan_time<NS, NR>(th, tk, T, P, Td, Pd, pt);
                hy_tan_col<NS, NR>(th, dthc, pt, tk, s, cl);
                auto jvp_c = [&](const HyPoint<NS, NR> &pp, const double (&ss)[NS], double (&fp)[NS]) {
                    HyTanPt<NS, NR> pq;
                    HyTanCol<NS, NR> cq;
                    from_point(pp, pq);
                    hy_tan_col<NS, NR>(th, dthc, pq, tk, ss, cq);       // its time part is dead code here
#pragma unroll
                    for (int i = 0; i < NS; ++i) fp[i] = cq.fp[i];
                };
                auto mixed_c = [&](const double (&v)[NS], const double tau, double (&out)[NS]) {
                    HyTanV<NS, NR> pv;
                    hy_tan_v<NS, NR>(th, pt, tk, v, pv);
                    hy_tan_mixed<NS, NR>(th, dthc, pt, pv, cl, v, out);
#pragma unroll
                    for (int i = 0; i < NS; ++i) out[i] = fma(tau, cl.ftp[i], out[i]);
                };
                mixed_c(k1, 1.0, mx);                                           // J' k1 + ft'
#pragma unroll
                for (int i = 0; i < NS; ++i) k1p[i] = fma(gam, mx[i], f0p[i]);
                lu_solve_lds<NS, LUS>(As, dinv, piv, wp, k1p);
#pragma unroll
                for (int i = 0; i < NS; ++i) s1[i] = fma(0.5 * dt, k1p[i], s[i]);
                jvp_c(p1, s1, f1p);
                mixed_c(dk, 0.0, mx);                                           // J' (k2 - k1)
#pragma unroll
                for (int i = 0; i < NS; ++i) dkp[i] = fma(gam, mx[i], f1p[i] - k1p[i]);
                lu_solve_lds<NS, LUS>(As, dinv, piv, wp, dkp);
#pragma unroll
                for (int i = 0; i < NS; ++i) { k2p[i] = k1p[i] + dkp[i]; snew[i] = fma(dt, k2p[i], s[i]); }
                jvp_c(p2, snew, f2p);
                mixed_c(k3, 1.0 / d_, mx);                                      // J' k3 + (dt / gam) ft'
#pragma unroll
                for (int i = 0; i < NS; ++i) k3p[i] = fma(gam, mx[i], f2p[i] - c32 * (k2p[i] - f1p[i]) - 2.0 * (k1p[i] - f0p[i]));
                lu_solve_lds<NS, LUS>(As, dinv, piv, wp, k3p);
                // the dual-inclusive norm: value and the group's partials per component
                double es = 0.0;
                bool fin = okf;
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const double k2i = k1[i] + dk[i];
                    const double ev = dt * (1.0 / 6.0) * (k1[i] - 2.0 * k2i + k3[i]);
                    const double de = dt * (1.0 / 6.0) * (k1p[i] - 2.0 * k2p[i] + k3p[i]);
                    const double na = fma(u[i], u[i], group_sum(s[i] * s[i]));
                    const double nb = fma(unew[i], unew[i], group_sum(snew[i] * snew[i]));
                    const double ee = fma(ev, ev, group_sum(de * de));
                    const double scl = fma(kc->rtol[i], sqrt(fmax(na, nb)), kc->atol[i]);
                    es += ee / (scl * scl);
                    fin = fin && isfinite(unew[i]) && isfinite(ev);
                }
                es *= inv_div;
                if (!(fin && isfinite(es))) { rc = 3; break; }
                const bool ee_zero = (es == 0.0);
                const double lEE = 0.5 * flog(ee_zero ? 1.0 : es);
                const double lq11 = kc->beta1 * lEE;
                double q = ee_zero ? 1.0 / kc->qmax : fmax(1.0 / kc->qmax, fmin(1.0 / kc->qmin, exp(lq11 - kc->beta2 * lqold) / kc->gamma));
                if (es <= 1.0) {
                    ++nacc;
                    while (jsave < nsave) {
                        const double ts = ts_lds[jsave];
                        if (!(ts <= tnew)) break;
                        const bool at_end = (ts == tnew);
                        const double Th = at_end ? 1.0 : (ts - t) / dt;
                        const double c1 = at_end ? 0.0 : Th * (1.0 - Th) * inv12d;
                        const double c2 = at_end ? 1.0 : Th * (Th - 2.0 * d_) * inv12d;
                        double v[NS], vp[NS];
#pragma unroll
                        for (int i = 0; i < NS; ++i) {
                            const double k2i = k1[i] + dk[i];
                            v[i] = at_end ? unew[i] : fma(dt, fma(c1, k1[i], c2 * k2i), u[i]);
                            vp[i] = at_end ? snew[i] : fma(dt, fma(c1, k1p[i], c2 * k2p[i]), s[i]);
                        }
                        save_point(v, vp, jsave);
                        ++jsave;
                    }
#pragma unroll
                    for (int i = 0; i < NS; ++i) { u[i] = unew[i]; s[i] = snew[i]; f0[i] = p2.f[i]; f0p[i] = f2p[i]; }
                    p0 = p2;
                    t = tnew;
                    if (q >= kc->qsteady_min && q <= kc->qsteady_max) q = 1.0;
                    lqold = ee_zero ? lqinit : fmax(lEE, lqinit);
                    dt = fmin(dt / q, dtmax);
                    if (jsave >= nsave) rc = 0;
                } else {
                    ++nrej;
                    dt = dt / fmin(1.0 / kc->qmin, exp(lq11) / kc->gamma);
                }
            }
        }
        {
            const double denom = (double)prm.n_obs * (double)jsave;
            const double inv = jsave > 0 ? 1.0 / denom : 0.0;
            if (nch == 1) prm.gtraj[(size_t)traj * C + col] = gsum * inv;          // d loss_b / d p_k of this chunk's k = column
            else if (col < ndir) prm.gtraj[(size_t)traj * sp.n_total + cid * C + col] = gsum * inv;
            if (col == 0 && nch == 1) {
                prm.loss[b] = loss_sum * inv;
                prm.retcode[b] = rc;
                prm.n_saved[b] = jsave;
                prm.n_accept[b] = nacc;
                prm.n_reject[b] = nrej;
            }
        }
        traj += groups_total;
    }
}

}  // namespace crnn

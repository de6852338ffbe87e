// tools/ubench/hy_tan_probe.hip -- compile-time probe (tools/kres_one.sh tools/ubench/hy_tan_probe.hip [-DCOLS=4]): registers / scratch of
// one Rosenbrock23 step's tangent arithmetic in the column-loop shape (hychem_tan.hpp): one point, three primal directions, COLS columns
// per lane with theta and the columns' d theta in LDS.  Against it, -DNESTED: the same work through hy_f over nested duals, one column.
#include <hip/hip_runtime.h>
#ifdef NESTED
#include "hychem_sens_kernel.hpp"
#else
#include "hychem_tan.hpp"
#endif
#ifndef COLS
#define COLS 4
#endif
namespace crnn {
#ifndef NESTED
__global__ __launch_bounds__(256) void hy_tan_probe(const double *theta, const double *dtheta, const double *cst, const double *in, double *out) {
    constexpr int NS = 9, NR = 10, NTH = HyTanLay<NS, NR>::NTH;
    __shared__ double th[NTH], dth[COLS * 3 * NTH], imw[NS], gsc[NS];
    for (int i = threadIdx.x; i < NTH; i += 256) th[i] = theta[i];
    for (int i = threadIdx.x; i < COLS * 3 * NTH; i += 256) dth[i] = dtheta[i];
    if (threadIdx.x < NS) { imw[threadIdx.x] = cst[4 + threadIdx.x]; gsc[threadIdx.x] = cst[16 + threadIdx.x]; }
    __syncthreads();
    const HyTanConst k{cst[0], cst[1], cst[2], cst[3], imw, gsc};
    const int t = blockIdx.x * 256 + threadIdx.x;
    const double *p = in + (size_t)t * 64;
    double u[NS], v[3][NS];
    for (int i = 0; i < NS; ++i) { u[i] = p[i]; v[0][i] = p[9 + i]; v[1][i] = p[18 + i]; v[2][i] = p[27 + i]; }
    HyTanPt<NS, NR> pt;
    HyTanV<NS, NR> pv[3];
    hy_tan_point<NS, NR>(th, k, u, p[60], p[61], p[62], p[63], pt);
#pragma unroll
    for (int q = 0; q < 3; ++q) hy_tan_v<NS, NR>(th, pt, k, v[q], pv[q]);
    double acc = 0.0;
    const double *dc = dth + (threadIdx.x % 3) * COLS * NTH;
#pragma unroll 1
    for (int c = 0; c < COLS; ++c) {
        double s[NS], mx[NS];
        for (int i = 0; i < NS; ++i) s[i] = p[36 + i] * (c + 1);
        HyTanCol<NS, NR> col;
        hy_tan_col<NS, NR>(th, dc + c * NTH, pt, k, s, col);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            hy_tan_mixed<NS, NR>(th, dc + c * NTH, pt, pv[q], col, v[q], mx);
            for (int i = 0; i < NS; ++i) acc += mx[i] + col.ftp[i];
        }
        for (int i = 0; i < NS; ++i) acc += col.fp[i];
    }
    out[t] = acc;
}
#else
__global__ __launch_bounds__(256) void hy_tan_probe(const double *theta, const double *dtheta, const double *cst, const double *in, double *out) {
    constexpr int NS = 9, NR = 10, NTH = LayH<NS, NR>::NTH;
    __shared__ double th[NTH], dth[12 * NTH], kc_lds[kNConst];
    for (int i = threadIdx.x; i < NTH; i += 256) th[i] = theta[i];
    for (int i = threadIdx.x; i < 12 * NTH; i += 256) dth[i] = dtheta[i];
    for (int i = threadIdx.x; i < kNConst; i += 256) kc_lds[i] = cst[i];
    __syncthreads();
    const KConst *kc = reinterpret_cast<const KConst *>(kc_lds);
    const int t = blockIdx.x * 256 + threadIdx.x;
    const double *p = in + (size_t)t * 64;
    const double *dthc = dth + (threadIdx.x % 12) * NTH;
    typedef Du<double> D1;
    typedef Du<Du<double>> D2;
    auto th1 = [&](const int m) -> D1 { return D1(th[m], dthc[m]); };
    auto th2 = [&](const int m) -> D2 { return D2(D1(th[m], dthc[m]), D1(0.0, 0.0)); };
    double acc = 0.0;
    {
        D1 ud[NS], fd[NS];
        for (int i = 0; i < NS; ++i) ud[i] = D1(p[i], p[36 + i]);
        hy_f<NS, NR, D1>(th1, kc, kc->inv_R, ud, D1(p[60]), D1(p[61]), fd);
        for (int i = 0; i < NS; ++i) acc += fd[i].d;
    }
    for (int q = 0; q < 3; ++q) {
        D2 ud[NS], fd[NS];
        for (int i = 0; i < NS; ++i) ud[i] = D2(D1(p[i], p[36 + i]), D1(p[9 + 9 * q + i], 0.0));
        hy_f<NS, NR, D2>(th2, kc, kc->inv_R, ud, D2(D1(p[60], 0.0), D1(p[62], 0.0)), D2(D1(p[61], 0.0), D1(p[63], 0.0)), fd);
        for (int i = 0; i < NS; ++i) acc += fd[i].d.d;
    }
    out[t] = acc;
}
#endif
}  // namespace crnn

// crnn_amd/csrc/svgd_kernel.hpp -- gfx950: the Stein variational gradient descent move of the Bayesian cathode ensemble
// (Cathode_NCM333_UQ/src_333/network.jl:67-87 svgd_kernel, crnn_cathode.jl:36-50):
//     d_ij = |p_i - p_j|,  h = sqrt(0.5 median(d_ij, i > j)^2 / log(N + 1))   (median trick, h < 0 on input)
//     K = exp(-d^2 / (2 h^2)),  data = K lnpgrad,  repulsion = (-K p + p .* rowsum(K)) / h^2
//     p <- p + stepsize (data + repulsion) / N
// The step that follows the hot path in BASELINE config 5.  N x N x dim is small (4096^2 x 17): the
// kernels are sized for latency, not for a roofline: (1) exact median by radix select on the bit patterns of the
// non-negative distances -- pair distances are recomputed in every pass (17 FMAs) instead of being stored, histograms
// are privatised in LDS; (2) one fused pass forming rowsum(K), K p and K lnpgrad per row over column chunks, with a
// fixed-order second pass over the chunk partials (bitwise reproducible).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace crnn {

constexpr int kSvgdMaxDim = 32;
constexpr int kSvgdBins = 8192;   // 13-bit digits (the first pass takes the top 12 bits: sign + exponent)

__device__ __forceinline__ double svgd_dist(const double *__restrict__ p, int dim, int64_t i, int64_t j) {
    double s = 0.0;
    for (int k = 0; k < dim; ++k) {
        const double d = p[i * dim + k] - p[j * dim + k];
        s = fma(d, d, s);
    }
    return sqrt(s);
}

// One radix-select pass over the strict lower triangle: histogram of digit (key >> shift) & (nbins-1) among the pairs
// whose higher bits equal `prefix` (prefix_shift = shift + digit bits; 64 -> no prefix yet).
__global__ __launch_bounds__(256) void svgd_hist_kernel(const double *__restrict__ p, int64_t N, int dim, int shift, int nbits,
                                                        int prefix_shift, unsigned long long prefix,
                                                        unsigned int *__restrict__ hist) {
    __shared__ unsigned int h[kSvgdBins];
    for (int b = threadIdx.x; b < kSvgdBins; b += 256) h[b] = 0;
    __syncthreads();
    const unsigned long long mask = (1ULL << nbits) - 1ULL;
    const int64_t npairs = N * (N - 1) / 2;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < npairs; q += (int64_t)gridDim.x * 256) {
        // pair index q -> (i, j), i > j:  i = floor((1 + sqrt(1 + 8 q)) / 2), corrected for rounding
        int64_t i = (int64_t)((1.0 + sqrt(1.0 + 8.0 * (double)q)) * 0.5);
        while (i * (i - 1) / 2 > q) --i;
        while ((i + 1) * i / 2 <= q) ++i;
        const int64_t j = q - i * (i - 1) / 2;
        const unsigned long long key = (unsigned long long)__double_as_longlong(svgd_dist(p, dim, i, j));
        if (prefix_shift >= 64 || (key >> prefix_shift) == prefix) atomicAdd(&h[(key >> shift) & mask], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < kSvgdBins; b += 256)
        if (h[b]) atomicAdd(&hist[b], h[b]);
}

// partial[(row * nchunk + c) * (1 + 2 dim)] = [ sum_j K_ij | sum_j K_ij p_j | sum_j K_ij g_j ] over column chunk c
__global__ __launch_bounds__(256) void svgd_rows_kernel(const double *__restrict__ p, const double *__restrict__ g, int64_t N,
                                                        int dim, double inv2h2, int nchunk, double *__restrict__ partial) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y;
    const int64_t j0 = (N * c) / nchunk, j1 = (N * (c + 1)) / nchunk;
    __shared__ double pj[64 * kSvgdMaxDim], gj[64 * kSvgdMaxDim];
    double pi[kSvgdMaxDim], ap[kSvgdMaxDim], ag[kSvgdMaxDim], ak = 0.0;
    for (int k = 0; k < kSvgdMaxDim; ++k) { pi[k] = 0.0; ap[k] = 0.0; ag[k] = 0.0; }
    if (row < N) for (int k = 0; k < dim; ++k) pi[k] = p[row * dim + k];
    for (int64_t t0 = j0; t0 < j1; t0 += 64) {
        const int nt = (int)((j1 - t0) < 64 ? (j1 - t0) : 64);
        __syncthreads();
        for (int idx = threadIdx.x; idx < nt * dim; idx += 256) { pj[idx] = p[t0 * dim + idx]; gj[idx] = g[t0 * dim + idx]; }
        __syncthreads();
        if (row < N) {
            for (int jj = 0; jj < nt; ++jj) {
                double s = 0.0;
                for (int k = 0; k < dim; ++k) { const double d = pi[k] - pj[jj * dim + k]; s = fma(d, d, s); }
                // the reference squares the rounded distance (pairwise_dists = distances.^2)
                const double dd = sqrt(s);
                const double kij = exp(-(dd * dd) * inv2h2);
                ak += kij;
                for (int k = 0; k < dim; ++k) { ap[k] = fma(kij, pj[jj * dim + k], ap[k]); ag[k] = fma(kij, gj[jj * dim + k], ag[k]); }
            }
        }
    }
    if (row < N) {
        double *o = partial + ((size_t)row * nchunk + c) * (1 + 2 * dim);
        o[0] = ak;
        for (int k = 0; k < dim; ++k) { o[1 + k] = ap[k]; o[1 + dim + k] = ag[k]; }
    }
}

// fixed-order combination of the chunk partials and the move itself
__global__ __launch_bounds__(256) void svgd_update_kernel(const double *__restrict__ p, const double *__restrict__ partial,
                                                          int64_t N, int dim, int nchunk, double inv_h2, double step_over_n,
                                                          double *__restrict__ p_new, double *__restrict__ data_term,
                                                          double *__restrict__ repulsion) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= N * dim) return;
    const int64_t row = idx / dim;
    const int k = (int)(idx - row * dim);
    double sk = 0.0, sp = 0.0, sg = 0.0;
    for (int c = 0; c < nchunk; ++c) {
        const double *o = partial + ((size_t)row * nchunk + c) * (1 + 2 * dim);
        sk += o[0]; sp += o[1 + k]; sg += o[1 + dim + k];
    }
    const double rep = (p[idx] * sk - sp) * inv_h2;
    if (data_term) data_term[idx] = sg;
    if (repulsion) repulsion[idx] = rep;
    p_new[idx] = p[idx] + step_over_n * (sg + rep);
}

}  // namespace crnn

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

// ---- device-resident selection (round 3): the radix select keeps its state in device memory, so the ten passes of the
// two order statistics, the bandwidth and the move are enqueued back to back without a host round trip.
struct SvgdSel {
    unsigned long long prefix;   // bits of the order statistic found so far
    long long rank;              // its rank among the pairs that share the prefix
    int prefix_shift;            // 64: no prefix yet
    int bad;                     // sticky: a move found h not finite and positive and left the particles where they were
    double median[2];            // the two middle order statistics (as values)
    double h;                    // bandwidth
    double inv2h2, inv_h2;       // 1/(2 h^2), 1/h^2
};

// start of a select: rank = `rank`, no prefix
__global__ void svgd_sel_init_kernel(SvgdSel *st, long long rank) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { st->prefix = 0ULL; st->rank = rank; st->prefix_shift = 64; }
}

// the hist pass with prefix / prefix_shift taken from the device state
__global__ __launch_bounds__(256) void svgd_hist_dev_kernel(const double *__restrict__ p, int64_t N, int dim, int shift, int nbits,
                                                            const SvgdSel *__restrict__ st, unsigned int *__restrict__ hist) {
    __shared__ unsigned int h[kSvgdBins];
    for (int b = threadIdx.x; b < kSvgdBins; b += 256) h[b] = 0;
    __syncthreads();
    const unsigned long long prefix = st->prefix;
    const int prefix_shift = st->prefix_shift;
    const unsigned long long mask = (1ULL << nbits) - 1ULL;
    const int64_t npairs = N * (N - 1) / 2;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < npairs; q += (int64_t)gridDim.x * 256) {
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

// one block: find the bin that holds the rank, extend the prefix, zero the histogram for the next pass; after the last
// pass (final_slot >= 0) the prefix IS the order statistic: store it
__global__ __launch_bounds__(1024) void svgd_pick_kernel(unsigned int *__restrict__ hist, int nbits, int shift, SvgdSel *st, int final_slot) {
    __shared__ unsigned long long part[1024];
    __shared__ int found_bin;
    __shared__ unsigned long long found_before;
    const int nb = 1 << nbits, tid = threadIdx.x;
    const int per = kSvgdBins / 1024;     // 8 consecutive bins per thread
    unsigned int loc[kSvgdBins / 1024];
    unsigned long long s = 0;
    for (int k = 0; k < per; ++k) { const int b = tid * per + k; loc[k] = b < nb ? hist[b] : 0u; s += loc[k]; }
    part[tid] = s;
    if (tid == 0) { found_bin = -1; found_before = 0; }
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {     // inclusive scan
        const unsigned long long v = tid >= off ? part[tid - off] : 0ULL;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    const unsigned long long rank = (unsigned long long)st->rank;
    const unsigned long long before = part[tid] - s;      // pairs in the bins of lower threads
    if (rank >= before && rank < part[tid]) {              // the rank falls into this thread's bins (exactly one thread)
        unsigned long long cum = before;
        for (int k = 0; k < per; ++k) {
            if (rank < cum + loc[k]) { found_bin = tid * per + k; found_before = cum; break; }
            cum += loc[k];
        }
    }
    __syncthreads();
    for (int k = 0; k < per; ++k) hist[tid * per + k] = 0u;
    if (tid == 0) {
        const int b = found_bin < 0 ? 0 : found_bin;       // (an empty ensemble cannot get here)
        st->prefix = (st->prefix << nbits) | (unsigned long long)b;
        st->rank = (long long)(rank - found_before);
        st->prefix_shift = shift;
        if (final_slot >= 0) st->median[final_slot] = __longlong_as_double((long long)st->prefix);
    }
}

// ---- the pair distances, formed ONCE (round 3): dist[i (i-1)/2 + j] = |p_i - p_j| for i > j.  Recomputing them in each
// of the ten histogram passes cost 0.5 ms per pass (strided row loads, a sqrt-and-correct index inversion per pair); stored,
// a pass streams 8 bytes per pair.  One block per 64 x 64 tile of the lower triangle, rows staged in LDS.
template <int DIM>
__global__ __launch_bounds__(256) void svgd_pairdist_kernel(const double *__restrict__ p, int64_t N, int dim_rt, double *__restrict__ dist) {
    const int dim = DIM > 0 ? DIM : dim_rt;
    constexpr int LD = DIM > 0 ? DIM : kSvgdMaxDim;
    __shared__ double pi[64 * LD], pj[64 * LD];
    // block -> tile (ti, tj), tj <= ti: blockIdx.x = ti (ti + 1) / 2 + tj
    int ti = (int)((sqrt(8.0 * (double)blockIdx.x + 1.0) - 1.0) * 0.5);
    while ((int64_t)ti * (ti + 1) / 2 > (int64_t)blockIdx.x) --ti;
    while ((int64_t)(ti + 1) * (ti + 2) / 2 <= (int64_t)blockIdx.x) ++ti;
    const int tj = (int)((int64_t)blockIdx.x - (int64_t)ti * (ti + 1) / 2);
    const int64_t i0 = (int64_t)ti * 64, j0 = (int64_t)tj * 64;
    for (int idx = threadIdx.x; idx < 64 * dim; idx += 256) {
        const int r = idx / dim, k = idx - r * dim;
        pi[r * LD + k] = (i0 + r < N) ? p[(i0 + r) * dim + k] : 0.0;
        pj[r * LD + k] = (j0 + r < N) ? p[(j0 + r) * dim + k] : 0.0;
    }
    __syncthreads();
    const int jj = threadIdx.x & 63;              // consecutive threads -> consecutive j: coalesced stores
    for (int ii = threadIdx.x >> 6; ii < 64; ii += 4) {
        const int64_t i = i0 + ii, j = j0 + jj;
        if (i < N && j < i) {
            double s = 0.0;
            if constexpr (DIM > 0) {
#pragma unroll
                for (int k = 0; k < DIM; ++k) { const double d = pi[ii * LD + k] - pj[jj * LD + k]; s = fma(d, d, s); }
            } else {
                for (int k = 0; k < dim; ++k) { const double d = pi[ii * LD + k] - pj[jj * LD + k]; s = fma(d, d, s); }
            }
            dist[i * (i - 1) / 2 + j] = sqrt(s);
        }
    }
}

// one radix-select pass over the stored distances.  Nearly all keys share their leading digits (the first pass sees two or
// three distinct exponents): a thread counts runs of its current digit in a register and touches the LDS histogram only
// when the digit changes.
__global__ __launch_bounds__(256) void svgd_hist_buf_kernel(const double *__restrict__ dist, int64_t npairs, int shift, int nbits,
                                                            const SvgdSel *__restrict__ st, unsigned int *__restrict__ hist) {
    __shared__ unsigned int h[kSvgdBins];
    for (int b = threadIdx.x; b < kSvgdBins; b += 256) h[b] = 0;
    __syncthreads();
    const unsigned long long prefix = st->prefix;
    const int prefix_shift = st->prefix_shift;
    const unsigned long long mask = (1ULL << nbits) - 1ULL;
    unsigned int run_bin = 0xffffffffu, run = 0;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < npairs; q += (int64_t)gridDim.x * 256) {
        const unsigned long long key = (unsigned long long)__double_as_longlong(dist[q]);
        if (prefix_shift >= 64 || (key >> prefix_shift) == prefix) {
            const unsigned int b = (unsigned int)((key >> shift) & mask);
            if (b == run_bin) ++run;
            else {
                if (run) atomicAdd(&h[run_bin], run);
                run_bin = b; run = 1;
            }
        }
    }
    if (run) atomicAdd(&h[run_bin], run);
    __syncthreads();
    for (int b = threadIdx.x; b < kSvgdBins; b += 256)
        if (h[b]) atomicAdd(&hist[b], h[b]);
}

// With the lower middle order statistic known (median[0], rank r), the upper one (rank r + 1) is the same value if enough
// keys equal it, else the smallest key above it: one pass counting keys < and == median[0] and taking the minimum above.
// cnt = { #less, #equal, min key above (bit pattern; 0xff..ff: none) }, zeroed / set by svgd_next_init_kernel.
__global__ void svgd_next_init_kernel(unsigned long long *cnt) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { cnt[0] = 0ULL; cnt[1] = 0ULL; cnt[2] = ~0ULL; }
}
__global__ __launch_bounds__(256) void svgd_next_kernel(const double *__restrict__ dist, int64_t npairs, const SvgdSel *__restrict__ st,
                                                        unsigned long long *cnt) {
    const unsigned long long lo = (unsigned long long)__double_as_longlong(st->median[0]);
    unsigned long long nl = 0, ne = 0, mg = ~0ULL;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < npairs; q += (int64_t)gridDim.x * 256) {
        const unsigned long long key = (unsigned long long)__double_as_longlong(dist[q]);
        nl += key < lo; ne += key == lo;
        if (key > lo && key < mg) mg = key;
    }
    __shared__ unsigned long long sl[256], se[256], sm[256];
    sl[threadIdx.x] = nl; se[threadIdx.x] = ne; sm[threadIdx.x] = mg;
    __syncthreads();
    for (int s_ = 128; s_ > 0; s_ >>= 1) {
        if ((int)threadIdx.x < s_) {
            sl[threadIdx.x] += sl[threadIdx.x + s_]; se[threadIdx.x] += se[threadIdx.x + s_];
            if (sm[threadIdx.x + s_] < sm[threadIdx.x]) sm[threadIdx.x] = sm[threadIdx.x + s_];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { atomicAdd(&cnt[0], sl[0]); atomicAdd(&cnt[1], se[0]); atomicMin(&cnt[2], sm[0]); }
}
__global__ void svgd_next_pick_kernel(SvgdSel *st, const unsigned long long *cnt, long long rank_hi) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const bool same = (unsigned long long)rank_hi < cnt[0] + cnt[1];
        st->median[1] = same ? st->median[0] : __longlong_as_double((long long)cnt[2]);
    }
}

// h = sqrt(0.5 median^2 / log(N + 1)), median = mean of the two middle order statistics (Julia's median); h_given >= 0: as is
__global__ void svgd_bandwidth_kernel(SvgdSel *st, double log_np1, double h_given) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double h = h_given;
        if (!(h_given >= 0.0)) {
            const double med = 0.5 * (st->median[0] + st->median[1]);
            h = sqrt(0.5 * (med * med) / log_np1);
        }
        st->h = h;
        st->inv2h2 = 0.5 / (h * h);
        st->inv_h2 = 1.0 / (h * h);
    }
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

// The same with the dimension known at compile time (DIM = 17: the cathode's particles) -- the accumulators stay in
// registers (the generic kernel indexes 32-wide arrays with a run-time bound: 784 B/lane of scratch) -- and the bandwidth
// taken from the device state.
template <int DIM>
__global__ __launch_bounds__(256) void svgd_rows_dim_kernel(const double *__restrict__ p, const double *__restrict__ g, int64_t N,
                                                            const SvgdSel *__restrict__ st, int nchunk, double *__restrict__ partial) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y;
    const int64_t j0 = (N * c) / nchunk, j1 = (N * (c + 1)) / nchunk;
    const double inv2h2 = st->inv2h2;
    __shared__ double pj[64 * DIM], gj[64 * DIM];
    double pi[DIM], ap[DIM], ag[DIM], ak = 0.0;
#pragma unroll
    for (int k = 0; k < DIM; ++k) { pi[k] = row < N ? p[row * DIM + k] : 0.0; ap[k] = 0.0; ag[k] = 0.0; }
    for (int64_t t0 = j0; t0 < j1; t0 += 64) {
        const int nt = (int)((j1 - t0) < 64 ? (j1 - t0) : 64);
        __syncthreads();
        for (int idx = threadIdx.x; idx < nt * DIM; idx += 256) { pj[idx] = p[t0 * DIM + idx]; gj[idx] = g[t0 * DIM + idx]; }
        __syncthreads();
        if (row < N) {
            for (int jj = 0; jj < nt; ++jj) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < DIM; ++k) { const double d = pi[k] - pj[jj * DIM + k]; s = fma(d, d, s); }
                const double dd = sqrt(s);            // the reference squares the rounded distance
                const double kij = exp(-(dd * dd) * inv2h2);
                ak += kij;
#pragma unroll
                for (int k = 0; k < DIM; ++k) { ap[k] = fma(kij, pj[jj * DIM + k], ap[k]); ag[k] = fma(kij, gj[jj * DIM + k], ag[k]); }
            }
        }
    }
    if (row < N) {
        double *o = partial + ((size_t)row * nchunk + c) * (1 + 2 * DIM);
        o[0] = ak;
#pragma unroll
        for (int k = 0; k < DIM; ++k) { o[1 + k] = ap[k]; o[1 + DIM + k] = ag[k]; }
    }
}

// generic dimension, bandwidth from the device state
__global__ __launch_bounds__(256) void svgd_rows_dev_kernel(const double *__restrict__ p, const double *__restrict__ g, int64_t N,
                                                            int dim, const SvgdSel *__restrict__ st, int nchunk, double *__restrict__ partial);

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

__global__ __launch_bounds__(256) void svgd_rows_dev_kernel(const double *__restrict__ p, const double *__restrict__ g, int64_t N,
                                                            int dim, const SvgdSel *__restrict__ st, int nchunk, double *__restrict__ partial) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y;
    const int64_t j0 = (N * c) / nchunk, j1 = (N * (c + 1)) / nchunk;
    const double inv2h2 = st->inv2h2;
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

// the move with the bandwidth from the device state; p_new may alias nothing (p is read by every thread's row sums only
// through `partial`, so p_new == p would also be safe: each thread reads and writes its own element)
__global__ __launch_bounds__(256) void svgd_update_dev_kernel(const double *__restrict__ p, const double *__restrict__ partial,
                                                              int64_t N, int dim, int nchunk, SvgdSel *st,          // (not const: the sticky `bad` flag is written here)
                                                              double step_over_n, double *p_new, double *__restrict__ data_term,
                                                              double *__restrict__ repulsion) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= N * dim) return;
    const double inv_h2 = st->inv_h2;
    const int64_t row = idx / dim;
    const int k = (int)(idx - row * dim);
    double sk = 0.0, sp = 0.0, sg = 0.0;
    for (int c = 0; c < nchunk; ++c) {
        const double *o = partial + ((size_t)row * nchunk + c) * (1 + 2 * dim);
        sk += o[0]; sp += o[1 + k]; sg += o[1 + dim + k];
    }
    const double pv = p[idx];
    const double rep = (pv * sk - sp) * inv_h2;
    if (data_term) data_term[idx] = sg;
    if (repulsion) repulsion[idx] = rep;
    // a bandwidth that is not finite and positive (coincident particles, NaN positions) would turn every particle into NaN/Inf:
    // the particles stay where they are and the state says so (the host checks `bad` where it looks at the loop)
    const double hh = st->h;
    const bool ok = hh > 0.0 && hh < INFINITY;
    if (!ok && idx == 0) st->bad = 1;
    p_new[idx] = ok ? pv + step_over_n * (sg + rep) : pv;
}

}  // namespace crnn

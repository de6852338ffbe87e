// micro-benchmarks: dependent vs independent FP64 FMA issue, LDS read latency, rcp, one wave per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
__global__ void k_dep(double* out, double a, double b) {
    double x = out[threadIdx.x];
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < N / 16; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) x = fma(x, a, b);
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x; if (threadIdx.x == 0) out[1024] = (double)(t1 - t0) / N;
}
template <int W>
__global__ void k_ind(double* out, double a, double b) {
    double x[W];
    for (int w = 0; w < W; ++w) x[w] = out[threadIdx.x] + w;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < N / 16; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j)
#pragma unroll
            for (int w = 0; w < W; ++w) x[w] = fma(x[w], a, b);
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0; for (int w = 0; w < W; ++w) s += x[w];
    out[threadIdx.x] = s; if (threadIdx.x == 0) out[1024] = (double)(t1 - t0) / (N * W);
}
__global__ void k_lds(double* out) {
    __shared__ double sh[1024];
    sh[threadIdx.x] = (double)((threadIdx.x * 7 + 3) & 63);
    __syncthreads();
    int idx = threadIdx.x & 63;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < N; ++i) idx = (int)sh[idx];
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = idx; if (threadIdx.x == 0) out[1024] = (double)(t1 - t0) / N;
}
__global__ void k_rcp(double* out) {
    double x = out[threadIdx.x] + 1.5;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < N / 8; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) x = __builtin_amdgcn_rcp(x) + 1.0;
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x; if (threadIdx.x == 0) out[1024] = (double)(t1 - t0) / N;
}
__global__ void k_f32dep(float* out, float a, float b) {
    float x = out[threadIdx.x];
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < N / 16; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) x = fmaf(x, a, b);
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x; if (threadIdx.x == 0) out[2048] = (float)(t1 - t0) / N;
}
int main() {
    double* d; hipMalloc(&d, 4096 * 8); hipMemset(d, 0, 4096 * 8);
    double r;
    auto rd = [&](const char* n) { hipDeviceSynchronize(); hipMemcpy(&r, d + 1024, 8, hipMemcpyDeviceToHost); printf("%-28s %.2f cycles (s_memtime ticks)/op\n", n, r); };
    for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_dep, 1, 64, 0, 0, d, 0.999, 0.001); rd("fma_f64 dependent chain");
    hipLaunchKernelGGL(k_ind<2>, 1, 64, 0, 0, d, 0.999, 0.001); rd("fma_f64 2 independent");
    hipLaunchKernelGGL(k_ind<4>, 1, 64, 0, 0, d, 0.999, 0.001); rd("fma_f64 4 independent");
    hipLaunchKernelGGL(k_ind<8>, 1, 64, 0, 0, d, 0.999, 0.001); rd("fma_f64 8 independent");
    hipLaunchKernelGGL(k_lds, 1, 64, 0, 0, d); rd("ds_read_b64 dependent");
    hipLaunchKernelGGL(k_rcp, 1, 64, 0, 0, d); rd("rcp_f64+add dependent");
    }
    // 4 waves on one CU (one per SIMD) vs 8 waves
    hipLaunchKernelGGL(k_ind<4>, 1, 256, 0, 0, d, 0.999, 0.001); rd("fma 4 indep, 4 waves/CU");
    hipLaunchKernelGGL(k_ind<4>, 1, 512, 0, 0, d, 0.999, 0.001); rd("fma 4 indep, 8 waves/CU");
    hipLaunchKernelGGL(k_dep, 1, 512, 0, 0, d, 0.999, 0.001); rd("fma dep, 8 waves/CU");
    return 0;
}

// tools/ubench/dp_chain.hip -- issue cost of dependent / independent FP64 FMA chains, DPP moves and LDS broadcast reads for ONE
// wavefront per SIMD on gfx950 (the regime of the HyChem kernels: 512 registers per lane).  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP>
__global__ void fma_chain(double *out, int n, double a, double b, unsigned long long *cyc) {
    double x[ILP];
    for (int i = 0; i < ILP; ++i) x[i] = threadIdx.x + i;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int k = 0; k < n; ++k) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < ILP; ++i) x[i] = __builtin_fma(x[i], a, b);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < ILP; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
__global__ void dpp_chain(double *out, int n, double a, unsigned long long *cyc) {
    double x = threadIdx.x;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int k = 0; k < n; ++k) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int lo = __double2loint(x), hi = __double2hiint(x);
            lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, false);
            hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, false);
            x = __builtin_fma(__hiloint2double(hi, lo), a, x);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
__global__ void lds_chain(double *out, int n, unsigned long long *cyc) {
    __shared__ double tab[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) tab[i] = 1.0 / (i + 1);
    __syncthreads();
    double x = threadIdx.x;
    int idx = 0;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int k = 0; k < n; ++k) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const double v = tab[(idx + r) & 255];   // broadcast read, dependent use
            x = __builtin_fma(x, v, v);
            asm volatile("" : "+v"(idx));
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    double *out; unsigned long long *cyc, h;
    hipMalloc(&out, 1024 * 64 * 8); hipMalloc(&cyc, 8);
    const int n = 4096;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto report = [&](const char *name, int per_iter, float ms) {
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-28s %8.3f ms  %6.2f ns/op  (counter ticks/op %.3f)\n", name, ms, ms * 1e6 / ((double)n * 16 * per_iter), (double)h / ((double)n * 16 * per_iter));
    };
#define RUN(name, per_iter, ...) do { __VA_ARGS__; hipDeviceSynchronize(); hipEventRecord(e0); __VA_ARGS__; hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); report(name, per_iter, ms); } while (0)
    RUN("fma chain ilp1", 1, hipLaunchKernelGGL(fma_chain<1>, dim3(1024), dim3(64), 0, 0, out, n, 0.999, 1e-3, cyc));
    RUN("fma chain ilp2", 2, hipLaunchKernelGGL(fma_chain<2>, dim3(1024), dim3(64), 0, 0, out, n, 0.999, 1e-3, cyc));
    RUN("fma chain ilp4", 4, hipLaunchKernelGGL(fma_chain<4>, dim3(1024), dim3(64), 0, 0, out, n, 0.999, 1e-3, cyc));
    RUN("fma chain ilp8", 8, hipLaunchKernelGGL(fma_chain<8>, dim3(1024), dim3(64), 0, 0, out, n, 0.999, 1e-3, cyc));
    RUN("fma ilp4, 2 waves/SIMD", 4, hipLaunchKernelGGL(fma_chain<4>, dim3(2048), dim3(64), 0, 0, out, n, 0.999, 1e-3, cyc));
    RUN("fma ilp1, 2 waves/SIMD", 1, hipLaunchKernelGGL(fma_chain<1>, dim3(2048), dim3(64), 0, 0, out, n, 0.999, 1e-3, cyc));
    RUN("dpp swap + fma chain (3 ops)", 1, hipLaunchKernelGGL(dpp_chain, dim3(1024), dim3(64), 0, 0, out, n, 1e-9, cyc));
    RUN("lds bcast read + fma chain", 1, hipLaunchKernelGGL(lds_chain, dim3(1024), dim3(64), 0, 0, out, n, cyc));
    return 0;
}

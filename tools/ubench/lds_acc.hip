// What does a lane-private FP64 accumulator in LDS cost a wavefront that has the SIMD to itself?
// One 256-lane block per CU (one wavefront per SIMD, as in ros23_adj_kernel), NACC accumulators per lane laid out [m][lane].
// Each "step" does FMAS dependent-free FMAs per accumulator update plus the update itself, in three flavours:
//   0  none (arithmetic only)            1  ds_add_f64 (fire and forget)
//   2  ds_read_b64 + fma + ds_write_b64  3  accumulators in registers
// Prints cycles per update (s_memtime of wave 0 / updates).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr int BLOCK = 256, NACC = 36;
template <int MODE, int FMAS>
__global__ __launch_bounds__(BLOCK) void k(double *out, unsigned long long *cyc, int steps, double seed) {
    __shared__ double acc_lds[NACC * BLOCK];
    double *a = acc_lds + threadIdx.x;
    for (int m = 0; m < NACC; ++m) a[m * BLOCK] = 0.0;
    double reg[NACC];
    for (int m = 0; m < NACC; ++m) reg[m] = 0.0;
    double x = seed + threadIdx.x * 1e-9, y = 1.0 + seed;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int m = 0; m < NACC; ++m) {
            double v[FMAS > 0 ? FMAS : 1];
#pragma unroll
            for (int q = 0; q < FMAS; ++q) v[q] = fma(x, y + q, (double)(m + q));   // independent of each other
            double val = x;
#pragma unroll
            for (int q = 0; q < FMAS; ++q) val += v[q];
            if (MODE == 1) unsafeAtomicAdd(&a[m * BLOCK], val);
            else if (MODE == 2) a[m * BLOCK] += val;
            else if (MODE == 3) reg[m] += val;
            else asm volatile("" ::"v"(val));
        }
        x = fma(x, 0.999999, 1e-7);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    double r = 0.0;
    for (int m = 0; m < NACC; ++m) r += a[m * BLOCK] + reg[m];
    out[blockIdx.x * BLOCK + threadIdx.x] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE, int FMAS>
void run(const char *name, double *o, unsigned long long *c) {
    const int steps = 2000;
    hipLaunchKernelGGL((k<MODE, FMAS>), 256, BLOCK, 0, 0, o, c, 10, 0.5);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, FMAS>), 256, BLOCK, 0, 0, o, c, steps, 0.5);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%-28s FMAS %d : %7.1f ns/update (event), %8.1f counter ticks/update\n", name, FMAS, ms * 1e6 / ((double)steps * NACC), (double)h / ((double)steps * NACC));
}
int main() {
    double *o; unsigned long long *c; hipMalloc(&o, 8 * 256 * BLOCK); hipMalloc(&c, 8);
    run<0, 2>("none", o, c); run<1, 2>("ds_add_f64", o, c); run<2, 2>("ds_read+add+ds_write", o, c); run<3, 2>("registers", o, c);
    run<0, 8>("none", o, c); run<1, 8>("ds_add_f64", o, c); run<2, 8>("ds_read+add+ds_write", o, c); run<3, 8>("registers", o, c);
    run<0, 20>("none", o, c); run<1, 20>("ds_add_f64", o, c); run<2, 20>("ds_read+add+ds_write", o, c); run<3, 20>("registers", o, c);
    return 0;
}

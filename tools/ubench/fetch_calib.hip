// FETCH_SIZE / WRITE_SIZE calibration in the solver's own access pattern (MI355X_MICROARCH.md, HBM section:
// "calibrate on a known byte count in your own access pattern before trusting an absolute").
// Pattern: wavefront = 16 groups of 4 lanes; the 4 lanes of a group read the same double; groups read
// consecutive doubles of an IC-fastest row; 300 rows per trajectory, rows B*8 bytes apart; persistent groups.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_read(const double* __restrict__ data, double* __restrict__ out, long B, int rows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 2, chunk = lane & 3;
    const long ngroups = (long)gridDim.x * 4 * 16;
    double acc = 0.0;
    for (long b = ((long)blockIdx.x * 4 + wave) * 16 + grp; b < B; b += ngroups) {
        for (int r = 0; r < rows; ++r) acc += data[(size_t)r * B + b];
        if (chunk == 0) out[b] = acc;   // 8 B written per trajectory
    }
}
// Write pattern of the adjoint's step tape: every lane appends 72-byte records (9 doubles) to its own lane-contiguous tape.
__global__ __launch_bounds__(256) void k_tape(double* __restrict__ tape, int steps, int cap) {
    double* t = tape + (size_t)(blockIdx.x * 256 + threadIdx.x) * cap * 9;
    for (int s = 0; s < steps; ++s)
        for (int i = 0; i < 9; ++i) t[(size_t)s * 9 + i] = (double)(s + i);
}
// ... and the reverse sweep's read of it (records walked backwards)
__global__ __launch_bounds__(256) void k_tape_read(const double* __restrict__ tape, double* __restrict__ out, int steps, int cap) {
    const double* t = tape + (size_t)(blockIdx.x * 256 + threadIdx.x) * cap * 9;
    double acc = 0.0;
    for (int s = steps - 1; s >= 0; --s)
        for (int i = 0; i < 9; ++i) acc += t[(size_t)s * 9 + i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main() {
    const long B = 65536; const int rows = 300;
    double *d, *o; hipMalloc(&d, sizeof(double) * B * rows); hipMalloc(&o, sizeof(double) * B);
    hipMemset(d, 0, sizeof(double) * B * rows);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k_read, 256, 256, 0, 0, d, o, B, rows);
    hipDeviceSynchronize();
    printf("k_read      known bytes per launch: read %ld write %ld\n", B * rows * 8L, B * 8L);
    const int steps = 30, cap = 64;
    double* tp; hipMalloc(&tp, sizeof(double) * B * cap * 9);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k_tape, 256, 256, 0, 0, tp, steps, cap);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k_tape_read, 256, 256, 0, 0, tp, o, steps, cap);
    hipDeviceSynchronize();
    printf("k_tape      known bytes per launch: read 0 write %ld\n", B * steps * 72L);
    printf("k_tape_read known bytes per launch: read %ld write %ld\n", B * steps * 72L, B * 8L);
    return 0;
}

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
int main() {
    const long B = 65536; const int rows = 300;
    double *d, *o; hipMalloc(&d, sizeof(double) * B * rows); hipMalloc(&o, sizeof(double) * B);
    hipMemset(d, 0, sizeof(double) * B * rows);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k_read, 256, 256, 0, 0, d, o, B, rows);
    hipDeviceSynchronize();
    printf("known bytes per launch: read %ld write %ld\n", B * rows * 8L, B * 8L);
    return 0;
}

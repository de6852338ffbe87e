// tools/ubench/mfma_f64_layout.hip -- operand / result layout of v_mfma_f64_16x16x4_f64 on gfx950, found with one-hot inputs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k(const double *A, const double *B, double *D) {   // A: 64 lane values, B: 64 lane values, D: [64][4]
    const int l = threadIdx.x;
    v4d c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(A[l], B[l], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];
}
int main() {
    double hA[64], hB[64], hD[256], *dA, *dB, *dD;
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048);
    // A[i][k] = 100 i + 10 k + 1 assumed at lane i + 16 k; B[k][j] = (k == kk) one-hot over k, all j -> D[i][j] = A[i][kk]
    for (int kk = 0; kk < 4; ++kk) {
        for (int l = 0; l < 64; ++l) { hA[l] = 100 * (l % 16) + 10 * (l / 16) + 1; hB[l] = (l / 16 == kk) ? 1.0 + 0.001 * (l % 16) : 0.0; }
        hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        hipMemcpy(hD, dD, 2048, hipMemcpyDeviceToHost);
        if (kk == 1) {
            printf("kk=1: D(lane, r) = A[i][1] * (1 + 0.001 j) -> i = int(D/100), j from the fraction\n");
            for (int l = 0; l < 64; l += 1) { printf("lane %2d:", l); for (int r = 0; r < 4; ++r) printf(" %10.3f", hD[l * 4 + r]); printf("\n"); }
        }
    }
    return 0;
}

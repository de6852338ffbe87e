// tests/simt/calib.hip -- TEST INFRASTRUCTURE: every cross-lane / matrix / special-function primitive the kernels under crnn_amd/csrc use, applied
// once to caller-supplied values by one wavefront.  The same source is built twice -- hipcc for gfx950, and as host C++ against the SIMT shim
// (tests/simt/hip/hip_runtime.h) -- and tests/test_simt_calibration.py compares the two results bit for bit on an MI355X: after that,
// "green under emulation" means "green on the device as far as these primitives go" (VERDICT r5 item 7).  The call forms are the kernels' own:
//   update_dpp quad_perm 0xB1 / 0xF5 / 0xA0 on the halves of a double   ros23_adj2_kernel.hpp:45-52, hychem2_kernel.hpp:51-59
//   __shfl over groups of 9 and 12 lanes, sources across group boundaries   cathode_sens_auto_kernel.hpp:73, hychem_sens_kernel.hpp:165
//   __shfl_down tree, __shfl_xor butterfly                                  hychem_kernel.hpp:961, ros23_adj_kernel.hpp:533
//   ballot_w64 converged and inside a divergent branch; readfirstlane converged and inside a divergent branch (first ACTIVE lane)
//   v_mfma_f64_16x16x4f64: A and B operands per lane, 4 accumulators per lane   hychem2_kernel.hpp:1107
//   rcp, rsq (one-ulp results: compared to 1 ulp), ldexp, frexp_mant, frexp_exp (exact)   ros23_kernel.hpp
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CALIB_SLOTS 32

__device__ __forceinline__ double dpp_pair(double a, int which) {
    const int lo = __double2loint(a), hi = __double2hiint(a);
    if (which == 0)
        return __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true), __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true));
    if (which == 1)
        return __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0xF5, 0xF, 0xF, false), __builtin_amdgcn_update_dpp(0, lo, 0xF5, 0xF, 0xF, true));
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0xA0, 0xF, 0xF, false), __builtin_amdgcn_update_dpp(0, lo, 0xA0, 0xF, 0xF, true));
}

__device__ __forceinline__ unsigned long long bits(double x) { return (unsigned long long)__double_as_longlong(x); }

// in [4][64] doubles (a, b, c, positive d), idx [64] source lanes; out [64][CALIB_SLOTS] 64-bit words
__global__ __launch_bounds__(64) void calib_kernel(const double *__restrict__ in, const int *__restrict__ idx, unsigned long long *__restrict__ out) {
    const int lane = threadIdx.x;
    const double a = in[lane], b = in[64 + lane], c = in[128 + lane], d = in[192 + lane];
    const int src = idx[lane];
    unsigned long long *o = out + (size_t)lane * CALIB_SLOTS;
    int s = 0;
    o[s++] = bits(dpp_pair(a, 0));
    o[s++] = bits(dpp_pair(a, 1));
    o[s++] = bits(dpp_pair(a, 2));
    o[s++] = (unsigned long long)(unsigned)__builtin_amdgcn_update_dpp(0, src, 0xB1, 0xF, 0xF, true);
    o[s++] = bits(__shfl(a, src));
    {   // the group sums of the nine- and twelve-lane kernels: sources run over the group's lanes, the last group is short / idle lanes exist
        const int g9 = lane / 9 * 9, g12 = lane / 12 * 12;
        double s9 = 0.0, s12 = 0.0;
        for (int q = 0; q < 9; ++q) s9 += __shfl(b, g9 + q);
        for (int q = 0; q < 12; ++q) s12 += __shfl(b, g12 + q);
        o[s++] = bits(s9);
        o[s++] = bits(s12);
    }
    {
        double v = c;
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        o[s++] = bits(v);
        double m = c;
        for (int k = 32; k >= 1; k >>= 1) m = max(m, __shfl_xor(m, k));
        o[s++] = bits(m);
    }
    o[s++] = __builtin_amdgcn_ballot_w64(a > 0.0);
    o[s++] = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(src + 1000 * lane);
    {   // inside a divergent branch: the ballot sees the active lanes only, readfirstlane returns the first ACTIVE lane's value
        unsigned long long bal = 0, rfl = 0;
        if (lane >= 5 && b > 0.0) {
            bal = __builtin_amdgcn_ballot_w64(c > 0.0);
            rfl = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(lane * 7 + 3);
        }
        o[s++] = bal;
        o[s++] = rfl;
    }
    {
        typedef double d4 __attribute__((ext_vector_type(4)));
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(c, d, acc, 0, 0, 0);
        o[s++] = bits(acc[0]); o[s++] = bits(acc[1]); o[s++] = bits(acc[2]); o[s++] = bits(acc[3]);
    }
    o[s++] = bits(__builtin_amdgcn_rcp(d));
    o[s++] = bits(__builtin_amdgcn_rsq(d));
    o[s++] = bits(__builtin_amdgcn_ldexp(a, src - 32));
    o[s++] = bits(__builtin_amdgcn_frexp_mant(a));
    o[s++] = (unsigned long long)(long long)__builtin_amdgcn_frexp_exp(a);
    while (s < CALIB_SLOTS) o[s++] = 0;
}

#define CALIB_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_ + 1000; } while (0)

extern "C" int crnn_calib_slots(void) { return CALIB_SLOTS; }

extern "C" int crnn_calib_run(const double *h_in /* [4][64] */, const int *h_idx /* [64] */, unsigned long long *h_out /* [64][CALIB_SLOTS] */) {
    double *d_in = nullptr; int *d_idx = nullptr; unsigned long long *d_out = nullptr;
    CALIB_TRY(hipMalloc((void **)&d_in, sizeof(double) * 256));
    CALIB_TRY(hipMalloc((void **)&d_idx, sizeof(int) * 64));
    CALIB_TRY(hipMalloc((void **)&d_out, sizeof(unsigned long long) * 64 * CALIB_SLOTS));
    CALIB_TRY(hipMemcpy(d_in, h_in, sizeof(double) * 256, hipMemcpyHostToDevice));
    CALIB_TRY(hipMemcpy(d_idx, h_idx, sizeof(int) * 64, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(calib_kernel, dim3(1), dim3(64), 0, 0, d_in, d_idx, d_out);
    CALIB_TRY(hipGetLastError());
    CALIB_TRY(hipDeviceSynchronize());
    CALIB_TRY(hipMemcpy(h_out, d_out, sizeof(unsigned long long) * 64 * CALIB_SLOTS, hipMemcpyDeviceToHost));
    (void)hipFree(d_in); (void)hipFree(d_idx); (void)hipFree(d_out);
    return 0;
}

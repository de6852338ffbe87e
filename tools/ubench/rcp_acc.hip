// accuracy of v_rcp_f64 on gfx950, raw and after one / two Newton steps (tools/ubench/rcp_acc.hip)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double *a, double *r0, double *r1, double *r2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = __builtin_amdgcn_rcp(a[i]);
    r0[i] = x;
    double e = fma(-a[i], x, 1.0); x = fma(x, e, x); r1[i] = x;
    e = fma(-a[i], x, 1.0); x = fma(x, e, x); r2[i] = x;
}
int main() {
    const int n = 1 << 22;
    std::vector<double> a(n), r0(n), r1(n), r2(n);
    unsigned long long s = 88172645463325252ULL;
    for (int i = 0; i < n; ++i) {
        s ^= s >> 12; s ^= s << 25; s ^= s >> 27;
        double u = (double)((s * 2685821657736338717ULL) >> 11) / 9007199254740992.0;
        a[i] = (i & 1) ? 1.0 + u : std::ldexp(0.5 + u, (int)(s % 80) - 40);
    }
    double *da, *d0, *d1, *d2;
    hipMalloc(&da, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(da, d0, d1, d2, n);
    hipMemcpy(r0.data(), d0, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(r2.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double m0 = 0, m1 = 0, m2 = 0;
    for (int i = 0; i < n; ++i) {
        long double ex = 1.0L / (long double)a[i];
        m0 = std::fmax(m0, (double)fabsl(((long double)r0[i] - ex) / ex));
        m1 = std::fmax(m1, (double)fabsl(((long double)r1[i] - ex) / ex));
        m2 = std::fmax(m2, (double)fabsl(((long double)r2[i] - ex) / ex));
    }
    printf("max relative error: raw v_rcp_f64 %.3e (%.2f ulp)  +1 Newton %.3e (%.2f ulp)  +2 Newton %.3e (%.2f ulp)\n", m0, m0 / 1.11e-16,
           m1, m1 / 1.11e-16, m2, m2 / 1.11e-16);
    return 0;
}

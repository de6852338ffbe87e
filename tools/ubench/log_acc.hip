// accuracy of the library's flog / flog_vec / fexp_vec against long double (tools/ubench/log_acc.hip)
#include "../../crnn_amd/csrc/ros23_kernel.hpp"
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double *a, double *l, double *e, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double c[2] = {a[i], a[i] * 1.7}, x[2];
    crnn::flog_vec<2>(c, x);
    l[i] = x[0];
    double z[1] = {x[0] * 3.0}, r[1];
    crnn::fexp_vec<1>(z, r);
    e[i] = r[0];
}
int main() {
    const int n = 1 << 22;
    std::vector<double> a(n), l(n), e(n);
    unsigned long long s = 88172645463325252ULL;
    for (int i = 0; i < n; ++i) {
        s ^= s >> 12; s ^= s << 25; s ^= s >> 27;
        double u = (double)((s * 2685821657736338717ULL) >> 11) / 9007199254740992.0;
        a[i] = (i & 1) ? 0.5 + 1.5 * u : std::ldexp(0.5 + u, (int)(s % 40) - 20);   // clamped concentrations: 1e-6 .. 10
    }
    double *da, *dl, *de;
    hipMalloc(&da, n * 8); hipMalloc(&dl, n * 8); hipMalloc(&de, n * 8);
    hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(da, dl, de, n);
    hipMemcpy(l.data(), dl, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(e.data(), de, n * 8, hipMemcpyDeviceToHost);
    double ml = 0, me = 0;
    for (int i = 0; i < n; ++i) {
        long double ex = logl((long double)a[i]);
        double ulp = std::ldexp(1.0, std::ilogb((double)fabsl(ex)) - 52);
        if (fabsl(ex) > 1e-300L) ml = std::fmax(ml, (double)(fabsl((long double)l[i] - ex) / ulp));
        long double ee = expl(3.0L * (long double)l[i]);
        me = std::fmax(me, (double)(fabsl((long double)e[i] - ee) / ee) / 1.11e-16);
    }
    printf("flog_vec: max error %.2f ulp;  fexp_vec: max relative error %.2f x 2^-53\n", ml, me);
    return 0;
}

// crnn_amd/csrc/p2vec.hpp -- parameter maps p -> theta and their Jacobians,
// compiled for host (crnn_p2vec) and device (the p2vec kernel of the fused
// training step) from one source.
//
//   CASE1  case1/case1.jl:70-78      w_b = p[1:nr] + b0,  w_out = reshape(p[nr+1:end], ns, nr),
//                                    w_in = clamp(-w_out, 0, 2.5)
//   CASE2  case2/case2.jl:91-99      slope = p[end]*100, w_b = p[1:nr]*slope, Ea = |p[..]*slope|,
//                                    w_in = [clamp(-w_out, 0, 4); Ea']
//   HYCHEM HyChem/crnn_pyrolysis_mass.jl:78-90 (see below)
//   ROBER  rober_crnn.jl:85-96       slope = |p[end]|, w_b = p[1:nr]*10*slope,
//                                    w_out = -w_in_raw * 10^w_out_raw, w_in = clamp(w_in_raw, 0, 2.5)
//
// Derivative conventions are ForwardDiff's: clamp' = 1 on the closed window,
// abs'(0) = +1.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define CRNN_HD __host__ __device__
#else
#define CRNN_HD
#endif

namespace crnn {

enum { PMAP_IDENTITY = 0, PMAP_CASE1 = 1, PMAP_CASE2 = 2, PMAP_ROBER = 3, PMAP_HYCHEM = 4 };

// has_temp = number of extra feature rows of w_in: 0, 1 (case2: -1/(R T)) or 2 (HyChem: -1/(R T), log T)
CRNN_HD inline int n_theta_of(int ns, int nr, int has_temp) { return nr * (ns + has_temp + 1 + ns); }

CRNN_HD inline int n_params_of(int pmap, int ns, int nr, int has_temp) {
    switch (pmap) {
    case PMAP_IDENTITY: return n_theta_of(ns, nr, has_temp);
    case PMAP_CASE1: return nr * (ns + 1);
    case PMAP_CASE2: return nr * (ns + 2) + 1;
    case PMAP_ROBER: return nr * (2 * ns + 1) + 1;
    case PMAP_HYCHEM: return nr * (2 * ns + 3) + 1;
    default: return -1;
    }
}

CRNN_HD inline double clampd(double v, double lo, double hi) { return v > hi ? hi : (v < lo ? lo : v); }
CRNN_HD inline double dclampd(double v, double lo, double hi) { return (v > hi || v < lo) ? 0.0 : 1.0; }
CRNN_HD inline double dabsd(double v) { return signbit(v) ? -1.0 : 1.0; }

// Writes only the structurally non-zero entries of dth (column-major nth x P);
// the caller zero-fills dth first.  dth may be null.
// (j0, jstep, i0, istep): the reactions j0, j0 + jstep, ... and, of each, the species i0, i0 + istep, ... (the entries that belong to
// a reaction alone go with i0 = 0).  Defaults: everything, serially (host).  On the device nr * ns threads call it with
// (t / ns, nr, t % ns, ns): one (species, reaction) pair each -- one thread writing all of theta and d theta / d p was 8 us of the
// launch between two solve launches.
CRNN_HD inline int p2vec_eval(int pmap, int ns, int nr, int has_temp, const double *p, double *th, double *dth,
                              int j0 = 0, int jstep = 1, int i0 = 0, int istep = 1) {
    const int n = ns + has_temp;
    const int nth = n_theta_of(ns, nr, has_temp);
    const int P = n_params_of(pmap, ns, nr, has_temp);
    const int o_in = 0, o_b = n * nr, o_out = (n + 1) * nr;
    if (P < 0) return -1;
#define DTH(row, col) dth[(row) + (int64_t)nth * (col)]
    if (pmap == PMAP_IDENTITY) {
        for (int k = j0 * ns + i0; k < nth; k += (jstep > 1 ? jstep * ns : 1)) { th[k] = p[k]; if (dth) DTH(k, k) = 1.0; }
    } else if (pmap == PMAP_CASE1) {
        if (has_temp) return -1;
        for (int j = j0; j < nr; j += jstep) {
            if (i0 == 0) {
                th[o_b + j] = p[j] + (-10.0);
                if (dth) DTH(o_b + j, j) = 1.0;
            }
            for (int i = i0; i < ns; i += istep) {
                const int k = nr + i + ns * j;
                const double wo = p[k];
                th[o_out + i + ns * j] = wo;
                th[o_in + i + n * j] = clampd(-wo, 0.0, 2.5);
                if (dth) { DTH(o_out + i + ns * j, k) = 1.0; DTH(o_in + i + n * j, k) = -dclampd(-wo, 0.0, 2.5); }
            }
        }
    } else if (pmap == PMAP_CASE2) {
        if (!has_temp) return -1;
        const double slope = p[P - 1] * 100.0;
        for (int j = j0; j < nr; j += jstep) {
            if (i0 == 0) {
                th[o_b + j] = p[j] * slope;
                if (dth) { DTH(o_b + j, j) = slope; DTH(o_b + j, P - 1) = p[j] * 100.0; }
            }
            for (int i = i0; i < ns; i += istep) {
                const int k = nr + i + ns * j;
                const double wo = p[k];
                th[o_out + i + ns * j] = wo;
                th[o_in + i + n * j] = clampd(-wo, 0.0, 4.0);
                if (dth) { DTH(o_out + i + ns * j, k) = 1.0; DTH(o_in + i + n * j, k) = -dclampd(-wo, 0.0, 4.0); }
            }
            if (i0 == 0) {
                const int ke = nr * (ns + 1) + j;
                const double v = p[ke] * slope;
                th[o_in + ns + n * j] = fabs(v);
                if (dth) { DTH(o_in + ns + n * j, ke) = dabsd(v) * slope; DTH(o_in + ns + n * j, P - 1) = dabsd(v) * p[ke] * 100.0; }
            }
        }
    } else if (pmap == PMAP_ROBER) {
        if (has_temp) return -1;
        const double ps = p[P - 1];
        const double slope = fabs(ps);
        const double ln10 = 2.302585092994045684;
        for (int j = j0; j < nr; j += jstep) {
            if (i0 == 0) {
                th[o_b + j] = p[j] * (10.0 * slope);
                if (dth) { DTH(o_b + j, j) = 10.0 * slope; DTH(o_b + j, P - 1) = p[j] * 10.0 * dabsd(ps); }
            }
            for (int i = i0; i < ns; i += istep) {
                const int ko = nr + i + ns * j;
                const int ki = nr * (ns + 1) + i + ns * j;
                const double wi_raw = p[ki], wo_raw = p[ko];
                const double pw = pow(10.0, wo_raw);
                th[o_out + i + ns * j] = -wi_raw * pw;
                th[o_in + i + n * j] = clampd(wi_raw, 0.0, 2.5);
                if (dth) {
                    DTH(o_out + i + ns * j, ki) = -pw;
                    DTH(o_out + i + ns * j, ko) = -wi_raw * pw * ln10;
                    DTH(o_in + i + n * j, ki) = dclampd(wi_raw, 0.0, 2.5);
                }
            }
        }
    } else if (pmap == PMAP_HYCHEM) {
        // HyChem/crnn_pyrolysis_mass.jl:78-90: slope = p[end]*10; w_b = p[1:nr]*slope; w_in_b = p[nr+1:2nr];
        // w_in_Ea = p[2nr+1:3nr]*slope; w_out = -w_in_raw .* 10^w_out_raw; w_in = [clamp(w_in_raw,0,2.5); Ea'; b']
        if (has_temp != 2) return -1;
        const double slope = p[P - 1] * 10.0;
        const double ln10 = 2.302585092994045684;
        for (int j = j0; j < nr; j += jstep) {
            if (i0 == 0) {
                th[o_b + j] = p[j] * slope;
                th[o_in + (ns + 1) + n * j] = p[nr + j];
                th[o_in + ns + n * j] = p[2 * nr + j] * slope;
                if (dth) {
                    DTH(o_b + j, j) = slope; DTH(o_b + j, P - 1) = p[j] * 10.0;
                    DTH(o_in + (ns + 1) + n * j, nr + j) = 1.0;
                    DTH(o_in + ns + n * j, 2 * nr + j) = slope; DTH(o_in + ns + n * j, P - 1) = p[2 * nr + j] * 10.0;
                }
            }
            for (int i = i0; i < ns; i += istep) {
                const int ko = 3 * nr + i + ns * j, ki = nr * (ns + 3) + i + ns * j;
                const double wo_raw = p[ko], wi_raw = p[ki];
                const double pw = pow(10.0, wo_raw);
                th[o_out + i + ns * j] = -wi_raw * pw;
                th[o_in + i + n * j] = clampd(wi_raw, 0.0, 2.5);
                if (dth) {
                    DTH(o_out + i + ns * j, ki) = -pw;
                    DTH(o_out + i + ns * j, ko) = -wi_raw * pw * ln10;
                    DTH(o_in + i + n * j, ki) = dclampd(wi_raw, 0.0, 2.5);
                }
            }
        }
    } else {
        return -1;
    }
#undef DTH
    return 0;
}

// Flux.Optimise chain: [norm clip] -> [ExpDecay] -> ADAM -> WeightDecay -> p .-= delta
// (case2/case2.jl:31-32,197; rober_crnn.jl:19,221-224).  EPS = 1e-8 as in
// Flux.Optimise.  state = [m(P) | v(P) | beta1^t, beta2^t, eta_expdecay, ncalls].
struct OptCfg {
    int32_t use_expdecay, decay_step;
    double ed_eta0, ed_decay, ed_clip;
    double eta, beta1, beta2, wd, grad_clip_norm;
};

CRNN_HD inline void opt_init(const OptCfg &o, int P, double *state) {
    for (int k = 0; k < 2 * P + 4; ++k) state[k] = 0.0;
    state[2 * P + 0] = o.beta1;
    state[2 * P + 1] = o.beta2;
    state[2 * P + 2] = o.ed_eta0;
}

// serial update (host, and thread 0 on device: P <= a few hundred)
CRNN_HD inline void opt_update(const OptCfg &o, int P, double *p, const double *grad, double gscale, double *state) {
    double *m = state, *v = state + P, *bp = state + 2 * P;
    double *ed_eta = state + 2 * P + 2, *ncalls = state + 2 * P + 3;
    const double eps = 1e-8;
    double gn = 0.0;
    bool clip = false;
    if (o.grad_clip_norm > 0) {
        for (int k = 0; k < P; ++k) { double g = grad[k] * gscale; gn += g * g; }
        gn = sqrt(gn);
        clip = gn > o.grad_clip_norm;
    }
    double eta_ed = 1.0;
    if (o.use_expdecay) {
        *ncalls += 1.0;
        if (fmod(*ncalls, (double)o.decay_step) == 0.0) {
            double e = *ed_eta * o.ed_decay;
            *ed_eta = e > o.ed_clip ? e : o.ed_clip;
        }
        eta_ed = *ed_eta;
    }
    for (int k = 0; k < P; ++k) {
        double g = grad[k] * gscale;
        if (clip) g = g / gn * o.grad_clip_norm;
        g *= eta_ed;
        m[k] = o.beta1 * m[k] + (1.0 - o.beta1) * g;
        v[k] = o.beta2 * v[k] + (1.0 - o.beta2) * g * g;
        double delta = m[k] / (1.0 - bp[0]) / (sqrt(v[k] / (1.0 - bp[1])) + eps) * o.eta;
        delta += o.wd * p[k];
        p[k] -= delta;
    }
    bp[0] *= o.beta1;
    bp[1] *= o.beta2;
}

}  // namespace crnn

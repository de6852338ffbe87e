/* oracle/sanitize_main.c -- TEST INFRASTRUCTURE (sanitizers on the CPU restatement).
 * Includes the oracle as one translation unit and drives every solver family on small seeded problems under
 * -fsanitize=address,undefined (tests/test_oracle_sanitizers.py builds and runs it).  Prints "OK" and exits 0. */
#include "crnn_oracle.c"
#include <stdio.h>

static double urand(unsigned long long *s) {   /* xorshift64*: deterministic, no libc state */
    *s ^= *s >> 12; *s ^= *s << 25; *s ^= *s >> 27;
    return (double)((*s * 2685821657736338717ULL) >> 11) / 9007199254740992.0;
}

int main(void) {
    unsigned long long seed = 88172645463325252ULL;
    int fails = 0;
    /* ---- CRNN family: case2-shaped (6 species + T, 3 reactions) and robertson-shaped, all three steppers ---- */
    for (int shape = 0; shape < 2; ++shape) {
        for (int solver = 0; solver < 3; ++solver) {
            orc_problem pb;
            orc_problem_defaults(&pb);
            if (shape == 0) { pb.ns = 6; pb.nr = 3; pb.has_temp = 1; pb.lb = 1e-6; pb.ub = 10.0; pb.inv_R = -1.0 / 1.98720425864083e-3; pb.clamp_pred = 1; }
            else { pb.ns = 3; pb.nr = 6; pb.has_temp = 0; pb.lb = 1e-8; pb.maxiters = 200000; pb.atol[1] = 1e-8; }
            pb.n_obs = pb.ns;
            orc_set_solver(&pb, solver);
            const int kind = shape == 0 ? 2 : 3, n = pb.ns + pb.has_temp, P = orc_n_params(kind, pb.ns, pb.nr), nth = orc_n_theta(&pb);
            double p[64], th[ORC_MAXTH], *dth = (double *)malloc(sizeof(double) * (size_t)nth * P);
            for (int k = 0; k < P; ++k) p[k] = 0.2 * (urand(&seed) - 0.5);
            if (shape == 0) { p[0] += 0.8; p[1] += 0.8; p[2] += 0.8; p[21] += 0.8; p[22] += 0.8; p[23] += 0.8; p[24] = 0.1; }
            else p[P - 1] = 0.3;
            if (orc_p2vec(kind, pb.ns, pb.nr, p, th, dth) != 0) ++fails;
            enum { D = 12 };
            double ts[D], u0[ORC_MAXN], data[ORC_MAXN * D], pred[ORC_MAXN * D], grad[64] = {0}, loss = 0;
            for (int j = 0; j < D; ++j) ts[j] = shape == 0 ? 50.0 * j / (D - 1) : pow(10.0, 3.0 * j / (D - 1));
            for (int i = 0; i < pb.ns; ++i) u0[i] = shape == 0 ? (i < 2 ? 0.2 + 2.0 * urand(&seed) : 0.0) : (i == 1 ? 1e-8 : 0.5 + urand(&seed));
            if (pb.has_temp) u0[pb.ns] = 323.0 + 20.0 * urand(&seed);
            for (int j = 0; j < pb.ns * D; ++j) data[j] = urand(&seed);
            int32_t nsv = 0;
            orc_stats st = {0, 0};
            int64_t sa[3];
            int rc = solver == 2 ? orc_solve_one_auto(&pb, th, dth, P, u0, ts, D, data, pred, NULL, &loss, grad, &nsv, &st, sa)
                                 : orc_solve_one(&pb, th, dth, P, u0, ts, D, data, pred, NULL, &loss, grad, &nsv, &st);
            if (rc < 0 || rc > 3 || nsv < 1 || !(loss == loss)) ++fails;
            if (solver == 0) {   /* the discrete-adjoint gradient path (grad_adjoint = 1): step tape, transposed solves */
                double grad2[64] = {0}, loss2 = 0;
                pb.grad_adjoint = 1;
                const int rc2 = orc_solve_one(&pb, th, dth, P, u0, ts, D, data, pred, NULL, &loss2, grad2, &nsv, &st);
                pb.grad_adjoint = 0;
                if (rc2 != rc || loss2 != loss) ++fails;
                for (int k = 0; k < P; ++k) if (!(fabs(grad2[k] - grad[k]) <= 1e-9 * (1.0 + fabs(grad[k])))) { ++fails; break; }
            }
            (void)n;
            free(dth);
        }
    }
    /* ---- batch driver (OpenMP path) ---- */
    {
        orc_problem pb;
        orc_problem_defaults(&pb);
        pb.ns = 5; pb.nr = 4; pb.lb = 1e-5; pb.ub = 10.0; pb.clamp_pred = 1; pb.n_obs = 5;
        const int P = orc_n_params(1, 5, 4), nth = orc_n_theta(&pb);
        double p[64], th[ORC_MAXTH], *dth = (double *)malloc(sizeof(double) * (size_t)nth * P);
        for (int k = 0; k < P; ++k) p[k] = 0.1 * (urand(&seed) - 0.5);
        orc_p2vec(1, 5, 4, p, th, dth);
        enum { B = 7, D = 9 };
        double ts[D], u0[5 * B], data[D * 5 * B], loss[B], grad[64];
        int32_t ret[B], nsv[B];
        int64_t stats[2];
        for (int j = 0; j < D; ++j) ts[j] = 40.0 * j / (D - 1);
        for (int k = 0; k < 5 * B; ++k) u0[k] = (k < 2 * B) ? 0.2 + urand(&seed) : 0.0;
        for (int k = 0; k < D * 5 * B; ++k) data[k] = urand(&seed);
        if (orc_solve_batch(&pb, th, dth, P, u0, ts, D, data, B, 1, B - 2, NULL, loss, grad, ret, nsv, stats, 2) != 0) ++fails;
        free(dth);
    }
    /* ---- optimiser chain ---- */
    {
        orc_opt o = {1, 5, 5e-3, 0.5, 1e-4, 0.005, 0.9, 0.999, 1e-6, 10.0};
        double state[2 * 25 + 4], p[25], g[25];
        orc_opt_init(&o, 25, state);
        for (int k = 0; k < 25; ++k) { p[k] = urand(&seed); g[k] = urand(&seed) - 0.5; }
        for (int it = 0; it < 12; ++it) orc_opt_update(&o, 25, p, g, state);
    }
    /* ---- cathode ---- */
    {
        orc_cathode c;
        orc_cathode_defaults(&c);
        c.rtol = 1e-3; c.atol = 1e-10;
        double th[17] = {1.0, 1.0, 1.0, 1.2, 1.4, 1.6, 0, 0, 0, 20.0, 22.0, 25.0, 0.5, 0.7, 1.0, 1.0, 1.0};
        enum { D = 20 };
        double ts[D], dbar[D], d2bar[D], hrr[D], loss = 0, grad[17];
        for (int j = 0; j < D; ++j) { ts[j] = 60.0 * j; dbar[j] = urand(&seed); d2bar[j] = dbar[j] * dbar[j] + 0.01; }
        int32_t nsv = 0;
        orc_stats st = {0, 0};
        int rc = orc_cathode_solve_one(&c, th, ts, D, dbar, d2bar, hrr, &loss, grad, &nsv, &st);
        if (rc < 0 || rc > 3) ++fails;
        c.solver = 2; c.qsteady_max = 1.0;     /* AutoTsit5 restatement */
        rc = orc_cathode_solve_one(&c, th, ts, D, dbar, d2bar, hrr, &loss, grad, &nsv, &st);
        if (rc < 0 || rc > 3) ++fails;
    }
    /* ---- HyChem ---- */
    {
        orc_hychem c;
        orc_hychem_defaults(&c);
        const int ns = 9, nr = 10, NP = nr * (2 * ns + 3) + 1, nth = nr * (2 * ns + 3);
        double *p = (double *)malloc(sizeof(double) * NP), *th = (double *)malloc(sizeof(double) * nth);
        double *dth = (double *)malloc(sizeof(double) * (size_t)nth * NP);
        for (int k = 0; k < NP; ++k) p[k] = 0.1 * (urand(&seed) - 0.5);
        p[NP - 1] = 0.1;
        const int P = NP;
        if (orc_hychem_p2vec(p, ns, nr, th, dth) != 0) ++fails;
        enum { D = 10 };
        double ts[D], Tt[D], Pt[D], u0[9], data[9 * D], pred[9 * D], loss = 0;
        double *grad = (double *)calloc((size_t)P, sizeof(double));
        for (int j = 0; j < D; ++j) { ts[j] = 1e-3 * j; Tt[j] = 1300.0; Pt[j] = 2.0 * 101325.0; }
        for (int i = 0; i < 9; ++i) u0[i] = 1e-8;
        u0[0] = 0.05; u0[8] = 0.95;
        for (int k = 0; k < 9 * D; ++k) data[k] = urand(&seed);
        for (int i = 0; i < 9; ++i) { c.scale[i] = 1.0; c.inv_yscale[i] = 1.0; }
        int32_t nsv = 0;
        orc_stats st = {0, 0};
        int rc = orc_hychem_solve_one(&c, th, dth, 3, u0, ts, D, D, Tt, Pt, data, pred, &loss, grad, &nsv, &st);
        if (rc < 0 || rc > 3) ++fails;
        free(p); free(th); free(dth); free(grad);
    }
    if (fails) { printf("FAIL %d\n", fails); return 1; }
    printf("OK\n");
    return 0;
}

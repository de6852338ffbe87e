/* examples/case2_train.c -- the C ABI of include/crnn_hip.h from plain C (no Python, no torch): what a compiled host of
 * the reference's case2 script (case2/case2.jl) would do.
 *
 *   build:  gcc -std=c99 -O2 -Iinclude examples/case2_train.c -o case2_train -Lcrnn_amd/csrc -lcrnn_hip -lm \
 *               -Wl,-rpath,$PWD/crnn_amd/csrc
 *   run  :  ./case2_train [n_exp] [n_steps]        (needs an MI355X; crnn_ctx_create fails loudly without one)
 *
 * Synthetic ensemble as in case2.jl:62-83: u0[1:2] ~ U(0.2, 2.2), T ~ U(323, 343) K; the data are the trajectories of a
 * perturbed parameter vector integrated by the library itself (tight tolerance), so that training from the unperturbed
 * start has something to learn.  Then `n_steps` device-resident optimiser steps (case2.jl:190-198) and the epoch-end loss.
 * Prints one line per step and exits 0 if the loss went down. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "crnn_hip.h"

#define CHECK(call)                                                                                 \
    do {                                                                                            \
        int32_t rc_ = (call);                                                                       \
        if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #call, (int)rc_, crnn_last_error(ctx)); return 1; } \
    } while (0)

/* The host's own collective handed to the library (crnn_comm_set_allreduce): a one-process host has nothing to add, a
 * multi-process one would call MPI_Allreduce(MPI_IN_PLACE, d_buf, n, MPI_DOUBLE, MPI_SUM, comm) on the device pointer here. */
static int n_collectives_seen = 0;
static int32_t host_allreduce(void *d_buf, int32_t n, void *hip_stream, void *user) {
    (void)d_buf; (void)hip_stream; (void)user;
    n_collectives_seen += (n == 25 + 6);      /* [grad(P) | n_overflow | loss_sum, n_ok, n_accept, n_reject, n_traj] */
    return 0;
}

static double urand(unsigned long long *s) {
    *s ^= *s >> 12; *s ^= *s << 25; *s ^= *s >> 27;
    return (double)((*s * 2685821657736338717ULL) >> 11) / 9007199254740992.0;
}

int main(int argc, char **argv) {
    const int64_t B = argc > 1 ? atoll(argv[1]) : 4096;
    const int n_steps = argc > 2 ? atoi(argv[2]) : 20;
    enum { NS = 6, N = 7, D = 50, P = 25 };
    crnn_ctx *ctx = NULL;
    crnn_config cfg;
    if (crnn_config_preset(&cfg, CRNN_PRESET_CASE2) != 0) { fprintf(stderr, "preset: %s\n", crnn_last_error(NULL)); return 1; }
    cfg.n_save = D;
    if (crnn_ctx_create(&cfg, &ctx) != 0) { fprintf(stderr, "crnn_ctx_create: %s\n", crnn_last_error(NULL)); return 2; }

    /* tsteps = range(0, 50, length = 50); u0_list [B x 7] column-major = IC-fastest */
    double ts[D];
    for (int j = 0; j < D; ++j) ts[j] = 50.0 * j / (D - 1);
    unsigned long long seed = 0x9E3779B97F4A7C15ULL;
    double *u0 = (double *)calloc((size_t)N * B, sizeof(double));
    double *data = (double *)calloc((size_t)D * NS * B, sizeof(double));
    double *pred = (double *)malloc(sizeof(double) * (size_t)D * N * B);
    double yscale[NS];
    for (int64_t b = 0; b < B; ++b) {
        u0[0 * B + b] = 0.2 + 2.0 * urand(&seed);
        u0[1 * B + b] = 0.2 + 2.0 * urand(&seed);
        u0[6 * B + b] = 323.0 + 20.0 * urand(&seed);
    }
    for (int i = 0; i < NS; ++i) yscale[i] = 1.0;
    CHECK(crnn_ctx_set_data(ctx, u0, data, ts, yscale, NULL, NS, B));

    /* a plausible mechanism: p_true; the optimiser starts from p0 = p_true with the rate parameters perturbed */
    double p_true[P], p0[P], theta[64], *dtheta = NULL;
    for (int k = 0; k < P; ++k) p_true[k] = 0.0;
    p_true[0] = 0.9; p_true[1] = 0.85; p_true[2] = 0.4;                 /* w_b / slope */
    {   /* w_out (6 x 3, column-major): TG + ROH -> DG + R'CO2R, DG + ROH -> MG + R'CO2R, MG + ROH -> GL + R'CO2R */
        const double wo[18] = {-1, -1, 1, 0, 0, 1,   0, -1, -1, 1, 0, 1,   0, -1, 0, -1, 1, 1};
        for (int k = 0; k < 18; ++k) p_true[3 + k] = wo[k];
    }
    p_true[21] = 0.73; p_true[22] = 0.72; p_true[23] = 0.32;           /* Ea / slope */
    p_true[24] = 0.2;                                                  /* slope / 100 */
    for (int k = 0; k < P; ++k) p0[k] = p_true[k];
    p0[0] += 0.05; p0[1] -= 0.04; p0[2] += 0.03; p0[21] -= 0.02; p0[22] += 0.02;

    /* data = predict_neuralode(u0, p_true) for all experiments (species rows only), yscale = max range per species */
    CHECK(crnn_p2vec(cfg.param_map, cfg.ns, cfg.nr, p_true, theta, dtheta));
    CHECK(crnn_solve(ctx, theta, NULL, 0, 0, B, D, pred, NULL, NULL, NULL, NULL, NULL));
    for (int i = 0; i < NS; ++i) {
        double lo = 1e300, hi = -1e300;
        for (int j = 0; j < D; ++j)
            for (int64_t b = 0; b < B; ++b) {
                const double v = pred[((size_t)j * N + i) * B + b];
                data[((size_t)j * NS + i) * B + b] = v;
                if (v < lo) lo = v;
                if (v > hi) hi = v;
            }
        yscale[i] = (hi - lo) + 1e-6;                                   /* max_min(...) + lb, case2.jl:71-73,83 */
    }
    CHECK(crnn_ctx_set_data(ctx, u0, data, ts, yscale, NULL, NS, B));

    /* opt = Optimiser(ExpDecay(...), ADAMW(...)); the loop of case2.jl:190-198, all ICs per step */
    crnn_opt_config opt;
    CHECK(crnn_opt_preset(&opt, CRNN_PRESET_CASE2));
    CHECK(crnn_train_init(ctx, &opt, p0));
    CHECK(crnn_comm_set_allreduce(ctx, host_allreduce, NULL));
    /* index order: the batch sums are then a function of (p, data) alone, which is what the bit-for-bit restart check below
     * needs; the default (queued by the previous step's step counts) is 10-20 % faster at this size and differs in the last bits */
    CHECK(crnn_ctx_set_queue_order(ctx, CRNN_QUEUE_INDEX));
    printf("library build: %s\n", crnn_build_info());
    double loss_first = 0.0, loss = 0.0;
    double p_ck[P], st_ck[2 * P + 4];
    for (int it = 0; it < n_steps; ++it) {
        if (it == n_steps / 2) {   /* `@save ... p opt`: parameters + optimiser state (ADAM moments, beta powers, ExpDecay) */
            CHECK(crnn_get_params(ctx, p_ck));
            CHECK(crnn_get_opt_state(ctx, st_ck));
        }
        CHECK(crnn_train_step(ctx, 0, B, D, &loss));
        if (it == 0) loss_first = loss;
        crnn_stats st;
        CHECK(crnn_last_stats(ctx, &st));
        printf("step %3d  loss %.6e  ok %lld/%lld  steps/traj %.1f  kernel %.3f ms\n", it, loss, (long long)st.n_ok,
               (long long)st.n_traj, (double)st.n_accept / (double)st.n_traj, st.kernel_ms);
    }
    double p_end[P], l_mean = 0.0, g[P];
    CHECK(crnn_get_params(ctx, p_end));
    if (crnn_opt_state_len(P) != 2 * P + 4 || n_collectives_seen != n_steps || crnn_comm_collectives(ctx) != n_steps) {
        fprintf(stderr, "collective / state bookkeeping is off: %d %lld\n", n_collectives_seen, (long long)crnn_comm_collectives(ctx));
        return 4;
    }
    {   /* `@load` and resume: from the mid-run checkpoint the second half repeats bit for bit */
        double p_re[P];
        CHECK(crnn_set_params(ctx, p_ck));
        CHECK(crnn_set_opt_state(ctx, st_ck));
        for (int it = n_steps / 2; it < n_steps; ++it) CHECK(crnn_train_step(ctx, 0, B, D, NULL));
        CHECK(crnn_get_params(ctx, p_re));
        for (int k = 0; k < P; ++k)
            if (p_re[k] != p_end[k]) { fprintf(stderr, "restart diverged at p[%d]\n", k); return 5; }
        printf("restart from the mid-run checkpoint reproduces the final parameters bit for bit\n");
    }
    CHECK(crnn_loss_grad(ctx, p_end, 0, B, D, &l_mean, g, NULL));       /* epoch-end loss (case2.jl:199-201) + gradient */
    double gn = 0.0;
    for (int k = 0; k < P; ++k) gn += g[k] * g[k];
    printf("final mean loss %.6e  |grad| %.3e  (first step %.6e)\n", l_mean, sqrt(gn), loss_first);
    crnn_ctx_destroy(ctx);
    free(u0); free(data); free(pred);
    return l_mean < loss_first ? 0 : 3;
}

/*
 * oracle/crnn_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C (double precision, scalar) restatement of the CRNN neural-ODE hot
 * path of DENG-MIT/CRNN.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this file's shared object; the product path
 * (crnn_amd/csrc) never links, imports or calls it.
 *
 * What it restates (paths relative to /root/reference):
 *   RHS   crnn / crnn!          case2/case2.jl:113-118, case1/case1.jl:80-83,
 *                               robertson/rober_crnn.jl:113-116
 *   p2vec                       case2/case2.jl:91-99, case1/case1.jl:70-78,
 *                               robertson/rober_crnn.jl:85-96
 *   solve(prob, Rosenbrock23)   call sites case2/case2.jl:126,
 *                               robertson/rober_crnn.jl:125-127
 *   clamp.(Array(sol),-ub,ub)   case2/case2.jl:126, case1/case1.jl:94-95
 *   loss_neuralode (mae)        case2/case2.jl:132-137,
 *                               robertson/rober_crnn.jl:139-144
 *   ForwardDiff.gradient        case2/case2.jl:195, robertson/rober_crnn.jl:219
 *   update!(opt,p,grad)         case2/case2.jl:31-32,197,
 *                               robertson/rober_crnn.jl:19,221-224
 *
 * PARITY STATUS: pinned, since round 6, to numbers the reference's own solver
 * stack computed -- for case2 (see "(round 6)" below): the loss at the saved p
 * to 5e-6 and the first 25 recorded epochs of training (500 gradient + update
 * steps) to 1e-4 ... 2e-2 per epoch.  Step-for-step identity with
 * OrdinaryDiffEq's internals (individual step sizes) is not observable in
 * anything the reference holds and stays *unpinned*; robertson, HyChem and the
 * cathode have no seeded, recorded run and are pinned through the same code
 * paths only (and robertson distributionally, "(round 5)" below).  The arithmetic
 * of `solve`, `ForwardDiff.gradient` and `update!` lives in un-vendored Julia
 * packages (OrdinaryDiffEq / DiffEqBase / ForwardDiff / Flux; no Manifest for
 * case1, case2, robertson -- README.md:15-21 says only "Julia 1.6").  Julia is
 * not installed here, so the reference cannot be executed.  The stepper below
 * restates the *published* algorithm (Shampine & Reichelt ode23s triple as
 * implemented by OrdinaryDiffEq's Rosenbrock23: d = 1/(2+sqrt 2),
 * c32 = 6+sqrt 2, Hermite-type dense output, PI step controller, Hairer
 * initial step) from memory of those packages.  What IS pinned (tests/):
 *   - closed-form RHS / p2vec against independent NumPy restatements,
 *   - trajectories against SciPy Radau (rtol 1e-12) golden vectors,
 *   - gradients against central finite differences and against SciPy-based
 *     finite differences of the converged solution,
 *   - the classical Robertson known answers,
 *   - the reference's own checkpoint parameter vectors
 *     (case2/checkpoint/mymodel.bson, robertson/checkpoint/mymodel.bson)
 *     mapping through p2vec to the physically known mechanisms,
 *   - (round 5) the training metrics the reference's checkpoints recorded
 *     with ITS solver stack at the saved p -- l_loss_train / l_loss_val
 *     (case2/case2.jl:159-160,178; robertson/rober_crnn.jl:176-177,201) and
 *     l_grad, the epoch mean of ||ForwardDiff.gradient||_2
 *     (rober_crnn.jl:218,229,178) -- as a DISTRIBUTIONAL pin: the same
 *     metrics formed here on experiments re-drawn from the reference's design
 *     reproduce the recorded values (robertson loss inside the reference's
 *     own last-50-epoch band, gradient norm within 12 %; case2 loss within
 *     4 %), and a 1 % change of p does not (tests/test_ckpt_history_pin.py).
 *     Coarse, but computed by the reference itself: it bounds any error of
 *     p2vec + RHS + stiff solve + loss (+ gradient) far below 1 % in p.
 *   - (round 6) case2's EXACT experiments: case2/case2.jl:11 seeds Julia's
 *     RNG, so u0_list (:60), the 30 noise draws (:79), the initial p (:86)
 *     and every epoch's randperm (:194) are a deterministic stream, restated
 *     in tests/golden/julia_rng.py (Julia 1.6 MersenneTwister; pinned to the
 *     values Julia's documentation prints).  On those experiments
 *       l_loss_train[end] = 1.6512280e-2, l_loss_val[end] = 1.3958350e-2
 *     (computed by OrdinaryDiffEq at the saved p) are reproduced by
 *     orc_solve_batch to 3.4e-6 / 4.7e-6 (solver = Tsit5 / the composite,
 *     which never leaves Tsit5 here; Rosenbrock23: 1.1e-3), and
 *       l_loss_train[1:25], l_loss_val[1:25]
 *     by replaying the training loop -- chunked ForwardDiff-style gradient
 *     with errnorm_sens = 2 + the Flux optimiser chain -- to <= 5e-4 on the
 *     first six epochs, median 7e-4 and <= 2e-2 over 25 (what a 0.1 % change of
 *     rtol moves the replay by); errnorm_sens = 1 is 1e-2 off from epoch 1,
 *     the primal-only norm 1e-1 off by epoch 2.  tests/test_case2_stream_pin.py.
 *
 * Gradient method: forward tangents pushed through every arithmetic operation
 * of the accepted Rosenbrock23 steps with the step sizes held as plain (non
 * dual) numbers -- i.e. what ForwardDiff.gradient does to the adaptive solver
 * (discretise-then-differentiate).  The error norm that drives step-size
 * control uses the primal values only (errnorm_sens = 0) or, optionally, the
 * ForwardDiff-style norm that includes the partials (errnorm_sens = 1,
 * [UNVERIFIED-DEP] DiffEqBase ODE_DEFAULT_NORM for Dual arrays).
 * grad_adjoint = 1 (Rosenbrock23, errnorm_sens = 0) forms the SAME derivative
 * backwards -- the discrete adjoint of the accepted steps, the algorithm the GPU
 * kernels run -- with dense matrices; the two agree to 1e-15 of max |grad|
 * (tests/test_oracle_golden.py) and bench.py times both as the CPU baseline.
 *
 * Restated here and NOT on the device (end of round 4; "oracle first": the checkers of kernels that do not exist yet, and the
 * measurement of what the device's choices leave out -- INTEGRATION.md quotes the numbers):
 *   - Rosenbrock23(autodiff = false): J by FiniteDiff's forward differences (orc_jac_fd; the device offers it for primal launches)
 *     and ForwardDiff's derivative THROUGH the difference quotients (orc_jac_fd_dir); HyChem's finite-difference J and time
 *     derivative on the T(t), P(t) tables (orc_hychem.jac_fd, primal solves);
 *   - the gradient of configs 4 and 5 as the reference really evaluates it -- ForwardDiff's chunks through the AutoTsit5 composite with
 *     the chunk's partials in the error estimate of every algorithm: HyChem solver 2 with errnorm_sens; cathode solver 2 / 3 with
 *     errnorm_sens, including tangent copies through TRBDF2's Newton iteration (cath_cp).
 *
 * Layout conventions (identical to the C ABI in include/crnn_hip.h):
 *   theta = [ w_in (n x nr, column-major) | w_b (nr) | w_out (ns x nr, col-major) ]
 *   n = ns + has_temp; state u = [species..., T]
 *   batched arrays are "IC-fastest": u0[i*B + b], data[(j*n_obs + i)*B + b],
 *   pred[(j*n + i)*B + b].
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAXN 12
#define ORC_MAXR 16
#define ORC_MAXTH (ORC_MAXR * (2 * ORC_MAXN + 1))

typedef struct orc_problem {
    int32_t ns, nr, has_temp;   /* n = ns + has_temp */
    int32_t n_obs;              /* observed species count */
    int32_t i_obs[ORC_MAXN];    /* 0-based indices of observed species */
    int32_t clamp_pred;         /* pred = clamp(sol, -ub, ub)  (case1/case2) */
    int32_t loss_kind;          /* 0 = MAE, 1 = MSE */
    int32_t maxiters;
    int32_t errnorm_sens;       /* 0 primal-only norm, 1 ForwardDiff-style (/ length(u)), 2 ForwardDiff-style (/ totallength(u)) */
    int32_t solver;             /* 0 Rosenbrock23, 1 Tsit5 (case1/case1.jl:28), 2 AutoTsit5(Rosenbrock23) (case2/case2.jl:26) */
    int32_t grad_adjoint;       /* 1: Rosenbrock23 gradients by the discrete adjoint of the accepted steps (solve_one_adj) instead
                                   of forward tangents -- the SAME derivative, the algorithm the GPU runs; CPU-baseline timing
                                   and an independent check of the adjoint formulas (bench.py, tests/test_oracle_golden.py) */
    int32_t jac_fd;             /* 1: Rosenbrock23(autodiff = false) (case2/case2.jl:26): W = I - gam J with J by forward differences of the
                                   right-hand side (orc_jac_fd); forward tangents differentiate through the quotient (orc_jac_fd_dir) */
    double lb, ub;              /* log-clamp window; ub may be +inf */
    double inv_R;               /* -1/R for the Arrhenius row (has_temp) */
    double rate_scale[ORC_MAXN];/* dydt_scale (robertson), else 1 */
    double atol[ORC_MAXN], rtol[ORC_MAXN];
    double yscale[ORC_MAXN];    /* per observed species */
    double t0;
    /* step-size controller (OrdinaryDiffEq defaults for Rosenbrock23) */
    double gamma, qmin, qmax, beta1, beta2, qsteady_min, qsteady_max, qoldinit;
    double dtmin;
} orc_problem;

int orc_sizeof_problem(void) { return (int)sizeof(orc_problem); }

/* PIController defaults per algorithm: beta2 = 2/(5 order), beta1 = 7/(10 order); the steady band [1, 6/5]
   exists only for the implicit family (qsteady_max = 1 for explicit methods). [UNVERIFIED-DEP] */
void orc_set_solver(orc_problem *pb, int solver) {
    pb->solver = solver;
    if (solver == 1 || solver == 2) { pb->beta1 = 7.0 / 50.0; pb->beta2 = 2.0 / 25.0; pb->qsteady_max = 1.0; }   /* composite: qsteady_max_default of a non-implicit algorithm type */
    else { pb->beta1 = 7.0 / 20.0; pb->beta2 = 2.0 / 10.0; pb->qsteady_max = 1.2; }
}

void orc_problem_defaults(orc_problem *pb) {
    memset(pb, 0, sizeof(*pb));
    for (int i = 0; i < ORC_MAXN; ++i) {
        pb->rate_scale[i] = 1.0; pb->yscale[i] = 1.0;
        pb->atol[i] = 1e-6; pb->rtol[i] = 1e-3; pb->i_obs[i] = i;
    }
    pb->maxiters = 100000;
    pb->ub = INFINITY;
    /* PIController defaults: beta2 = 2/(5*order), beta1 = 7/(10*order), order 2;
       gamma 9/10, qmin 1/5, qmax 10, qsteady in [1, 6/5] for implicit algs. */
    pb->gamma = 0.9; pb->qmin = 0.2; pb->qmax = 10.0;
    pb->beta1 = 7.0 / 20.0; pb->beta2 = 2.0 / 10.0;
    pb->qsteady_min = 1.0; pb->qsteady_max = 1.2; pb->qoldinit = 1e-4;
    pb->dtmin = 0.0;
}

/* ------------------------------------------------------------------------ */
/* RHS family  du = scale .* (w_out * exp(w_in' * x + w_b)),                */
/*   x_i = log(clamp(u_i, lb, ub)) (i < ns),  x_ns = inv_R / T (has_temp).  */
/* Follows case2/case2.jl:114-118; case1/case1.jl:80-83;                    */
/* robertson/rober_crnn.jl:113-116.                                         */
/* ------------------------------------------------------------------------ */
static inline int N_(const orc_problem *pb) { return pb->ns + pb->has_temp; }
static inline const double *W_IN(const orc_problem *pb, const double *th) { (void)pb; return th; }
static inline const double *W_B(const orc_problem *pb, const double *th) { return th + N_(pb) * pb->nr; }
static inline const double *W_OUT(const orc_problem *pb, const double *th) { return th + (N_(pb) + 1) * pb->nr; }
int orc_n_theta(const orc_problem *pb) { return pb->nr * (N_(pb) + 1 + pb->ns); }

/* x, dx/du (g) and d2x/du2 (h).  ForwardDiff differentiates clamp as 1 inside
   the closed window and 0 outside. */
static void feat(const orc_problem *pb, const double *u, double *x, double *g, double *h) {
    for (int i = 0; i < pb->ns; ++i) {
        double ui = u[i];
        int inside = (ui >= pb->lb) && (ui <= pb->ub);
        double c = ui < pb->lb ? pb->lb : (ui > pb->ub ? pb->ub : ui);
        x[i] = log(c);
        if (g) g[i] = inside ? 1.0 / ui : 0.0;
        if (h) h[i] = inside ? -1.0 / (ui * ui) : 0.0;
    }
    if (pb->has_temp) {
        double T = u[pb->ns];
        x[pb->ns] = pb->inv_R / T;
        if (g) g[pb->ns] = -pb->inv_R / (T * T);
        if (h) h[pb->ns] = 2.0 * pb->inv_R / (T * T * T);
    }
}

void orc_rhs(const orc_problem *pb, const double *th, const double *u, double *du) {
    int n = N_(pb), ns = pb->ns, nr = pb->nr;
    const double *w_in = W_IN(pb, th), *w_b = W_B(pb, th), *w_out = W_OUT(pb, th);
    double x[ORC_MAXN], r[ORC_MAXR];
    feat(pb, u, x, NULL, NULL);
    for (int j = 0; j < nr; ++j) {
        double z = w_b[j];
        for (int i = 0; i < n; ++i) z += w_in[i + n * j] * x[i];
        r[j] = exp(z);
    }
    for (int i = 0; i < ns; ++i) {
        double a = 0.0;
        for (int j = 0; j < nr; ++j) a += w_out[i + ns * j] * r[j];
        du[i] = a * pb->rate_scale[i];
    }
    if (pb->has_temp) du[ns] = 0.0;
}

/* Analytic Jacobian J[i + n*c] = d du_i / d u_c. */
void orc_jac(const orc_problem *pb, const double *th, const double *u, double *J) {
    int n = N_(pb), ns = pb->ns, nr = pb->nr;
    const double *w_in = W_IN(pb, th), *w_b = W_B(pb, th), *w_out = W_OUT(pb, th);
    double x[ORC_MAXN], g[ORC_MAXN], r[ORC_MAXR];
    feat(pb, u, x, g, NULL);
    for (int j = 0; j < nr; ++j) {
        double z = w_b[j];
        for (int i = 0; i < n; ++i) z += w_in[i + n * j] * x[i];
        r[j] = exp(z);
    }
    for (int c = 0; c < n; ++c)
        for (int i = 0; i < n; ++i) {
            double a = 0.0;
            if (i < ns)
                for (int j = 0; j < nr; ++j) a += w_out[i + ns * j] * r[j] * w_in[c + n * j];
            J[i + n * c] = (i < ns) ? a * g[c] * pb->rate_scale[i] : 0.0;
        }
}

/* Rosenbrock23(autodiff = false) (case2/case2.jl:26, robertson/rober_crnn_lm.jl:34): OrdinaryDiffEq fills J with
   FiniteDiff.finite_difference_jacobian!(J, f, u, Val(:forward)) -- column c = (f(u + eps_c e_c) - f(u)) / eps_c with
   eps_c = max(relstep |u_c|, absstep), relstep = absstep = sqrt(eps(Float64)) (FiniteDiff's default_relstep / compute_epsilon for
   forward differences) [UNVERIFIED-DEP: FiniteDiff.jl is not vendored with the reference; restated from its published algorithm].
   f0 = f(u) is the value the stepper already holds.  The temperature state gets its column too (its row is zero: du_T = 0), as
   FiniteDiff differences every state.  The companion time derivative dT = (f(t + eps_t) - f(t)) / eps_t is exactly zero for these
   right-hand sides (they do not read t) and is not formed. */
void orc_jac_fd(const orc_problem *pb, const double *th, const double *u, const double *f0, double *J) {
    const int n = N_(pb);
    const double rel = 1.4901161193847656e-08;   /* sqrt(2^-52) */
    double up[ORC_MAXN], fp[ORC_MAXN];
    memcpy(up, u, sizeof(double) * n);
    for (int c = 0; c < n; ++c) {
        const double eps = fmax(rel * fabs(u[c]), rel);
        up[c] = u[c] + eps;
        orc_rhs(pb, th, up, fp);
        for (int i = 0; i < n; ++i) J[i + n * c] = (fp[i] - f0[i]) / eps;
        up[c] = u[c];
    }
}

/* Directional derivative of f along (su, dth): out = f_u su + f_theta dth. */
void orc_rhs_jvp(const orc_problem *pb, const double *th, const double *dth,
                 const double *u, const double *su, double *out) {
    int n = N_(pb), ns = pb->ns, nr = pb->nr;
    const double *w_in = W_IN(pb, th), *w_b = W_B(pb, th), *w_out = W_OUT(pb, th);
    const double *dw_in = W_IN(pb, dth), *dw_b = W_B(pb, dth), *dw_out = W_OUT(pb, dth);
    double x[ORC_MAXN], g[ORC_MAXN], r[ORC_MAXR], dr[ORC_MAXR];
    feat(pb, u, x, g, NULL);
    for (int j = 0; j < nr; ++j) {
        double z = w_b[j], dz = dw_b[j];
        for (int i = 0; i < n; ++i) {
            z += w_in[i + n * j] * x[i];
            dz += dw_in[i + n * j] * x[i] + w_in[i + n * j] * g[i] * su[i];
        }
        r[j] = exp(z);
        dr[j] = r[j] * dz;
    }
    for (int i = 0; i < ns; ++i) {
        double a = 0.0;
        for (int j = 0; j < nr; ++j) a += dw_out[i + ns * j] * r[j] + w_out[i + ns * j] * dr[j];
        out[i] = a * pb->rate_scale[i];
    }
    if (pb->has_temp) out[ns] = 0.0;
}

/* Directional derivative of the finite-difference Jacobian along (su, dth) -- what ForwardDiff.gradient does to
   Rosenbrock23(autodiff = false) (case2/case2.jl:26 with :195): the Duals go THROUGH FiniteDiff's quotient.  With x = u + su eps_D,
   column c = (f(x + e e_c) - f(x)) / e, e = max(rel |x_c|, rel):
     d column c = [ f'(u + e e_c; su + e' e_c, dth) - f'(u; su, dth) ] / e  -  J_fd[:, c] e' / e,
   e' = rel sign(u_c) su_c where rel |u_c| > rel (the first argument of max wins: |u_c| > 1), else 0 (the absolute increment is a
   constant); abs'(0) = +1 as everywhere here.  df0 = f'(u; su, dth) is the tangent the stepper holds.  [UNVERIFIED-DEP] as orc_jac_fd. */
void orc_jac_fd_dir(const orc_problem *pb, const double *th, const double *dth, const double *u, const double *su,
                    const double *df0, const double *Jfd, double *dJ) {
    const int n = N_(pb);
    const double rel = 1.4901161193847656e-08;
    double up[ORC_MAXN], sp[ORC_MAXN], dfp[ORC_MAXN];
    memcpy(up, u, sizeof(double) * n);
    memcpy(sp, su, sizeof(double) * n);
    for (int c = 0; c < n; ++c) {
        const double a = rel * fabs(u[c]);
        const double eps = fmax(a, rel);
        const double deps = (a > rel) ? rel * (signbit(u[c]) ? -1.0 : 1.0) * su[c] : 0.0;
        up[c] = u[c] + eps;
        sp[c] = su[c] + deps;
        orc_rhs_jvp(pb, th, dth, up, sp, dfp);
        for (int i = 0; i < n; ++i) dJ[i + n * c] = (dfp[i] - df0[i]) / eps - Jfd[i + n * c] * (deps / eps);
        up[c] = u[c];
        sp[c] = su[c];
    }
}

/* Directional derivative of the Jacobian along (su, dth): dJ (n x n). */
void orc_jac_dir(const orc_problem *pb, const double *th, const double *dth,
                 const double *u, const double *su, double *dJ) {
    int n = N_(pb), ns = pb->ns, nr = pb->nr;
    const double *w_in = W_IN(pb, th), *w_b = W_B(pb, th), *w_out = W_OUT(pb, th);
    const double *dw_in = W_IN(pb, dth), *dw_b = W_B(pb, dth), *dw_out = W_OUT(pb, dth);
    double x[ORC_MAXN], g[ORC_MAXN], h[ORC_MAXN], r[ORC_MAXR], dr[ORC_MAXR];
    feat(pb, u, x, g, h);
    for (int j = 0; j < nr; ++j) {
        double z = w_b[j], dz = dw_b[j];
        for (int i = 0; i < n; ++i) {
            z += w_in[i + n * j] * x[i];
            dz += dw_in[i + n * j] * x[i] + w_in[i + n * j] * g[i] * su[i];
        }
        r[j] = exp(z);
        dr[j] = r[j] * dz;
    }
    /* J[i,c] = scale_i * g_c * sum_j w_out[i,j] r_j w_in[c,j] */
    for (int c = 0; c < n; ++c) {
        double dg = h[c] * su[c];
        for (int i = 0; i < n; ++i) {
            if (i >= ns) { dJ[i + n * c] = 0.0; continue; }
            double a = 0.0, da = 0.0;
            for (int j = 0; j < nr; ++j) {
                double wo = w_out[i + ns * j], wi = w_in[c + n * j];
                a += wo * r[j] * wi;
                da += dw_out[i + ns * j] * r[j] * wi + wo * dr[j] * wi + wo * r[j] * dw_in[c + n * j];
            }
            dJ[i + n * c] = pb->rate_scale[i] * (da * g[c] + a * dg);
        }
    }
}

/* ------------------------------------------------------------------------ */
/* p2vec variants and their Jacobians d theta / d p (n_theta x P col-major) */
/* kind: 1 = case1 (case1/case1.jl:70-78), 2 = case2 (case2/case2.jl:91-99),*/
/*       3 = robertson (robertson/rober_crnn.jl:85-96).                     */
/* abs'(0) = +1 and clamp' = 1 on the closed window, as ForwardDiff does.   */
/* ------------------------------------------------------------------------ */
int orc_n_params(int kind, int ns, int nr) {
    switch (kind) {
    case 1: return nr * (ns + 1);
    case 2: return nr * (ns + 2) + 1;
    case 3: return nr * (2 * ns + 1) + 1;
    default: return -1;
    }
}

static double clampd(double v, double lo, double hi) { return v > hi ? hi : (v < lo ? lo : v); }
static double dclamp(double v, double lo, double hi) { return (v > hi || v < lo) ? 0.0 : 1.0; }
static double dabs_(double v) { return signbit(v) ? -1.0 : 1.0; }

int orc_p2vec(int kind, int ns, int nr, const double *p, double *th, double *dth /* may be NULL */) {
    int has_temp = (kind == 2);
    int n = ns + has_temp;
    int nth = nr * (n + 1 + ns);
    int P = orc_n_params(kind, ns, nr);
    if (P < 0) return -1;
    double *w_in = th, *w_b = th + n * nr, *w_out = th + (n + 1) * nr;
    if (dth) memset(dth, 0, sizeof(double) * (size_t)nth * (size_t)P);
#define DTH(row, col) dth[(row) + (size_t)nth * (col)]
    int o_in = 0, o_b = n * nr, o_out = (n + 1) * nr;
    if (kind == 1) {
        const double b0 = -10.0; /* case1/case1.jl:70 */
        for (int j = 0; j < nr; ++j) {
            w_b[j] = p[j] + b0;
            if (dth) DTH(o_b + j, j) = 1.0;
            for (int i = 0; i < ns; ++i) {
                int k = nr + i + ns * j;
                double wo = p[k];
                w_out[i + ns * j] = wo;
                w_in[i + n * j] = clampd(-wo, 0.0, 2.5);
                if (dth) {
                    DTH(o_out + i + ns * j, k) = 1.0;
                    DTH(o_in + i + n * j, k) = -dclamp(-wo, 0.0, 2.5);
                }
            }
        }
    } else if (kind == 2) {
        double slope = p[P - 1] * 100.0;
        for (int j = 0; j < nr; ++j) {
            w_b[j] = p[j] * slope;
            if (dth) { DTH(o_b + j, j) = slope; DTH(o_b + j, P - 1) = p[j] * 100.0; }
            for (int i = 0; i < ns; ++i) {
                int k = nr + i + ns * j;
                double wo = p[k];
                w_out[i + ns * j] = wo;
                w_in[i + n * j] = clampd(-wo, 0.0, 4.0);
                if (dth) {
                    DTH(o_out + i + ns * j, k) = 1.0;
                    DTH(o_in + i + n * j, k) = -dclamp(-wo, 0.0, 4.0);
                }
            }
            int ke = nr * (ns + 1) + j;
            double v = p[ke] * slope;
            w_in[ns + n * j] = fabs(v);
            if (dth) {
                DTH(o_in + ns + n * j, ke) = dabs_(v) * slope;
                DTH(o_in + ns + n * j, P - 1) = dabs_(v) * p[ke] * 100.0;
            }
        }
    } else if (kind == 3) {
        double ps = p[P - 1];
        double slope = fabs(ps);
        const double ln10 = 2.302585092994045684;
        for (int j = 0; j < nr; ++j) {
            w_b[j] = p[j] * (10.0 * slope);
            if (dth) { DTH(o_b + j, j) = 10.0 * slope; DTH(o_b + j, P - 1) = p[j] * 10.0 * dabs_(ps); }
            for (int i = 0; i < ns; ++i) {
                int ko = nr + i + ns * j;
                int ki = nr * (ns + 1) + i + ns * j;
                double wi_raw = p[ki], wo_raw = p[ko];
                double pw = pow(10.0, wo_raw);
                w_out[i + ns * j] = -wi_raw * pw;
                w_in[i + n * j] = clampd(wi_raw, 0.0, 2.5);
                if (dth) {
                    DTH(o_out + i + ns * j, ki) = -pw;
                    DTH(o_out + i + ns * j, ko) = -wi_raw * pw * ln10;
                    DTH(o_in + i + n * j, ki) = dclamp(wi_raw, 0.0, 2.5);
                }
            }
        }
    }
#undef DTH
    return 0;
}

/* ------------------------------------------------------------------------ */
/* Dense LU with partial pivoting (n <= ORC_MAXN).                          */
/* ------------------------------------------------------------------------ */
static long g_lu_swaps = 0;   /* row exchanges since the last orc_lu_swaps(1): lets a test show that a scenario pivots */
long orc_lu_swaps(int reset) { long v = g_lu_swaps; if (reset) g_lu_swaps = 0; return v; }
static int lu_factor(int n, double *A, int *piv) {
    for (int k = 0; k < n; ++k) {
        int p = k; double best = fabs(A[k + n * k]);
        for (int i = k + 1; i < n; ++i) { double v = fabs(A[i + n * k]); if (v > best) { best = v; p = i; } }
        piv[k] = p;
        if (p != k) ++g_lu_swaps;
        if (p != k) for (int c = 0; c < n; ++c) { double t = A[k + n * c]; A[k + n * c] = A[p + n * c]; A[p + n * c] = t; }
        double d = A[k + n * k];
        if (d == 0.0) return -1;
        double inv = 1.0 / d;
        for (int i = k + 1; i < n; ++i) A[i + n * k] *= inv;
        for (int c = k + 1; c < n; ++c) {
            double a = A[k + n * c];
            for (int i = k + 1; i < n; ++i) A[i + n * c] -= A[i + n * k] * a;
        }
    }
    return 0;
}
static void lu_solve(int n, const double *A, const int *piv, double *b) {
    for (int k = 0; k < n; ++k) { int p = piv[k]; if (p != k) { double t = b[k]; b[k] = b[p]; b[p] = t; } }
    for (int k = 0; k < n; ++k) { double a = b[k]; for (int i = k + 1; i < n; ++i) b[i] -= A[i + n * k] * a; }
    for (int k = n - 1; k >= 0; --k) { b[k] /= A[k + n * k]; double a = b[k]; for (int i = 0; i < k; ++i) b[i] -= A[i + n * k] * a; }
}

static void matvec(int n, const double *A, const double *v, double *out) {
    for (int i = 0; i < n; ++i) { double a = 0.0; for (int c = 0; c < n; ++c) a += A[i + n * c] * v[c]; out[i] = a; }
}

static double rms_scaled(const orc_problem *pb, int n, const double *v, const double *ua, const double *ub_) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) {
        double m = fmax(fabs(ua[i]), fabs(ub_[i]));
        double e = v[i] / (pb->atol[i] + pb->rtol[i] * m);
        s += e * e;
    }
    return sqrt(s / n);
}

/* Hairer initial step as in OrdinaryDiffEq's ode_determine_initdt
   [UNVERIFIED-DEP] (order = 2 for Rosenbrock23). */
static double init_dt(const orc_problem *pb, const double *th, const double *u0, const double *f0, double tspan_len, int order) {
    int n = N_(pb);
    double sk[ORC_MAXN], d0 = 0, d1 = 0;
    for (int i = 0; i < n; ++i) {
        sk[i] = pb->atol[i] + fabs(u0[i]) * pb->rtol[i];
        d0 += (u0[i] / sk[i]) * (u0[i] / sk[i]);
        d1 += (f0[i] / sk[i]) * (f0[i] / sk[i]);
    }
    d0 = sqrt(d0 / n); d1 = sqrt(d1 / n);
    double dtmax = tspan_len;
    double dt0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * (d0 / d1);
    dt0 = fmin(dt0, dtmax);
    double u1[ORC_MAXN] = {0}, f1[ORC_MAXN] = {0};
    for (int i = 0; i < n; ++i) u1[i] = u0[i] + dt0 * f0[i];
    orc_rhs(pb, th, u1, f1);
    double d2 = 0;
    for (int i = 0; i < n; ++i) { double e = (f1[i] - f0[i]) / sk[i]; d2 += e * e; }
    d2 = sqrt(d2 / n) / dt0;
    double dm = fmax(d1, d2);
    double dt1 = (dm <= 1e-15) ? fmax(1e-6, dt0 * 1e-3) : pow(10.0, -(2.0 + log10(dm)) / (double)order);
    return fmax(pb->dtmin, fmin(fmin(100.0 * dt0, dt1), dtmax));
}

/* The same initial step when the state carries partials (errnorm_sens != 0, P directions): OrdinaryDiffEq promotes u0
   to Dual numbers with zero partials, f0 = f(u0, p) and f1 = f(u0 + dt0 f0, p) carry the partials of p, and every
   `internalnorm` of ode_determine_initdt is the dual-inclusive one -- d1 and d2 grow by the partials, d0 does not (but
   shares the divisor).  df0: n x P tangents of f0.  [UNVERIFIED-DEP] like the norm itself. */
static double init_dt_sens(const orc_problem *pb, const double *th, const double *dth, int P, const double *u0,
                           const double *f0, const double *df0, double tspan_len, int order) {
    const int n = N_(pb), nth = orc_n_theta(pb);
    const double div = pb->errnorm_sens == 2 ? (double)n * (1.0 + (double)P) : (double)n;
    double sk[ORC_MAXN], d0 = 0, d1 = 0;
    for (int i = 0; i < n; ++i) {
        sk[i] = pb->atol[i] + fabs(u0[i]) * pb->rtol[i];
        d0 += (u0[i] / sk[i]) * (u0[i] / sk[i]);
        d1 += (f0[i] / sk[i]) * (f0[i] / sk[i]);
        for (int k = 0; k < P; ++k) { double e = df0[i + (size_t)n * k] / sk[i]; d1 += e * e; }
    }
    d0 = sqrt(d0 / div); d1 = sqrt(d1 / div);
    const double dtmax = tspan_len;
    double dt0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * (d0 / d1);
    dt0 = fmin(dt0, dtmax);
    double u1[ORC_MAXN] = {0}, f1[ORC_MAXN] = {0}, s1[ORC_MAXN], df1[ORC_MAXN];
    for (int i = 0; i < n; ++i) u1[i] = u0[i] + dt0 * f0[i];
    orc_rhs(pb, th, u1, f1);
    double d2 = 0;
    for (int i = 0; i < n; ++i) { double e = (f1[i] - f0[i]) / sk[i]; d2 += e * e; }
    for (int k = 0; k < P; ++k) {
        for (int i = 0; i < n; ++i) s1[i] = dt0 * df0[i + (size_t)n * k];
        orc_rhs_jvp(pb, th, dth + (size_t)nth * k, u1, s1, df1);
        for (int i = 0; i < n; ++i) { double e = (df1[i] - df0[i + (size_t)n * k]) / sk[i]; d2 += e * e; }
    }
    d2 = sqrt(d2 / div) / dt0;
    const double dm = fmax(d1, d2);
    const double dt1 = (dm <= 1e-15) ? fmax(1e-6, dt0 * 1e-3) : pow(10.0, -(2.0 + log10(dm)) / (double)order);
    return fmax(pb->dtmin, fmin(fmin(100.0 * dt0, dt1), dtmax));
}

/* ------------------------------------------------------------------------ */
/* One trajectory: adaptive Rosenbrock23 + forward tangents + loss.         */
/*   u0[n], tsave[nsave] (ascending, tsave[nsave-1] = end of tspan),        */
/*   data[j*n_obs + i], dth: n_theta x P (col-major) or NULL,               */
/*   pred (n x nsave, col-major) or NULL, dpred (n x nsave x P) or NULL.    */
/* retcode: 0 ok, 1 maxiters, 2 dt<dtmin, 3 non-finite.                     */
/* ------------------------------------------------------------------------ */
typedef struct orc_stats { int64_t naccept, nreject; } orc_stats;

/* step tape of the adjoint path: (t, dt, u[n]) of every accepted step, appended by solve_one_ws when armed */
typedef struct orc_tape { double *rec; size_t n_rec, cap; int width; double t_end; } orc_tape;
static _Thread_local orc_tape *g_tape = NULL;
static void tape_push(orc_tape *tp, double t, double dt, const double *u, int n) {
    if (tp->n_rec == tp->cap) {
        tp->cap = tp->cap ? 2 * tp->cap : 64;
        tp->rec = (double *)realloc(tp->rec, sizeof(double) * tp->cap * (size_t)tp->width);
    }
    double *r = tp->rec + tp->n_rec * (size_t)tp->width;
    r[0] = t; r[1] = dt;
    memcpy(r + 2, u, sizeof(double) * (size_t)n);
    tp->n_rec++;
}

/* workspace: n*P*7 + P doubles (zeroed here) */
static int solve_one_ws(const orc_problem *pb, const double *th, const double *dth, int P,
                        const double *u0, const double *tsave, int nsave,
                        const double *data, double *pred, double *dpred,
                        double *loss_out, double *grad /* [P] accumulated += */,
                        int32_t *n_saved_out, orc_stats *st, double *ws) {
    const int n = N_(pb), nobs = pb->n_obs;
    const int nth = orc_n_theta(pb);
    const double d = 1.0 / (2.0 + sqrt(2.0));
    const double c32 = 6.0 + sqrt(2.0);
    const double tend = tsave[nsave - 1];
    double t = pb->t0;
    double u[ORC_MAXN], f0[ORC_MAXN];
    double *S = NULL, *dk1 = NULL, *dk2 = NULL, *dk3 = NULL, *Snew = NULL, *gtr = NULL, *df0 = NULL, *df2 = NULL;
    if (P > 0) {
        memset(ws, 0, sizeof(double) * ((size_t)n * P * 7 + P));
        S = ws;
        Snew = S + (size_t)n * P; dk1 = Snew + (size_t)n * P; dk2 = dk1 + (size_t)n * P;
        dk3 = dk2 + (size_t)n * P; df0 = dk3 + (size_t)n * P; df2 = df0 + (size_t)n * P;
        gtr = ws + (size_t)n * P * 7;
    }
    memcpy(u, u0, sizeof(double) * n);
    orc_rhs(pb, th, u, f0);
    for (int k = 0; k < P; ++k) {
        double zero[ORC_MAXN] = {0};
        orc_rhs_jvp(pb, th, dth + (size_t)nth * k, u, zero, df0 + (size_t)n * k);
    }
    double dt = (pb->errnorm_sens && P > 0) ? init_dt_sens(pb, th, dth, P, u, f0, df0, tend - pb->t0, 2)
                                           : init_dt(pb, th, u, f0, tend - pb->t0, 2);
    double qold = pb->qoldinit;
    int jsave = 0, retcode = 0, iter = 0;
    double loss_sum = 0.0;

    /* save_start: OrdinaryDiffEq stores u0 when tspan[1] is in saveat. */
#define SAVE_POINT(uvec, svec_expr_block)                                              \
    do {                                                                               \
        for (int i = 0; i < n; ++i) {                                                  \
            double v = (uvec)[i];                                                      \
            if (pb->clamp_pred) v = clampd(v, -pb->ub, pb->ub);                        \
            if (pred) pred[i + n * jsave] = v;                                         \
        }                                                                              \
        for (int io = 0; io < nobs; ++io) {                                            \
            int i = pb->i_obs[io];                                                     \
            double v = (uvec)[i];                                                      \
            double mask = 1.0;                                                         \
            if (pb->clamp_pred) { mask = dclamp(v, -pb->ub, pb->ub); v = clampd(v, -pb->ub, pb->ub); } \
            /* mae(data./yscale, pred./yscale) = mean |data - pred| / yscale */        \
            double r_ = (data[io + nobs * jsave] - v) / pb->yscale[io];                \
            double w_;                                                                 \
            if (pb->loss_kind == 0) { loss_sum += fabs(r_); w_ = -dabs_(r_); }         \
            else { loss_sum += r_ * r_; w_ = -2.0 * r_; }                              \
            w_ *= mask / pb->yscale[io];                                               \
            for (int k = 0; k < P; ++k) { double sv; svec_expr_block; gtr[k] += w_ * sv; \
                if (dpred) dpred[i + n * (jsave + (size_t)nsave * k)] = mask * sv; }   \
        }                                                                              \
        ++jsave;                                                                       \
    } while (0)

    if (nsave > 0 && tsave[0] == pb->t0) SAVE_POINT(u, sv = 0.0);

    double J[ORC_MAXN * ORC_MAXN], W[ORC_MAXN * ORC_MAXN], dJ[ORC_MAXN * ORC_MAXN];
    int piv[ORC_MAXN];
    while (jsave < nsave) {
        if (++iter > pb->maxiters) { retcode = 1; break; }
        int last = 0;
        if (t + dt * (1.0 + 1e-13) >= tend) { dt = tend - t; last = 1; }
        if (!(dt > pb->dtmin) || t + dt == t) { retcode = 2; break; }
        const double gam = d * dt;
        if (pb->jac_fd) orc_jac_fd(pb, th, u, f0, J); else orc_jac(pb, th, u, J);
        for (int c = 0; c < n; ++c) for (int i = 0; i < n; ++i) W[i + n * c] = (i == c ? 1.0 : 0.0) - gam * J[i + n * c];
        if (lu_factor(n, W, piv) != 0) { retcode = 3; break; }
        double k1[ORC_MAXN], k2[ORC_MAXN], k3[ORC_MAXN], u1[ORC_MAXN], f1[ORC_MAXN], unew[ORC_MAXN], f2[ORC_MAXN], tmp[ORC_MAXN];
        memcpy(k1, f0, sizeof(double) * n); lu_solve(n, W, piv, k1);
        for (int i = 0; i < n; ++i) u1[i] = u[i] + 0.5 * dt * k1[i];
        orc_rhs(pb, th, u1, f1);
        for (int i = 0; i < n; ++i) tmp[i] = f1[i] - k1[i];
        lu_solve(n, W, piv, tmp);
        for (int i = 0; i < n; ++i) { k2[i] = tmp[i] + k1[i]; unew[i] = u[i] + dt * k2[i]; }
        orc_rhs(pb, th, unew, f2);
        for (int i = 0; i < n; ++i) k3[i] = f2[i] - c32 * (k2[i] - f1[i]) - 2.0 * (k1[i] - f0[i]);
        lu_solve(n, W, piv, k3);
        double ev[ORC_MAXN];
        for (int i = 0; i < n; ++i) ev[i] = dt / 6.0 * (k1[i] - 2.0 * k2[i] + k3[i]);
        int finite = 1;
        for (int i = 0; i < n; ++i) if (!isfinite(unew[i]) || !isfinite(ev[i])) finite = 0;
        if (!finite) { retcode = 3; break; }
        /* Primal-only norm: tangents are evaluated lazily, for accepted steps
           only.  ForwardDiff-style norm: they are needed before the test. */
        const int sens_norm = (pb->errnorm_sens && P > 0);
        double EEst = rms_scaled(pb, n, ev, u, unew);
        int accept = (EEst <= 1.0);
        if (P > 0 && (sens_norm || accept)) {
            for (int k = 0; k < P; ++k) {
                const double *dthk = dth + (size_t)nth * k;
                double *s = S + (size_t)n * k, *a1 = dk1 + (size_t)n * k, *a2 = dk2 + (size_t)n * k, *a3 = dk3 + (size_t)n * k;
                double *sn = Snew + (size_t)n * k, *b0 = df0 + (size_t)n * k, *b2 = df2 + (size_t)n * k;
                if (pb->jac_fd) orc_jac_fd_dir(pb, th, dthk, u, s, b0, J, dJ); else orc_jac_dir(pb, th, dthk, u, s, dJ);
                /* W k1' = f0' + gam * dJ k1 */
                matvec(n, dJ, k1, tmp);
                for (int i = 0; i < n; ++i) a1[i] = b0[i] + gam * tmp[i];
                lu_solve(n, W, piv, a1);
                double s1[ORC_MAXN], df1[ORC_MAXN], dd[ORC_MAXN], kd[ORC_MAXN];
                for (int i = 0; i < n; ++i) s1[i] = s[i] + 0.5 * dt * a1[i];
                orc_rhs_jvp(pb, th, dthk, u1, s1, df1);
                /* W (k2-k1)' = f1' - k1' + gam * dJ (k2-k1) */
                for (int i = 0; i < n; ++i) kd[i] = k2[i] - k1[i];
                matvec(n, dJ, kd, tmp);
                for (int i = 0; i < n; ++i) dd[i] = df1[i] - a1[i] + gam * tmp[i];
                lu_solve(n, W, piv, dd);
                for (int i = 0; i < n; ++i) { a2[i] = a1[i] + dd[i]; sn[i] = s[i] + dt * a2[i]; }
                orc_rhs_jvp(pb, th, dthk, unew, sn, b2);
                if (sens_norm) {
                    /* W k3' = f2' - c32 (k2' - f1') - 2 (k1' - f0') + gam * dJ k3 */
                    matvec(n, dJ, k3, tmp);
                    for (int i = 0; i < n; ++i)
                        a3[i] = b2[i] - c32 * (a2[i] - df1[i]) - 2.0 * (a1[i] - b0[i]) + gam * tmp[i];
                    lu_solve(n, W, piv, a3);
                }
            }
            if (sens_norm) {
                /* [UNVERIFIED-DEP] DiffEqBase norm on Dual arrays: sum of squares of
                   value and partials, divided by the number of components; the
                   per-component scale uses the 2-norm of (value, partials). */
                double ssum = 0.0;
                for (int i = 0; i < n; ++i) {
                    double na = u[i] * u[i], nb = unew[i] * unew[i], ee = ev[i] * ev[i];
                    for (int k = 0; k < P; ++k) {
                        double s_ = S[i + (size_t)n * k], sn_ = Snew[i + (size_t)n * k];
                        na += s_ * s_; nb += sn_ * sn_;
                        double de = dt / 6.0 * (dk1[i + (size_t)n * k] - 2.0 * dk2[i + (size_t)n * k] + dk3[i + (size_t)n * k]);
                        ee += de * de;
                    }
                    double sc = pb->atol[i] + pb->rtol[i] * sqrt(fmax(na, nb));
                    ssum += ee / (sc * sc);
                }
                /* errnorm_sens 1: sqrt(sum(sse, u) / length(u)) (early DiffEqBase 6); 2: / totallength(u) = n (1 + partials per Dual)
                   (the cathode Manifest pins 6.189, which has it; case1, case2, robertson name no versions, README.md:15-21, but the case2
                   checkpoint's recorded history is reproduced with 2 and not with 1: tests/test_case2_stream_pin.py).  P counts every
                   partial of the Dual: a caller passes the zero-padded directions of ForwardDiff's last chunk too. */
                EEst = sqrt(ssum / (pb->errnorm_sens == 2 ? (double)n * (1.0 + (double)P) : (double)n));
                if (!isfinite(EEst)) { retcode = 3; break; }
                accept = (EEst <= 1.0);
            }
        }
        /* PI controller (OrdinaryDiffEq PIController) */
        double q, q11 = 0.0;
        if (EEst == 0.0) q = 1.0 / pb->qmax;
        else {
            q11 = pow(EEst, pb->beta1);
            q = q11 / pow(qold, pb->beta2);
            q = fmax(1.0 / pb->qmax, fmin(1.0 / pb->qmin, q / pb->gamma));
        }
        if (accept) {
            if (st) st->naccept++;
            if (q >= pb->qsteady_min && q <= pb->qsteady_max) q = 1.0;
            qold = fmax(EEst, pb->qoldinit);
            double tnew = last ? tend : t + dt;
            if (g_tape) { tape_push(g_tape, t, dt, u, n); g_tape->t_end = tnew; }
            /* saveat via the Rosenbrock23 dense output:
               u(t+Theta dt) = u + dt (c1 k1 + c2 k2), c1 = Th(1-Th)/(1-2d), c2 = Th(Th-2d)/(1-2d) */
            while (jsave < nsave && tsave[jsave] <= tnew) {
                double ts = tsave[jsave];
                if (ts == tnew) {
                    SAVE_POINT(unew, sv = Snew[i + (size_t)n * k]);
                } else {
                    double Th = (ts - t) / dt;
                    double c1 = Th * (1.0 - Th) / (1.0 - 2.0 * d), c2 = Th * (Th - 2.0 * d) / (1.0 - 2.0 * d);
                    double ui[ORC_MAXN];
                    for (int i = 0; i < n; ++i) ui[i] = u[i] + dt * (c1 * k1[i] + c2 * k2[i]);
                    SAVE_POINT(ui, sv = S[i + (size_t)n * k] + dt * (c1 * dk1[i + (size_t)n * k] + c2 * dk2[i + (size_t)n * k]));
                }
            }
            memcpy(u, unew, sizeof(double) * n);
            memcpy(f0, f2, sizeof(double) * n);
            if (P > 0) {
                memcpy(S, Snew, sizeof(double) * (size_t)n * P);
                memcpy(df0, df2, sizeof(double) * (size_t)n * P);
            }
            t = tnew;
            dt = dt / q;
            double dtmax = tend - pb->t0;
            if (dt > dtmax) dt = dtmax;
        } else {
            if (st) st->nreject++;
            dt = dt / fmin(1.0 / pb->qmin, q11 / pb->gamma);
        }
    }
#undef SAVE_POINT
    /* mae / mse over the saved prefix (robertson/rober_crnn.jl:141: data[:, 1:size(pred)[2]]) */
    double denom = (double)nobs * (double)jsave;
    double loss = jsave > 0 ? loss_sum / denom : 0.0;
    if (loss_out) *loss_out = loss;
    if (n_saved_out) *n_saved_out = jsave;
    if (grad && jsave > 0) for (int k = 0; k < P; ++k) grad[k] += gtr[k] / denom;
    return retcode;
}


/* ------------------------------------------------------------------------ */
/* Tsit5 (Tsitouras 2011) as OrdinaryDiffEq implements it: 7 stages, FSAL,  */
/* embedded 4th-order error estimate, "free" 4th-order dense output.        */
/* The tableau below was written from memory of Tsit5ConstantCache and is   */
/* verified by tests/test_oracle_golden.py against all 17 order conditions  */
/* up to order 5, the embedded-pair conditions and the continuous order     */
/* conditions of the interpolant.  Reference call site: case1/case1.jl:28,  */
/* 94-95 (alg = Tsit5(), maxiters = 10000).                                 */
/* ------------------------------------------------------------------------ */
static const double TS_C[7] = {0.0, 0.161, 0.327, 0.9, 0.9800255409045097, 1.0, 1.0};
static const double TS_A[7][6] = {
    {0, 0, 0, 0, 0, 0},
    {0.161, 0, 0, 0, 0, 0},
    {-0.008480655492356989, 0.335480655492357, 0, 0, 0, 0},
    {2.8971530571054935, -6.359448489975075, 4.3622954328695815, 0, 0, 0},
    {5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525, 0, 0},
    {5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383, 0},
    {0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774}};
static const double TS_BT[7] = {-0.00178001105222577714, -0.0008164344596567469, 0.007880878010261995,
                                -0.1447110071732629, 0.5823571654525552, -0.45808210592918697, 0.015151515151515152};
void orc_tsit5_tableau(double *c, double *a /*7x6 row-major*/, double *bt) {
    memcpy(c, TS_C, sizeof(TS_C)); memcpy(a, TS_A, sizeof(TS_A)); memcpy(bt, TS_BT, sizeof(TS_BT));
}
void orc_tsit5_dense(double T, double *b) {
    b[0] = -1.0530884977290216 * T * (T - 1.3299890189751412) * (T * T - 1.4364028541716351 * T + 0.7139816917074209);
    b[1] = 0.1017 * T * T * (T * T - 2.1966568338249754 * T + 1.2949852507374631);
    b[2] = 2.490627285651252793 * T * T * (T * T - 2.38535645472061657 * T + 1.57803468208092486);
    b[3] = -16.54810288924490272 * (T - 1.21712927295533244) * (T - 0.61620406037800089) * T * T;
    b[4] = 47.37952196281928122 * (T - 1.203071208372362603) * (T - 0.658047292653547382) * T * T;
    b[5] = -34.87065786149660974 * (T - 1.2) * (T - 0.666666666666666667) * T * T;
    b[6] = 2.5 * (T - 1.0) * (T - 0.6) * T * T;
}

/* workspace: n*P*9 + P doubles */
static int solve_one_tsit5(const orc_problem *pb, const double *th, const double *dth, int P,
                           const double *u0, const double *tsave, int nsave,
                           const double *data, double *pred, double *dpred,
                           double *loss_out, double *grad, int32_t *n_saved_out, orc_stats *st, double *ws) {
    const int n = N_(pb), nobs = pb->n_obs;
    const int nth = orc_n_theta(pb);
    const double tend = tsave[nsave - 1];
    double t = pb->t0;
    double u[ORC_MAXN], k[7][ORC_MAXN];
    double *S = NULL, *Snew = NULL, *dk = NULL, *gtr = NULL;   /* dk: 7 blocks of n*P */
    if (P > 0) {
        memset(ws, 0, sizeof(double) * ((size_t)n * P * 9 + P));
        S = ws; Snew = S + (size_t)n * P; dk = Snew + (size_t)n * P; gtr = ws + (size_t)n * P * 9;
    }
#define DK(i) (dk + (size_t)(i) * n * P)
    memcpy(u, u0, sizeof(double) * n);
    orc_rhs(pb, th, u, k[0]);
    for (int c = 0; c < P; ++c) {
        double zero[ORC_MAXN] = {0};
        orc_rhs_jvp(pb, th, dth + (size_t)nth * c, u, zero, DK(0) + (size_t)n * c);
    }
    double dt = (pb->errnorm_sens && P > 0) ? init_dt_sens(pb, th, dth, P, u, k[0], DK(0), tend - pb->t0, 5)
                                           : init_dt(pb, th, u, k[0], tend - pb->t0, 5);
    double qold = pb->qoldinit;
    int jsave = 0, retcode = 0, iter = 0;
    double loss_sum = 0.0;
#define SAVE_POINT(uvec, svec_expr_block)                                              \
    do {                                                                               \
        for (int i = 0; i < n; ++i) {                                                  \
            double v = (uvec)[i];                                                      \
            if (pb->clamp_pred) v = clampd(v, -pb->ub, pb->ub);                        \
            if (pred) pred[i + n * jsave] = v;                                         \
        }                                                                              \
        for (int io = 0; io < nobs; ++io) {                                            \
            int i = pb->i_obs[io];                                                     \
            double v = (uvec)[i];                                                      \
            double mask = 1.0;                                                         \
            if (pb->clamp_pred) { mask = dclamp(v, -pb->ub, pb->ub); v = clampd(v, -pb->ub, pb->ub); } \
            double r_ = (data[io + nobs * jsave] - v) / pb->yscale[io];                \
            double w_;                                                                 \
            if (pb->loss_kind == 0) { loss_sum += fabs(r_); w_ = -dabs_(r_); }         \
            else { loss_sum += r_ * r_; w_ = -2.0 * r_; }                              \
            w_ *= mask / pb->yscale[io];                                               \
            for (int c = 0; c < P; ++c) { double sv; svec_expr_block; gtr[c] += w_ * sv; \
                if (dpred) dpred[i + n * (jsave + (size_t)nsave * c)] = mask * sv; }   \
        }                                                                              \
        ++jsave;                                                                       \
    } while (0)
    if (nsave > 0 && tsave[0] == pb->t0) SAVE_POINT(u, sv = 0.0);

    while (jsave < nsave) {
        if (++iter > pb->maxiters) { retcode = 1; break; }
        int last = 0;
        if (t + dt * (1.0 + 1e-13) >= tend) { dt = tend - t; last = 1; }
        if (!(dt > pb->dtmin) || t + dt == t) { retcode = 2; break; }
        double g[ORC_MAXN], unew[ORC_MAXN];
        for (int s_ = 1; s_ < 7; ++s_) {          /* stages 2..7 (stage 7 is evaluated at u_{n+1}: FSAL) */
            for (int i = 0; i < n; ++i) {
                double a = 0.0;
                for (int j = 0; j < s_; ++j) a += TS_A[s_][j] * k[j][i];
                g[i] = u[i] + dt * a;
            }
            if (s_ == 6) memcpy(unew, g, sizeof(double) * n);
            orc_rhs(pb, th, g, k[s_]);
        }
        double ev[ORC_MAXN];
        int finite = 1;
        for (int i = 0; i < n; ++i) {
            double a = 0.0;
            for (int j = 0; j < 7; ++j) a += TS_BT[j] * k[j][i];
            ev[i] = dt * a;
            if (!isfinite(unew[i]) || !isfinite(ev[i])) finite = 0;
        }
        if (!finite) { retcode = 3; break; }
        double EEst = rms_scaled(pb, n, ev, u, unew);
        int accept = (EEst <= 1.0);
        /* Primal-only norm: tangents lazily, for accepted steps only.  ForwardDiff-style norm (errnorm_sens = 1):
           they are needed before the test, for every attempt. */
        const int sens_norm = (pb->errnorm_sens && P > 0);
        if ((accept || sens_norm) && P > 0) {
            for (int c = 0; c < P; ++c) {
                const double *dthc = dth + (size_t)nth * c;
                const double *s = S + (size_t)n * c;
                double gs[ORC_MAXN], gu[ORC_MAXN];
                for (int s_ = 1; s_ < 7; ++s_) {
                    for (int i = 0; i < n; ++i) {
                        double a = 0.0, b = 0.0;
                        for (int j = 0; j < s_; ++j) { a += TS_A[s_][j] * k[j][i]; b += TS_A[s_][j] * DK(j)[i + (size_t)n * c]; }
                        gu[i] = u[i] + dt * a; gs[i] = s[i] + dt * b;
                    }
                    if (s_ == 6) memcpy(Snew + (size_t)n * c, gs, sizeof(double) * n);
                    orc_rhs_jvp(pb, th, dthc, gu, gs, DK(s_) + (size_t)n * c);
                }
            }
            if (sens_norm) {
                /* [UNVERIFIED-DEP] DiffEqBase norm on Dual arrays, as in the Rosenbrock23 stepper above: value and
                   partials enter the sum of squares, the per-component scale uses the 2-norm of (value, partials);
                   the error estimate's partials are dt sum_j btilde_j k_j'. */
                double ssum = 0.0;
                for (int i = 0; i < n; ++i) {
                    double na = u[i] * u[i], nb = unew[i] * unew[i], ee = ev[i] * ev[i];
                    for (int c = 0; c < P; ++c) {
                        double s_ = S[i + (size_t)n * c], sn_ = Snew[i + (size_t)n * c], a = 0.0;
                        for (int j = 0; j < 7; ++j) a += TS_BT[j] * DK(j)[i + (size_t)n * c];
                        na += s_ * s_; nb += sn_ * sn_; ee += (dt * a) * (dt * a);
                    }
                    double sc = pb->atol[i] + pb->rtol[i] * sqrt(fmax(na, nb));
                    ssum += ee / (sc * sc);
                }
                /* errnorm_sens 1 / 2: as in the Rosenbrock23 stepper above (the recorded case2 history selects 2) */
                EEst = sqrt(ssum / (pb->errnorm_sens == 2 ? (double)n * (1.0 + (double)P) : (double)n));
                if (!isfinite(EEst)) { retcode = 3; break; }
                accept = (EEst <= 1.0);
            }
        }
        double q, q11 = 0.0;
        if (EEst == 0.0) q = 1.0 / pb->qmax;
        else {
            q11 = pow(EEst, pb->beta1);
            q = q11 / pow(qold, pb->beta2);
            q = fmax(1.0 / pb->qmax, fmin(1.0 / pb->qmin, q / pb->gamma));
        }
        if (accept) {
            if (st) st->naccept++;
            if (q >= pb->qsteady_min && q <= pb->qsteady_max) q = 1.0;
            qold = fmax(EEst, pb->qoldinit);
            double tnew = last ? tend : t + dt;
            while (jsave < nsave && tsave[jsave] <= tnew) {
                double ts = tsave[jsave];
                if (ts == tnew) {
                    SAVE_POINT(unew, sv = Snew[i + (size_t)n * c]);
                } else {
                    double Th = (ts - t) / dt, bth[7], ui[ORC_MAXN];
                    orc_tsit5_dense(Th, bth);
                    for (int i = 0; i < n; ++i) {
                        double a = 0.0;
                        for (int j = 0; j < 7; ++j) a += bth[j] * k[j][i];
                        ui[i] = u[i] + dt * a;
                    }
                    SAVE_POINT(ui, { double a_ = 0.0; for (int j = 0; j < 7; ++j) a_ += bth[j] * DK(j)[i + (size_t)n * c];
                                     sv = S[i + (size_t)n * c] + dt * a_; });
                }
            }
            memcpy(u, unew, sizeof(double) * n);
            memcpy(k[0], k[6], sizeof(double) * n);
            if (P > 0) {
                memcpy(S, Snew, sizeof(double) * (size_t)n * P);
                memcpy(DK(0), DK(6), sizeof(double) * (size_t)n * P);
            }
            t = tnew;
            dt = dt / q;
            double dtmax = tend - pb->t0;
            if (dt > dtmax) dt = dtmax;
        } else {
            if (st) st->nreject++;
            dt = dt / fmin(1.0 / pb->qmin, q11 / pb->gamma);
        }
    }
#undef SAVE_POINT
#undef DK
    double denom = (double)nobs * (double)jsave;
    double loss = jsave > 0 ? loss_sum / denom : 0.0;
    if (loss_out) *loss_out = loss;
    if (n_saved_out) *n_saved_out = jsave;
    if (grad && jsave > 0) for (int c = 0; c < P; ++c) grad[c] += gtr[c] / denom;
    return retcode;
}

/* ------------------------------------------------------------------------ */
/* AutoTsit5(Rosenbrock23()): OrdinaryDiffEq's stiffness-switching composite */
/* (case2/case2.jl:26, HyChem/crnn_pyrolysis_mass.jl:29).  The package is   */
/* not in the reference tree; restated from its published algorithm         */
/* (AutoSwitch, composite_algs.jl) [UNVERIFIED-DEP]:                        */
/*   * start on Tsit5; before every attempt after the first, test           */
/*       stiffness = |eigen_est * dt| / 3.5068  >  9/10                      */
/*     with dt the step size about to be tried and eigen_est from the last  */
/*     attempt (accepted or not); count consecutive positives (+) and       */
/*     negatives (-);                                                       */
/*   * on Tsit5, more than 10 positives in a row: dt *= 2, switch to        */
/*     Rosenbrock23; on Rosenbrock23, more than 3 negatives in a row:       */
/*     dt /= 2, switch back;                                                */
/*   * eigen_est: Tsit5 max_i |k7_i - k6_i| / |g7_i - g6_i| (Hairer II p.22 */
/*     in the Inf norm; a component that does not move gives 0/0 = NaN,     */
/*     which Julia's `maximum` propagates, and NaN > 9/10 is false -- so a  */
/*     state vector with a constant component, like case2's temperature,    */
/*     never switches); Rosenbrock23: opnorm(J(u_n), Inf);                  */
/*   * the PI exponents follow the running algorithm (beta2 = 2/(5 order),  */
/*     beta1 = 7/(10 order)); qold, gamma, qmin, qmax and the steady band   */
/*     of pb are shared.                                                    */
/* st_alg (optional): [0] Tsit5 accepted steps, [1] Rosenbrock23 accepted   */
/* steps, [2] number of switches.                                           */
/* ------------------------------------------------------------------------ */
#define AS_MAXSTIFF 10
#define AS_MAXNONSTIFF 3
#define AS_TOL 0.9
#define AS_DTFAC 2.0
#define AS_STAB 3.5068
static int solve_one_auto(const orc_problem *pb, const double *th, const double *dth, int P,
                          const double *u0, const double *tsave, int nsave,
                          const double *data, double *pred, double *dpred,
                          double *loss_out, double *grad, int32_t *n_saved_out, orc_stats *st, double *ws,
                          int64_t *st_alg) {
    const int n = N_(pb), nobs = pb->n_obs;
    const int nth = orc_n_theta(pb);
    const double d = 1.0 / (2.0 + sqrt(2.0));
    const double c32 = 6.0 + sqrt(2.0);
    const double tend = tsave[nsave - 1];
    double t = pb->t0;
    double u[ORC_MAXN], k[7][ORC_MAXN];
    double *S = NULL, *Snew = NULL, *dk = NULL, *gtr = NULL;
    if (P > 0) {
        memset(ws, 0, sizeof(double) * ((size_t)n * P * 9 + P));
        S = ws; Snew = S + (size_t)n * P; dk = Snew + (size_t)n * P; gtr = ws + (size_t)n * P * 9;
    }
#define DK(i) (dk + (size_t)(i) * n * P)
    memcpy(u, u0, sizeof(double) * n);
    orc_rhs(pb, th, u, k[0]);
    for (int c = 0; c < P; ++c) {
        double zero[ORC_MAXN] = {0};
        orc_rhs_jvp(pb, th, dth + (size_t)nth * c, u, zero, DK(0) + (size_t)n * c);
    }
    double dt = init_dt(pb, th, u, k[0], tend - pb->t0, 5);
    double qold = pb->qoldinit;
    int jsave = 0, retcode = 0, iter = 0;
    int alg = 0, cnt = 0, have_est = 0;
    double eigen_est = 0.0;
    double loss_sum = 0.0;
#define SAVE_POINT(uvec, svec_expr_block)                                              \
    do {                                                                               \
        for (int i = 0; i < n; ++i) {                                                  \
            double v = (uvec)[i];                                                      \
            if (pb->clamp_pred) v = clampd(v, -pb->ub, pb->ub);                        \
            if (pred) pred[i + n * jsave] = v;                                         \
        }                                                                              \
        for (int io = 0; io < nobs; ++io) {                                            \
            int i = pb->i_obs[io];                                                     \
            double v = (uvec)[i];                                                      \
            double mask = 1.0;                                                         \
            if (pb->clamp_pred) { mask = dclamp(v, -pb->ub, pb->ub); v = clampd(v, -pb->ub, pb->ub); } \
            double r_ = (data[io + nobs * jsave] - v) / pb->yscale[io];                \
            double w_;                                                                 \
            if (pb->loss_kind == 0) { loss_sum += fabs(r_); w_ = -dabs_(r_); }         \
            else { loss_sum += r_ * r_; w_ = -2.0 * r_; }                              \
            w_ *= mask / pb->yscale[io];                                               \
            for (int c = 0; c < P; ++c) { double sv; svec_expr_block; gtr[c] += w_ * sv; \
                if (dpred) dpred[i + n * (jsave + (size_t)nsave * c)] = mask * sv; }   \
        }                                                                              \
        ++jsave;                                                                       \
    } while (0)
    if (nsave > 0 && tsave[0] == pb->t0) SAVE_POINT(u, sv = 0.0);

    double J[ORC_MAXN * ORC_MAXN], W[ORC_MAXN * ORC_MAXN], dJ[ORC_MAXN * ORC_MAXN];
    int piv[ORC_MAXN];
    while (jsave < nsave) {
        if (++iter > pb->maxiters) { retcode = 1; break; }
        /* choose_algorithm! (loopheader!) */
        if (have_est) {
            const double stiffness = fabs(eigen_est * dt / AS_STAB);
            const int stiff = stiffness > AS_TOL;      /* false for NaN */
            cnt = stiff ? (cnt < 0 ? 1 : cnt + 1) : (cnt > 0 ? -1 : cnt - 1);
            if (alg == 0 && cnt > AS_MAXSTIFF) { dt *= AS_DTFAC; alg = 1; if (st_alg) st_alg[2]++; }
            else if (alg == 1 && cnt < -AS_MAXNONSTIFF) { dt /= AS_DTFAC; alg = 0; if (st_alg) st_alg[2]++; }
        }
        const double beta1 = alg == 0 ? 7.0 / 50.0 : 7.0 / 20.0, beta2 = alg == 0 ? 2.0 / 25.0 : 2.0 / 10.0;
        int last = 0;
        if (t + dt * (1.0 + 1e-13) >= tend) { dt = tend - t; last = 1; }
        if (!(dt > pb->dtmin) || t + dt == t) { retcode = 2; break; }
        double unew[ORC_MAXN], ev[ORC_MAXN];
        double u1[ORC_MAXN], f1[ORC_MAXN], k2[ORC_MAXN], k3[ORC_MAXN], tmp[ORC_MAXN];   /* Rosenbrock23 stages (k1 in k[1]) */
        const double gam = d * dt;
        if (alg == 0) {
            double g[ORC_MAXN], g6[ORC_MAXN];
            for (int s_ = 1; s_ < 7; ++s_) {
                for (int i = 0; i < n; ++i) {
                    double a = 0.0;
                    for (int j = 0; j < s_; ++j) a += TS_A[s_][j] * k[j][i];
                    g[i] = u[i] + dt * a;
                }
                if (s_ == 5) memcpy(g6, g, sizeof(double) * n);
                if (s_ == 6) memcpy(unew, g, sizeof(double) * n);
                orc_rhs(pb, th, g, k[s_]);
            }
            for (int i = 0; i < n; ++i) {
                double a = 0.0;
                for (int j = 0; j < 7; ++j) a += TS_BT[j] * k[j][i];
                ev[i] = dt * a;
            }
            double est = 0.0; int isnan_ = 0;
            for (int i = 0; i < n; ++i) {
                double q_ = fabs((k[6][i] - k[5][i]) / (unew[i] - g6[i]));
                if (q_ != q_) isnan_ = 1; else if (q_ > est) est = q_;
            }
            eigen_est = isnan_ ? NAN : est;
        } else {
            orc_jac(pb, th, u, J);
            double est = 0.0;
            for (int i = 0; i < n; ++i) { double a = 0.0; for (int c = 0; c < n; ++c) a += fabs(J[i + n * c]); if (a > est) est = a; }
            eigen_est = est;
            for (int c = 0; c < n; ++c) for (int i = 0; i < n; ++i) W[i + n * c] = (i == c ? 1.0 : 0.0) - gam * J[i + n * c];
            if (lu_factor(n, W, piv) != 0) { retcode = 3; break; }
            memcpy(k[1], k[0], sizeof(double) * n); lu_solve(n, W, piv, k[1]);
            for (int i = 0; i < n; ++i) u1[i] = u[i] + 0.5 * dt * k[1][i];
            orc_rhs(pb, th, u1, f1);
            for (int i = 0; i < n; ++i) tmp[i] = f1[i] - k[1][i];
            lu_solve(n, W, piv, tmp);
            for (int i = 0; i < n; ++i) { k2[i] = tmp[i] + k[1][i]; unew[i] = u[i] + dt * k2[i]; }
            orc_rhs(pb, th, unew, k[6]);
            for (int i = 0; i < n; ++i) k3[i] = k[6][i] - c32 * (k2[i] - f1[i]) - 2.0 * (k[1][i] - k[0][i]);
            lu_solve(n, W, piv, k3);
            for (int i = 0; i < n; ++i) ev[i] = dt / 6.0 * (k[1][i] - 2.0 * k2[i] + k3[i]);
        }
        have_est = 1;
        int finite = 1;
        for (int i = 0; i < n; ++i) if (!isfinite(unew[i]) || !isfinite(ev[i])) finite = 0;
        if (!finite) { retcode = 3; break; }
        const double EEst = rms_scaled(pb, n, ev, u, unew);
        const int accept = (EEst <= 1.0);
        if (accept && P > 0) {
            for (int c = 0; c < P; ++c) {
                const double *dthc = dth + (size_t)nth * c;
                const double *s = S + (size_t)n * c;
                if (alg == 0) {
                    double gs[ORC_MAXN], gu[ORC_MAXN];
                    for (int s_ = 1; s_ < 7; ++s_) {
                        for (int i = 0; i < n; ++i) {
                            double a = 0.0, b = 0.0;
                            for (int j = 0; j < s_; ++j) { a += TS_A[s_][j] * k[j][i]; b += TS_A[s_][j] * DK(j)[i + (size_t)n * c]; }
                            gu[i] = u[i] + dt * a; gs[i] = s[i] + dt * b;
                        }
                        if (s_ == 6) memcpy(Snew + (size_t)n * c, gs, sizeof(double) * n);
                        orc_rhs_jvp(pb, th, dthc, gu, gs, DK(s_) + (size_t)n * c);
                    }
                } else {
                    double *a1 = DK(1) + (size_t)n * c, *a2 = DK(2) + (size_t)n * c, *b0 = DK(0) + (size_t)n * c;
                    double *sn = Snew + (size_t)n * c, *b2 = DK(6) + (size_t)n * c;
                    orc_jac_dir(pb, th, dthc, u, s, dJ);
                    matvec(n, dJ, k[1], tmp);
                    for (int i = 0; i < n; ++i) a1[i] = b0[i] + gam * tmp[i];
                    lu_solve(n, W, piv, a1);
                    double s1[ORC_MAXN], df1[ORC_MAXN], dd[ORC_MAXN], kd[ORC_MAXN];
                    for (int i = 0; i < n; ++i) s1[i] = s[i] + 0.5 * dt * a1[i];
                    orc_rhs_jvp(pb, th, dthc, u1, s1, df1);
                    for (int i = 0; i < n; ++i) kd[i] = k2[i] - k[1][i];
                    matvec(n, dJ, kd, tmp);
                    for (int i = 0; i < n; ++i) dd[i] = df1[i] - a1[i] + gam * tmp[i];
                    lu_solve(n, W, piv, dd);
                    for (int i = 0; i < n; ++i) { a2[i] = a1[i] + dd[i]; sn[i] = s[i] + dt * a2[i]; }
                    orc_rhs_jvp(pb, th, dthc, unew, sn, b2);
                }
            }
        }
        double q, q11 = 0.0;
        if (EEst == 0.0) q = 1.0 / pb->qmax;
        else {
            q11 = pow(EEst, beta1);
            q = q11 / pow(qold, beta2);
            q = fmax(1.0 / pb->qmax, fmin(1.0 / pb->qmin, q / pb->gamma));
        }
        if (accept) {
            if (st) st->naccept++;
            if (st_alg) st_alg[alg]++;
            if (q >= pb->qsteady_min && q <= pb->qsteady_max) q = 1.0;
            qold = fmax(EEst, pb->qoldinit);
            double tnew = last ? tend : t + dt;
            while (jsave < nsave && tsave[jsave] <= tnew) {
                double ts = tsave[jsave];
                if (ts == tnew) {
                    SAVE_POINT(unew, sv = Snew[i + (size_t)n * c]);
                } else if (alg == 0) {
                    double Th = (ts - t) / dt, bth[7], ui[ORC_MAXN];
                    orc_tsit5_dense(Th, bth);
                    for (int i = 0; i < n; ++i) {
                        double a = 0.0;
                        for (int j = 0; j < 7; ++j) a += bth[j] * k[j][i];
                        ui[i] = u[i] + dt * a;
                    }
                    SAVE_POINT(ui, { double a_ = 0.0; for (int j = 0; j < 7; ++j) a_ += bth[j] * DK(j)[i + (size_t)n * c];
                                     sv = S[i + (size_t)n * c] + dt * a_; });
                } else {
                    double Th = (ts - t) / dt;
                    double c1 = Th * (1.0 - Th) / (1.0 - 2.0 * d), c2 = Th * (Th - 2.0 * d) / (1.0 - 2.0 * d);
                    double ui[ORC_MAXN];
                    for (int i = 0; i < n; ++i) ui[i] = u[i] + dt * (c1 * k[1][i] + c2 * k2[i]);
                    SAVE_POINT(ui, sv = S[i + (size_t)n * c] + dt * (c1 * DK(1)[i + (size_t)n * c] + c2 * DK(2)[i + (size_t)n * c]));
                }
            }
            memcpy(u, unew, sizeof(double) * n);
            memcpy(k[0], k[6], sizeof(double) * n);
            if (P > 0) {
                memcpy(S, Snew, sizeof(double) * (size_t)n * P);
                memcpy(DK(0), DK(6), sizeof(double) * (size_t)n * P);
            }
            t = tnew;
            dt = dt / q;
            double dtmax = tend - pb->t0;
            if (dt > dtmax) dt = dtmax;
        } else {
            if (st) st->nreject++;
            dt = dt / fmin(1.0 / pb->qmin, q11 / pb->gamma);
        }
    }
#undef SAVE_POINT
#undef DK
    double denom = (double)nobs * (double)jsave;
    double loss = jsave > 0 ? loss_sum / denom : 0.0;
    if (loss_out) *loss_out = loss;
    if (n_saved_out) *n_saved_out = jsave;
    if (grad && jsave > 0) for (int c = 0; c < P; ++c) grad[c] += gtr[c] / denom;
    return retcode;
}

/* ------------------------------------------------------------------------ */
/* Rosenbrock23 gradient by the DISCRETE ADJOINT of the accepted steps: the  */
/* same derivative as the forward tangents of solve_one_ws (dt, the accept / */
/* reject decisions and the saveat weights held fixed), formed backwards.   */
/* Forward sweep = solve_one_ws without tangents, recording (t, dt, u) per   */
/* accepted step; reverse sweep, per step with lam = d loss / d u_{n+1}:     */
/*   re-form k1 = W^-1 f(u_n), u_mid, dk = W^-1 (f(u_mid) - k1);             */
/*   seeds of the save points inside the step: A, B1, B2 (d/d u_n, k1, k2);  */
/*   kb2 = dt lam + B2, v = W^-T kb2, kb1 = B1 + kb2 - v;                    */
/*   u_mid: gb = f_u(u_mid)^T v, thb += f_theta(u_mid)^T v, kb1 += dt/2 gb;  */
/*   w = W^-T kb1;                                                           */
/*   u_n: lam <- lam + A + gb + d/du [w.f + gam (v.J dk + w.J k1)],          */
/*        thb += d/dtheta [ the same bracket ]                               */
/* with J = D_sc W_out D_r W_in^T D_g, so every contraction is O(n nr).      */
/* This is what crnn_amd/csrc/ros23_adj_kernel.hpp runs on the GPU; here it  */
/* is written with dense n x n matrices and the temperature as a state.      */
/* ------------------------------------------------------------------------ */
static void rates_at(const orc_problem *pb, const double *th, const double *u, double *x, double *g, double *h, double *r, double *f) {
    const int n = N_(pb), ns = pb->ns, nr = pb->nr;
    const double *w_in = W_IN(pb, th), *w_b = W_B(pb, th), *w_out = W_OUT(pb, th);
    feat(pb, u, x, g, h);
    for (int j = 0; j < nr; ++j) {
        double z = w_b[j];
        for (int i = 0; i < n; ++i) z += w_in[i + n * j] * x[i];
        r[j] = exp(z);
    }
    for (int i = 0; i < n; ++i) f[i] = 0.0;
    for (int i = 0; i < ns; ++i) {
        double a = 0.0;
        for (int j = 0; j < nr; ++j) a += w_out[i + ns * j] * r[j];
        f[i] = a * pb->rate_scale[i];
    }
}

static int solve_one_adj(const orc_problem *pb, const double *th, const double *dth, int P,
                         const double *u0, const double *tsave, int nsave, const double *data, double *pred,
                         double *loss_out, double *grad /* [P] accumulated += */, int32_t *n_saved_out, orc_stats *st) {
    const int n = N_(pb), ns = pb->ns, nr = pb->nr, nobs = pb->n_obs, nth = orc_n_theta(pb);
    const double d = 1.0 / (2.0 + sqrt(2.0));
    const double *w_in = W_IN(pb, th), *w_out = W_OUT(pb, th);
    orc_tape tp = {NULL, 0, 0, n + 2, pb->t0};
    int32_t jsave_end = 0;
    double loss = 0.0;
    g_tape = &tp;
    const int rc = solve_one_ws(pb, th, NULL, 0, u0, tsave, nsave, data, pred, NULL, &loss, NULL, &jsave_end, st, NULL);
    g_tape = NULL;
    if (loss_out) *loss_out = loss;
    if (n_saved_out) *n_saved_out = jsave_end;
    if (P > 0 && grad && jsave_end > 0 && tp.n_rec > 0) {
        double thb[ORC_MAXTH];
        for (int m = 0; m < nth; ++m) thb[m] = 0.0;
        double *b_in = thb, *b_b = thb + n * nr, *b_out = thb + (n + 1) * nr;
        double lam[ORC_MAXN] = {0};
        int jsave = jsave_end;
        const int jlo = (nsave > 0 && tsave[0] == pb->t0) ? 1 : 0;   /* the saved initial point carries no gradient */
        double tnew = tp.t_end;
        for (long s = (long)tp.n_rec - 1; s >= 0; --s) {
            const double *rec = tp.rec + (size_t)s * tp.width;
            const double tn = rec[0], h = rec[1], *un = rec + 2;
            const double gam = d * h;
            double x0[ORC_MAXN], g0[ORC_MAXN], h0[ORC_MAXN], r0[ORC_MAXR], f0[ORC_MAXN];
            double x1[ORC_MAXN], g1[ORC_MAXN], h1[ORC_MAXN], r1[ORC_MAXR], f1[ORC_MAXN];
            double J[ORC_MAXN * ORC_MAXN], W[ORC_MAXN * ORC_MAXN], Wt[ORC_MAXN * ORC_MAXN];
            int piv[ORC_MAXN], pivt[ORC_MAXN];
            rates_at(pb, th, un, x0, g0, h0, r0, f0);
            orc_jac(pb, th, un, J);
            for (int c = 0; c < n; ++c) for (int i = 0; i < n; ++i) {
                W[i + n * c] = (i == c ? 1.0 : 0.0) - gam * J[i + n * c];
                Wt[c + n * i] = W[i + n * c];
            }
            if (lu_factor(n, W, piv) != 0 || lu_factor(n, Wt, pivt) != 0) break;
            double k1[ORC_MAXN], dk[ORC_MAXN], u1[ORC_MAXN];
            memcpy(k1, f0, sizeof(double) * n); lu_solve(n, W, piv, k1);
            for (int i = 0; i < n; ++i) u1[i] = un[i] + 0.5 * h * k1[i];
            rates_at(pb, th, u1, x1, g1, h1, r1, f1);
            for (int i = 0; i < n; ++i) dk[i] = f1[i] - k1[i];
            lu_solve(n, W, piv, dk);
            /* loss seeds of the save points inside (tn, tnew] */
            double A[ORC_MAXN] = {0}, B1[ORC_MAXN] = {0}, B2[ORC_MAXN] = {0};
            while (jsave > jlo && tsave[jsave - 1] > tn) {
                const double ts = tsave[jsave - 1];
                const int at_end = (ts == tnew);
                const double Th = at_end ? 1.0 : (ts - tn) / h;
                const double c1 = at_end ? 0.0 : Th * (1.0 - Th) / (1.0 - 2.0 * d);
                const double c2 = at_end ? 1.0 : Th * (Th - 2.0 * d) / (1.0 - 2.0 * d);
                for (int io = 0; io < nobs; ++io) {
                    const int i = pb->i_obs[io];
                    double v = un[i] + h * (c1 * k1[i] + c2 * (k1[i] + dk[i]));
                    double mask = 1.0;
                    if (pb->clamp_pred) { mask = dclamp(v, -pb->ub, pb->ub); v = clampd(v, -pb->ub, pb->ub); }
                    const double r_ = (data[io + nobs * (jsave - 1)] - v) / pb->yscale[io];
                    double w_ = (pb->loss_kind == 0) ? -dabs_(r_) : -2.0 * r_;
                    w_ *= mask / pb->yscale[io];
                    A[i] += w_; B1[i] += w_ * h * c1; B2[i] += w_ * h * c2;
                }
                --jsave;
            }
            /* adjoint of the step */
            double v[ORC_MAXN], ub[ORC_MAXN], kb1[ORC_MAXN];
            for (int i = 0; i < n; ++i) { v[i] = h * lam[i] + B2[i]; ub[i] = lam[i] + A[i]; kb1[i] = B1[i] + v[i]; }
            lu_solve(n, Wt, pivt, v);
            for (int i = 0; i < n; ++i) kb1[i] -= v[i];
            double av[ORC_MAXR], rho1[ORC_MAXR];
            for (int j = 0; j < nr; ++j) {
                double a = 0.0;
                for (int i = 0; i < ns; ++i) a += v[i] * pb->rate_scale[i] * w_out[i + ns * j];
                av[j] = a; rho1[j] = a * r1[j];
            }
            for (int c = 0; c < n; ++c) {
                double um = 0.0;
                for (int j = 0; j < nr; ++j) um += rho1[j] * w_in[c + n * j];
                const double gb = um * g1[c];
                ub[c] += gb; kb1[c] += 0.5 * h * gb;
            }
            lu_solve(n, Wt, pivt, kb1);     /* w */
            double s1[ORC_MAXN] = {0}, s2[ORC_MAXN] = {0};
            for (int j = 0; j < nr; ++j) {
                double aw = 0.0, q1 = 0.0, qd = 0.0;
                for (int i = 0; i < ns; ++i) aw += kb1[i] * pb->rate_scale[i] * w_out[i + ns * j];
                for (int c = 0; c < n; ++c) { const double wg = w_in[c + n * j] * g0[c]; q1 += wg * k1[c]; qd += wg * dk[c]; }
                const double c1j = 1.0 + gam * q1, czd = gam * qd;
                const double pv = av[j] * gam * r0[j], pw = aw * r0[j], gpw = gam * pw;
                const double beta = pw * c1j + pv * qd;
                b_b[j] += beta + rho1[j];
                for (int c = 0; c < n; ++c) {
                    const double mm = pv * dk[c] + gpw * k1[c];
                    b_in[c + n * j] += rho1[j] * x1[c] + beta * x0[c] + g0[c] * mm;
                    s1[c] += beta * w_in[c + n * j];
                    s2[c] += w_in[c + n * j] * mm;
                }
                const double ca = r0[j] * czd + r1[j], cb = r0[j] * c1j;
                for (int i = 0; i < ns; ++i)
                    b_out[i + ns * j] += pb->rate_scale[i] * (v[i] * ca + kb1[i] * cb);
            }
            for (int c = 0; c < n; ++c) lam[c] = ub[c] + g0[c] * s1[c] + h0[c] * s2[c];
            tnew = tn;
        }
        const double denom = (double)nobs * (double)jsave_end;
        for (int k = 0; k < P; ++k) {
            double a = 0.0;
            for (int m = 0; m < nth; ++m) a += dth[m + (size_t)nth * k] * thb[m];
            grad[k] += a / denom;
        }
    }
    free(tp.rec);
    return rc;
}

static int solve_dispatch(const orc_problem *pb, const double *th, const double *dth, int P,
                          const double *u0, const double *tsave, int nsave,
                          const double *data, double *pred, double *dpred,
                          double *loss_out, double *grad, int32_t *n_saved_out, orc_stats *st, double *ws) {
    if (pb->jac_fd && pb->solver != 0) return -7;   /* the finite-difference W is restated for plain Rosenbrock23 (primal solves and forward tangents) */
    if (pb->jac_fd) return solve_one_ws(pb, th, dth, P, u0, tsave, nsave, data, pred, dpred, loss_out, grad, n_saved_out, st, ws);   /* never the analytic-J adjoint */
    if (pb->solver == 2 && pb->errnorm_sens && P > 0) return -8;   /* the CRNN composite below carries tangents on accepted steps only: no dual norm (it used to be ignored silently; case2's composite IS Tsit5 -- solver 1 -- see solve_one_auto's header) */
    if (pb->solver == 2) return solve_one_auto(pb, th, dth, P, u0, tsave, nsave, data, pred, dpred, loss_out, grad, n_saved_out, st, ws, NULL);
    if (pb->solver == 1) return solve_one_tsit5(pb, th, dth, P, u0, tsave, nsave, data, pred, dpred, loss_out, grad, n_saved_out, st, ws);
    if (pb->grad_adjoint && P > 0 && !pb->errnorm_sens && !dpred)
        return solve_one_adj(pb, th, dth, P, u0, tsave, nsave, data, pred, loss_out, grad, n_saved_out, st);
    return solve_one_ws(pb, th, dth, P, u0, tsave, nsave, data, pred, dpred, loss_out, grad, n_saved_out, st, ws);
}

/* composite only: also reports how the accepted steps split between the two algorithms and the number of switches */
int orc_solve_one_auto(const orc_problem *pb, const double *th, const double *dth, int P,
                       const double *u0, const double *tsave, int nsave,
                       const double *data, double *pred, double *dpred,
                       double *loss_out, double *grad, int32_t *n_saved_out, orc_stats *st, int64_t *st_alg /*[3]*/) {
    const int n = N_(pb);
    double *ws = P > 0 ? (double *)malloc(sizeof(double) * ((size_t)n * P * 9 + P)) : NULL;
    if (st_alg) st_alg[0] = st_alg[1] = st_alg[2] = 0;
    if (pb->jac_fd) { free(ws); return -7; }   /* the composite's stiff branch is restated with the analytic J only */
    if (pb->errnorm_sens && P > 0) { free(ws); return -8; }   /* as in solve_dispatch */
    int rc = solve_one_auto(pb, th, dth, P, u0, tsave, nsave, data, pred, dpred, loss_out, grad, n_saved_out, st, ws, st_alg);
    free(ws);
    return rc;
}

int orc_solve_one(const orc_problem *pb, const double *th, const double *dth, int P,
                  const double *u0, const double *tsave, int nsave,
                  const double *data, double *pred, double *dpred,
                  double *loss_out, double *grad /* [P] accumulated += */,
                  int32_t *n_saved_out, orc_stats *st) {
    const int n = N_(pb);
    double *ws = P > 0 ? (double *)malloc(sizeof(double) * ((size_t)n * P * 9 + P)) : NULL;
    int rc = solve_dispatch(pb, th, dth, P, u0, tsave, nsave, data, pred, dpred, loss_out, grad, n_saved_out, st, ws);
    free(ws);
    return rc;
}

/* ------------------------------------------------------------------------ */
/* Batch driver over B initial conditions (IC-fastest layout), OpenMP over  */
/* trajectories: the stand-in for an EnsembleThreads() run.                 */
/* grad[P] = sum over trajectories of d loss_b / d p (NOT divided by B).    */
/* ------------------------------------------------------------------------ */
int orc_solve_batch(const orc_problem *pb, const double *th, const double *dth, int P,
                    const double *u0 /*[n*B]*/, const double *tsave, int nsave,
                    const double *data /*[nsave*n_obs*B]*/, int64_t B, int64_t first, int64_t count,
                    double *pred /*[nsave*n*B] or NULL*/, double *loss /*[B]*/,
                    double *grad /*[P]*/, int32_t *retcode /*[B]*/, int32_t *n_saved /*[B]*/,
                    int64_t *stats /* [2]: accepted, rejected */, int nthreads) {
    const int n = N_(pb), nobs = pb->n_obs;
    int64_t acc = 0, rej = 0;
    if (grad) memset(grad, 0, sizeof(double) * (size_t)P);
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel reduction(+ : acc, rej)
    {
        double *g_loc = P > 0 ? (double *)calloc((size_t)P, sizeof(double)) : NULL;
        double *d_loc = (double *)malloc(sizeof(double) * (size_t)nobs * nsave);
        double *p_loc = pred ? (double *)malloc(sizeof(double) * (size_t)n * nsave) : NULL;
        double *ws = (grad && P > 0) ? (double *)malloc(sizeof(double) * ((size_t)n * P * 9 + P)) : NULL;
#pragma omp for schedule(dynamic, 8)
        for (int64_t b = first; b < first + count; ++b) {
            double u[ORC_MAXN];
            for (int i = 0; i < n; ++i) u[i] = u0[(size_t)i * B + b];
            for (int j = 0; j < nsave; ++j)
                for (int i = 0; i < nobs; ++i) d_loc[i + nobs * j] = data[((size_t)j * nobs + i) * B + b];
            orc_stats st = {0, 0};
            double l = 0; int32_t ns_ = 0;
            if (p_loc) memset(p_loc, 0, sizeof(double) * (size_t)n * nsave);
            int rc = solve_dispatch(pb, th, dth, grad ? P : 0, u, tsave, nsave, d_loc, p_loc, NULL, &l, g_loc, &ns_, &st, ws);
            if (loss) loss[b] = l;
            if (retcode) retcode[b] = rc;
            if (n_saved) n_saved[b] = ns_;
            if (pred) for (int j = 0; j < nsave; ++j) for (int i = 0; i < n; ++i) pred[((size_t)j * n + i) * B + b] = p_loc[i + n * j];
            acc += st.naccept; rej += st.nreject;
        }
#pragma omp critical
        { if (grad && g_loc) for (int k = 0; k < P; ++k) grad[k] += g_loc[k]; }
        free(g_loc); free(d_loc); free(p_loc); free(ws);
    }
    if (stats) { stats[0] = acc; stats[1] = rej; }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* Flux.Optimise restatement (old "implicit" optimisers; EPS = 1e-8)        */
/* [UNVERIFIED-DEP]:  Optimiser(ExpDecay(eta0,decay,step,clip), ADAM(eta,   */
/* beta), WeightDecay(wd)) applied left to right, then p .-= delta.         */
/* case2/case2.jl:31-32,197; robertson/rober_crnn.jl:19,221-224 (norm clip) */
/* state = [m(P) | v(P) | beta1^t, beta2^t, expdecay_eta, n_calls]          */
/* ------------------------------------------------------------------------ */
typedef struct orc_opt {
    int32_t use_expdecay; int32_t decay_step;
    double ed_eta0, ed_decay, ed_clip;
    double eta, beta1, beta2, wd;
    double grad_clip_norm; /* <=0: off (robertson: 10) */
} orc_opt;

int orc_opt_state_len(int P) { return 2 * P + 4; }
void orc_opt_init(const orc_opt *o, int P, double *state) {
    memset(state, 0, sizeof(double) * (size_t)(2 * P + 4));
    state[2 * P + 0] = o->beta1; state[2 * P + 1] = o->beta2;
    state[2 * P + 2] = o->ed_eta0; state[2 * P + 3] = 0.0;
}
void orc_opt_update(const orc_opt *o, int P, double *p, const double *grad_in, double *state) {
    double *m = state, *v = state + P, *bp = state + 2 * P, *ed_eta = state + 2 * P + 2, *ncalls = state + 2 * P + 3;
    const double eps = 1e-8;
    double gn = 0.0, cs = 1.0;
    if (o->grad_clip_norm > 0) {
        for (int k = 0; k < P; ++k) gn += grad_in[k] * grad_in[k];
        gn = sqrt(gn);
        if (gn > o->grad_clip_norm) cs = o->grad_clip_norm / gn;
    }
    double eta_ed = 1.0;
    if (o->use_expdecay) {
        *ncalls += 1.0;
        if (fmod(*ncalls, (double)o->decay_step) == 0.0) *ed_eta = fmax(*ed_eta * o->ed_decay, o->ed_clip);
        eta_ed = *ed_eta;
    }
    for (int k = 0; k < P; ++k) {
        double g = (o->grad_clip_norm > 0 && cs != 1.0) ? grad_in[k] / gn * o->grad_clip_norm : grad_in[k];
        g *= eta_ed;
        m[k] = o->beta1 * m[k] + (1.0 - o->beta1) * g;
        v[k] = o->beta2 * v[k] + (1.0 - o->beta2) * g * g;
        double delta = m[k] / (1.0 - bp[0]) / (sqrt(v[k] / (1.0 - bp[1])) + eps) * o->eta;
        delta += o->wd * p[k];
        p[k] -= delta;
    }
    bp[0] *= o->beta1; bp[1] *= o->beta2;
}

/* ======================================================================== */
/* Cathode-UQ path (BASELINE config 5).  TEST INFRASTRUCTURE like the rest.  */
/*   crnn!        Cathode_NCM333_UQ/src_333/network.jl:152-165                */
/*   T(t)         getsampletemp, network.jl:143-149 (T0 = 373.15, :189)       */
/*   HRR_getter   network.jl:167-175                                          */
/*   pred_n_ode   network.jl:196-218 (tspan = [ts[1], ts[end]], saveat = ts)  */
/*   loss         network.jl:262-266: sum((pred - data).^2)/n_replicas/D      */
/*   gradient     ForwardDiff.gradient, network.jl:232                        */
/* theta (17, already multiplied by p_scales as the reference does inside the */
/* RHS): [lnA(3) | Ea(3) (the exponent is R/T * Ea * 1e5) | b(3) | dH(3) |    */
/* order n(3) | nu2, nu3].  The reference's solver here is                    */
/* AutoTsit5(TRBDF2(autodiff=true)); this restatement integrates with the     */
/* non-autonomous Rosenbrock23 of this file (same tolerances): results agree  */
/* with the reference only to within the solver tolerance -- parity unpinned. */
/* Tangents are taken by complex-step differentiation of the closed forms, so */
/* the oracle shares no derivative code with the HIP kernel.                  */
/* ======================================================================== */
#include <complex.h>
typedef double complex cplx;

typedef struct orc_cathode {
    double lb_clamp, T0, beta;      /* beta in K/min: T = T0 + beta/60 t */
    double atol, rtol;
    int32_t maxiters;
    int32_t solver;   /* 0 Rosenbrock23; 2 AutoTsit5 composite with Rosenbrock23 as its stiff algorithm;
                         3 AutoTsit5(TRBDF2(autodiff = true)) -- THE REFERENCE'S ALGORITHM (network.jl:195, used :205-212): the
                           explicit branch, the switching rule (see solve_one_auto above) and TRBDF2 with OrdinaryDiffEq's Newton
                           machinery restated (cath_trbdf2_* below; primal only);
                         4 TRBDF2 alone (to test the stepper by itself) */
    double gamma, qmin, qmax, beta1, beta2, qsteady_min, qsteady_max, qoldinit;
    /* errnorm_sens != 0 (Rosenbrock23, gradient solves): ForwardDiff.gradient(x -> loss_neuralode(x, i_exp), p_temp) (network.jl:232)
       pushes Duals through the adaptive solve in chunks of 9 (+ 8 and one zero partial) of the 17 normalised parameters, and the
       error norm weighs the partials (solve_one_ws has the formula; 1: / length(u), 2: / totallength(u) -- the cathode Manifest pins
       DiffEqBase 6.189, the totallength era).  One call = ONE chunk: the directions [dir_lo, dir_lo + dir_n) of theta, each scaled by
       dir_scale[k] = d theta_k / d p_k = p_scales[k] (network.jl:152-157) in the norm, dual_partials partials per Dual (9).  The
       gradient is still returned with respect to theta; entries outside the chunk are zero. */
    int32_t errnorm_sens, dir_lo, dir_n, dual_partials;
    double dir_scale[17];
    int32_t trbdf2_est;   /* TRBDF2's smoothed error estimate (smooth_est = true, the default): 0 = `W \ tmp` with the W the Newton
                             iteration holds, W = J - I/(gamma dt) -- what the package's perform_step! reads like; 1 = Shampine's
                             (I - gamma dt J)^-1 tmp (differs by the factor gamma dt).  [UNVERIFIED-DEP] */
    int32_t pad_;
} orc_cathode;

void orc_cathode_defaults(orc_cathode *c) {
    memset(c, 0, sizeof(*c));
    c->lb_clamp = 1e-16; c->T0 = 373.15; c->beta = 10.0; c->atol = 1e-12; c->rtol = 1e-3; c->maxiters = 2500000;
    c->gamma = 0.9; c->qmin = 0.2; c->qmax = 10.0; c->beta1 = 7.0 / 20.0; c->beta2 = 2.0 / 10.0;
    c->qsteady_min = 1.0; c->qsteady_max = 1.2; c->qoldinit = 1e-4;
}
int orc_sizeof_cathode(void) { return (int)sizeof(orc_cathode); }

static const double CATH_R = -1.0 / 8.314;   /* network.jl:151 */

static cplx cclampc(cplx v, double lo, double hi) {
    double re = creal(v);
    return re > hi ? (cplx)hi : (re < lo ? (cplx)lo : v);
}
/* rates r_j(t, u; theta) in complex arithmetic */
static void cath_rates(const orc_cathode *c, const cplx *th, const cplx *u, double t, cplx *r) {
    double T = c->T0 + c->beta / 60.0 * t;
    for (int j = 0; j < 3; ++j) {
        cplx lx = clog(cclampc(u[j], c->lb_clamp, 10.0));
        r[j] = cexp(th[6 + j] * log(T) + (CATH_R / T) * (th[3 + j] * 1e5) + th[12 + j] * lx + th[j]);
    }
}
static void cath_rhs(const orc_cathode *c, const cplx *th, const cplx *u, double t, cplx *du) {
    cplx r[3];
    cath_rates(c, th, u, t, r);
    du[0] = -r[0];
    du[1] = -r[1] + th[15] * r[0];
    du[2] = -r[2] + th[16] * r[1];
}
static void cath_rhs_real(const orc_cathode *c, const double *th, const double *u, double t, double *du) {
    cplx thc[17], uc[3], d[3];
    for (int k = 0; k < 17; ++k) thc[k] = th[k];
    for (int i = 0; i < 3; ++i) uc[i] = u[i];
    cath_rhs(c, thc, uc, t, d);
    for (int i = 0; i < 3; ++i) du[i] = creal(d[i]);
}
/* analytic Jacobian and time derivative in complex arithmetic (so that their directional derivatives come for free) */
static void cath_jac_ft(const orc_cathode *c, const cplx *th, const cplx *u, double t, cplx *J /*3x3 col-major*/, cplx *ft) {
    double T = c->T0 + c->beta / 60.0 * t, Td = c->beta / 60.0;
    cplx r[3], a[3], rho[3];
    cath_rates(c, th, u, t, r);
    for (int j = 0; j < 3; ++j) {
        double re = creal(u[j]);
        cplx g = (re >= c->lb_clamp && re <= 10.0) ? 1.0 / u[j] : 0.0;
        a[j] = r[j] * th[12 + j] * g;
        rho[j] = r[j] * (th[6 + j] * (Td / T) - (th[3 + j] * 1e5) * CATH_R * Td / (T * T));
    }
    for (int k = 0; k < 9; ++k) J[k] = 0.0;
    J[0 + 3 * 0] = -a[0];
    J[1 + 3 * 0] = th[15] * a[0]; J[1 + 3 * 1] = -a[1];
    J[2 + 3 * 1] = th[16] * a[1]; J[2 + 3 * 2] = -a[2];
    ft[0] = -rho[0];
    ft[1] = -rho[1] + th[15] * rho[0];
    ft[2] = -rho[2] + th[16] * rho[1];
}
void orc_cathode_rhs(const orc_cathode *c, const double *th, const double *u, double t, double *du) { cath_rhs_real(c, th, u, t, du); }

static void csolve3(const double *W, const int *piv, cplx *b) {   /* real LU applied to a complex rhs */
    for (int k = 0; k < 3; ++k) { int p = piv[k]; if (p != k) { cplx tt = b[k]; b[k] = b[p]; b[p] = tt; } }
    for (int k = 0; k < 3; ++k) { cplx a = b[k]; for (int i = k + 1; i < 3; ++i) b[i] -= W[i + 3 * k] * a; }
    for (int k = 2; k >= 0; --k) { b[k] /= W[k + 3 * k]; cplx a = b[k]; for (int i = 0; i < k; ++i) b[i] -= W[i + 3 * k] * a; }
}

/* ------------------------------------------------------------------------------------------------------------------ *
 * TRBDF2 (Bank et al. / Hosea & Shampine) as OrdinaryDiffEqSDIRK's `TRBDF2(autodiff = true)` runs it -- the stiff algorithm of
 * the reference's composite (network.jl:195; Cathode_NCM333_UQ/Manifest.toml pins OrdinaryDiffEqSDIRK 1.7.0,
 * OrdinaryDiffEqNonlinearSolve, OrdinaryDiffEqDifferentiation 1.16.0 -- none of them vendored).  EVERYTHING BELOW IS
 * [UNVERIFIED-DEP]: restated from the packages' published source as remembered, not executable here (no Julia).  Branch by
 * branch:
 *   tableau      gamma = 2 - sqrt 2, d = 1 - sqrt2/2 (= gamma/2), omega = sqrt2/4, btilde = ((1 - sqrt2)/3, 1/3, (sqrt2 - 2)/3),
 *                alpha1 = -sqrt2/2, alpha2 = 1 + sqrt2/2 (TRBDF2Tableau)
 *   stages       in z = dt f form.  zprev = dt fsalfirst;  stage 1: tmp = uprev + d zprev, guess z = zprev,
 *                solve z = dt f(tmp + d z, t + gamma dt) -> z_g;  stage 2: tmp = uprev + omega zprev + omega z_g, guess
 *                z = alpha1 zprev + alpha2 z_g (Shampine), solve z = dt f(tmp + d z, t + dt);  u = tmp + d z;
 *                fsallast = z / dt (NOT a fresh f(u): the next TRBDF2 step starts from it; an algorithm switch re-evaluates f)
 *   Newton       NLNewton defaults: kappa = 1/100, max_iter = 10, fast_convergence_cutoff = 1/5, new_W_dt_cutoff = 1/5, no
 *                relaxation.  W = J - I/(gamma dt) (gamma = d; LU with partial pivoting); iteration
 *                dz = W \ ((dt f(tmp + d z) - z)/(gamma dt)), z <- z - dz, ndz = RMS(dz / (atol + rtol max(|uprev|, |ustep|)));
 *                first iteration converged if ndz < 1e-5; from the second theta = ndz/ndz_prev, divergence if theta > 2,
 *                eta = theta/(1 - theta), converged if eta >= 0 and eta ndz < kappa; round-off guard |theta - 1| <= 10 eps;
 *                ten iterations without convergence = divergence; a divergence with a Jacobian that is not the current step's
 *                -> status TryAgain, new J, once more from the current iterate; otherwise the step fails:
 *                dt <- dt / failfactor (2), no controller call, EEst untouched.
 *   J / W reuse  do_newJW: new J and W at integrator.iter <= 1 and at the cache's first call; else errorfail = (EEst of the
 *                previous attempt > 1), freshJ = (J was formed at this t) && !errorfail; if !freshJ:
 *                smallstepchange = |W_gammadt/(gamma dt) - 1| <= 1/5, jbad = (status == TryAgain && smallstepchange);
 *                wbad = !smallstepchange || (first stage && errorfail) || status == Divergence; (new_jac, new_W) = (jbad,
 *                jbad || wbad).  J = df/du at (uprev, t) by ForwardDiff (exact derivative arithmetic: the analytic J here),
 *                held for both stages and, stale, for later steps; calc_J! of a composite also sets eigen_est = opnorm(J, Inf),
 *                so the switch-back test of AutoSwitch sees the norm of the LAST FORMED Jacobian.
 *   error        tmp = btilde1 zprev + btilde2 z_g + btilde3 z; smooth_est: est = W \ tmp (trbdf2_est above); EEst = RMS(est /
 *                (atol + rtol max(|uprev|, |u|))); order 2 -> PI exponents 7/20, 2/10.
 *   saveat       TRBDF2 has no interpolant of its own: third-order Hermite on (uprev, u, k1 = fsalfirst, k2 = fsallast).
 * ------------------------------------------------------------------------------------------------------------------ */
enum { NL_FASTCONV = 2, NL_CONV = 1, NL_DIV = -2, NL_TRYAGAIN = -4 };
typedef struct cath_nl {
    double J[9], W[9];
    cplx Jk[17][9];   /* the Jacobian of the tangent copies (theta + i h e_k, u + i h s_k), formed with J and reused with it */
    int piv[3];
    double J_t, W_gdt, eta_old;
    int status, firstcall, new_W;
    int64_t n_newton, n_jac, n_w, n_fail;   /* statistics */
} cath_nl;
static void cath_nl_init(cath_nl *nl) {
    memset(nl, 0, sizeof(*nl));
    nl->J_t = NAN; nl->W_gdt = 0.0; nl->eta_old = 1.0; nl->status = NL_DIV; nl->firstcall = 1;
}
/* The tangent copies that ride through a nonlinear solve (gradient of the TRBDF2 branch, primal error norm): copy k carries theta + i h e_k
   and the state's tangent in its imaginary part.  They follow the primal's iteration step for step -- the same W (real: dt, the pivots and every
   decision are the primal's), the same number of iterations -- with the partial of W moved to the right-hand side as in the Rosenbrock23 branch:
   (W + i h W') (dz + i h dz') = rhs + i h rhs'  ->  W dz' = rhs' - W' dz,  W' = J' (the imaginary part of the copy's J at (uprev, t_J)).
   That is ForwardDiff's arithmetic through NLNewton with a Dual-valued W, to first order in h -- EXCEPT that ForwardDiff's convergence
   tests see the norm of Dual arrays (partials included) and these copies do not enter any test: the primal-norm mode of the other
   branches (errnorm_sens = 0).  [UNVERIFIED-DEP] like the solver itself. */
typedef struct cath_cp {
    const int *act;        /* [17]: copy carried? */
    cplx (*thk)[17];       /* [18][17] */
    cplx (*uprev)[3];      /* [18][3] */
    cplx (*tmp)[3];        /* [18][3] */
    cplx (*z)[3];          /* [18][3], in/out */
} cath_cp;
static void csolve3(const double *W, const int *piv, cplx *b);
static void cath_jac_real(const orc_cathode *c, const double *th, const double *u, double t, double *J) {
    cplx thc[17], uc[3], Jc[9], ftc[3];
    for (int k = 0; k < 17; ++k) thc[k] = th[k];
    for (int i = 0; i < 3; ++i) uc[i] = u[i];
    cath_jac_ft(c, thc, uc, t, Jc, ftc);
    for (int k = 0; k < 9; ++k) J[k] = creal(Jc[k]);
}
/* one nlsolve! call: stage equation z = dt f(tmp + gam z, t + cst dt), z in/out.  Returns 0 (converged) or -1 (step fails). */
static int cath_trbdf2_nlsolve(const orc_cathode *c, const double *th, cath_nl *nl, const double *uprev, double t, double dt,
                               int integ_iter, double EEst_prev, int isfs, double gam, double cst, const double *tmp, double *z,
                               double *eigen_est, const cath_cp *cp) {
    const double kappa = 1.0 / 100.0, gW = gam * dt;
    double eta = nl->eta_old;
    for (int redo = 0; redo < 3; ++redo) {
        /* ---- update_W! -> calc_W! -> do_newJW */
        int new_jac, new_W;
        if (integ_iter <= 1 || nl->firstcall) { new_jac = 1; new_W = 1; }
        else {
            const int errorfail = EEst_prev > 1.0;
            const int freshJ = (t == nl->J_t) && !errorfail;
            int jbad, small;
            if (freshJ) { jbad = 0; small = 1; }
            else {
                const double W_igdt = 1.0 / nl->W_gdt, igdt = 1.0 / gW;
                small = fabs(igdt / W_igdt - 1.0) <= 0.2;
                jbad = (nl->status == NL_TRYAGAIN) && small;
            }
            const int wbad = (!small) || (isfs && errorfail) || nl->status == NL_DIV;
            new_jac = jbad; new_W = jbad || wbad;
        }
        if (new_jac) {
            cath_jac_real(c, th, uprev, t, nl->J);
            if (cp) for (int k = 0; k < 17; ++k) if (cp->act[k]) { cplx ftd[3]; cath_jac_ft(c, cp->thk[k], cp->uprev[k], t, nl->Jk[k], ftd); }
            nl->J_t = t; nl->n_jac++;
            double est = 0.0;   /* calc_J! of a composite: eigen_est = opnorm(J, Inf) */
            for (int i = 0; i < 3; ++i) { double a = 0.0; for (int cc = 0; cc < 3; ++cc) a += fabs(nl->J[i + 3 * cc]); if (a > est) est = a; }
            *eigen_est = est;
        }
        if (new_W) {
            const double inv = 1.0 / gW;
            for (int cc = 0; cc < 3; ++cc) for (int i = 0; i < 3; ++i) nl->W[i + 3 * cc] = nl->J[i + 3 * cc] - (i == cc ? inv : 0.0);
            if (lu_factor(3, nl->W, nl->piv) != 0) { nl->status = NL_DIV; return -1; }
            nl->W_gdt = gW; nl->n_w++;
        }
        nl->new_W = new_W;
        /* ---- initialize!, the iteration */
        const double inv_gdt = 1.0 / (dt * gam), tstep = t + cst * dt;
        nl->status = NL_DIV;   /* check_div: what a loop that runs out of iterations leaves behind */
        eta = new_W ? pow(fmax(nl->eta_old, 2.220446049250313e-16), 0.8) : nl->eta_old;
        double ndz = 0.0, ndzprev = 0.0;
        for (int it = 1; it <= 10; ++it) {
            double ustep[3], f[3], dz[3], zt[3];
            for (int i = 0; i < 3; ++i) ustep[i] = tmp[i] + gam * z[i];
            cath_rhs_real(c, th, ustep, tstep, f);
            for (int i = 0; i < 3; ++i) dz[i] = (dt * f[i] - z[i]) * inv_gdt;
            lu_solve(3, nl->W, nl->piv, dz);
            nl->n_newton++;
            cplx dzk[17][3];
            if (cp) for (int k = 0; k < 17; ++k) if (cp->act[k]) {
                cplx us[3], fk[3];
                for (int i = 0; i < 3; ++i) us[i] = cp->tmp[k][i] + gam * cp->z[k][i];
                cath_rhs(c, cp->thk[k], us, tstep, fk);
                for (int i = 0; i < 3; ++i) {
                    dzk[k][i] = (dt * fk[i] - cp->z[k][i]) * inv_gdt;
                    for (int cc = 0; cc < 3; ++cc) dzk[k][i] -= (nl->Jk[k][i + 3 * cc] - creal(nl->Jk[k][i + 3 * cc])) * dz[cc];
                }
                csolve3(nl->W, nl->piv, dzk[k]);
            }
            double s_ = 0.0;
            for (int i = 0; i < 3; ++i) { const double e = dz[i] / (c->atol + c->rtol * fmax(fabs(uprev[i]), fabs(ustep[i]))); s_ += e * e; }
            ndzprev = ndz; ndz = sqrt(s_ / 3.0);
            for (int i = 0; i < 3; ++i) zt[i] = z[i] - dz[i];
            if (!isfinite(ndz)) { nl->status = NL_DIV; break; }
            double theta = 0.0;
            if (it > 1) {
                theta = ndz / ndzprev;
                if (fabs(theta - 1.0) <= 10.0 * 2.220446049250313e-16) {
                    if (ndz <= 1.0) { nl->status = NL_CONV; break; }
                    nl->status = NL_DIV; break;
                }
                if (theta > 2.0) { nl->status = NL_DIV; break; }
            }
            for (int i = 0; i < 3; ++i) z[i] = zt[i];   /* apply_step! */
            if (cp) for (int k = 0; k < 17; ++k) if (cp->act[k]) for (int i = 0; i < 3; ++i) cp->z[k][i] -= dzk[k][i];
            if (it > 1) eta = theta / (1.0 - theta);
            if ((it == 1 && ndz < 1e-5) || (it > 1 && eta >= 0.0 && eta * ndz < kappa)) { nl->status = NL_CONV; break; }
        }
        if (nl->status == NL_DIV && !(t == nl->J_t)) { nl->status = NL_TRYAGAIN; continue; }   /* @goto REDO */
        break;
    }
    nl->eta_old = eta;
    nl->firstcall = 0;   /* postamble! */
    if (nl->status < 0) { nl->n_fail++; return -1; }
    return 0;
}
_Thread_local int64_t orc_cathode_last_nl[4] = {0, 0, 0, 0};   /* Newton iterations, Jacobians, W factorisations, failed solves */
void orc_cathode_nl_stats(int64_t *out) { for (int i = 0; i < 4; ++i) out[i] = orc_cathode_last_nl[i]; }

/* One (particle, heating-rate) trajectory: HRR prediction, MSE loss against the replica statistics
   dbar[i] = mean_k data[i,k], d2bar[i] = mean_k data[i,k]^2, gradient wrt theta (17).
   Complex-step: for direction e_k the whole step is evaluated at theta + i*h*e_k, u + i*h*s_k with the REAL
   factorisation of W (dt and the pivots are real), which is exactly the first-order tangent. */
_Thread_local int64_t orc_cathode_last_tsit5_steps = 0;   /* accepted Tsit5 steps of the calling thread's last solve (composite) */
int64_t orc_cathode_tsit5_steps(void) { return orc_cathode_last_tsit5_steps; }
int orc_cathode_solve_one(const orc_cathode *c, const double *th, const double *ts, int D,
                          const double *dbar, const double *d2bar, double *hrr /*[D] or NULL*/,
                          double *loss_out, double *grad /*[17] or NULL*/, int32_t *n_saved_out, orc_stats *st) {
    const double d = 1.0 / (2.0 + sqrt(2.0)), c32 = 6.0 + sqrt(2.0), h = 1e-30;
    const int P = grad ? 17 : 0;
    const int composite = (c->solver == 2 || c->solver == 3), trbdf2 = (c->solver == 3 || c->solver == 4);
    const int sens = (c->errnorm_sens != 0 && grad != NULL);
    /* errnorm_sens: every algorithm's error estimate carries the chunk's partials (Tsit5: dt sum_j bt_j k_j'; TRBDF2: the smoothed estimate's
       partial W^-1 (tmp' - J' est)); TRBDF2's Newton convergence tests stay on the primal (cath_cp) */
    if (sens && (c->dir_n < 1 || c->dir_lo < 0 || c->dir_lo + c->dir_n > 17)) return -1;
#define ACT(k) ((k) < P && (!sens || ((k) >= c->dir_lo && (k) < c->dir_lo + c->dir_n)))
    const double sens_div = c->errnorm_sens == 2 ? 3.0 * (1.0 + (double)c->dual_partials) : 3.0;
    const double t0 = ts[0], tend = ts[D - 1];
    double t = t0;
    cplx u[18][3];   /* u[P] = primal (imag 0); u[k] = primal + i h s_k */
    cplx thk[18][17];
    for (int k = 0; k <= 17; ++k) {
        for (int m = 0; m < 17; ++m) thk[k][m] = th[m] + ((k < 17 && m == k) ? I * h : 0.0);
        u[k][0] = 1.0; u[k][1] = 0.0; u[k][2] = 0.0;      /* u0 = (1,0,0), network.jl:186-187 */
    }
    const int PR = 17;  /* index of the primal copy */
    cplx f0[18][3];
    for (int k = 0; k <= 17; ++k) if (k == PR || ACT(k)) cath_rhs(c, thk[k], u[k], t, f0[k]);
    /* initial step (Hairer, order 2) on the primal; with errnorm_sens every norm of it is the dual-inclusive one (init_dt_sens) */
    double dt;
    {
        double sk[3], d0 = 0, d1 = 0, d2 = 0, ur[3], fr[3], u1[3], f1[3];
        for (int i = 0; i < 3; ++i) { ur[i] = creal(u[PR][i]); fr[i] = creal(f0[PR][i]); sk[i] = c->atol + fabs(ur[i]) * c->rtol;
            d0 += (ur[i] / sk[i]) * (ur[i] / sk[i]); d1 += (fr[i] / sk[i]) * (fr[i] / sk[i]); }
        if (sens) for (int k = 0; k < 17; ++k) if (ACT(k)) for (int i = 0; i < 3; ++i) { double e = c->dir_scale[k] * cimag(f0[k][i]) / h / sk[i]; d1 += e * e; }
        const double dv = sens ? sens_div : 3.0;
        d0 = sqrt(d0 / dv); d1 = sqrt(d1 / dv);
        double dtmax = tend - t0;
        double dt0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
        dt0 = fmin(dt0, dtmax);
        for (int i = 0; i < 3; ++i) u1[i] = ur[i] + dt0 * fr[i];
        cath_rhs_real(c, th, u1, t + dt0, f1);
        for (int i = 0; i < 3; ++i) { double e = (f1[i] - fr[i]) / sk[i]; d2 += e * e; }
        if (sens) for (int k = 0; k < 17; ++k) if (ACT(k)) {
            cplx u1k[3], f1k[3];
            for (int i = 0; i < 3; ++i) u1k[i] = u[k][i] + dt0 * f0[k][i];
            cath_rhs(c, thk[k], u1k, t + dt0, f1k);
            for (int i = 0; i < 3; ++i) { double e = c->dir_scale[k] * cimag(f1k[i] - f0[k][i]) / h / sk[i]; d2 += e * e; }
        }
        d2 = sqrt(d2 / dv) / dt0;
        double dm = fmax(d1, d2);
        double dt1 = dm <= 1e-15 ? fmax(1e-6, dt0 * 1e-3) : pow(10.0, -(2.0 + log10(dm)) / (composite ? 5.0 : 2.0));   /* the order of the starting algorithm */
        dt = fmin(fmin(100 * dt0, dt1), dtmax);
    }
    double qold = c->qoldinit, loss_sum = 0.0, g[17];
    for (int k = 0; k < 17; ++k) g[k] = 0.0;
    int jsave = 0, retcode = 0, iter = 0;
#define CATH_SAVE(UARR_EXPR, tsv)                                                                                \
    do {                                                                                                        \
        cplx hv[18];                                                                                            \
        for (int k = 0; k <= 17; ++k) if (k == PR || ACT(k)) {                                                  \
            cplx uu[3], r[3];                                                                                   \
            for (int i = 0; i < 3; ++i) uu[i] = (UARR_EXPR);                                                    \
            cath_rates(c, thk[k], uu, (tsv), r);                                                                \
            hv[k] = r[0] * thk[k][9] + r[1] * thk[k][10] + r[2] * thk[k][11];                                   \
        }                                                                                                       \
        double hp = creal(hv[PR]), e = hp - dbar[jsave];                                                        \
        if (hrr) hrr[jsave] = hp;                                                                               \
        loss_sum += e * e + (d2bar[jsave] - dbar[jsave] * dbar[jsave]);                                         \
        for (int k = 0; k < P; ++k) if (ACT(k)) g[k] += 2.0 * e * cimag(hv[k]) / h;                             \
        ++jsave;                                                                                                \
    } while (0)
    CATH_SAVE(u[k][i], t0);   /* saveat contains tspan[1] */
    static _Thread_local cplx KT[7][18][3];   /* Tsit5 stage slopes of all copies */
    int alg = composite ? 0 : 1, cnt = 0, have_est = 0;
    int64_t n_ts5 = 0;
    double eigen_est = 0.0;
    cath_nl nl;                 /* TRBDF2's nonlinear-solver cache: lives across steps and across algorithm switches */
    cath_nl_init(&nl);
    double EEst_prev = 1.0;     /* integrator.EEst: the last attempt's estimate, whichever algorithm made it (do_newJW's errorfail) */
    double zg_[3] = {0, 0, 0}, z_[3] = {0, 0, 0};
    while (jsave < D) {
        if (++iter > c->maxiters) { retcode = 1; break; }
        if (composite && have_est) {   /* choose_algorithm!, as in solve_one_auto */
            const int stiff = fabs(eigen_est * dt / AS_STAB) > AS_TOL;
            cnt = stiff ? (cnt < 0 ? 1 : cnt + 1) : (cnt > 0 ? -1 : cnt - 1);
            if (alg == 0 && cnt > AS_MAXSTIFF) { dt *= AS_DTFAC; alg = 1; }
            else if (alg == 1 && cnt < -AS_MAXNONSTIFF) {
                dt /= AS_DTFAC; alg = 0;
                /* initialize!(Tsit5 cache): fsalfirst = f(uprev, t) afresh -- TRBDF2 left z/dt there, not a function value */
                if (trbdf2) for (int k = 0; k <= 17; ++k) if (k == PR || ACT(k)) cath_rhs(c, thk[k], u[k], t, f0[k]);
            }
        }
        int last = 0;
        if (t + dt * (1.0 + 1e-13) >= tend) { dt = tend - t; last = 1; }
        if (!(dt > 0.0) || t + dt == t) { retcode = 2; break; }
        const double gam = d * dt;
        cplx Jc[9], ftc[3];
        double W[9]; int piv[3];
        cplx k1[18][3], k2[18][3], k3[18][3], un[18][3], f2[18][3];
        int finite = 1;
        double ev[3], EEst = 0.0;
        int stepfail = 0;
        if (alg == 0 && composite) {
            /* ---- Tsit5 attempt (non-autonomous: stage s at t + c_s dt) ---- */
            for (int pass = 0; pass < 2; ++pass) {
                for (int k = 0; k <= 17; ++k) {
                    if (pass == 0 ? (k != PR) : !ACT(k)) continue;
                    cplx g[3], g6[3] = {0, 0, 0};
                    for (int i = 0; i < 3; ++i) KT[0][k][i] = f0[k][i];
                    for (int s_ = 1; s_ < 7; ++s_) {
                        for (int i = 0; i < 3; ++i) {
                            cplx a = 0.0;
                            for (int j = 0; j < s_; ++j) a += TS_A[s_][j] * KT[j][k][i];
                            g[i] = u[k][i] + dt * a;
                        }
                        if (s_ == 5) for (int i = 0; i < 3; ++i) g6[i] = g[i];
                        if (s_ == 6) for (int i = 0; i < 3; ++i) un[k][i] = g[i];
                        cath_rhs(c, thk[k], g, s_ == 6 ? (last ? tend : t + dt) : t + TS_C[s_] * dt, KT[s_][k]);
                    }
                    for (int i = 0; i < 3; ++i) f2[k][i] = KT[6][k][i];
                    if (k == PR) {
                        double est = 0.0; int isnan_ = 0;
                        for (int i = 0; i < 3; ++i) {
                            cplx a = 0.0;
                            for (int j = 0; j < 7; ++j) a += TS_BT[j] * KT[j][k][i];
                            ev[i] = dt * creal(a);
                            if (!isfinite(creal(un[k][i])) || !isfinite(ev[i])) finite = 0;
                            double q_ = fabs(creal(KT[6][k][i] - KT[5][k][i]) / creal(un[k][i] - g6[i]));
                            if (q_ != q_) isnan_ = 1; else if (q_ > est) est = q_;
                        }
                        eigen_est = isnan_ ? NAN : est;
                    }
                }
                if (pass == 0) {
                    if (!finite) break;
                    double s_ = 0.0;
                    for (int i = 0; i < 3; ++i) { double m = fmax(fabs(creal(u[PR][i])), fabs(creal(un[PR][i]))); double e = ev[i] / (c->atol + c->rtol * m); s_ += e * e; }
                    EEst = sqrt(s_ / 3.0);
                    if (!sens && (!(EEst <= 1.0) || P == 0)) break;
                } else if (sens) {   /* the dual-inclusive norm of the Tsit5 attempt (partials with respect to p_k = theta_k / dir_scale[k]) */
                    double ssum = 0.0;
                    for (int i = 0; i < 3; ++i) {
                        double na = creal(u[PR][i]) * creal(u[PR][i]), nb = creal(un[PR][i]) * creal(un[PR][i]), ee = ev[i] * ev[i];
                        for (int k = 0; k < 17; ++k) if (ACT(k)) {
                            const double sc = c->dir_scale[k] / h;
                            cplx a = 0.0;
                            for (int j = 0; j < 7; ++j) a += TS_BT[j] * KT[j][k][i];
                            const double s_ = sc * cimag(u[k][i]), sn_ = sc * cimag(un[k][i]), de = sc * dt * cimag(a);
                            na += s_ * s_; nb += sn_ * sn_; ee += de * de;
                        }
                        const double scl = c->atol + c->rtol * sqrt(fmax(na, nb));
                        ssum += ee / (scl * scl);
                    }
                    EEst = sqrt(ssum / sens_div);
                    if (!isfinite(EEst)) finite = 0;
                }
            }
        } else if (trbdf2) {
            /* ---- TRBDF2 attempt (primal only; see the block comment above cath_trbdf2_nlsolve) ---- */
            const double s2 = sqrt(2.0), tg = 2.0 - s2, dd = 1.0 - s2 / 2.0, om = s2 / 4.0;
            const double bt1 = (1.0 - s2) / 3.0, bt2 = 1.0 / 3.0, bt3 = (s2 - 2.0) / 3.0, al1 = -s2 / 2.0, al2 = 1.0 + s2 / 2.0;
            double up[3], zp[3], tmp[3];
            cplx zpk[18][3], zgk[18][3], zk[18][3], tmpk[18][3];
            int actv[17];
            for (int k = 0; k < 17; ++k) actv[k] = ACT(k) ? 1 : 0;
            cath_cp cpy = { actv, thk, u, tmpk, zgk };
            const cath_cp *cp = P > 0 ? &cpy : NULL;
            for (int i = 0; i < 3; ++i) { up[i] = creal(u[PR][i]); zp[i] = dt * creal(f0[PR][i]); }
            for (int i = 0; i < 3; ++i) { zg_[i] = zp[i]; tmp[i] = up[i] + dd * zp[i]; }
            if (cp) for (int k = 0; k < 17; ++k) if (actv[k]) for (int i = 0; i < 3; ++i) { zpk[k][i] = dt * f0[k][i]; zgk[k][i] = zpk[k][i]; tmpk[k][i] = u[k][i] + dd * zpk[k][i]; }
            stepfail = cath_trbdf2_nlsolve(c, th, &nl, up, t, dt, iter, EEst_prev, 1, dd, tg, tmp, zg_, &eigen_est, cp) != 0;
            if (!stepfail) {
                for (int i = 0; i < 3; ++i) { z_[i] = al1 * zp[i] + al2 * zg_[i]; tmp[i] = up[i] + om * zp[i] + om * zg_[i]; }
                if (cp) for (int k = 0; k < 17; ++k) if (actv[k]) for (int i = 0; i < 3; ++i) { zk[k][i] = al1 * zpk[k][i] + al2 * zgk[k][i]; tmpk[k][i] = u[k][i] + om * zpk[k][i] + om * zgk[k][i]; }
                cpy.z = zk;
                stepfail = cath_trbdf2_nlsolve(c, th, &nl, up, t, dt, iter, EEst_prev, 0, dd, 1.0, tmp, z_, &eigen_est, cp) != 0;
            }
            if (!stepfail) {
                double est[3], s_ = 0.0;
                if (cp) for (int k = 0; k < 17; ++k) if (actv[k]) for (int i = 0; i < 3; ++i) { un[k][i] = tmpk[k][i] + dd * zk[k][i]; f2[k][i] = zk[k][i] / dt; }
                for (int i = 0; i < 3; ++i) {
                    un[PR][i] = tmp[i] + dd * z_[i];
                    f2[PR][i] = z_[i] / dt;                      /* fsallast = z ./ dt */
                    est[i] = bt1 * zp[i] + bt2 * zg_[i] + bt3 * z_[i];
                }
                lu_solve(3, nl.W, nl.piv, est);                   /* smooth_est: get_W(nlsolver) \ tmp */
                for (int i = 0; i < 3; ++i) {
                    if (c->trbdf2_est == 1) est[i] /= nl.W_gdt;   /* Shampine's (I - gamma dt J)^-1 tmp, up to sign */
                    ev[i] = est[i];
                    if (!isfinite(creal(un[PR][i])) || !isfinite(ev[i])) finite = 0;
                    const double m = fmax(fabs(up[i]), fabs(creal(un[PR][i])));
                    const double e = ev[i] / (c->atol + c->rtol * m);
                    s_ += e * e;
                }
                EEst = sqrt(s_ / 3.0);
                if (sens) {   /* the smoothed estimate's partials: est_k = W^-1 (tmp_k - J_k' est), then the dual-inclusive norm */
                    double ssum = 0.0, na[3], nb[3], ee[3];
                    for (int i = 0; i < 3; ++i) { na[i] = up[i] * up[i]; nb[i] = creal(un[PR][i]) * creal(un[PR][i]); ee[i] = ev[i] * ev[i]; }
                    for (int k = 0; k < 17; ++k) if (actv[k]) {
                        cplx ek[3];
                        /* ev = W \ tmp, or that divided by W_gdt (trbdf2_est = 1): J' multiplies the unscaled solution */
                        for (int i = 0; i < 3; ++i) {
                            ek[i] = bt1 * zpk[k][i] + bt2 * zgk[k][i] + bt3 * zk[k][i];
                            for (int cc = 0; cc < 3; ++cc) ek[i] -= (nl.Jk[k][i + 3 * cc] - creal(nl.Jk[k][i + 3 * cc])) * (c->trbdf2_est == 1 ? ev[cc] * nl.W_gdt : ev[cc]);
                        }
                        csolve3(nl.W, nl.piv, ek);
                        const double sc = c->dir_scale[k] / h;
                        for (int i = 0; i < 3; ++i) {
                            const double de = sc * cimag(ek[i]) / (c->trbdf2_est == 1 ? nl.W_gdt : 1.0);
                            const double s_ = sc * cimag(u[k][i]), sn_ = sc * cimag(un[k][i]);
                            na[i] += s_ * s_; nb[i] += sn_ * sn_; ee[i] += de * de;
                        }
                    }
                    for (int i = 0; i < 3; ++i) { const double scl = c->atol + c->rtol * sqrt(fmax(na[i], nb[i])); ssum += ee[i] / (scl * scl); }
                    EEst = sqrt(ssum / sens_div);
                    if (!isfinite(EEst)) finite = 0;
                }
            }
        } else {
        cath_jac_ft(c, thk[PR], u[PR], t, Jc, ftc);
        {   /* eigen_est of the stiff algorithm: opnorm(J, Inf) */
            double est = 0.0;
            for (int i = 0; i < 3; ++i) { double a = 0.0; for (int cc = 0; cc < 3; ++cc) a += fabs(creal(Jc[i + 3 * cc])); if (a > est) est = a; }
            eigen_est = est;
        }
        for (int cc = 0; cc < 3; ++cc) for (int i = 0; i < 3; ++i) W[i + 3 * cc] = (i == cc ? 1.0 : 0.0) - gam * creal(Jc[i + 3 * cc]);
        if (lu_factor(3, W, piv) != 0) { retcode = 3; break; }
        for (int pass = 0; pass < 2; ++pass) {       /* pass 0: primal (and accept test), pass 1: tangents if accepted */
            for (int k = 0; k <= 17; ++k) {
                if (pass == 0 ? (k != PR) : !ACT(k)) continue;
                cplx Jk[9], ftk[3], b[3], u1[3], f1[3], tmp[3];
                cath_jac_ft(c, thk[k], u[k], t, Jk, ftk);
                /* W_k k1 = f0 + gam ft, with W_k = W + i h W': solve with the real W and move i h W' k1 to the rhs:
                   to first order in h this is one fixed-point sweep: k1 = W^-1 (rhs + gam (J_k - J) k1_real) */
                for (int i = 0; i < 3; ++i) b[i] = f0[k][i] + gam * ftk[i];
                if (k != PR) for (int i = 0; i < 3; ++i) for (int cc = 0; cc < 3; ++cc) b[i] += gam * (Jk[i + 3 * cc] - creal(Jk[i + 3 * cc])) * creal(k1[PR][cc]);
                csolve3(W, piv, b);
                for (int i = 0; i < 3; ++i) { k1[k][i] = b[i]; u1[i] = u[k][i] + 0.5 * dt * b[i]; }
                cath_rhs(c, thk[k], u1, t + 0.5 * dt, f1);
                for (int i = 0; i < 3; ++i) tmp[i] = f1[i] - k1[k][i];
                if (k != PR) for (int i = 0; i < 3; ++i) for (int cc = 0; cc < 3; ++cc) tmp[i] += gam * (Jk[i + 3 * cc] - creal(Jk[i + 3 * cc])) * creal(k2[PR][cc] - k1[PR][cc]);
                csolve3(W, piv, tmp);
                for (int i = 0; i < 3; ++i) { k2[k][i] = tmp[i] + k1[k][i]; un[k][i] = u[k][i] + dt * k2[k][i]; }
                cath_rhs(c, thk[k], un[k], t + dt, f2[k]);
                if (k == PR || sens) {   /* the third stage: the primal's for the error estimate, with errnorm_sens the tangents' too */
                    for (int i = 0; i < 3; ++i) b[i] = f2[k][i] - c32 * (k2[k][i] - f1[i]) - 2.0 * (k1[k][i] - f0[k][i]) + dt * ftk[i];
                    if (k != PR) for (int i = 0; i < 3; ++i) for (int cc = 0; cc < 3; ++cc) b[i] += gam * (Jk[i + 3 * cc] - creal(Jk[i + 3 * cc])) * creal(k3[PR][cc]);
                    csolve3(W, piv, b);
                    for (int i = 0; i < 3; ++i) k3[k][i] = b[i];
                    if (k == PR) for (int i = 0; i < 3; ++i) { ev[i] = dt / 6.0 * creal(k1[k][i] - 2.0 * k2[k][i] + k3[k][i]);
                        if (!isfinite(creal(un[k][i])) || !isfinite(ev[i])) finite = 0; }
                }
            }
            if (pass == 0) {
                if (!finite) break;
                double s_ = 0.0;
                for (int i = 0; i < 3; ++i) { double m = fmax(fabs(creal(u[PR][i])), fabs(creal(un[PR][i]))); double e = ev[i] / (c->atol + c->rtol * m); s_ += e * e; }
                EEst = sqrt(s_ / 3.0);
                if (!sens && (!(EEst <= 1.0) || P == 0)) break;
            } else if (sens) {
                /* the dual-inclusive norm (solve_one_ws): value and the chunk's partials, each partial taken with respect to the
                   normalised parameter p_k = theta_k / dir_scale[k] */
                double ssum = 0.0;
                for (int i = 0; i < 3; ++i) {
                    double na = creal(u[PR][i]) * creal(u[PR][i]), nb = creal(un[PR][i]) * creal(un[PR][i]), ee = ev[i] * ev[i];
                    for (int k = 0; k < 17; ++k) if (ACT(k)) {
                        const double sc = c->dir_scale[k] / h;
                        const double s_ = sc * cimag(u[k][i]), sn_ = sc * cimag(un[k][i]);
                        const double de = sc * dt / 6.0 * cimag(k1[k][i] - 2.0 * k2[k][i] + k3[k][i]);
                        na += s_ * s_; nb += sn_ * sn_; ee += de * de;
                    }
                    const double scl = c->atol + c->rtol * sqrt(fmax(na, nb));
                    ssum += ee / (scl * scl);
                }
                EEst = sqrt(ssum / sens_div);
                if (!isfinite(EEst)) finite = 0;
            }
        }
        }
        have_est = 1;
        if (stepfail) {   /* the Newton iteration failed: force_stepfail -> dt / failfactor, no controller, EEst as it was */
            if (st) st->nreject++;
            dt = dt / 2.0;
            continue;
        }
        if (!finite) { retcode = 3; break; }
        EEst_prev = EEst;
        int accept = (EEst <= 1.0);
        const double b1_ = composite ? (alg == 0 ? 7.0 / 50.0 : 7.0 / 20.0) : c->beta1;
        const double b2_ = composite ? (alg == 0 ? 2.0 / 25.0 : 2.0 / 10.0) : c->beta2;
        double q, q11 = 0.0;
        if (EEst == 0.0) q = 1.0 / c->qmax;
        else { q11 = pow(EEst, b1_); q = q11 / pow(qold, b2_); q = fmax(1.0 / c->qmax, fmin(1.0 / c->qmin, q / c->gamma)); }
        if (accept) {
            if (st) st->naccept++;
            if (alg == 0 && composite) n_ts5++;
            if (q >= c->qsteady_min && q <= c->qsteady_max) q = 1.0;
            qold = fmax(EEst, c->qoldinit);
            double tnew = last ? tend : t + dt;
            while (jsave < D && ts[jsave] <= tnew) {
                double tsv = ts[jsave];
                if (tsv == tnew) { CATH_SAVE(un[k][i], tsv); }
                else if (trbdf2 && alg == 1) {   /* Hermite on (uprev, u, fsalfirst, fsallast) */
                    const double Th = (tsv - t) / dt;
                    CATH_SAVE((1.0 - Th) * u[k][i] + Th * un[k][i] + Th * (Th - 1.0) * ((1.0 - 2.0 * Th) * (un[k][i] - u[k][i])
                              + (Th - 1.0) * dt * f0[k][i] + Th * dt * f2[k][i]), tsv);
                }
                else if (alg == 0 && composite) {
                    double bth[7];
                    orc_tsit5_dense((tsv - t) / dt, bth);
                    CATH_SAVE(u[k][i] + dt * (bth[0] * KT[0][k][i] + bth[1] * KT[1][k][i] + bth[2] * KT[2][k][i] + bth[3] * KT[3][k][i]
                                              + bth[4] * KT[4][k][i] + bth[5] * KT[5][k][i] + bth[6] * KT[6][k][i]), tsv);
                }
                else {
                    double Th = (tsv - t) / dt;
                    double c1 = Th * (1.0 - Th) / (1.0 - 2.0 * d), c2 = Th * (Th - 2.0 * d) / (1.0 - 2.0 * d);
                    CATH_SAVE(u[k][i] + dt * (c1 * k1[k][i] + c2 * k2[k][i]), tsv);
                }
            }
            for (int k = 0; k <= 17; ++k) if (k == PR || ACT(k)) for (int i = 0; i < 3; ++i) { u[k][i] = un[k][i]; f0[k][i] = f2[k][i]; }
            t = tnew;
            dt = fmin(dt / q, tend - t0);
        } else {
            if (st) st->nreject++;
            dt = dt / fmin(1.0 / c->qmin, q11 / c->gamma);
        }
    }
#undef CATH_SAVE
#undef ACT
    /* loss = sum(...)/n_replicas/size(exp_data)[1]: divided by the FULL number of rows (network.jl:266) */
    if (loss_out) *loss_out = loss_sum / (double)D;
    if (grad) for (int k = 0; k < 17; ++k) grad[k] = g[k] / (double)D;
    if (n_saved_out) *n_saved_out = jsave;
    orc_cathode_last_tsit5_steps = n_ts5;
    orc_cathode_last_nl[0] = nl.n_newton; orc_cathode_last_nl[1] = nl.n_jac; orc_cathode_last_nl[2] = nl.n_w; orc_cathode_last_nl[3] = nl.n_fail;
    return retcode;
}

/* AutoSwitch census (round 3): every (particle, heating rate) trajectory of an ensemble through the composite restatement
   (c0->solver = 2), primal only, OpenMP over trajectories.  out = { trajectories, those whose accepted steps were ALL
   Tsit5 steps (the stiff branch never ran), failed solves, sum of accepted steps, sum of accepted Tsit5 steps, largest
   accepted-step count }.  tools/cathode_autoswitch_census.py runs it on BASELINE config 5's 4 096 x 256 ensemble. */
int orc_cathode_census(const orc_cathode *c0, const double *theta /*[n_part][17]*/, int64_t n_part, const double *beta /*[n_sets]*/,
                       int n_sets, const double *ts /*[n_sets][Dmax]*/, const int32_t *D, int Dmax, int64_t *out /*[6]*/, int nthreads) {
    int64_t n_never = 0, n_fail = 0, s_acc = 0, s_ts5 = 0, m_acc = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    double *zero = (double *)calloc((size_t)Dmax, sizeof(double));
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : n_never, n_fail, s_acc, s_ts5) reduction(max : m_acc)
    for (int64_t tr = 0; tr < n_part * n_sets; ++tr) {
        const int64_t ip = tr / n_sets;
        const int is = (int)(tr % n_sets);
        orc_cathode c = *c0;
        c.beta = beta[is];
        double loss = 0;
        int32_t nsv = 0;
        orc_stats st = {0, 0};
        const int rc = orc_cathode_solve_one(&c, theta + 17 * ip, ts + (size_t)is * Dmax, D[is], zero, zero, NULL, &loss, NULL, &nsv, &st);
        const int64_t t5 = orc_cathode_last_tsit5_steps;
        if (rc != 0) ++n_fail;
        if (t5 == st.naccept) ++n_never;
        s_acc += st.naccept; s_ts5 += t5;
        if (st.naccept > m_acc) m_acc = st.naccept;
    }
    free(zero);
    out[0] = n_part * n_sets; out[1] = n_never; out[2] = n_fail; out[3] = s_acc; out[4] = s_ts5; out[5] = m_acc;
    return 0;
}

/* ======================================================================================================== *
 * HyChem pyrolysis (BASELINE config 4): HyChem/crnn_pyrolysis_mass.jl
 *   p2vec                       :78-90
 *   Y2density, Y2C, crnn!       :107-131     (T = itpT(t), P = itpP(t): LinearInterpolation on tsteps, :103-104)
 *   predict_n_ode / loss_n_ode  :135-147     (saveat = tsteps[1:sample], mae(pred ./ yscale, data ./ yscale))
 *   ForwardDiff.gradient        :201
 * Stepper: non-autonomous Rosenbrock23 (solver 0: the reference's AutoTsit5(Rosenbrock23(autodiff=false)) in its stiff branch)
 * or that composite itself (solver 2), dT = df/dt analytic on the current table segment (the reference takes a finite difference).  Tangents: complex step
 * per direction (theta + i h dtheta_k, u + i h s_k) through the same step arithmetic with the real W factorisation,
 * dt held real -- ForwardDiff's arithmetic, and deliberately a different mechanism from the device's analytic adjoint.
 * PARITY UNPINNED for solver internals (no Manifest, no reference tests); pinned against Radau + sensitivities of
 * the NumPy restatement (tests/golden/fixtures_hychem.json).
 * theta = [ w_in ((ns+2) x nr col-major; row ns: x (-1/(R T)), row ns+1: x log T) | w_b | w_out (ns x nr) ].
 * ======================================================================================================== */
typedef struct orc_hychem {
    int32_t ns, nr, maxiters;
    int32_t solver;   /* 0 Rosenbrock23; 2 AutoTsit5(Rosenbrock23(autodiff=false)) -- the reference's `ode_solver`
                         (crnn_pyrolysis_mass.jl:29, used :138-139): Tsit5 with stage times t + c_s dt on the T/P tables, the
                         AutoSwitch rule of solve_one_auto, Rosenbrock23 (analytic J: the reference's autodiff=false takes finite
                         differences) as the stiff algorithm; set qsteady_max = 1 with it (a composite is not an implicit type) */
    int32_t errnorm_sens, dual_partials;   /* errnorm_sens != 0 (Rosenbrock23, and the AutoTsit5(Rosenbrock23) composite): the `ndir` directions of a call are ONE ForwardDiff chunk
                         (ForwardDiff.gradient(x -> loss_n_ode(x, sample), p), crnn_pyrolysis_mass.jl:201: 211 parameters in chunks of 12)
                         whose partials weigh in the error norm (solve_one_ws has the formula; 1: / length(u), 2: / totallength(u) with
                         dual_partials partials per Dual -- the zero-padded ones of the last chunk count) */
    int32_t jac_fd;   /* 1: the stiff algorithm as the reference configures it, Rosenbrock23(autodiff = false) (crnn_pyrolysis_mass.jl:29): J by
                         FiniteDiff's forward differences of the right-hand side (orc_jac_fd has the increment) AND the time derivative
                         dT = (f(u, t + e_t) - f(u, t)) / e_t, e_t = max(sqrt(eps) |t|, sqrt(eps)), on the T(t), P(t) tables; primal solves
                         (ndir = 0) with solver 0 or 2 [UNVERIFIED-DEP] */
    double lb, ub, inv_R, Ru, atol, rtol;
    double mw[12], scale[12], inv_yscale[12];
    double gamma, qmin, qmax, beta1, beta2, qsteady_min, qsteady_max, qoldinit;
} orc_hychem;

int orc_sizeof_hychem(void) { return (int)sizeof(orc_hychem); }

void orc_hychem_defaults(orc_hychem *c) {
    memset(c, 0, sizeof(*c));
    c->ns = 9; c->nr = 10; c->maxiters = 10000;                                   /* :21-23,55 */
    c->lb = 1e-8; c->ub = 10.0; c->atol = 1e-8; c->rtol = 1e-3;                   /* :26-28,123,126 */
    c->inv_R = (double)(-1.0f / 1.98720425864083e-3f);                            /* Float32 literal, :106,128 */
    c->Ru = 8.31446261815324e3;                                                   /* :108 */
    const double mw[9] = {136.238, 2.016, 16.043, 26.038, 28.054, 28.014, 56.108, 1.008, 15.035};   /* :58 */
    for (int i = 0; i < 12; ++i) { c->mw[i] = i < 9 ? mw[i] : 1.0; c->scale[i] = 1.0; c->inv_yscale[i] = 1.0; }
    c->gamma = 0.9; c->qmin = 0.2; c->qmax = 10.0; c->beta1 = 7.0 / 20.0; c->beta2 = 2.0 / 10.0;
    c->qsteady_min = 1.0; c->qsteady_max = 1.2; c->qoldinit = 1e-4;
}

/* p -> theta and d theta / d p (dth[k * nth + m], zeroed here); ForwardDiff's clamp' = 1 on the closed window */
int orc_hychem_p2vec(const double *p, int ns, int nr, double *th, double *dth) {
    const int n = ns + 2, nth = nr * (n + 1 + ns), P = nr * (2 * ns + 3) + 1;
    const double slope = p[P - 1] * 10.0, ln10 = log(10.0);
    if (dth) memset(dth, 0, sizeof(double) * (size_t)nth * P);
#define DTH(m, k) dth[(size_t)(k) * nth + (m)]
    const int o_b = n * nr, o_out = (n + 1) * nr;
    for (int j = 0; j < nr; ++j) {
        th[o_b + j] = p[j] * slope;                                   /* w_b */
        th[(ns + 1) + n * j] = p[nr + j];                             /* w_in_b -> row ns+1 */
        th[ns + n * j] = p[2 * nr + j] * slope;                       /* w_in_Ea -> row ns */
        if (dth) {
            DTH(o_b + j, j) = slope; DTH(o_b + j, P - 1) = p[j] * 10.0;
            DTH((ns + 1) + n * j, nr + j) = 1.0;
            DTH(ns + n * j, 2 * nr + j) = slope; DTH(ns + n * j, P - 1) = p[2 * nr + j] * 10.0;
        }
        for (int i = 0; i < ns; ++i) {
            const int ko = 3 * nr + i + ns * j, ki = nr * (ns + 3) + i + ns * j;
            const double wo = p[ko], wi = p[ki], pw = pow(10.0, wo);
            th[o_out + i + ns * j] = -wi * pw;
            th[i + n * j] = wi < 0.0 ? 0.0 : (wi > 2.5 ? 2.5 : wi);
            if (dth) {
                DTH(o_out + i + ns * j, ki) = -pw;
                DTH(o_out + i + ns * j, ko) = -wi * pw * ln10;
                DTH(i + n * j, ki) = (wi >= 0.0 && wi <= 2.5) ? 1.0 : 0.0;
            }
        }
    }
#undef DTH
    return 0;
}

static cplx hy_clamp(cplx x, double lo, double hi) { double re = creal(x); return re < lo ? lo : (re > hi ? hi : x); }

/* table segment of t: i = max{ i : ts[i] <= t }, clipped to [0, D-2] */
static int hy_seg(const double *ts, int D, double t) {
    int i = 0;
    while (i + 1 < D - 1 && ts[i + 1] <= t) ++i;
    return i;
}
static void hy_tab(const double *ts, int D, const double *tab, double t, double *val, double *slope) {
    int i = hy_seg(ts, D, t);
    double s = (tab[i + 1] - tab[i]) / (ts[i + 1] - ts[i]);
    *val = tab[i] + (t - ts[i]) * s;
    if (slope) *slope = s;
}

/* f, and optionally J = df/du (col-major ns x ns) and ft = df/dt for given (T, P, dT/dt, dP/dt); complex-analytic */
static void hy_eval(const orc_hychem *c, const cplx *th, const cplx *u, double T, double P, double Td, double Pd,
                    cplx *f, cplx *J, cplx *ft) {
    const int ns = c->ns, nr = c->nr, n = ns + 2;
    const cplx *w_in = th, *w_b = th + n * nr, *w_out = th + (n + 1) * nr;
    cplx Y[12], x[14], r[16], om[12];
    double cY[12], cC[12];
    cplx S = 0.0;
    for (int i = 0; i < ns; ++i) { Y[i] = hy_clamp(u[i], c->lb, c->ub); cY[i] = (creal(u[i]) >= c->lb && creal(u[i]) <= c->ub) ? 1.0 : 0.0; S += Y[i] / c->mw[i]; }
    const cplx rho = P / (c->Ru * T * S);
    for (int i = 0; i < ns; ++i) {
        cplx C = rho * (Y[i] / c->mw[i]) * 1e3;
        cC[i] = (creal(C) >= c->lb && creal(C) <= c->ub) ? 1.0 : 0.0;
        x[i] = clog(hy_clamp(C, c->lb, c->ub));
    }
    x[ns] = c->inv_R / T;
    x[ns + 1] = log(T);
    for (int j = 0; j < nr; ++j) {
        cplx z = w_b[j];
        for (int m = 0; m < n; ++m) z += w_in[m + n * j] * x[m];
        r[j] = cexp(z);
    }
    for (int i = 0; i < ns; ++i) {
        cplx a = 0.0;
        for (int j = 0; j < nr; ++j) a += w_out[i + ns * j] * r[j];
        om[i] = a;
        f[i] = a * c->mw[i] / rho * c->scale[i];
    }
    if (J) {
        /* d log rho / du_c = -cY_c / (mw_c S);  d x_i / du_c = cC_i (d log rho/du_c + delta_ic cY_c / Y_c) */
        for (int cc = 0; cc < ns; ++cc) {
            const cplx dl = -cY[cc] / (c->mw[cc] * S);
            for (int i = 0; i < ns; ++i) {
                cplx a = 0.0;
                for (int j = 0; j < nr; ++j) {
                    cplx dz = 0.0;
                    for (int m = 0; m < ns; ++m) dz += w_in[m + n * j] * cC[m] * (dl + (m == cc ? cY[cc] / Y[cc] : 0.0));
                    a += w_out[i + ns * j] * r[j] * dz;
                }
                J[i + ns * cc] = a * c->mw[i] / rho * c->scale[i] - f[i] * dl;
            }
        }
    }
    if (ft) {
        const double ld = Pd / P - Td / T;           /* d log rho / dt */
        for (int i = 0; i < ns; ++i) {
            cplx a = 0.0;
            for (int j = 0; j < nr; ++j) {
                cplx dz = w_in[ns + n * j] * (-c->inv_R * Td / (T * T)) + w_in[ns + 1 + n * j] * (Td / T);
                for (int m = 0; m < ns; ++m) dz += w_in[m + n * j] * cC[m] * ld;
                a += w_out[i + ns * j] * r[j] * dz;
            }
            ft[i] = a * c->mw[i] / rho * c->scale[i] - f[i] * ld;
        }
    }
    (void)om;
}

/* Closed-form tangents of the HyChem right-hand side in REAL arithmetic -- the formulas a hand-written dual-norm kernel would carry instead of
   nested dual numbers (hychem_sens_kernel.hpp evaluates the right-hand side over Du<Du<double>>: 5.2 KB of scratch per lane), checked against
   the complex-step evaluation above (tests/test_hychem.py).  For a direction (su, dth) at the point (u, T, P, Td, Pd):
     fp    = f'(u; su, dth)                                     (first-order tangent)
     mixed = d/d(su, dth) [ J(u) v + tau ft(u) ]                (the stage equations' W' = -gam J' and d_t f' terms; v, tau held fixed)
   With Y = clamp(u), S = sum Y_i / M_i, L = log rho = log P - log(Ru T S), x_m = log clamp(1e3 rho Y_m / M_m), a_m / cY_m the clamp
   indicators, sY = cY su:   L' = -S'/S;  x_m' = a_m (L' + sY_m / Y_m);  z_j' = dwb_j + sum_m dwi_mj x_m + wi_mj x_m';  r_j' = r_j z_j';
     f_i = K_i om_i / rho,  om_i = sum_j wo_ij r_j:      f_i' = K_i / rho (sum_j dwo_ij r_j + wo_ij r_j') - f_i L'
     (J v)_i = K_i / rho A_i - f_i Lv,  Lv = -sum_c cY_c v_c / (M_c S),  A_i = sum_j wo_ij r_j zv_j,  zv_j = sum_m wi_mj xv_m,
               xv_m = a_m (Lv + cY_m v_m / Y_m):         Lv' = Lv L';  xv_m' = a_m (Lv L' - v_m sY_m / Y_m^2);
               (J v)_i' = K_i / rho (A_i' - A_i L') - f_i' Lv - f_i Lv'
     ft_i = K_i / rho B_i - f_i ld,  ld = Pd/P - Td/T,  B_i = sum_j wo_ij r_j zt_j,  zt_j = wi_Ej e1 + wi_Lj e2 + ld sum_m wi_mj a_m:
               ft_i' = K_i / rho (B_i' - B_i L') - f_i' ld   (e1 = -inv_R Td / T^2, e2 = Td / T; the indicators are piecewise constant) */
void orc_hychem_tangents(const orc_hychem *c, const double *th, const double *dth, const double *u, const double *su, const double *v,
                         double tau, double T, double P, double Td, double Pd, double *fp, double *mixed) {
    const int ns = c->ns, nr = c->nr, n = ns + 2;
    const double *wi = th, *wb = th + n * nr, *wo = th + (n + 1) * nr;
    const double *dwi = dth, *dwb = dth + n * nr, *dwo = dth + (n + 1) * nr;
    double Y[12], cY[12], a[12], x[14], xp[12], xv[12], xvp[12], sY[12];
    double S = 0.0, Sp = 0.0, Sv = 0.0;
    for (int i = 0; i < ns; ++i) {
        cY[i] = (u[i] >= c->lb && u[i] <= c->ub) ? 1.0 : 0.0;
        Y[i] = u[i] < c->lb ? c->lb : (u[i] > c->ub ? c->ub : u[i]);
        sY[i] = cY[i] * su[i];
        S += Y[i] / c->mw[i]; Sp += sY[i] / c->mw[i]; Sv += cY[i] * v[i] / c->mw[i];
    }
    const double rho = P / (c->Ru * T * S), Lp = -Sp / S, Lv = -Sv / S, Lvp = Lv * Lp;
    for (int m = 0; m < ns; ++m) {
        const double C = rho * (Y[m] / c->mw[m]) * 1e3;
        a[m] = (C >= c->lb && C <= c->ub) ? 1.0 : 0.0;
        x[m] = log(C < c->lb ? c->lb : (C > c->ub ? c->ub : C));
        xp[m] = a[m] * (Lp + sY[m] / Y[m]);
        xv[m] = a[m] * (Lv + cY[m] * v[m] / Y[m]);
        xvp[m] = a[m] * (Lvp - v[m] * sY[m] / (Y[m] * Y[m]));
    }
    x[ns] = c->inv_R / T; x[ns + 1] = log(T);
    const double ld = Pd / P - Td / T, e1 = -c->inv_R * Td / (T * T), e2 = Td / T;
    double r[16], rp[16], zv[16], zvp[16], zt[16], ztp[16];
    for (int j = 0; j < nr; ++j) {
        double z = wb[j], zp = dwb[j], sa = 0.0, dsa = 0.0;
        zv[j] = 0.0; zvp[j] = 0.0;
        for (int m = 0; m < n; ++m) { z += wi[m + n * j] * x[m]; zp += dwi[m + n * j] * x[m]; }
        for (int m = 0; m < ns; ++m) {
            zp += wi[m + n * j] * xp[m];
            zv[j] += wi[m + n * j] * xv[m];
            zvp[j] += dwi[m + n * j] * xv[m] + wi[m + n * j] * xvp[m];
            sa += wi[m + n * j] * a[m]; dsa += dwi[m + n * j] * a[m];
        }
        r[j] = exp(z); rp[j] = r[j] * zp;
        zt[j] = wi[ns + n * j] * e1 + wi[ns + 1 + n * j] * e2 + ld * sa;
        ztp[j] = dwi[ns + n * j] * e1 + dwi[ns + 1 + n * j] * e2 + ld * dsa;
    }
    for (int i = 0; i < ns; ++i) {
        double om = 0.0, omp = 0.0, A = 0.0, Ap = 0.0, B = 0.0, Bp = 0.0;
        for (int j = 0; j < nr; ++j) {
            const double w = wo[i + ns * j], dw = dwo[i + ns * j];
            om += w * r[j]; omp += dw * r[j] + w * rp[j];
            A += w * r[j] * zv[j]; Ap += dw * r[j] * zv[j] + w * (rp[j] * zv[j] + r[j] * zvp[j]);
            B += w * r[j] * zt[j]; Bp += dw * r[j] * zt[j] + w * (rp[j] * zt[j] + r[j] * ztp[j]);
        }
        const double K = c->mw[i] * c->scale[i] / rho;
        const double f = K * om, fpi = K * omp - f * Lp;
        fp[i] = fpi;
        const double Jvp = K * (Ap - A * Lp) - fpi * Lv - f * Lvp;
        const double ftp = K * (Bp - B * Lp) - fpi * ld;
        mixed[i] = Jvp + tau * ftp;
    }
}

/* the same two quantities by the complex step through hy_eval (the arithmetic the solver's tangents use) */
void orc_hychem_tangents_cs(const orc_hychem *c, const double *th, const double *dth, const double *u, const double *su, const double *v,
                            double tau, double T, double P, double Td, double Pd, double *fp, double *mixed) {
    const int ns = c->ns, nth = c->nr * (2 * ns + 3);
    const double h = 1e-30;
    cplx thc[256], uc[12] = {0}, fc[12], Jc[144], ftc[12];
    for (int m = 0; m < nth; ++m) thc[m] = th[m] + I * h * dth[m];
    for (int i = 0; i < ns; ++i) uc[i] = u[i] + I * h * su[i];
    hy_eval(c, thc, uc, T, P, Td, Pd, fc, Jc, ftc);
    for (int i = 0; i < ns; ++i) {
        cplx jv = 0.0;
        for (int cc = 0; cc < ns; ++cc) jv += Jc[i + ns * cc] * v[cc];
        fp[i] = cimag(fc[i]) / h;
        mixed[i] = cimag(jv) / h + tau * cimag(ftc[i]) / h;
    }
}

void orc_hychem_rhs(const orc_hychem *c, const double *th, const double *u, double T, double P, double Td, double Pd,
                    double *f, double *J, double *ft) {
    const int ns = c->ns, nth = c->nr * (2 * ns + 3);
    cplx thc[256], uc[12], fc[12], Jc[144], ftc[12];
    for (int m = 0; m < nth; ++m) thc[m] = th[m];
    for (int i = 0; i < ns; ++i) uc[i] = u[i];
    hy_eval(c, thc, uc, T, P, Td, Pd, fc, J ? Jc : NULL, ft ? ftc : NULL);
    for (int i = 0; i < ns; ++i) { f[i] = creal(fc[i]); if (ft) ft[i] = creal(ftc[i]); }
    if (J) for (int i = 0; i < ns * ns; ++i) J[i] = creal(Jc[i]);
}

static void hy_csolve(int n, const double *W, const int *piv, cplx *b) {   /* real LU applied to a complex rhs */
    for (int k = 0; k < n; ++k) { int p = piv[k]; if (p != k) { cplx tt = b[k]; b[k] = b[p]; b[p] = tt; } }
    for (int k = 0; k < n; ++k) { cplx a = b[k]; for (int i = k + 1; i < n; ++i) b[i] -= W[i + n * k] * a; }
    for (int k = n - 1; k >= 0; --k) { b[k] /= W[k + n * k]; cplx a = b[k]; for (int i = 0; i < k; ++i) b[i] -= W[i + n * k] * a; }
}

_Thread_local int64_t orc_hychem_last_tsit5_steps = 0;   /* accepted Tsit5 steps of the calling thread's last solve (composite) */
int64_t orc_hychem_tsit5_steps(void) { return orc_hychem_last_tsit5_steps; }
/* u0 [ns]; ts, Ttab, Ptab [D]; data [ns][D] (species-major rows of length Dfull); pred [ns][Dfull] or NULL;
 * dth [ndir][nth] or NULL; grad [ndir].  Returns the retcode. */
int orc_hychem_solve_one(const orc_hychem *c, const double *th, const double *dth, int ndir, const double *u0,
                         const double *ts, int D, int Dfull, const double *Ttab, const double *Ptab, const double *data,
                         double *pred, double *loss_out, double *grad, int32_t *n_saved_out, orc_stats *st) {
    const int ns = c->ns, nth = c->nr * (2 * ns + 3), K = ndir + 1, PR = ndir;
    const double d = 1.0 / (2.0 + sqrt(2.0)), c32 = 6.0 + sqrt(2.0), h = 1e-30;
    const double t0 = 0.0, tend = ts[D - 1];                          /* tspan = [0, tsteps[sample]], :137 */
    const int sens = (c->errnorm_sens != 0 && ndir > 0);
    if (c->jac_fd && ndir > 0) return -7;                             /* tangents through the finite-difference quotients are not restated here */
    if (sens && c->solver != 0 && c->solver != 2) return -1;          /* the dual-inclusive norm is restated for Rosenbrock23 and for the AutoTsit5(Rosenbrock23) composite */
    const double sens_div = c->errnorm_sens == 2 ? (double)ns * (1.0 + (double)c->dual_partials) : (double)ns;
    cplx *thk = (cplx *)malloc(sizeof(cplx) * (size_t)K * nth);
    cplx *ws = (cplx *)malloc(sizeof(cplx) * (size_t)K * ns * 7);
    cplx *u = ws, *f0 = ws + K * ns, *k1 = ws + 2 * K * ns, *k2 = ws + 3 * K * ns, *un = ws + 4 * K * ns, *f2 = ws + 5 * K * ns, *k3 = ws + 6 * K * ns;
    double *g = (double *)calloc((size_t)(ndir > 0 ? ndir : 1), sizeof(double));
    for (int k = 0; k < K; ++k) {
        for (int m = 0; m < nth; ++m) thk[(size_t)k * nth + m] = th[m] + (k < ndir ? I * h * dth[(size_t)k * nth + m] : 0.0);
        for (int i = 0; i < ns; ++i) u[k * ns + i] = u0[i];
    }
    double t = t0, Tn, Pn, Tdn, Pdn;
    hy_tab(ts, Dfull, Ttab, t, &Tn, &Tdn); hy_tab(ts, Dfull, Ptab, t, &Pn, &Pdn);
    for (int k = 0; k < K; ++k) hy_eval(c, thk + (size_t)k * nth, u + k * ns, Tn, Pn, 0, 0, f0 + k * ns, NULL, NULL);
    double dt;
    {   /* Hairer initial step, order 2, on the primal */
        double sk[12], d0 = 0, d1 = 0, d2 = 0, ur[12], fr[12], f1r[12];
        cplx u1[12] = {0}, f1[12] = {0};
        for (int i = 0; i < ns; ++i) { ur[i] = creal(u[PR * ns + i]); fr[i] = creal(f0[PR * ns + i]); sk[i] = c->atol + fabs(ur[i]) * c->rtol;
            d0 += (ur[i] / sk[i]) * (ur[i] / sk[i]); d1 += (fr[i] / sk[i]) * (fr[i] / sk[i]); }
        if (sens) for (int k = 0; k < ndir; ++k) for (int i = 0; i < ns; ++i) { double e = cimag(f0[k * ns + i]) / h / sk[i]; d1 += e * e; }
        const double dv = sens ? sens_div : (double)ns;
        d0 = sqrt(d0 / dv); d1 = sqrt(d1 / dv);
        const double dtmax = tend - t0;
        double dt0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
        dt0 = fmin(dt0, dtmax);
        for (int i = 0; i < ns; ++i) u1[i] = ur[i] + dt0 * fr[i];
        double T1, P1; hy_tab(ts, Dfull, Ttab, t + dt0, &T1, NULL); hy_tab(ts, Dfull, Ptab, t + dt0, &P1, NULL);
        hy_eval(c, thk + (size_t)PR * nth, u1, T1, P1, 0, 0, f1, NULL, NULL);
        for (int i = 0; i < ns; ++i) { f1r[i] = creal(f1[i]); double e = (f1r[i] - fr[i]) / sk[i]; d2 += e * e; }
        if (sens) for (int k = 0; k < ndir; ++k) {   /* f1 = f(u0 + dt0 f0) with Duals: the tangent of u1 is dt0 f0' */
            cplx u1k[12], f1k[12];
            for (int i = 0; i < ns; ++i) u1k[i] = u[k * ns + i] + dt0 * f0[k * ns + i];
            hy_eval(c, thk + (size_t)k * nth, u1k, T1, P1, 0, 0, f1k, NULL, NULL);
            for (int i = 0; i < ns; ++i) { double e = cimag(f1k[i] - f0[k * ns + i]) / h / sk[i]; d2 += e * e; }
        }
        d2 = sqrt(d2 / dv) / dt0;
        double dm = fmax(d1, d2);
        double dt1 = dm <= 1e-15 ? fmax(1e-6, dt0 * 1e-3) : pow(10.0, -(2.0 + log10(dm)) / (c->solver == 2 ? 5.0 : 2.0));   /* order of the starting algorithm */
        dt = fmin(fmin(100 * dt0, dt1), dtmax);
    }
    double qold = c->qoldinit, loss_sum = 0.0;
    int jsave = 0, retcode = 0, iter = 0;
    const int composite = c->solver == 2;
    int alg = composite ? 0 : 1, cnt = 0, have_est = 0;
    int64_t n_ts5 = 0;
    double eigen_est = 0.0;
    cplx *KT = composite ? (cplx *)malloc(sizeof(cplx) * 7 * (size_t)K * ns) : NULL;   /* Tsit5 stage slopes of all copies: KT[(s * K + k) * ns + i] */
#define HY_SAVE(UEXPR)                                                                                           \
    do {                                                                                                         \
        for (int i = 0; i < ns; ++i) {                                                                           \
            int k = PR; cplx vp = (UEXPR);                                                                       \
            double rr = (data[(size_t)i * Dfull + jsave] - creal(vp)) * c->inv_yscale[i];                        \
            if (pred) pred[(size_t)i * Dfull + jsave] = creal(vp);                                               \
            loss_sum += fabs(rr);                                                                                \
            double w = (signbit(rr) ? 1.0 : -1.0) * c->inv_yscale[i];                                            \
            for (k = 0; k < ndir; ++k) { cplx vk = (UEXPR); g[k] += w * cimag(vk) / h; }                         \
        }                                                                                                        \
        ++jsave;                                                                                                 \
    } while (0)
    if (ts[0] == t0) HY_SAVE(u[k * ns + i]);
    double *W = (double *)malloc(sizeof(double) * ns * ns);
    cplx *Jk = (cplx *)malloc(sizeof(cplx) * ns * ns);
    int piv[12];
    while (jsave < D) {
        if (++iter > c->maxiters) { retcode = 1; break; }
        if (composite && have_est) {   /* choose_algorithm!, as in solve_one_auto */
            const int stiff = fabs(eigen_est * dt / AS_STAB) > AS_TOL;      /* false for NaN */
            cnt = stiff ? (cnt < 0 ? 1 : cnt + 1) : (cnt > 0 ? -1 : cnt - 1);
            if (alg == 0 && cnt > AS_MAXSTIFF) { dt *= AS_DTFAC; alg = 1; }
            else if (alg == 1 && cnt < -AS_MAXNONSTIFF) { dt /= AS_DTFAC; alg = 0; }
        }
        int last = 0;
        if (t + dt * (1.0 + 1e-13) >= tend) { dt = tend - t; last = 1; }
        if (!(dt > 0.0) || t + dt == t) { retcode = 2; break; }
        const double gam = d * dt, tm = t + 0.5 * dt, tnew = last ? tend : t + dt;
        double Tm, Pm, T2, P2;
        int finite = 1;
        double ev[12], EEst = 0.0;
        cplx ftk[12], fdum[12];
        if (alg == 0 && composite) {
            /* ---- Tsit5 attempt: stage s at t + c_s dt on the tables (the seventh at the step's end time) ---- */
            for (int pass = 0; pass < 2; ++pass) {
                for (int k = 0; k < K; ++k) {
                    if (pass == 0 ? (k != PR) : !(k < ndir)) continue;
                    cplx gk[12], g6[12];
                    const cplx *thc = thk + (size_t)k * nth;
                    for (int i = 0; i < ns; ++i) { KT[((size_t)0 * K + k) * ns + i] = f0[k * ns + i]; g6[i] = 0.0; }
                    for (int s_ = 1; s_ < 7; ++s_) {
                        for (int i = 0; i < ns; ++i) {
                            cplx a = 0.0;
                            for (int j = 0; j < s_; ++j) a += TS_A[s_][j] * KT[((size_t)j * K + k) * ns + i];
                            gk[i] = u[k * ns + i] + dt * a;
                        }
                        if (s_ == 5) for (int i = 0; i < ns; ++i) g6[i] = gk[i];
                        if (s_ == 6) for (int i = 0; i < ns; ++i) un[k * ns + i] = gk[i];
                        const double tq = s_ == 6 ? tnew : t + TS_C[s_] * dt;
                        double Tq, Pq;
                        hy_tab(ts, Dfull, Ttab, tq, &Tq, NULL); hy_tab(ts, Dfull, Ptab, tq, &Pq, NULL);
                        hy_eval(c, thc, gk, Tq, Pq, 0, 0, KT + ((size_t)s_ * K + k) * ns, NULL, NULL);
                    }
                    for (int i = 0; i < ns; ++i) f2[k * ns + i] = KT[((size_t)6 * K + k) * ns + i];
                    if (k == PR) {
                        double est = 0.0; int isnan_ = 0;
                        for (int i = 0; i < ns; ++i) {
                            cplx a = 0.0;
                            for (int j = 0; j < 7; ++j) a += TS_BT[j] * KT[((size_t)j * K + k) * ns + i];
                            ev[i] = dt * creal(a);
                            if (!isfinite(creal(un[k * ns + i])) || !isfinite(ev[i])) finite = 0;
                            const double q_ = fabs(creal(KT[((size_t)6 * K + k) * ns + i] - KT[((size_t)5 * K + k) * ns + i]) / creal(un[k * ns + i] - g6[i]));
                            if (q_ != q_) isnan_ = 1; else if (q_ > est) est = q_;
                        }
                        eigen_est = isnan_ ? NAN : est;
                    }
                }
                if (pass == 0) {
                    if (!finite) break;
                    double s_ = 0.0;
                    for (int i = 0; i < ns; ++i) { double m = fmax(fabs(creal(u[PR * ns + i])), fabs(creal(un[PR * ns + i]))); double e = ev[i] / (c->atol + c->rtol * m); s_ += e * e; }
                    EEst = sqrt(s_ / ns);
                    if (!sens && (!(EEst <= 1.0) || ndir == 0)) break;
                } else if (sens) {   /* the dual-inclusive norm of the Tsit5 attempt: the embedded error estimate's partials are dt sum_j bt_j k_j' */
                    double ssum = 0.0;
                    for (int i = 0; i < ns; ++i) {
                        double na = creal(u[PR * ns + i]) * creal(u[PR * ns + i]), nb = creal(un[PR * ns + i]) * creal(un[PR * ns + i]), ee = ev[i] * ev[i];
                        for (int k = 0; k < ndir; ++k) {
                            const double s_ = cimag(u[k * ns + i]) / h, sn_ = cimag(un[k * ns + i]) / h;
                            cplx a = 0.0;
                            for (int j = 0; j < 7; ++j) a += TS_BT[j] * KT[((size_t)j * K + k) * ns + i];
                            const double de = dt * cimag(a) / h;
                            na += s_ * s_; nb += sn_ * sn_; ee += de * de;
                        }
                        const double scl = c->atol + c->rtol * sqrt(fmax(na, nb));
                        ssum += ee / (scl * scl);
                    }
                    EEst = sqrt(ssum / sens_div);
                    if (!isfinite(EEst)) finite = 0;
                }
            }
        } else {
        hy_tab(ts, Dfull, Ttab, t, &Tn, &Tdn); hy_tab(ts, Dfull, Ptab, t, &Pn, &Pdn);
        hy_tab(ts, Dfull, Ttab, tm, &Tm, NULL); hy_tab(ts, Dfull, Ptab, tm, &Pm, NULL);
        hy_tab(ts, Dfull, Ttab, tnew, &T2, NULL); hy_tab(ts, Dfull, Ptab, tnew, &P2, NULL);
        hy_eval(c, thk + (size_t)PR * nth, u + PR * ns, Tn, Pn, Tdn, Pdn, fdum, Jk, ftk);
        double ftfd[12];
        if (c->jac_fd) {   /* Rosenbrock23(autodiff = false): J and dT by forward differences (FiniteDiff's default increments) */
            const double rel = 1.4901161193847656e-08;
            cplx up[12], fp[12];
            for (int i = 0; i < ns; ++i) up[i] = creal(u[PR * ns + i]);
            for (int cc = 0; cc < ns; ++cc) {
                const double uc = creal(u[PR * ns + cc]), eps = fmax(rel * fabs(uc), rel);
                up[cc] = uc + eps;
                hy_eval(c, thk + (size_t)PR * nth, up, Tn, Pn, 0, 0, fp, NULL, NULL);
                for (int i = 0; i < ns; ++i) Jk[i + ns * cc] = (creal(fp[i]) - creal(f0[PR * ns + i])) / eps;
                up[cc] = uc;
            }
            const double et = fmax(rel * fabs(t), rel);
            double Te, Pe;
            hy_tab(ts, Dfull, Ttab, t + et, &Te, NULL); hy_tab(ts, Dfull, Ptab, t + et, &Pe, NULL);
            hy_eval(c, thk + (size_t)PR * nth, up, Te, Pe, 0, 0, fp, NULL, NULL);
            for (int i = 0; i < ns; ++i) ftfd[i] = (creal(fp[i]) - creal(f0[PR * ns + i])) / et;
        }
        {   /* eigen_est of the stiff algorithm: opnorm(J, Inf) */
            double est = 0.0;
            for (int i = 0; i < ns; ++i) { double a = 0.0; for (int cc = 0; cc < ns; ++cc) a += fabs(creal(Jk[i + ns * cc])); if (a > est) est = a; }
            eigen_est = est;
        }
        for (int i = 0; i < ns * ns; ++i) W[i] = ((i % ns) == (i / ns) ? 1.0 : 0.0) - gam * creal(Jk[i]);
        if (lu_factor(ns, W, piv) != 0) { retcode = 3; break; }
        for (int pass = 0; pass < 2; ++pass) {
            for (int k = 0; k < K; ++k) {
                if (pass == 0 ? (k != PR) : !(k < ndir)) continue;
                cplx b[12], u1[12], f1[12], tmp[12];
                const cplx *thc = thk + (size_t)k * nth;
                if (!c->jac_fd) hy_eval(c, thc, u + k * ns, Tn, Pn, Tdn, Pdn, fdum, Jk, ftk);
                else for (int i = 0; i < ns; ++i) ftk[i] = ftfd[i];     /* primal only: Jk already holds the difference quotients */
                for (int i = 0; i < ns; ++i) b[i] = f0[k * ns + i] + gam * ftk[i];
                if (k != PR) for (int i = 0; i < ns; ++i) for (int cc = 0; cc < ns; ++cc)
                    b[i] += gam * (Jk[i + ns * cc] - creal(Jk[i + ns * cc])) * creal(k1[PR * ns + cc]);
                hy_csolve(ns, W, piv, b);
                for (int i = 0; i < ns; ++i) { k1[k * ns + i] = b[i]; u1[i] = u[k * ns + i] + 0.5 * dt * b[i]; }
                hy_eval(c, thc, u1, Tm, Pm, 0, 0, f1, NULL, NULL);
                for (int i = 0; i < ns; ++i) tmp[i] = f1[i] - k1[k * ns + i];
                if (k != PR) for (int i = 0; i < ns; ++i) for (int cc = 0; cc < ns; ++cc)
                    tmp[i] += gam * (Jk[i + ns * cc] - creal(Jk[i + ns * cc])) * creal(k2[PR * ns + cc] - k1[PR * ns + cc]);
                hy_csolve(ns, W, piv, tmp);
                for (int i = 0; i < ns; ++i) { k2[k * ns + i] = tmp[i] + k1[k * ns + i]; un[k * ns + i] = u[k * ns + i] + dt * k2[k * ns + i]; }
                hy_eval(c, thc, un + k * ns, T2, P2, 0, 0, f2 + k * ns, NULL, NULL);
                if (k == PR || sens) {   /* the third stage: the primal's for the error estimate, with errnorm_sens the tangents' too */
                    for (int i = 0; i < ns; ++i) b[i] = f2[k * ns + i] - c32 * (k2[k * ns + i] - f1[i]) - 2.0 * (k1[k * ns + i] - f0[k * ns + i]) + dt * ftk[i];
                    if (k != PR) for (int i = 0; i < ns; ++i) for (int cc = 0; cc < ns; ++cc)
                        b[i] += gam * (Jk[i + ns * cc] - creal(Jk[i + ns * cc])) * creal(k3[PR * ns + cc]);
                    hy_csolve(ns, W, piv, b);
                    for (int i = 0; i < ns; ++i) k3[k * ns + i] = b[i];
                    if (k == PR) for (int i = 0; i < ns; ++i) { ev[i] = dt / 6.0 * creal(k1[k * ns + i] - 2.0 * k2[k * ns + i] + b[i]);
                        if (!isfinite(creal(un[k * ns + i])) || !isfinite(ev[i])) finite = 0; }
                }
            }
            if (pass == 0) {
                if (!finite) break;
                double s_ = 0.0;
                for (int i = 0; i < ns; ++i) { double m = fmax(fabs(creal(u[PR * ns + i])), fabs(creal(un[PR * ns + i]))); double e = ev[i] / (c->atol + c->rtol * m); s_ += e * e; }
                EEst = sqrt(s_ / ns);
                if (!sens && (!(EEst <= 1.0) || ndir == 0)) break;
            } else if (sens) {   /* the dual-inclusive norm (solve_one_ws): value and the chunk's partials */
                double ssum = 0.0;
                for (int i = 0; i < ns; ++i) {
                    double na = creal(u[PR * ns + i]) * creal(u[PR * ns + i]), nb = creal(un[PR * ns + i]) * creal(un[PR * ns + i]), ee = ev[i] * ev[i];
                    for (int k = 0; k < ndir; ++k) {
                        const double s_ = cimag(u[k * ns + i]) / h, sn_ = cimag(un[k * ns + i]) / h;
                        const double de = dt / 6.0 * cimag(k1[k * ns + i] - 2.0 * k2[k * ns + i] + k3[k * ns + i]) / h;
                        na += s_ * s_; nb += sn_ * sn_; ee += de * de;
                    }
                    const double scl = c->atol + c->rtol * sqrt(fmax(na, nb));
                    ssum += ee / (scl * scl);
                }
                EEst = sqrt(ssum / sens_div);
                if (!isfinite(EEst)) finite = 0;
            }
        }
        }
        have_est = 1;
        if (!finite) { retcode = 3; break; }
        const int accept = (EEst <= 1.0);
        const double b1_ = composite ? (alg == 0 ? 7.0 / 50.0 : 7.0 / 20.0) : c->beta1;
        const double b2_ = composite ? (alg == 0 ? 2.0 / 25.0 : 2.0 / 10.0) : c->beta2;
        double q, q11 = 0.0;
        if (EEst == 0.0) q = 1.0 / c->qmax;
        else { q11 = pow(EEst, b1_); q = q11 / pow(qold, b2_); q = fmax(1.0 / c->qmax, fmin(1.0 / c->qmin, q / c->gamma)); }
        if (accept) {
            if (st) st->naccept++;
            if (alg == 0 && composite) n_ts5++;
            if (q >= c->qsteady_min && q <= c->qsteady_max) q = 1.0;
            qold = fmax(EEst, c->qoldinit);
            while (jsave < D && ts[jsave] <= tnew) {
                const double tsv = ts[jsave];
                if (tsv == tnew) { HY_SAVE(un[k * ns + i]); }
                else if (alg == 0 && composite) {
                    double bth[7];
                    orc_tsit5_dense((tsv - t) / dt, bth);
#define KTI(s_) KT[((size_t)(s_) * K + k) * ns + i]
                    HY_SAVE(u[k * ns + i] + dt * (bth[0] * KTI(0) + bth[1] * KTI(1) + bth[2] * KTI(2) + bth[3] * KTI(3) + bth[4] * KTI(4) + bth[5] * KTI(5) + bth[6] * KTI(6)));
#undef KTI
                }
                else {
                    const double Th = (tsv - t) / dt;
                    const double c1 = Th * (1.0 - Th) / (1.0 - 2.0 * d), c2 = Th * (Th - 2.0 * d) / (1.0 - 2.0 * d);
                    HY_SAVE(u[k * ns + i] + dt * (c1 * k1[k * ns + i] + c2 * k2[k * ns + i]));
                }
            }
            for (int k = 0; k < K; ++k) if (k == PR || k < ndir) for (int i = 0; i < ns; ++i) { u[k * ns + i] = un[k * ns + i]; f0[k * ns + i] = f2[k * ns + i]; }
            t = tnew;
            dt = fmin(dt / q, tend - t0);
        } else {
            if (st) st->nreject++;
            dt = dt / fmin(1.0 / c->qmin, q11 / c->gamma);
        }
    }
#undef HY_SAVE
    const double den = (double)ns * (double)jsave;
    if (loss_out) *loss_out = jsave > 0 ? loss_sum / den : 0.0;
    if (grad) for (int k = 0; k < ndir; ++k) grad[k] = jsave > 0 ? g[k] / den : 0.0;
    if (n_saved_out) *n_saved_out = jsave;
    orc_hychem_last_tsit5_steps = n_ts5;
    free(W); free(Jk); free(thk); free(ws); free(g); free(KT);
    return retcode;
}

/* ======================================================================== */
/* SVGD move of the Bayesian cathode ensemble.  TEST INFRASTRUCTURE.        */
/*   svgd_kernel   Cathode_NCM333_UQ/src_333/network.jl:67-87               */
/*   update        Cathode_NCM333_UQ/src_333/crnn_cathode.jl:36-50          */
/* p, lnpgrad, p_new, data_term, repulsion: [N x dim] row-major.            */
/* Scalar loops, dense N x N matrices, qsort for the median (Julia's        */
/* `median` of an even-length vector is the mean of the middle pair).       */
/* ======================================================================== */
static int svgd_cmp_double(const void *a, const void *b) {
    const double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

int orc_svgd_update(const double *p, const double *lnpgrad, int N, int dim, double stepsize, double h,
                    double *p_new, double *h_out, double *data_term /* may be NULL */, double *repulsion /* may be NULL */) {
    if (N < 2 || dim < 1) return -1;
    double *dist = (double *)malloc(sizeof(double) * (size_t)N * N);           /* pairwise(Euclidean(), p, dims=1)   :68 */
    double *K = (double *)malloc(sizeof(double) * (size_t)N * N);
    double *dxk = (double *)malloc(sizeof(double) * (size_t)N * dim);
    double *dat = (double *)malloc(sizeof(double) * (size_t)N * dim);
    if (!dist || !K || !dxk || !dat) { free(dist); free(K); free(dxk); free(dat); return -2; }
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            double s = 0.0;
            for (int k = 0; k < dim; ++k) { const double d = p[(size_t)i * dim + k] - p[(size_t)j * dim + k]; s += d * d; }
            dist[(size_t)i * N + j] = sqrt(s);
        }
    if (h < 0) {                                                                /* median trick                      :72-75 */
        const size_t npairs = (size_t)N * (N - 1) / 2;
        double *low = (double *)malloc(sizeof(double) * npairs);               /* strictly lower triangle           :69 */
        if (!low) { free(dist); free(K); free(dxk); free(dat); return -2; }
        size_t q = 0;
        for (int j = 0; j < N; ++j) for (int i = j + 1; i < N; ++i) low[q++] = dist[(size_t)i * N + j];
        qsort(low, npairs, sizeof(double), svgd_cmp_double);
        const double med = (npairs & 1) ? low[npairs / 2] : 0.5 * (low[npairs / 2 - 1] + low[npairs / 2]);
        free(low);
        h = med * med;
        h = sqrt(0.5 * h / log((double)N + 1.0));
    }
    if (h_out) *h_out = h;
    for (size_t q = 0; q < (size_t)N * N; ++q) K[q] = exp(-(dist[q] * dist[q]) / (h * h) / 2.0);   /* Kxy       :77 */
    for (int i = 0; i < N; ++i) {
        double sumk = 0.0;                                                      /* sum(Kxy, dims=2)                 :80 */
        for (int j = 0; j < N; ++j) sumk += K[(size_t)i * N + j];
        for (int k = 0; k < dim; ++k) {
            double kp = 0.0, kg = 0.0;
            for (int j = 0; j < N; ++j) {
                kp += K[(size_t)i * N + j] * p[(size_t)j * dim + k];             /* Kxy * p                          :79 */
                kg += K[(size_t)i * N + j] * lnpgrad[(size_t)j * dim + k];       /* kxy * lnpgrad   crnn_cathode.jl:41 */
            }
            dxk[(size_t)i * dim + k] = (-kp + p[(size_t)i * dim + k] * sumk) / (h * h);   /* :82-85 */
            dat[(size_t)i * dim + k] = kg;
        }
    }
    for (size_t q = 0; q < (size_t)N * dim; ++q) {
        const double gcur = (dat[q] + dxk[q]) / (double)N;                       /* grad_p_curr     crnn_cathode.jl:43 */
        p_new[q] = p[q] + stepsize * gcur;                                       /*                 crnn_cathode.jl:49 */
        if (data_term) data_term[q] = dat[q];
        if (repulsion) repulsion[q] = dxk[q];
    }
    free(dist); free(K); free(dxk); free(dat);
    return 0;
}

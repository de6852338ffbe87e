/*
 * include/crnn_hip.h -- C ABI of libcrnn_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the neural-ODE hot path of DENG-MIT/CRNN.  The
 * reference has no FFI of its own (it is a set of Julia scripts); the boundary
 * is cut at the `solve(prob, alg, u0=u0, p=p)` call inside `predict_neuralode`
 * (reference case2/case2.jl:126, robertson/rober_crnn.jl:125-127,
 * case1/case1.jl:94-95) and at `ForwardDiff.gradient(x -> loss_neuralode(x,
 * i_exp), p)` + `update!(opt, p, grad)` (case2/case2.jl:195-197,
 * robertson/rober_crnn.jl:219-224).  Every entry point below names the
 * reference lines it replaces.  A Julia host binds these with `ccall`
 * (julia/CRNNHip.jl); the Python host in crnn_amd/ binds them
 * with ctypes.
 *
 * Conventions
 *   - plain pointers and sizes only; all floating point is IEEE double;
 *   - every function returns int32: 0 = ok, <0 = error (crnn_last_error());
 *     solver failures of individual trajectories are NOT errors: they are
 *     reported per trajectory in retcode[] / n_saved[] exactly as the reference
 *     prints "ode solver failed" and carries on with the truncated solution
 *     (robertson/rober_crnn.jl:130-134);
 *   - host arrays are caller-owned and only read/written during the call;
 *     device buffers are library-owned behind the opaque crnn_ctx;
 *   - a ctx is bound to one HIP device; use one ctx per GPU / per process;
 *   - batched arrays are "IC-fastest" (species-major), matching Julia's
 *     column-major u0_list[n_exp, ns+1], ode_data_list[n_exp, ns, datasize]
 *     (case2/case2.jl:62,69):
 *        u0  [i*B + b]                 i < n
 *        data[(j*n_obs + i)*B + b]     j < n_save, i < n_obs
 *        pred[(j*n     + i)*B + b]     j < n_save, i < n
 *   - theta = [ w_in (n x nr, column-major) | w_b (nr) | w_out (ns x nr, column-major) ],
 *     n = ns + has_temp, n_theta = nr*(n + 1 + ns): the "effective weights"
 *     p2vec returns (case2/case2.jl:91-99).
 */
#ifndef CRNN_HIP_H
#define CRNN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* v4 (round 4): + crnn_cathode_set_solver, crnn_cathode_set_errnorm_sens, crnn_cathode_last_chunk_stats; crnn_cathode_set_tape_every
 * takes 2; CRNN_SOLVER_AUTOTSIT5 is accepted on the HyChem preset; the lb of the case1 / case2 presets is the Float32 value.  No
 * struct layout changed. */
/* v5 (round 6): + crnn_tape_retries (added in round 5 without a bump), crnn_hychem_block_cap.  No struct layout changed. */
#define CRNN_ABI_VERSION 5
#define CRNN_MAX_N 12   /* max ODE states  */
#define CRNN_MAX_NR 16  /* max reactions   */

/* p2vec variants (parameter vector p -> effective weights theta) */
enum {
    CRNN_PMAP_IDENTITY = 0, /* p IS theta */
    CRNN_PMAP_CASE1 = 1,    /* case1/case1.jl:70-78 */
    CRNN_PMAP_CASE2 = 2,    /* case2/case2.jl:91-99 */
    CRNN_PMAP_ROBER = 3,    /* robertson/rober_crnn.jl:85-96 */
    CRNN_PMAP_HYCHEM = 4    /* HyChem/crnn_pyrolysis_mass.jl:78-90 */
};
/* right-hand-side families */
enum {
    CRNN_RHS_CRNN = 0,      /* du = scale .* (w_out exp(w_in' [log clamp(u); inv_R/T] + w_b))  (case1, case2, robertson) */
    CRNN_RHS_HYCHEM = 1     /* HyChem/crnn_pyrolysis_mass.jl:121-131: mass fractions, density coupling, T(t)/P(t) tables;
                               feature rows = [log clamp(C) (ns); -1/(R T); log T], w_in is (ns+2) x nr */
};
enum { CRNN_LOSS_MAE = 0, CRNN_LOSS_MSE = 1 };
/* per-trajectory return codes (DiffEq retcodes Success / MaxIters / DtLessThanMin / Unstable) */
enum { CRNN_RET_SUCCESS = 0, CRNN_RET_MAXITERS = 1, CRNN_RET_DTMIN = 2, CRNN_RET_UNSTABLE = 3 };
/* time steppers (reference: alg = Rosenbrock23(...) rober_crnn.jl:33, Tsit5() case1.jl:28, AutoTsit5(Rosenbrock23())
 * case2.jl:26).  AUTOTSIT5 is OrdinaryDiffEq's stiffness-switching composite with its default AutoSwitch constants
 * (10 stiff / 3 non-stiff steps in a row, tolerances 9/10, dt factor 2, Tsit5 stability size 3.5068); it exists for the
 * discrete-adjoint gradient only (grad_mode AUTO or ADJOINT).  [UNVERIFIED-DEP: the switching rule is restated from
 * OrdinaryDiffEq's published algorithm, the package is not in the reference tree.]  Under that restatement a state vector
 * with a component that never moves -- the constant temperature of has_temp = 1 -- makes the stiffness estimate 0/0 = NaN,
 * and such a problem never leaves Tsit5 (crnn_amd/csrc/auto_adj_kernel.hpp). */
enum { CRNN_SOLVER_ROSENBROCK23 = 0, CRNN_SOLVER_TSIT5 = 1, CRNN_SOLVER_AUTOTSIT5 = 2 };
/* How the loss gradient (ForwardDiff.gradient, case2/case2.jl:195) is formed.  Both differentiate the accepted steps
 * with dt held fixed and agree to rounding:
 *   FORWARD  P tangent columns pushed through every step (cost ~ (1+P) primal solves; any stepper);
 *   ADJOINT  the accepted steps are recorded on a tape and reversed once (cost ~ 2 primal solves; every stepper);
 *   AUTO     ADJOINT where available, else FORWARD. */
enum { CRNN_GRAD_AUTO = 0, CRNN_GRAD_FORWARD = 1, CRNN_GRAD_ADJOINT = 2 };
/* internal to the adjoint path: a trajectory ran out of tape; the library then repeats the call with FORWARD and this
 * code never reaches the caller */
enum { CRNN_RET_TAPE_OVERFLOW = 5 };
/* presets for crnn_config_preset.  NOTE on CASE2: the reference script asks for AutoTsit5(Rosenbrock23(autodiff=false))
 * (case2/case2.jl:26); the preset selects CRNN_SOLVER_ROSENBROCK23, the stepper BASELINE.json's headline is quoted on.
 * crnn_config_set_solver(cfg, CRNN_SOLVER_AUTOTSIT5) selects the composite (on case2 it never leaves Tsit5: see above). */
enum { CRNN_PRESET_CASE1 = 1, CRNN_PRESET_CASE2 = 2, CRNN_PRESET_ROBER = 3, CRNN_PRESET_HYCHEM = 4 };

/* Problem descriptor: what `ODEProblem(crnn, u0, tspan; saveat, atol, rtol)` plus
 * the script-top constants carry in the reference (case2/case2.jl:15-35,113,120-121). */
typedef struct crnn_config {
    int32_t abi_version;          /* = CRNN_ABI_VERSION */
    int32_t ns, nr;               /* species, reactions */
    int32_t has_temp;             /* 1: trailing temperature state with du_T = 0 (case2) */
    int32_t param_map;            /* CRNN_PMAP_* used by the p-level entry points */
    int32_t n_save;               /* D = length(tsteps) */
    int32_t loss_kind;            /* CRNN_LOSS_* */
    int32_t clamp_pred;           /* pred = clamp.(Array(sol), -ub, ub) (case1/2) */
    int32_t maxiters;             /* solver iterations (accepted+rejected) per trajectory */
    int32_t errnorm_sens;         /* error norm of the step-size controller in GRADIENT calls.  0: primal values only (a gradient
                                     call takes the step sequence of a plain solve; every gradient algorithm).  1: ForwardDiff's
                                     dual-inclusive norm, chunked like ForwardDiff.pickchunksize -- what the reference's
                                     ForwardDiff.gradient through the adaptive solver does (case2/case2.jl:195); Rosenbrock23 and
                                     Tsit5 -- and AUTOTSIT5 on a shape with a temperature state (case2's own `alg`, :26), where the
                                     composite never leaves Tsit5 and the Tsit5 kernel runs --, forward tangents (grad_mode AUTO or FORWARD); round 4: also the HyChem right-hand side
                                     (crnn_pyrolysis_mass.jl:201: 211 parameters in 18 chunks of 12) -- round 5: at speed
                                     (hychem_sens2_kernel: the rows of p2vec's Jacobian are sparse, one reaction each; a caller's
                                     dense directions run the general hychem_sens_kernel), and with solver = AUTOTSIT5 through the
                                     reference's own composite, the partials in both algorithms' norms (:29).  crnn_solve then
                                     treats its n_dir directions as ONE chunk (n_dir <= 9 case2, 12 case1 / robertson / HyChem);
                                     crnn_loss_grad / crnn_train_step run ForwardDiff's chunks, loss and statistics from a
                                     final plain solve.  The squared norm is divided by length(u) (early DiffEqBase 6).  2: the
                                     same with totallength(u) = n (1 + partials per Dual) as divisor.  The reference pins no
                                     DiffEqBase version for case1 / case2 / robertson / HyChem, but its case2 checkpoint says
                                     which form ran: replaying the recorded training history from the script's seeded RNG stream
                                     lands within 5e-4 per epoch with mode 2 and 1e-2 off with mode 1
                                     (tests/test_case2_stream_pin.py); the cathode Manifest pins a version with form 2 as well.
                                     Use 2 to reproduce the reference. */
    int32_t device;               /* HIP device ordinal */
    int32_t cols_per_lane;        /* kernel tuning: tangent columns per lane, 0 = auto */
    int32_t solver;               /* CRNN_SOLVER_*; set it with crnn_config_set_solver (also sets the controller) */
    int32_t grad_mode;            /* CRNN_GRAD_* */
    int32_t tape_steps;           /* adjoint: accepted steps recordable per trajectory; 0 = auto (min(maxiters, memory budget)) */
    int32_t rhs_kind;             /* CRNN_RHS_* */
    double lb, ub;                /* log(clamp(u, lb, ub)); ub may be +inf */
    double inv_R;                 /* -1/R (case2/case2.jl:113); unused when has_temp = 0 */
    double t0;                    /* tspan[1] */
    double atol[CRNN_MAX_N];      /* per state (robertson uses a vector, rober_crnn.jl:34) */
    double rtol[CRNN_MAX_N];
    double rate_scale[CRNN_MAX_N];/* dydt_scale (rober_crnn.jl:82,115; crnn_pyrolysis_mass.jl:120); 1 otherwise */
    double mw[CRNN_MAX_N];        /* HyChem: molar masses l_MW (crnn_pyrolysis_mass.jl:58) */
    double gas_const;             /* HyChem: 8.31446261815324e3 J/(kmol K) (:108) */
    /* PI step-size controller (OrdinaryDiffEq defaults for Rosenbrock23) */
    double gamma, qmin, qmax, beta1, beta2, qsteady_min, qsteady_max, qoldinit, dtmin;
} crnn_config;

typedef struct crnn_stats {
    int64_t n_traj;     /* trajectories integrated by this call */
    int64_t n_ok;       /* with retcode == CRNN_RET_SUCCESS */
    int64_t n_accept;   /* accepted Rosenbrock steps, summed */
    int64_t n_reject;   /* rejected steps, summed */
    double kernel_ms;   /* HIP-event time of the solve kernel(s) on the ctx stream */
} crnn_stats;

/* Flux.Optimise chain  Optimiser(ExpDecay?, ADAM, WeightDecay) with optional
 * gradient-norm clipping in front (case2/case2.jl:31-32; rober_crnn.jl:19,221-223). */
typedef struct crnn_opt_config {
    int32_t use_expdecay;   /* 1: ExpDecay(ed_eta0, ed_decay, decay_step, ed_clip) first */
    int32_t decay_step;
    double ed_eta0, ed_decay, ed_clip;
    double eta, beta1, beta2; /* ADAM */
    double wd;                /* WeightDecay (ADAMW's third argument) */
    double grad_clip_norm;    /* > 0: grad <- grad/|grad|*clip when |grad| > clip */
} crnn_opt_config;

typedef struct crnn_ctx crnn_ctx;

int32_t crnn_abi_version(void);
/* "src=<16 hex digits> arch=gfx950": sha256 prefix of the sources (the sorted .hip and .hpp files of crnn_amd/csrc, then this header)
 * the loaded binary was compiled from -- a host can refuse (or rebuild) a stale library (crnn_amd/_lib.py does). */
const char *crnn_build_info(void);
/* Diagnostic builds only (-DCRNN_BOUNDS_CHECK: every indexed access of the adjoint kernels checked against its extent, see
 * ros23_adj_kernel.hpp): number of violations since the library was loaded and the site code of the first one; resets both.
 * A release build has no checks compiled in and returns -1 without touching the outputs. */
int32_t crnn_debug_bounds(uint32_t *violations, uint32_t *first_site);
/* sizeof(crnn_config) / crnn_stats / crnn_opt_config / crnn_cathode_config for which = 0 / 1 / 2 / 3: lets a binding
 * (Julia struct, ctypes.Structure) verify its mirror of the C structs at load time. */
int32_t crnn_sizeof(int32_t which);
const char *crnn_last_error(const crnn_ctx *ctx); /* ctx may be NULL: last error of a failed create */

/* Fill cfg with the reference's script-top constants for a preset
 * (case1/case1.jl:14-35, case2/case2.jl:15-35,113, robertson/rober_crnn.jl:19-37). */
int32_t crnn_config_preset(crnn_config *cfg, int32_t preset);
int32_t crnn_opt_preset(crnn_opt_config *o, int32_t preset);
/* Select the time stepper and the PI-controller defaults OrdinaryDiffEq attaches to it
 * (beta1 = 7/(10 q), beta2 = 2/(5 q), q = 2 | 5; steady band [1, 6/5] only for the implicit family). */
int32_t crnn_config_set_solver(crnn_config *cfg, int32_t solver);

/* ---- host-side parameter maps: p2vec and its Jacobian ------------------- */
int32_t crnn_n_params(int32_t param_map, int32_t ns, int32_t nr);  /* length of p */
/* length of theta for a config: nr * (ns + extra + 1 + ns), extra = has_temp (CRNN) or 2 (HyChem's -1/(RT), log T rows);
 * crnn_n_theta(ns, nr, extra) takes that number of extra feature rows as its third argument */
int32_t crnn_config_n_theta(const crnn_config *cfg);
int32_t crnn_n_theta(int32_t ns, int32_t nr, int32_t has_temp);
/* theta[n_theta]; dtheta[n_theta x n_params] column-major, may be NULL.
 * Replaces p2vec (case2/case2.jl:91-99 etc.) and the part of
 * ForwardDiff.gradient that differentiates through it. */
int32_t crnn_p2vec(int32_t param_map, int32_t ns, int32_t nr, const double *p,
                   double *theta, double *dtheta);

/* ---- context ------------------------------------------------------------ */
int32_t crnn_ctx_create(const crnn_config *cfg, crnn_ctx **out);
void crnn_ctx_destroy(crnn_ctx *ctx);
/* Run all ctx work on a caller-owned hipStream_t (e.g. torch's current stream). */
int32_t crnn_ctx_set_stream(crnn_ctx *ctx, void *hip_stream);
/* Upload the ensemble: u0_list, ode_data_list, tsteps, yscale, i_obs
 * (case2/case2.jl:62-83,131).  i_obs: 0-based, NULL = all species. */
int32_t crnn_ctx_set_data(crnn_ctx *ctx, const double *u0, const double *data,
                          const double *tsteps, const double *yscale,
                          const int32_t *i_obs, int32_t n_obs, int64_t B);
/* Same, from buffers already resident on ctx's device (no PCIe copy). */
int32_t crnn_ctx_set_data_device(crnn_ctx *ctx, const void *d_u0, const void *d_data,
                                 const double *tsteps, const double *yscale,
                                 const int32_t *i_obs, int32_t n_obs, int64_t B);
/* HyChem: per-trajectory temperature [K] and pressure [Pa] tables on the saveat grid, IC-fastest like data:
 * T[j*B + b], j < n_save (Tlist / Plist, crnn_pyrolysis_mass.jl:44-47,103-104; piecewise linear in t).
 * Call after crnn_ctx_set_data (B is taken from it). */
int32_t crnn_ctx_set_tables(crnn_ctx *ctx, const double *T, const double *P);
/* In which order the adjoint kernels take the ensemble's trajectories from their queue (one lane per trajectory, 64 per
 * wavefront).  AUTO (default): once a launch over the range has run, by that launch's step counts, longest first -- the
 * 64 trajectories of a wavefront then take about the same number of steps, and the steps of one iteration hold about the
 * same number of save points (case2 at 65 536: -7 % kernel time, at 4 096-32 768: -10...-20 %, AutoTsit5 robertson -25 %,
 * HyChem -12 %; ensembles larger than the resident lanes: -30 %).  Per-trajectory results do not depend on the order; the
 * batch sums (gradient, mean loss) do, in their last bits, through the composition of the 64-trajectory partial sums.
 * INDEX: always in index order -- the sums are then a function of the call's inputs alone (a run and its restart from a
 * checkpoint agree bit for bit: examples/case2_train.c). */
enum { CRNN_QUEUE_AUTO = 0, CRNN_QUEUE_INDEX = 1 };
int32_t crnn_ctx_set_queue_order(crnn_ctx *ctx, int32_t order);
/* How many lanes work on one trajectory in the Rosenbrock23 adjoint kernel.  1: one lane per trajectory (ros23_adj_kernel:
 * the least total work, the kernel for ensembles that fill the chip).  2: an adjacent lane pair per trajectory
 * (ros23_adj2_kernel: each lane holds half of the species, their weight rows and their gradient accumulators; species sums
 * cross the pair with one DPP step; ~45 % fewer instructions on a step's critical path) -- for shards smaller than the
 * chip, e.g. one GPU's share of a strongly-scaled batch, where the second lane of a pair would otherwise be idle.
 * 0 = AUTO (default): 2 where a two-lane instantiation exists for the shape (nr < ns, no rate scaling: case1, case2) and the
 * pairs fit the resident lanes (32 768 trajectories on 256 CUs), else 1.  Beyond that size two lanes still pay where step
 * counts spread widely (case2 at trained parameters, 65 536 trajectories: -4 %) and cost where they do not (-13 % at the
 * reference's initialiser): the host's call.  HyChem contexts: 1 = hychem_kernel, 2 (and AUTO, at every ensemble size) =
 * hychem2_kernel: a lane pair per trajectory, every vector and W's rows distributed over the pair, and the batch gradient
 * summed on chip (FP64 MFMAs over each batch of 32 trajectories) instead of in 210 HBM accumulators per trajectory.  Same
 * derivative either way; results agree to rounding (the species sums and the batch sums are formed in a different order),
 * per-trajectory outputs do not depend on the launch geometry. */
int32_t crnn_ctx_set_lanes_per_traj(crnn_ctx *ctx, int32_t lanes);
/* Lanes per trajectory the most recent adjoint gradient launch used (1 or 2; 0 if no adjoint launch has run, -1 for a null ctx). */
int32_t crnn_last_lanes_per_traj(const crnn_ctx *ctx);
/* HyChem: gradient launches this context had to repeat with fewer resident trajectories because a trajectory accepted more steps than
 * its share of the adjoint tape holds (tape sized automatically, crnn_config.tape_steps = 0).  The width that fitted is remembered for
 * calls of the same shape (and re-probed: crnn_hychem_block_cap), so a count that grows by more than one in 16 calls says the tape budget is
 * too small for this ensemble: set crnn_config.tape_steps.  0 if it never happened; -1 for a null ctx. */
int64_t crnn_tape_retries(const crnn_ctx *ctx);
/* HyChem: the resident-block limit gradient launches over the last overflowing range currently start from (0: none, full width; -1 for a
 * null ctx).  It is only a starting point: every 16th launch served from it tries four times the width again, so a limit set by one hard
 * parameter vector does not outlive it; crnn_ctx_set_data clears it. */
int32_t crnn_hychem_block_cap(const crnn_ctx *ctx);
/* Jacobian behind W = I - gam J of the Rosenbrock23 stepper in PRIMAL launches (crnn_solve with n_dir = 0: predict_neuralode,
 * loss_neuralode, the epoch-end loop).  ANALYTIC (default): the exact J -- what Rosenbrock23(autodiff = true) forms
 * (robertson/rober_crnn.jl:33).  FINITE_DIFF: what Rosenbrock23(autodiff = false) forms (the stiff algorithm inside case2's
 * AutoTsit5(Rosenbrock23(autodiff=false)), case2/case2.jl:26 -- a context with solver = AUTOTSIT5 has NO finite-difference
 * variant, the mode is refused there; robertson/rober_crnn_lm.jl:34): forward differences of the right-hand side, column c = (f(u + eps_c e_c) - f(u)) / eps_c with
 * eps_c = max(sqrt(eps) |u_c|, sqrt(eps)) (FiniteDiff.jl's default step for forward differences, restated -- FiniteDiff is
 * not vendored with the reference), ns more right-hand sides per attempt and a dense ns x ns factorisation.  Rosenbrock23 is a
 * W-method, so both are Rosenbrock23 solves of the same problem: results differ by ~1e-8 relative in J; in a loss at the
 * reference's tolerances by ~2e-9 on case2 and 3e-4 ... 1.3e-3 on robertson (stiffness 1e11: tests/test_gpu_primal.py).  Gradient launches are not affected: they differentiate the analytic-W step (the reference
 * pushes Duals through FiniteDiff's increments).  CRNN right-hand side: Rosenbrock23 contexts only.
 * HyChem contexts (round 5; HyChem/crnn_pyrolysis_mass.jl:29, AutoTsit5(Rosenbrock23(autodiff=false))): the primal launches of a
 * Rosenbrock23 or an AUTOTSIT5 context form J column by column from ns more right-hand sides AND the time derivative
 * dT = (f(u, t + e_t) - f(u, t)) / e_t, e_t = max(sqrt(eps) |t|, sqrt(eps)), on the T(t), P(t) tables (hychem_auto_kernel<..., JFD>):
 * a parity mode, ~3x the right-hand sides of an attempt; it moves a loss at the reference's tolerances by ~3e-4 on hot trajectories
 * (1e-5 inside the composite, 5e-9 at rtol 1e-8; tests/test_hychem.py). */
enum { CRNN_JAC_ANALYTIC = 0, CRNN_JAC_FINITE_DIFF = 1 };
int32_t crnn_ctx_set_jacobian(crnn_ctx *ctx, int32_t mode);

/* ---- the hot path at theta level ---------------------------------------- *
 * Integrates trajectories [first, first+count) of the uploaded ensemble with
 * the adaptive Rosenbrock23 stepper up to tsteps[n_save_active-1]
 * (robertson's random horizon, rober_crnn.jl:125,218), evaluates the loss and,
 * if n_dir > 0, pushes n_dir forward tangents (columns of dtheta, column-major
 * n_theta x n_dir) through every step -- what ForwardDiff.gradient does to the
 * solver (case2/case2.jl:195).
 *   pred  [n_save x n x B]  or NULL   predict_neuralode (case2/case2.jl:124-128)
 *   loss  [B]               or NULL   loss_neuralode    (case2/case2.jl:132-137)
 *   grad  [n_dir]           or NULL   SUM over the count trajectories of d loss_b / d dir
 *   retcode, n_saved [B]    or NULL   (entries outside [first, first+count) untouched)
 */
int32_t crnn_solve(crnn_ctx *ctx, const double *theta, const double *dtheta, int32_t n_dir,
                   int64_t first, int64_t count, int32_t n_save_active,
                   double *pred, double *loss, double *grad,
                   int32_t *retcode, int32_t *n_saved, crnn_stats *stats);

/* ---- the hot path at p level (device-resident p2vec + solve + reduction) - *
 * loss_mean = mean_b loss_neuralode(p, b), grad_p = d loss_mean / d p.       */
int32_t crnn_loss_grad(crnn_ctx *ctx, const double *p, int64_t first, int64_t count,
                       int32_t n_save_active, double *loss_mean, double *grad_p,
                       crnn_stats *stats);

/* ---- optimiser and fused training step ---------------------------------- */
int32_t crnn_opt_state_len(int32_t n_params);
/* Host restatement of update!(opt, p, grad): state = [m | v | beta1^t beta2^t eta_expdecay ncalls] */
int32_t crnn_opt_init(const crnn_opt_config *o, int32_t n_params, double *state);
int32_t crnn_opt_update(const crnn_opt_config *o, int32_t n_params, double *p,
                        const double *grad, double *state);

/* Device-resident training: p and optimiser state live on the GPU; one step =
 * p2vec kernel -> solve+tangent kernel -> deterministic reduction ->
 * [RCCL all-reduce of (grad | loss_sum | n_traj) when a communicator is
 * attached] -> optimiser kernel.  Batch semantics: one update per step with
 * the gradient of the mean loss over all ranks' trajectories.               */
int32_t crnn_train_init(crnn_ctx *ctx, const crnn_opt_config *o, const double *p0);
int32_t crnn_train_step(crnn_ctx *ctx, int64_t first, int64_t count, int32_t n_save_active,
                        double *loss_mean /* NULL = do not synchronise */);
/* The two halves of crnn_train_step, for hosts that run their own collective
 * (e.g. torch.distributed) on crnn_grad_buffer() between them. */
int32_t crnn_train_step_begin(crnn_ctx *ctx, int64_t first, int64_t count, int32_t n_save_active);
int32_t crnn_train_step_end(crnn_ctx *ctx, double *loss_mean);
/* device vector [grad_sum(P) | n_overflow | loss_sum, n_ok, n_accept, n_reject, n_traj] of the step in flight: P + 6 doubles,
 * the same layout whichever gradient algorithm produced it (also after a rank-local fall-back from the adjoint to forward
 * tangents), so ranks can always sum it element-wise.  n_overflow = trajectories that outran the adjoint tape. */
int32_t crnn_grad_buffer(crnn_ctx *ctx, void **d_ptr, int32_t *n_doubles);
int32_t crnn_get_params(crnn_ctx *ctx, double *p);
int32_t crnn_set_params(crnn_ctx *ctx, const double *p);
/* The device-resident optimiser state [m(P) | v(P) | beta1^t, beta2^t, eta_expdecay, ncalls] (crnn_opt_state_len doubles):
 * what the reference's `@save ... p opt ...` / `@load` keeps across a restart (case2/case2.jl:178-187,213). */
int32_t crnn_get_opt_state(crnn_ctx *ctx, double *state);
int32_t crnn_set_opt_state(crnn_ctx *ctx, const double *state);
/* update!(opt, p, grad) on the device with a caller-supplied gradient [n_params] (case2/case2.jl:197): the same optimiser
 * kernel crnn_train_step ends with. */
int32_t crnn_train_update(crnn_ctx *ctx, const double *grad);
int32_t crnn_last_stats(crnn_ctx *ctx, crnn_stats *stats); /* of the most recent solve; synchronises */
/* Per-trajectory step counts of the most recent solve over [first, first+count): what `sol.destats.naccept / nreject`
 * report for each `solve` of the ensemble (case2/case2.jl:126).  Either pointer may be NULL; the range must lie inside
 * the range that solve covered; synchronises. */
int32_t crnn_last_step_counts(crnn_ctx *ctx, int64_t first, int64_t count, int32_t *n_accept, int32_t *n_reject);
/* HIP-event durations (ms) of the solve kernel of the last n launches (n <= 64), oldest first;
 * synchronises the ctx stream.  The measurement bench.py's roofline figures come from. */
int32_t crnn_kernel_times(crnn_ctx *ctx, double *ms, int32_t n);
int32_t crnn_synchronize(crnn_ctx *ctx);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI ---------------------- */
#define CRNN_UNIQUE_ID_BYTES 128
int32_t crnn_comm_get_unique_id(char id[CRNN_UNIQUE_ID_BYTES]);           /* rank 0, then broadcast */
int32_t crnn_comm_init(crnn_ctx *ctx, const char id[CRNN_UNIQUE_ID_BYTES], int32_t rank, int32_t world);
int32_t crnn_comm_destroy(crnn_ctx *ctx);
/* A host that owns the process group (Julia MPI.jl, torch.distributed ...) can supply the collective instead: fn must
 * sum d_buf[0..n) (device doubles) in place over all ranks, ordered after the work already enqueued on hip_stream, and
 * return 0.  Used by crnn_train_step (also for its deferred replays) in place of the library's ncclAllReduce. */
typedef int32_t (*crnn_allreduce_fn)(void *d_buf, int32_t n, void *hip_stream, void *user);
int32_t crnn_comm_set_allreduce(crnn_ctx *ctx, crnn_allreduce_fn fn, void *user);
/* number of all-reduces the training loop has issued on this ctx (every rank must report the same number) */
int64_t crnn_comm_collectives(crnn_ctx *ctx);
/* In-place sum of a host vector over all ranks (gradient | loss | count). */
int32_t crnn_allreduce_grad(crnn_ctx *ctx, double *buf, int32_t n);

/* ======================================================================== *
 * Bayesian cathode CRNN (BASELINE config 5): per-particle parameters, heat-release-rate observable.
 * Replaces, per (particle, heating rate):
 *   pred_n_ode      Cathode_NCM333_UQ/src_333/network.jl:196-218   (solve + HRR_getter :167-175)
 *   loss_neuralode  network.jl:262-266    sum((hrr - data).^2) / n_replicas / D
 *   the loop body of dlnprob  network.jl:227-252   (loss and ForwardDiff.gradient per particle; the division by
 *   Normalizer.^2 and the SVGD update stay on the host)
 * theta: [n_part x 17] row-major, each row = p .* p_scales as the reference forms it inside crnn!:
 *   [lnA(3) | Ea(3) (exponent uses Ea*1e5) | b(3) | dH(3) | reaction order n(3) | nu2, nu3].
 * An observation set = one heating rate: its time grid ts (from the temperatures, dataset.jl:19-23), the replica
 * statistics dbar_i = mean_k data_ik and d2bar_i = mean_k data_ik^2 (all the MSE needs), beta in K/min.
 * Trajectory index = particle * n_sets + set.  Stepper: non-autonomous Rosenbrock23 for gradient launches; primal launches run
 * Rosenbrock23 or the reference's AutoTsit5(TRBDF2) composite (crnn_cathode_set_solver below).
 * ======================================================================== */
#define CRNN_CATHODE_NP 17
#define CRNN_CATHODE_MAX_D 128
#define CRNN_CATHODE_MAX_SETS 4096  /* heating rates per context (the first 8 are staged in LDS) */
typedef struct crnn_cathode_config {
    int32_t abi_version, device, maxiters;
    int32_t grad_mode;  /* CRNN_GRAD_*: AUTO / ADJOINT = discrete adjoint (forward tangents if a trajectory outruns the tape) */
    double lb_clamp;   /* config.yaml:6   1e-16 */
    double T0;         /* network.jl:189  373.15 K */
    double atol, rtol; /* config.yaml:7 lb_abstol 1e-12; reltol = DiffEq default 1e-3 */
    double gamma, qmin, qmax, beta1, beta2, qsteady_min, qsteady_max, qoldinit;
} crnn_cathode_config;
typedef struct crnn_cathode_ctx crnn_cathode_ctx;
int32_t crnn_cathode_config_default(crnn_cathode_config *cfg);
int32_t crnn_cathode_create(const crnn_cathode_config *cfg, crnn_cathode_ctx **out);
void crnn_cathode_destroy(crnn_cathode_ctx *ctx);
const char *crnn_cathode_last_error(const crnn_cathode_ctx *ctx);
/* ts, dbar, d2bar: [n_sets x Dmax] row-major (rows padded beyond D[s]); beta, D: [n_sets] */
int32_t crnn_cathode_set_obs(crnn_cathode_ctx *ctx, int32_t n_sets, int32_t Dmax, const int32_t *D, const double *ts,
                             const double *dbar, const double *d2bar, const double *beta);
/* loss [n_part*n_sets]; grad [n_part*n_sets x 17] or NULL; hrr [n_part*n_sets x Dmax] or NULL;
 * retcode, n_saved [n_part*n_sets] or NULL */
int32_t crnn_cathode_solve(crnn_cathode_ctx *ctx, const double *theta, int64_t n_part, double *loss, double *grad,
                           double *hrr, int32_t *retcode, int32_t *n_saved, crnn_stats *stats);

/* Particle-shard exchange (after dlnprob, crnn_cathode.jl:31): the N particles are partitioned contiguously over
 * the ranks (the first N % world ranks hold one more); every rank passes its rows [n_local x width] (e.g. [loss | lnpgrad],
 * width 18) and receives all N rows in `full` -- one ncclAllGather on the ctx stream.  Without a communicator
 * (single process) n_local must equal n_total and the rows are copied. */
int32_t crnn_cathode_comm_init(crnn_cathode_ctx *ctx, const char id[CRNN_UNIQUE_ID_BYTES], int32_t rank, int32_t world);
int32_t crnn_cathode_comm_destroy(crnn_cathode_ctx *ctx);
int32_t crnn_cathode_allgather(crnn_cathode_ctx *ctx, const double *local, int64_t n_local, int32_t width, int64_t n_total,
                               double *full);

/* The SVGD move that follows dlnprob (Cathode_NCM333_UQ/src_333/network.jl:67-87 svgd_kernel, crnn_cathode.jl:36-50):
 *   d_ij = |p_i - p_j|; h < 0: h = sqrt(0.5 median(d_ij, i > j)^2 / log(N + 1)); K = exp(-d^2 / (2 h^2));
 *   data = K lnpgrad; repulsion = (-K p + p .* rowsum(K)) / h^2; p_new = p + stepsize (data + repulsion) / N.
 * p, lnpgrad, p_new, data_term, repulsion: [N x dim] row-major host arrays (dim <= 32; data_term / repulsion may be NULL);
 * h_out receives the bandwidth used.  Stateless towards the caller: runs on HIP device `device`; the device workspace of the
 * previous call on that device is reused (nothing is allocated in steady state), the median is selected on the device (no
 * host round trip between the passes). */
int32_t crnn_svgd_update(int32_t device, const double *p, const double *lnpgrad, int64_t N, int32_t dim, double stepsize,
                         double h, double *p_new, double *h_out, double *data_term, double *repulsion);

/* Device-resident SVGD loop (crnn_cathode.jl:36-50: `for i_exp in randperm(...)  lnpgrad = dlnprob(p, i_exp); p = svgd update`):
 * the particles live on the device between iterations.
 *   crnn_cathode_set_particles   p [n_part x 17] normalised particles (row-major), p_scales [17]
 *                                (theta = p .* p_scales, network.jl:152-157); n_part >= 2
 *   crnn_cathode_svgd_step       one iteration for observation set (heating rate) i_set: every particle is integrated for that
 *                                rate with per-particle adjoint gradients (one launch), lnpgrad[:, k] = -(d loss / d p_k) / normalizer2[k]
 *                                (dlnprob, network.jl:234-250), then the SVGD move in place -- solve, chain rule, median select,
 *                                kernel sums and update are enqueued back to back on the ctx stream.  h < 0: median trick.
 *                                loss_mean / h_out / ms (kernel times in ms: {solve, SVGD move}) may all be NULL: then nothing is
 *                                read back beyond the 4-byte tape-overflow flag of the adjoint launch.
 *   crnn_cathode_get_particles   copies the current particles out */
int32_t crnn_cathode_set_particles(crnn_cathode_ctx *ctx, const double *p, const double *p_scales, int64_t n_part);
/* Layout of the adjoint's step tape.  1 (default): (t, dt, u) of every accepted step, 40 B per step -- the fastest.  2 / 4 / 8:
 * CHECKPOINTED: dt of every step and (t, u) of every 2nd / 4th / 8th (24 / 16 / 12 B per step); the reverse sweep re-forms the states
 * in between (same arithmetic as the forward sweep, gradients agree to rounding).  BASELINE config 5 on one MI355X
 * (4 096 x 256 trajectories, 338 steps each): HBM traffic 31 -> 9.4 GB per launch, tape capacity per trajectory 3.3x, kernel
 * time 35.7 -> 40.3 ms (the kernel is issue-bound, not bandwidth-bound: profiles/r03e_*).  Round 4: every 2nd step at two
 * wavefronts per SIMD like the full tape -- 40.3 ms too (full tape: 31.9): the re-formations cost what the traffic saves. */
int32_t crnn_cathode_set_tape_every(crnn_cathode_ctx *ctx, int32_t every);
/* The stepper of PRIMAL launches (crnn_cathode_solve with grad == NULL: pred_n_ode / HRR_getter / loss_neuralode, the epoch-end
 * loss loop).  The reference integrates this model with `alg = AutoTsit5(TRBDF2(autodiff = true))` (network.jl:195, used by
 * pred_n_ode :205-212):
 *   CRNN_CATH_SOLVER_ROSENBROCK23      (default) Rosenbrock23 throughout: 338 accepted steps per config-5 trajectory
 *   CRNN_CATH_SOLVER_AUTOTSIT5_TRBDF2  the reference's composite: Tsit5 + OrdinaryDiffEq's AutoSwitch + TRBDF2 (Newton iteration,
 *                                      Jacobian / W reuse, smoothed error estimate) -- 114 accepted steps per trajectory
 *   CRNN_CATH_SOLVER_AUTOTSIT5_ROS23   the same composite with Rosenbrock23 as its stiff algorithm
 * PARITY MODES, SLOWER than the default: 17.8 ms (TRBDF2) / 17.2 ms (ROS23) per 4 096 x 256 launch against Rosenbrock23's 15.6 -- a
 * Tsit5 attempt is six right-hand sides with their logarithms and exponentials, a Rosenbrock23 attempt two; a third of the steps does
 * not buy a third of the time (profiles/r04i).  They exist so that a primal launch can be the reference's own algorithm.
 * All restated from the published algorithms ([UNVERIFIED-DEP]: the packages are not in the reference tree; oracle/crnn_oracle.c
 * states every branch).  GRADIENT launches: the discrete adjoint (errnorm_sens = 0) and the dual-norm chunks (crnn_cathode_set_errnorm_sens)
 * run on the L-stable Rosenbrock23 whatever this setting (the adjoint through Tsit5's steps is unstable for this model:
 * cathode_auto_kernel.hpp) -- EXCEPT the dual-norm chunks under _AUTOTSIT5_TRBDF2 (round 5): they run through the composite itself, the
 * partials in both algorithms' error estimates and through TRBDF2's Newton iterations (cathode_sens_auto_kernel.hpp): a gradient call
 * is then network.jl:232 through :195, algorithm for algorithm; parity mode, nine lanes per trajectory.  Results of the three agree to
 * solver tolerance (at tight tolerance to 1e-8); two implementations of the composite agree to ~1e-8 in the loss and a fraction
 * of rtol in the heat-release curve, not step for step (explicit steps at their stability limit amplify round-off). */
#define CRNN_CATH_SOLVER_ROSENBROCK23 0
#define CRNN_CATH_SOLVER_AUTOTSIT5_TRBDF2 1
#define CRNN_CATH_SOLVER_AUTOTSIT5_ROS23 2
int32_t crnn_cathode_set_solver(crnn_cathode_ctx *ctx, int32_t solver);
/* The gradient as the reference evaluates it: `ForwardDiff.gradient(x -> loss_neuralode(x, i_exp), p_temp)` (network.jl:232) pushes
 * Duals through the adaptive solve, in ForwardDiff's chunks of the 17 normalised parameters (9, then 8 and a zero partial), every
 * chunk its own adaptive solve, the error norm weighing the chunk's partials (crnn_config.errnorm_sens has the norm; mode 1:
 * / length(u); mode 2: / totallength(u), the form of the DiffEqBase 6.189 this project's Manifest pins).  p_scales[17] = d theta / d p
 * (network.jl:152-157): the partials in the norm are those with respect to p.  mode != 0: a gradient call = two chunk launches
 * (forward tangents through every attempt, Rosenbrock23: cathode_sens_kernel.hpp) + the plain solve whose loss / curves / return
 * codes it reports; the gradient is still returned with respect to theta.  mode 0 (the default OF THIS C ABI: a context that was never
 * told otherwise): primal-only norm, discrete adjoint (about 2.2 times faster; the two gradients differ by a few 1e-3 of their largest entry
 * at reltol 1e-3 on most trajectories, and by orders of magnitude on a few per cent of a particle cloud: profiles/r04m, r05c).  The host
 * mirrors of the reference surface (crnn_amd.CathodeUQ, CRNNHip.jl's Cathode) call this with mode 2 in their constructors -- `dlnprob`
 * there is the reference's gradient unless the caller passes errnorm_sens = 0; crnn_cathode_config.grad_mode and
 * crnn_cathode_set_tape_every configure the mode-0 gradient only, and the mirrors refuse them together with mode != 0.
 * crnn_cathode_last_chunk_stats: {accepted, rejected} steps summed over the trajectories, for chunk 1 and chunk 2 of the last call. */
int32_t crnn_cathode_set_errnorm_sens(crnn_cathode_ctx *ctx, int32_t mode, const double *p_scales /* [17] */);
int32_t crnn_cathode_last_chunk_stats(crnn_cathode_ctx *ctx, int64_t *out /* [4] */);
int32_t crnn_cathode_svgd_step(crnn_cathode_ctx *ctx, int32_t i_set, const double *normalizer2 /* [17] */, double stepsize, double h,
                               double *loss_mean, double *h_out, double *ms);
int32_t crnn_cathode_get_particles(crnn_cathode_ctx *ctx, double *p);

#ifdef __cplusplus
}
#endif
#endif /* CRNN_HIP_H */

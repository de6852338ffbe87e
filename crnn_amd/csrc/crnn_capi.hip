// crnn_amd/csrc/crnn_capi.hip -- implementation of include/crnn_hip.h for MI355X (gfx950).
// Host side: context / buffers / launch geometry / RCCL; device side: ros23_kernel.hpp.
#include "../../include/crnn_hip.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "p2vec.hpp"
#include "ros23_kernel.hpp"
#include "ros23_adj_kernel.hpp"
#include "ros23_sens_kernel.hpp"
#include "tsit5_sens_kernel.hpp"
#include "hychem_kernel.hpp"
#include "hychem2_kernel.hpp"
#include "hychem_auto_kernel.hpp"
#include "hychem_sens_kernel.hpp"
#include "hychem_sens2_kernel.hpp"
#include "tsit5_kernel.hpp"
#include "auto_adj_kernel.hpp"
#include "ros23_adj2_kernel.hpp"
#include "cathode_kernel.hpp"
#include "cathode_auto_kernel.hpp"
#include "cathode_sens_kernel.hpp"
#include "cathode_sens_auto_kernel.hpp"
#include "svgd_kernel.hpp"

namespace {

thread_local std::string g_last_error;

struct Ctx;
int32_t fail(Ctx *ctx, const std::string &msg);
int32_t check_pending(Ctx *c, double *loss_mean);   // deferred outcome of adjoint training steps (defined with the training loop)

#define HIP_TRY(ctx, expr)                                                                          \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return fail(ctx, std::string(#expr) + ": " + hipGetErrorString(e_));                    \
    } while (0)
#define NCCL_TRY(ctx, expr)                                                                         \
    do {                                                                                            \
        ncclResult_t r_ = (expr);                                                                   \
        if (r_ != ncclSuccess)                                                                      \
            return fail(ctx, std::string(#expr) + ": " + ncclGetErrorString(r_));                   \
    } while (0)

using KernelFn = void (*)(const crnn::SolveParams, const double *, const double *);

struct KernelEntry {
    int solver, ns, nr, has_t, use_scale, C, L;
    KernelFn fn;
    int drows = 0;   // dual-norm kernels: rows of d theta / d p the kernel stages (> P: all chunks of a gradient in one launch)
};

constexpr int kBlock = 256;

#define KENT(NS, NR, HT, SC, C, L) \
    { CRNN_SOLVER_ROSENBROCK23, NS, NR, HT, SC, C, L, (KernelFn)crnn::ros23_kernel<NS, NR, (HT) != 0, (SC) != 0, C, L, kBlock> }
#define KENT5(NS, NR, HT, SC, C, L) \
    { CRNN_SOLVER_TSIT5, NS, NR, HT, SC, C, L, (KernelFn)crnn::tsit5_kernel<NS, NR, (HT) != 0, (SC) != 0, C, L, kBlock> }

// Instantiated shapes: case2 (6 species + T, 3 reactions), robertson (3, 6,
// dydt_scale), case1 (5, 4).  C = tangent columns per lane, L = lanes per
// trajectory (L*C >= number of directions; 64/L trajectories per wavefront).
const KernelEntry kKernels[] = {
    KENT(6, 3, 1, 0, 0, 1), KENT(6, 3, 1, 0, 1, 25), KENT(6, 3, 1, 0, 3, 9), KENT(6, 3, 1, 0, 4, 7),
    KENT(6, 3, 1, 0, 5, 5), KENT(6, 3, 1, 0, 7, 4), KENT(6, 3, 1, 0, 7, 6),
    KENT(3, 6, 0, 1, 0, 1), KENT(3, 6, 0, 1, 1, 43), KENT(3, 6, 0, 1, 4, 11), KENT(3, 6, 0, 1, 6, 8), KENT(3, 6, 0, 1, 11, 4),
    KENT(5, 4, 0, 0, 0, 1), KENT(5, 4, 0, 0, 1, 24), KENT(5, 4, 0, 0, 4, 6), KENT(5, 4, 0, 0, 6, 4), KENT(5, 4, 0, 0, 8, 6),
    // explicit Tsit5 (case1's reference algorithm; the non-stiff branch of case2's AutoTsit5)
    KENT5(5, 4, 0, 0, 0, 1), KENT5(5, 4, 0, 0, 1, 24), KENT5(5, 4, 0, 0, 4, 6), KENT5(5, 4, 0, 0, 6, 4), KENT5(5, 4, 0, 0, 8, 6),
    KENT5(6, 3, 1, 0, 0, 1), KENT5(6, 3, 1, 0, 1, 25), KENT5(6, 3, 1, 0, 5, 5), KENT5(6, 3, 1, 0, 7, 6),
};

// errnorm_sens = 1 (ForwardDiff's dual-inclusive error norm): one launch = one ForwardDiff chunk of at most C*L partials.
// (C, L) are sized for ForwardDiff.pickchunksize: case2 P = 25 -> 9 + 9 + 7, robertson 43 -> 11 x 3 + 10, case1 24 -> 12 + 12.
constexpr int kSensBlock = 128;
#define KSENS(NS, NR, HT, SC, C, L, DROWS) \
    { CRNN_SOLVER_ROSENBROCK23, NS, NR, HT, SC, C, L, \
      (KernelFn)crnn::ros23_sens_kernel<NS, NR, (HT) != 0, (SC) != 0, C, L, kSensBlock, DROWS>, DROWS }
#define KSENS5(NS, NR, HT, SC, C, L, DROWS) \
    { CRNN_SOLVER_TSIT5, NS, NR, HT, SC, C, L, \
      (KernelFn)crnn::tsit5_sens_kernel<NS, NR, (HT) != 0, (SC) != 0, C, L, kSensBlock, DROWS>, DROWS }
const KernelEntry kSensKernels[] = {
    // DROWS = P + 1 where two blocks per CU still fit (case2: 25 parameters, robertson: 43): the chunks of a gradient in one launch
    KSENS(6, 3, 1, 0, 3, 3, 26), KSENS(3, 6, 0, 1, 4, 3, 44), KSENS(5, 4, 0, 0, 4, 3, 13),
    KSENS5(5, 4, 0, 0, 4, 3, 13), KSENS5(6, 3, 1, 0, 3, 3, 26),      // case1's Tsit5 (case1.jl:28); the non-stiff branch of case2's AutoTsit5
};

using AdjKernelFn = void (*)(const crnn::SolveParams, const double *, const crnn::AdjParams);
struct AdjEntry {
    int solver, ns, nr, has_t, use_scale;
    AdjKernelFn fn;
    int max_gen = 1;   // two-lane entries: AUTO uses the kernel while count <= max_gen * (resident lane pairs)
    AdjKernelFn fn_primal = nullptr;   // the forward sweep alone (ros23_adj_kernel<..., PRIMAL = true>): crnn_solve with no directions
    AdjKernelFn fn_primal_fd = nullptr;   // the same with W from forward differences (Rosenbrock23(autodiff = false): crnn_ctx_set_jacobian)
};
#define KADJ(NS, NR, HT, SC) \
    { CRNN_SOLVER_ROSENBROCK23, NS, NR, HT, SC, (AdjKernelFn)crnn::ros23_adj_kernel<NS, NR, (HT) != 0, (SC) != 0, kBlock>, 1, \
      (AdjKernelFn)crnn::ros23_adj_kernel<NS, NR, (HT) != 0, (SC) != 0, kBlock, true>, \
      (AdjKernelFn)crnn::ros23_adj_kernel<NS, NR, (HT) != 0, (SC) != 0, kBlock, true, true> }
#define KAUTO(SOLVER, NS, NR, HT, SC, COMPOSITE) \
    { SOLVER, NS, NR, HT, SC, (AdjKernelFn)crnn::auto_adj_kernel<NS, NR, (HT) != 0, (SC) != 0, kBlock, COMPOSITE>, 1, \
      (AdjKernelFn)crnn::auto_adj_kernel<NS, NR, (HT) != 0, (SC) != 0, kBlock, COMPOSITE, true> }
// Rosenbrock23 discrete adjoint with TWO lanes per trajectory (ros23_adj2_kernel.hpp): shapes with nr < ns, no rate scaling.
// One wavefront per SIMD (at most 512 registers per lane); a 256-register build at two per SIMD measured break-even in round 4 and was deleted.
// max_gen (measured, tools/kbench.py --lanes 1|2, MI355X): while the pairs fit the resident lanes the two-lane kernel always
// wins (case2 4 096-32 768 trajectories: 0.28-0.32 ms against 0.48; case1's shape with Rosenbrock23, 16 384: 0.139 against
// 0.176).  With TWO generations of pairs (32 769-65 536 trajectories) it depends on how widely the step counts spread -- the
// queue is longest-first, so the second generation is the short trajectories: case2 at the trained parameters (longest 48
// steps, mean 29): 49 152: 0.405 against 0.486 ms, 65 536: 0.483 against 0.503; at the reference's initialiser (22.6 steps,
// nearly uniform) 65 536: 0.373 against 0.33; case1 (6 steps each) 65 536: 0.304 against 0.215.  AUTO therefore stops at
// one generation; a host that knows its step counts spread asks for two lanes itself (crnn_ctx_set_lanes_per_traj).
#define KADJ2(NS, NR, HT, MAXGEN) \
    { CRNN_SOLVER_ROSENBROCK23, NS, NR, HT, 0, (AdjKernelFn)crnn::ros23_adj2_kernel<NS, NR, (HT) != 0, kBlock, 1>, MAXGEN }
const AdjEntry kAdj2Kernels[] = { KADJ2(6, 3, 1, 1), KADJ2(5, 4, 0, 1) };
// discrete-adjoint gradient kernels, one lane per trajectory: Rosenbrock23; Tsit5; the AutoTsit5(Rosenbrock23()) composite
// (with a constant temperature state it never leaves Tsit5 -- auto_adj_kernel.hpp -- and shares that instantiation)
const AdjEntry kAdjKernels[] = {
    KADJ(6, 3, 1, 0), KADJ(3, 6, 0, 1), KADJ(5, 4, 0, 0),
    KAUTO(CRNN_SOLVER_TSIT5, 6, 3, 1, 0, false), KAUTO(CRNN_SOLVER_TSIT5, 5, 4, 0, 0, false),
    KAUTO(CRNN_SOLVER_AUTOTSIT5, 6, 3, 1, 0, false), KAUTO(CRNN_SOLVER_AUTOTSIT5, 5, 4, 0, 0, true),
    KAUTO(CRNN_SOLVER_AUTOTSIT5, 3, 6, 0, 1, true),
};

struct Ctx {
    crnn_config cfg{};
    int n = 0, n_theta = 0, n_params = 0;
    int nfx = 0;                    // extra feature rows of w_in: has_temp (CRNN) or 2 (HyChem)
    bool hychem = false;
    double *d_tabs = nullptr;       // HyChem T/P tables, trajectory-major [B][2][D]
    int64_t tabs_B = 0;
    double *d_gacc = nullptr;       // HyChem gradient accumulators [ceil(count/64)][n_theta][64]
    size_t gacc_cap = 0;
    bool use_scale = false;
    std::string err;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    static constexpr int kRing = 64;
    static constexpr int kSpreadRing = 4;
    int32_t *d_spread = nullptr, *h_spread = nullptr;        // step-count spread of a launch {longest, median, count}: device slot, pinned host ring
    hipEvent_t ev_spread[kSpreadRing] = {};
    int64_t spread_first[kSpreadRing] = {}, spread_count[kSpreadRing] = {};
    uint64_t spread_n = 0;                                   // spreads recorded in a row (0: no history)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;           // events of the most recent solve kernel
    hipEvent_t ring0[kRing] = {}, ring1[kRing] = {};   // ring of (start, stop) pairs, one per launch
    int64_t n_launch = 0;
    int num_cu = 256;
    // ensemble
    int64_t B = 0;
    int n_obs = 0;
    int drow[CRNN_MAX_N];
    double inv_yscale[CRNN_MAX_N];
    double *d_u0 = nullptr, *d_data = nullptr, *d_tsave = nullptr;
    bool own_u0 = false, own_data = false;
    std::vector<double> tsave;
    // per-call device outputs
    double *d_pred = nullptr, *d_loss = nullptr;
    int32_t *d_ret = nullptr, *d_nsaved = nullptr, *d_nacc = nullptr, *d_nrej = nullptr;
    size_t pred_cap = 0;
    double *d_gtraj = nullptr;
    size_t gtraj_cap = 0;
    crnn::KConst *d_kc = nullptr;
    unsigned long long *d_queue = nullptr;
    bool kc_dirty = true;
    // weights
    double *d_theta = nullptr, *d_dtheta = nullptr;  // [n_theta], [n_theta * max_dir]
    int max_dir = 0;
    // adjoint tape
    double *d_tape = nullptr;
    size_t tape_doubles = 0;
    unsigned int *d_overflow = nullptr;
    // queue order of ensembles larger than the resident lanes (sort_steps_kernel): by the step counts of the previous launch
    int32_t *d_perm = nullptr;
    size_t perm_cap = 0;
    int queue_order = CRNN_QUEUE_AUTO;
    bool fuse_opt = false, opt_fused = false;   // crnn_train_step: the optimiser update rides in the reduction launch / did so
    bool perm_ready = false;                    // d_perm holds the queue order for the next launch over [perm_first, +perm_count)
    int64_t perm_first = 0, perm_count = 0;
    int64_t steps_first = 0, steps_count = 0;   // [first, first+count) whose d_nacc / d_nrej hold a completed launch's counts
    double *d_red_theta = nullptr;  // [n_theta + kExtra]
    int64_t n_fallback = 0;         // calls repeated with forward tangents after a tape overflow
    int adj_occ = 0;                // cached occupancy of the adjoint kernel
    int adj2_occ = 0;               // ... of the two-lanes-per-trajectory variant
    int lanes_per_traj = 0;         // crnn_ctx_set_lanes_per_traj: 0 = AUTO, 1, 2
    int jac_mode = CRNN_JAC_ANALYTIC;   // crnn_ctx_set_jacobian: W of the Rosenbrock23 primal launches
    int hysens_occ = 0;
    int hysens2_occ = 0, hysens2c_occ = 0;
    bool hy_dirs_sparse = false;   // the directions of the launch being prepared fit hychem_sens2_kernel's sparse description (set by the entry points)
    int hy_sens_kernel = 0;        // CRNN_HY_SENS_KERNEL.  0 / 1: hychem_sens_kernel (dense directions; the kernel a device has run) for every Rosenbrock23 dual-norm launch;
                                   // 2: hychem_sens2_kernel (sparse directions) where the directions fit -- opt-in until a device session has passed its parity tests
    int64_t hy_tape_retries = 0;    // HyChem launches repeated with fewer resident trajectories after a tape overflow
    int hy_block_cap = 0;           // the block count such a repetition found to fit: later launches over the same range start from it
    int hy_cap_uses = 0;            // launches served from the remembered width since it was last (re)established: every kHyCapReprobe-th one tries 4x wider
    int64_t hy_cap_first = -1, hy_cap_count = -1;   // (without it every later step overflowed, drained and relaunched again)
    int last_lanes = 0;             // lanes per trajectory of the most recent adjoint launch (0: another kernel family ran)
    // deferred outcome of adjoint training steps (crnn_train_step): see check_pending
    bool defer_next = false, last_deferred = false, force_forward = false;
    struct StepArgs { int64_t first, count; int32_t n_save; };
    std::vector<StepArgs> pending;  // steps enqueued since the host last looked
    double *d_poison = nullptr;     // [0] sticky flag: a step was skipped; [1] number of skipped steps
    size_t tape_budget = 0;         // bytes the tape may take (auto mode), fixed at the first gradient call
    // reduction
    double *d_partials = nullptr;
    size_t partials_cap = 0;
    double *d_red = nullptr;  // [npart_max]
    int npart_max = 0;
    double *d_red_asm = nullptr;    // errnorm_sens: the full [P + kTail] vector assembled from the chunk passes
    int red_asm_len = 0;
    int last_npart = 0, last_P = 0;
    // training state
    double *d_p = nullptr, *d_opt = nullptr;
    double *d_p_eval = nullptr;     // scratch copy of a caller's p (crnn_loss_grad) -- never the training parameters
    bool theta_current = false;     // d_theta/d_dtheta already hold p2vec(d_p) (written by the fused optimiser kernel)
    bool sens_one_launch = true;    // errnorm_sens gradients: the chunks of a gradient in one launch where the kernel stages all of d theta / d p
                                    // (CRNN_SENS_ONE_LAUNCH=0 at context creation: one launch per chunk, for measurements)
    bool flags_zeroed = false;      // queue head / overflow counter already zeroed on the stream by the optimiser kernel
    crnn::OptCfg opt{};
    bool train_ready = false;
    // comm
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    double *d_comm_buf = nullptr;
    int comm_buf_len = 0;
    crnn_allreduce_fn host_ar = nullptr;   // caller-supplied collective (crnn_comm_set_allreduce); takes precedence over comm
    void *host_ar_user = nullptr;
    int64_t n_collectives = 0;             // all-reduces of the reduced vector issued by the training loop (tests count them)
};

int32_t fail(Ctx *ctx, const std::string &msg) {
    g_last_error = msg;
    if (ctx) ctx->err = msg;
    return -1;
}

// The one exchange of a training step: in-place sum of d_red over the ranks, on the ctx stream.
int32_t allreduce_red(Ctx *c) {
    if (c->host_ar) {
        ++c->n_collectives;
        if (c->host_ar(c->d_red, c->last_npart, (void *)c->stream, c->host_ar_user) != 0)
            return fail(c, "the caller's all-reduce callback (crnn_comm_set_allreduce) reported failure");
        return 0;
    }
    if (c->comm && c->world > 1) {   // a one-rank communicator has nothing to add
        ++c->n_collectives;
        NCCL_TRY(c, ncclAllReduce(c->d_red, c->d_red, c->last_npart, ncclDouble, ncclSum, c->comm, c->stream));
    }
    return 0;
}

bool shape_match(const Ctx *c, const KernelEntry &k) {
    return k.solver == c->cfg.solver && k.ns == c->cfg.ns && k.nr == c->cfg.nr && k.has_t == c->cfg.has_temp &&
           k.use_scale == (c->use_scale ? 1 : 0);
}

// AutoTsit5(Rosenbrock23) on a state vector with a constant component (case2's temperature) never leaves Tsit5: OrdinaryDiffEq's stiffness
// estimate max_i |k7_i - k6_i| / |g7_i - g6_i| is 0/0 = NaN there and NaN > 9/10 is false (auto_adj_kernel.hpp; the reference's recorded case2
// history confirms it to 5e-6: tests/test_case2_stream_pin.py).  So `alg = AutoTsit5(Rosenbrock23())` (case2/case2.jl:26) with errnorm_sens IS
// the Tsit5 dual-norm gradient, and a host that keeps the reference's `alg` gets it instead of an error
static bool composite_is_tsit5(const crnn_config &cfg) {
    return cfg.solver == CRNN_SOLVER_AUTOTSIT5 && cfg.rhs_kind == CRNN_RHS_CRNN && cfg.has_temp != 0;
}

const KernelEntry *find_sens(const Ctx *c) {
    const int solver = composite_is_tsit5(c->cfg) ? CRNN_SOLVER_TSIT5 : c->cfg.solver;
    for (const auto &k : kSensKernels)
        if (k.solver == solver && k.ns == c->cfg.ns && k.nr == c->cfg.nr && k.has_t == c->cfg.has_temp && k.use_scale == (c->use_scale ? 1 : 0)) return &k;
    return nullptr;
}

// ForwardDiff.pickchunksize (DEFAULT_CHUNK_THRESHOLD = 12): the number of partials a Dual carries per pass
int fd_chunk_size(int P) {
    if (P <= 12) return P;
    const int nchunks = (P + 11) / 12;
    return (P + nchunks - 1) / nchunks;
}

const KernelEntry *find_primal(const Ctx *c) {
    for (const auto &k : kKernels)
        if (shape_match(c, k) && k.C == 0) return &k;
    return nullptr;
}

// Choose the (C, L) variant for P tangent directions: cfg.cols_per_lane if an
// instance with that C covers P, else the instance that minimises the modelled
// lane-work per trajectory (primal ~ 3 column-equivalents, 64/L trajectories per wave).
const KernelEntry *pick_kernel(const Ctx *c, int P) {
    if (P == 0) return find_primal(c);
    const KernelEntry *best = nullptr;
    double best_cost = 1e300;
    for (const auto &k : kKernels) {
        if (!shape_match(c, k) || k.C == 0 || k.C * k.L < P) continue;
        double cost = (3.0 + k.C) / (double)(64 / k.L);
        if (c->cfg.cols_per_lane > 0 && k.C == c->cfg.cols_per_lane) cost *= 1e-3;
        if (cost < best_cost) { best_cost = cost; best = &k; }
    }
    return best;
}

__global__ void p2vec_kernel(int pmap, int ns, int nr, int has_temp, const double *__restrict__ p, double *th, double *dth,
                             int nth, int P) {
    for (int i = threadIdx.x; i < nth * P; i += blockDim.x) dth[i] = 0.0;
    __syncthreads();
    const int t = threadIdx.x;
    if (t < nr * ns) crnn::p2vec_eval(pmap, ns, nr, has_temp, p, th, dth, t / ns, nr, t % ns, ns);   // one (species, reaction) pair per thread
}

// red = [grad_sum(P) | n_overflow | loss_sum, n_ok, n_accept, n_reject, n_traj]   (npart = P + kTail)
// Flux chain (p2vec.hpp opt_update) with one thread per parameter; the norm clip is a fixed-order LDS tree.
// Tail: theta, dtheta = p2vec(updated p) for the next step and zeroing of the next launch's queue head / overflow
// counter, so that a training step is [this kernel] -> solve -> reductions.
// (device function: called by opt_kernel and, when no collective sits between the reduction and the update, by block 0 of
// reduce_opt_sort_kernel; NT = threads of the block, sh = NT doubles of LDS)
// What opt_body reads that does not depend on the reduced gradient: requested before the reduction (block 0 of
// reduce_opt_sort_kernel) so that the values arrive while the partial rows are summed -- read where they are used, behind the block's
// barriers, they were five dependent memory round trips of a launch that sits alone between two solve launches.
struct OptPre { double poison0, ncalls, ed_eta, b1t, b2t, mk, vk, pk; };
template <int NT>
__device__ __forceinline__ void opt_prefetch(int P, const double *p, const double *state, const double *poison, OptPre &q) {
    const int tid = threadIdx.x;
    q.poison0 = poison[0];
    q.b1t = state[2 * P]; q.b2t = state[2 * P + 1];
    q.ed_eta = state[2 * P + 2]; q.ncalls = state[2 * P + 3];
    const int k = tid < P ? tid : 0;
    q.mk = state[k]; q.vk = state[P + k]; q.pk = p[k];
}
template <int NT>
__device__ __forceinline__ void opt_body(const crnn::OptCfg &o, int P, int npart, double *p, const double *red,
                                         double *state, int pmap, int ns, int nr, int has_temp, double *th, double *dth,
                                         int nth, unsigned long long *queue, unsigned int *overflow, double *poison, double *sh,
                                         const OptPre &q) {
    const int tid = threadIdx.x;
    // A gradient formed while some rank's adjoint tape overflowed (summed overflow count != 0) is not applied, and neither
    // is any later step until the host has repeated the skipped ones in order (sticky flag): p, the optimiser state and
    // theta stay as they are.  Only the overflow count gates this: a NaN gradient from any other cause is applied and
    // shows up in p, as it would in the reference.
    const double n_over = red[npart - crnn::kTail], ntraj = red[npart - 1], red_k = red[tid < P ? tid : 0];   // one round trip
    const bool skip = q.poison0 != 0.0 || n_over != 0.0;
    __syncthreads();
    if (skip) {
        if (tid == 0) { poison[0] = 1.0; poison[1] += 1.0; *queue = 0ULL; *overflow = 0u; }
        return;
    }
    const double gscale = ntraj > 0 ? 1.0 / ntraj : 0.0;
    double *m = state, *v = state + P, *bp = state + 2 * P;
    double *ed_eta = state + 2 * P + 2, *ncalls = state + 2 * P + 3;
    double gn = 0.0;
    bool clip = false;
    if (o.grad_clip_norm > 0) {
        double a = 0.0;
        for (int k = tid; k < P; k += NT) { const double g = (k == tid ? red_k : red[k]) * gscale; a = fma(g, g, a); }
        sh[tid] = a;
        __syncthreads();
        for (int s_ = NT / 2; s_ > 0; s_ >>= 1) {
            if (tid < s_) sh[tid] += sh[tid + s_];
            __syncthreads();
        }
        gn = sqrt(sh[0]);
        clip = gn > o.grad_clip_norm;
    }
    double eta_ed = 1.0;
    if (o.use_expdecay) {
        const double nc = q.ncalls + 1.0;
        double e = q.ed_eta;
        if (fmod(nc, (double)o.decay_step) == 0.0) { e = e * o.ed_decay; e = e > o.ed_clip ? e : o.ed_clip; }
        eta_ed = e;
        if (tid == 0) { *ncalls = nc; *ed_eta = e; }
    }
    const double b1t = q.b1t, b2t = q.b2t;
    __syncthreads();   // sh (the clip norm's tree) is re-used for the new p below
    for (int k = tid; k < P; k += NT) {
        const bool mine = (k == tid);
        double g = (mine ? red_k : red[k]) * gscale;
        if (clip) g = g / gn * o.grad_clip_norm;
        g *= eta_ed;
        const double mk = o.beta1 * (mine ? q.mk : m[k]) + (1.0 - o.beta1) * g;
        const double vk = o.beta2 * (mine ? q.vk : v[k]) + (1.0 - o.beta2) * g * g;
        m[k] = mk;
        v[k] = vk;
        double delta = mk / (1.0 - b1t) / (sqrt(vk / (1.0 - b2t)) + 1e-8) * o.eta;
        const double pk = mine ? q.pk : p[k];
        delta += o.wd * pk;
        const double pn = pk - delta;
        p[k] = pn;
        if (k < NT) sh[k] = pn;     // the new p for the one thread that forms theta (not read back from global memory)
    }
    for (int i = tid; i < nth * P; i += NT) dth[i] = 0.0;
    __syncthreads();
    if (tid < nr * ns) crnn::p2vec_eval(pmap, ns, nr, has_temp, P <= NT ? sh : p, th, dth, tid / ns, nr, tid % ns, ns);   // one (species, reaction) pair per thread
    if (tid == 0) {
        bp[0] = b1t * o.beta1;
        bp[1] = b2t * o.beta2;
        *queue = 0ULL;
        *overflow = 0u;
    }
}

__global__ __launch_bounds__(256) void opt_kernel(crnn::OptCfg o, int P, int npart, double *p, const double *__restrict__ red,
                                                  double *state, int pmap, int ns, int nr, int has_temp, double *th, double *dth,
                                                  int nth, unsigned long long *queue, unsigned int *overflow, double *poison) {
    __shared__ double sh[256];
    OptPre q;
    opt_prefetch<256>(P, p, state, poison, q);
    opt_body<256>(o, P, npart, p, red, state, pmap, ns, nr, has_temp, th, dth, nth, queue, overflow, poison, sh, q);
}

// One launch for the tail of a training step that needs no collective: block 0 = reduction + chain rule, then the
// optimiser update + p2vec of the new p; blocks 1 ... = the step-count sort for the next launch (ros23_adj_kernel.hpp).
__global__ __launch_bounds__(1024) void reduce_opt_sort_kernel(const double *__restrict__ partials, int nblk,
                                                               const double *dtheta_in, int nth, int P,       // = dth, = overflow:
                                                               double *__restrict__ red_theta, double *red,   // no restrict
                                                               const unsigned int *overflow_in,
                                                               const int32_t *__restrict__ n_accept, const int32_t *__restrict__ n_reject,
                                                               int64_t first, int count, int32_t *__restrict__ perm, int32_t *__restrict__ spread,
                                                               crnn::OptCfg o, int npart, double *p, double *state, int pmap, int ns,
                                                               int nr, int has_temp, double *th, double *dth,
                                                               unsigned long long *queue, unsigned int *overflow, double *poison) {
    __shared__ double sh[1024];
    __shared__ double part[16][64];
    __shared__ unsigned key[1024];
    if (blockIdx.x == 0) {
        OptPre q;
        opt_prefetch<1024>(P, p, state, poison, q);       // in flight during the reduction
        crnn::reduce_project_body(partials, nblk, dtheta_in, nth, P, red_theta, red, overflow_in, sh, part);
        __threadfence_block();
        __syncthreads();          // red[] is complete and visible to the whole block; dtheta_in is not read any more
        opt_body<1024>(o, P, npart, p, red, state, pmap, ns, nr, has_temp, th, dth, nth, queue, overflow, poison, sh, q);
    } else if (spread && blockIdx.x == gridDim.x - 1) {
        crnn::step_spread_block(n_accept, n_reject, first, count, spread, key);
    } else if (perm) {
        crnn::sort_steps_run(n_accept, n_reject, first, count, perm, (int)blockIdx.x - 1, (int)gridDim.x - 1 - (spread ? 1 : 0), key);
    }
}

int32_t upload_consts(Ctx *c) {
    if (!c->kc_dirty) return 0;
    crnn::KConst k{};
    k.lb = c->cfg.lb; k.ub = c->cfg.ub; k.inv_R = c->cfg.inv_R; k.t0 = c->cfg.t0;
    k.gamma = c->cfg.gamma; k.qmin = c->cfg.qmin; k.qmax = c->cfg.qmax;
    k.beta1 = c->cfg.beta1; k.beta2 = c->cfg.beta2;
    k.qsteady_min = c->cfg.qsteady_min; k.qsteady_max = c->cfg.qsteady_max;
    k.qoldinit = c->cfg.qoldinit; k.dtmin = c->cfg.dtmin;
    for (int i = 0; i < CRNN_MAX_N; ++i) {
        k.atol[i] = c->cfg.atol[i]; k.rtol[i] = c->cfg.rtol[i]; k.scale[i] = c->cfg.rate_scale[i];
        k.inv_yscale[i] = c->inv_yscale[i]; k.drow[i] = (double)c->drow[i];
        k.mw[i] = c->cfg.mw[i] > 0 ? c->cfg.mw[i] : 1.0;
        k.imw[i] = 1.0 / k.mw[i];
        k.gsc[i] = k.mw[i] * c->cfg.rate_scale[i];
    }
    k.Ru = c->cfg.gas_const;
    HIP_TRY(c, hipMemcpyAsync(c->d_kc, &k, sizeof(k), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));  // k is a stack object
    c->kc_dirty = false;
    return 0;
}

template <class T>
int32_t ensure(Ctx *c, T **ptr, size_t *cap, size_t need) {
    if (*cap >= need && *ptr) return 0;
    if (*ptr) HIP_TRY(c, hipFree(*ptr));
    *ptr = nullptr;
    HIP_TRY(c, hipMalloc((void **)ptr, need * sizeof(T)));
    *cap = need;
    return 0;
}

// pred buffer [n_save x n x B], zero-filled before every launch that writes it: the kernels store the saved columns only,
// so the trailing columns of a failed / truncated trajectory read as zero (api.py predict_neuralode documents that).
int32_t ensure_pred(Ctx *c) {
    const size_t need = (size_t)c->cfg.n_save * c->n * c->B;
    if (ensure(c, &c->d_pred, &c->pred_cap, need)) return -1;
    HIP_TRY(c, hipMemsetAsync(c->d_pred, 0, sizeof(double) * need, c->stream));
    return 0;
}

const AdjEntry *find_adjoint(const Ctx *c) {
    for (const auto &k : kAdjKernels)
        if (k.solver == c->cfg.solver && k.ns == c->cfg.ns && k.nr == c->cfg.nr && k.has_t == c->cfg.has_temp && k.use_scale == (c->use_scale ? 1 : 0))
            return &k;
    return nullptr;
}

const AdjEntry *find_adjoint2(const Ctx *c) {
    if (c->use_scale) return nullptr;
    for (const auto &k : kAdj2Kernels)
        if (k.solver == c->cfg.solver && k.ns == c->cfg.ns && k.nr == c->cfg.nr && k.has_t == c->cfg.has_temp) return &k;
    return nullptr;
}

void fill_params(Ctx *c, crnn::SolveParams &prm, int P, int64_t first, int64_t count, int n_save_active, bool want_pred) {
    prm.u0 = c->d_u0; prm.data = c->d_data; prm.tsave = c->d_tsave;
    prm.row_stride = (int64_t)c->cfg.n_save * c->n_obs;
    prm.pred = want_pred ? c->d_pred : nullptr;
    prm.loss = c->d_loss; prm.retcode = c->d_ret; prm.n_saved = c->d_nsaved;
    prm.n_accept = c->d_nacc; prm.n_reject = c->d_nrej;
    prm.gtraj = c->d_gtraj;
    prm.B = c->B; prm.first = first; prm.count = count;
    prm.n_save = n_save_active; prm.P = P;
    prm.maxiters = c->cfg.maxiters; prm.clamp_pred = c->cfg.clamp_pred; prm.loss_kind = c->cfg.loss_kind;
    prm.n_obs = c->n_obs;
    prm.kc = c->d_kc;
    prm.queue = c->d_queue;
}

// More trajectories than resident lanes: wavefronts take several 64-trajectory batches from the queue one after the other.
// Queue the trajectories by their last known step counts (those of the previous launch over a range that covers this one:
// in training p moves little from step to step) so that batches are homogeneous and the queue roughly longest-first.
// *perm stays null (index order) when the ensemble fits the resident lanes or no counts are known yet.
int32_t queue_by_steps(Ctx *c, size_t lanes, int64_t first, int64_t count, const int32_t **perm) {
    *perm = nullptr;
    // also when the ensemble fits the resident lanes: homogeneous wavefronts keep the save points of an iteration's steps
    // coherent (crnn_hip.h: crnn_ctx_set_queue_order); below two runs of 1024 the sort is not worth its launch
    const bool sortable = c->queue_order == CRNN_QUEUE_AUTO && ((size_t)count > lanes || count >= 2048) && count < ((int64_t)1 << 31) &&
                          first >= c->steps_first && first + count <= c->steps_first + c->steps_count;
    if (!sortable) return 0;
    if (ensure(c, &c->d_perm, &c->perm_cap, (size_t)count)) return -1;
    if (c->perm_ready && c->perm_first == first && c->perm_count == count) {   // sorted by the previous launch's reduction kernel
        c->perm_ready = false;
        *perm = c->d_perm;
        return 0;
    }
    hipLaunchKernelGGL(crnn::sort_steps_kernel, dim3((unsigned)((count + 1023) / 1024)), dim3(1024), 0, c->stream, c->d_nacc, c->d_nrej,
                       first, (int)count, c->d_perm);
    HIP_TRY(c, hipGetLastError());
    *perm = c->d_perm;
    return 0;
}

// Gradient by the discrete adjoint (ros23_adj_kernel.hpp): theta-space gradient per trajectory, fixed-order batch
// reduction, then the chain rule through the P given directions.  Returns 1 (not an error) when some trajectory ran
// out of tape: the caller repeats the call with forward tangents.
int32_t launch_adjoint(Ctx *c, const AdjEntry *k, const double *d_theta, const double *d_dtheta, int P, int64_t first,
                       int64_t count, int n_save_active, bool want_pred, bool defer) {
    const int nth = c->n_theta;
    const int npart_th = nth + crnn::kExtra, npart = P + crnn::kTail;
    if (c->adj_occ < 1) {
        HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&c->adj_occ, (const void *)k->fn, kBlock, 0));
        if (c->adj_occ < 1) c->adj_occ = 1;
    }
    // Two lanes per trajectory (ros23_adj2_kernel.hpp) where the pairs still fit the resident lanes: a shard smaller than
    // the chip then uses its idle lanes to shorten every step instead of leaving them empty (lanes_per_traj = AUTO), or
    // wherever the caller asks for it (crnn_ctx_set_lanes_per_traj).
    int G = 1;
    bool want_spread = false;
    const bool primal = (P == 0 && k->fn_primal != nullptr);   // forward sweep only: no tape, no reverse sweep, one lane per trajectory
                                                               // (the Tsit5 / AutoTsit5 tape kernels run their primal calls themselves, P = 0)
    if (const AdjEntry *k2 = (!primal && k->solver == CRNN_SOLVER_ROSENBROCK23 && c->lanes_per_traj != 1) ? find_adjoint2(c) : nullptr) {
        if (c->adj2_occ < 1) {
            HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&c->adj2_occ, (const void *)k2->fn, kBlock, 0));
            if (c->adj2_occ < 1) c->adj2_occ = 1;
        }
        const int64_t resident_pairs = (int64_t)c->num_cu * c->adj2_occ * (kBlock / 2);
        if (c->lanes_per_traj == 2 || count <= (int64_t)k2->max_gen * resident_pairs) { G = 2; k = k2; }
        else if (c->lanes_per_traj == 0 && c->queue_order == CRNN_QUEUE_AUTO && count <= 2 * (int64_t)k2->max_gen * resident_pairs) {
            // AUTO beyond one generation of pairs (round 4): two generations of pairs, longest first, last about
            // 0.6 (longest + median) step times of the one-lane kernel, one lane per trajectory lasts `longest` of them (one
            // wavefront per SIMD either way; ros23_adj2_kernel.hpp has the measurements) -- pairs win where the step counts
            // spread (a trained p: longest 48, median 29), single lanes where they do not (the initialiser: 25 / 23).  The
            // spread is that of the launch BEFORE the previous one over the same range (step_spread_block; its 12 bytes were
            // copied out behind that launch, so waiting for them never waits for the launch in flight): the choice is a
            // deterministic function of the run's own history, like the queue order (and only with it: in index order -- batch sums that
            // depend on the call's inputs alone -- the choice is one lane, as before).
            want_spread = true;
            if (c->spread_n >= 2) {
                const int slot = (int)((c->spread_n - 2) % Ctx::kSpreadRing);
                if (c->spread_first[slot] == first && c->spread_count[slot] == count) {
                    HIP_TRY(c, hipEventSynchronize(c->ev_spread[slot]));
                    const int32_t *h = c->h_spread + 4 * slot;
                    if (h[2] == (int32_t)count && 8 * (int64_t)h[1] <= 5 * (int64_t)h[0]) { G = 2; k = k2; }   // median <= 0.625 longest
                }
            }
        }
    }
    if (!primal) c->last_lanes = G;      // (crnn_last_lanes_per_traj reports gradient launches)
    const int occ = G == 2 ? c->adj2_occ : c->adj_occ;
    const int tpb = kBlock / G;                                  // trajectories per block
    const int tpw = 64 / G;                                      // trajectories per wavefront = per batch row
    const int64_t need_blocks = (count + tpb - 1) / tpb;
    const int nblk = (int)std::max<int64_t>(1, std::min<int64_t>(need_blocks, (int64_t)c->num_cu * occ));
    const size_t lanes = (size_t)nblk * tpb;                     // tape slots (one per resident trajectory)
    const size_t recw = (size_t)c->cfg.ns + 2;
    int64_t cap = c->cfg.tape_steps;
    if (cap <= 0) {  // auto: what fits in min(free/4, 16 GiB), at most maxiters (no trajectory accepts more steps)
        if (c->tape_budget == 0) {
            size_t fr = 0, tot = 0;
            HIP_TRY(c, hipMemGetInfo(&fr, &tot));
            c->tape_budget = std::min<size_t>(fr / 4, (size_t)16 << 30);
        }
        cap = (int64_t)(c->tape_budget / (lanes * recw * sizeof(double)));
        cap = std::max<int64_t>(cap, 64);
    }
    cap = std::min<int64_t>(cap, c->cfg.maxiters);
    if (!primal && c->tape_doubles < lanes * (size_t)cap * recw) {
        if (ensure(c, &c->d_tape, &c->tape_doubles, lanes * (size_t)cap * recw)) return -1;
    }
    const int nbatch = (int)((count + tpw - 1) / tpw);   // one partial row per batch of a wavefront (written by the solve kernel)
    if (ensure(c, &c->d_partials, &c->partials_cap, (size_t)nbatch * std::max(npart_th, npart))) return -1;
    if (c->npart_max < npart) {
        if (c->d_red) HIP_TRY(c, hipFree(c->d_red));
        c->d_red = nullptr;
        HIP_TRY(c, hipMalloc((void **)&c->d_red, sizeof(double) * npart));
        c->npart_max = npart;
    }
    if (want_pred && ensure_pred(c)) return -1;
    crnn::SolveParams prm{};
    fill_params(c, prm, P, first, count, n_save_active, want_pred);
    crnn::AdjParams adj{};
    adj.tape = c->d_tape; adj.tape_cap = (int32_t)cap; adj.overflow = c->d_overflow; adj.batch_partials = c->d_partials;
    adj.perm = nullptr;
    {   // Reverse sweep aligned by step index or by physical time (ros23_adj_kernel.hpp): by index when the save grid is
        // front-loaded like the steps behind a stiff transient (robertson's logarithmic grid: half of the points lie in the
        // first quarter of the horizon), by time on a uniform grid (case1 / case2).  Either way the results are the same.
        const int D = n_save_active;
        const double t_lo = c->cfg.t0, t_hi = c->tsave[D - 1];
        adj.align_rev = (D >= 4 && c->tsave[D / 2] - t_lo < 0.25 * (t_hi - t_lo)) ? 1 : 0;
        if (const char *e = getenv("CRNN_ADJ_ALIGN_REV")) { if (*e) adj.align_rev = atoi(e) != 0; }   // measurement override
    }
    // more trajectories than resident lanes: wavefronts take several batches from the queue one after the other; queue the
    // trajectories by their last known step counts so that batches are homogeneous (the counts of the previous launch over
    // the same range: in training p moves little from step to step)
    if (queue_by_steps(c, lanes, first, count, &adj.perm)) return -1;
#ifdef CRNN_ADJ_PROF
    static unsigned long long *d_aprof = nullptr;
    if (!d_aprof) HIP_TRY(c, hipMalloc((void **)&d_aprof, 16 * sizeof(unsigned long long)));
    adj.prof = d_aprof;
#endif
    if (upload_consts(c)) return -1;
    if (!c->flags_zeroed) {
        HIP_TRY(c, hipMemsetAsync(c->d_queue, 0, sizeof(unsigned long long), c->stream));
        HIP_TRY(c, hipMemsetAsync(c->d_overflow, 0, sizeof(unsigned int), c->stream));
    }
    c->flags_zeroed = false;
    c->ev0 = c->ring0[c->n_launch % Ctx::kRing];
    c->ev1 = c->ring1[c->n_launch % Ctx::kRing];
    ++c->n_launch;
    HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    const AdjKernelFn fn = !primal ? k->fn : (c->jac_mode == CRNN_JAC_FINITE_DIFF && k->fn_primal_fd) ? k->fn_primal_fd : k->fn_primal;
    hipLaunchKernelGGL(fn, dim3(nblk), dim3(kBlock), 0, c->stream, prm, d_theta, adj);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    // the next launch over this range will want the queue ordered by this launch's step counts (queue_by_steps): sort
    // them in the same launch as the reduction
    const bool sort_next = c->queue_order == CRNN_QUEUE_AUTO && ((size_t)count > lanes || count >= 2048) && count < ((int64_t)1 << 31);
    if (sort_next && ensure(c, &c->d_perm, &c->perm_cap, (size_t)count)) return -1;
    want_spread = want_spread && sort_next && !primal;
    if (want_spread && !c->h_spread) {
        // pinned, host-coherent, mapped: the sort launch's extra block writes the three numbers straight into the host ring (no copy
        // operation in the stream -- a 12-byte hipMemcpyAsync cost 30 us of every step); the event behind the launch orders the read
        HIP_TRY(c, hipHostMalloc((void **)&c->h_spread, 4 * Ctx::kSpreadRing * sizeof(int32_t), hipHostMallocMapped | hipHostMallocCoherent));
        HIP_TRY(c, hipHostGetDevicePointer((void **)&c->d_spread, c->h_spread, 0));
        for (int i = 0; i < Ctx::kSpreadRing; ++i) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_spread[i], hipEventDisableTiming));
    }
    int32_t *const spread = want_spread ? c->d_spread + 4 * (c->spread_n % Ctx::kSpreadRing) : nullptr;
    const unsigned sort_blocks = sort_next ? (unsigned)((count + 1023) / 1024) + (spread ? 1u : 0u) : 0u;
    if (c->fuse_opt && defer) {   // crnn_train_step without a communicator: reduction, optimiser update and the next launch's sort in one
        hipLaunchKernelGGL(reduce_opt_sort_kernel, dim3(1 + sort_blocks), dim3(1024), 0, c->stream,
                           c->d_partials, nbatch, d_dtheta, nth, P, c->d_red_theta, c->d_red, c->d_overflow, c->d_nacc, c->d_nrej,
                           first, (int)count, sort_next ? c->d_perm : nullptr, spread, c->opt, npart, c->d_p, c->d_opt, c->cfg.param_map,
                           c->cfg.ns, c->cfg.nr, c->nfx, c->d_theta, c->d_dtheta, c->d_queue, c->d_overflow, c->d_poison);
        c->opt_fused = true;
        c->perm_ready = sort_next; c->perm_first = first; c->perm_count = count;
    } else if (sort_next) {
        hipLaunchKernelGGL(crnn::reduce_project_sort_kernel, dim3(1 + sort_blocks), dim3(1024), 0, c->stream,
                           c->d_partials, nbatch, d_dtheta, nth, P, c->d_red_theta, c->d_red, c->d_overflow, c->d_nacc, c->d_nrej,
                           first, (int)count, c->d_perm, spread);
        c->perm_ready = true; c->perm_first = first; c->perm_count = count;
    } else {
        hipLaunchKernelGGL(crnn::reduce_project_kernel, dim3(1), dim3(1024), 0, c->stream, c->d_partials, nbatch, d_dtheta, nth, P,
                           c->d_red_theta, c->d_red, c->d_overflow);
        c->perm_ready = false;
    }
    HIP_TRY(c, hipGetLastError());
    if (spread) {   // the spread travels to the host behind the launch; launch n + 2 over the same range reads it
        const int slot = (int)(c->spread_n % Ctx::kSpreadRing);
        HIP_TRY(c, hipEventRecord(c->ev_spread[slot], c->stream));
        c->spread_first[slot] = first; c->spread_count[slot] = count;
        ++c->spread_n;
    } else if (!primal) c->spread_n = 0;    // another launch shape in between: the history starts over
    c->last_npart = npart;
    c->last_P = P;
    c->steps_first = first; c->steps_count = count;     // d_nacc / d_nrej of this range are current once the launch has run
#ifdef CRNN_ADJ2_PROF
    if (G == 2) {
        unsigned long long hp_[16];
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipMemcpyFromSymbol(hp_, HIP_SYMBOL(crnn::g_adj2_prof), sizeof(hp_)));
        double tot = 0;
        for (int q = 0; q < 14; ++q) tot += (double)hp_[q];
        fprintf(stderr, "[adj2_prof] wave 0 ticks %.0f:", tot);
        for (int q = 0; q < 14; ++q) fprintf(stderr, " %d:%.1f%%", q, 100.0 * (double)hp_[q] / (tot > 0 ? tot : 1));
        fprintf(stderr, "\n");
    }
#endif
#ifdef CRNN_ADJ_PROF
    {
        unsigned long long hp_[16];
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipMemcpy(hp_, d_aprof, sizeof(hp_), hipMemcpyDeviceToHost));
        double tot = 0;
        for (int k = 0; k < 16; ++k) tot += (double)hp_[k];
        fprintf(stderr, "[adj_prof] wave 0 ticks %.0f:", tot);
        for (int k = 0; k < 14; ++k) fprintf(stderr, " %d:%.1f%%", k, 100.0 * (double)hp_[k] / (tot > 0 ? tot : 1));
        fprintf(stderr, "\n");
    }
#endif
    if (defer) return 0;   // the device-resident training loop looks at the outcome later (check_pending)
    unsigned int ovf = 0;
    HIP_TRY(c, hipMemcpyAsync(&ovf, c->d_overflow, sizeof(ovf), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return ovf ? 1 : 0;
}

// HyChem: one kernel family (hychem_kernel.hpp).  P > 0: discrete-adjoint gradient in theta space (HBM accumulators)
// + chain rule; P == 0: primal + loss.  A trajectory that outruns the tape (there is no forward-tangent kernel for 211 parameters to
// fall back on): with the tape sized automatically the call is repeated with a quarter of the resident trajectories -- four
// times the records per lane from the same budget, the queue works through the ensemble in more generations -- until the
// records fit or a lane holds maxiters of them (no trajectory accepts more); with an explicit crnn_config.tape_steps it is an error.
int32_t launch_hychem(Ctx *c, const double *d_theta, const double *d_dtheta, int P, int64_t first, int64_t count,
                      int n_save_active, bool want_pred, int max_blocks = 0) {
    if (c->cfg.ns != 9 || c->cfg.nr != 10) return fail(c, "crnn_solve: the HyChem kernel is instantiated for ns = 9, nr = 10");
    if (!c->d_tabs || c->tabs_B != c->B) return fail(c, "crnn_solve: HyChem needs T/P tables (crnn_ctx_set_tables after crnn_ctx_set_data)");
    const int nth = c->n_theta;
    const int npart_th = nth + crnn::kExtra, npart = P + crnn::kTail;
    using KFn = void (*)(const crnn::SolveParams, const double *, const crnn::HyParams);
    // 128 trajectories per CU either way: one lane each in a 128-lane block (hychem_kernel: W's factors + parked state in LDS,
    // ~1.1 KB per trajectory) or a lane pair each in a 256-lane block (hychem2_kernel.hpp: every vector and W's rows distributed
    // over the pair, 512 registers per lane).  AUTO = the pair at every size: it works through each trajectory faster
    // (32 768: 8.12 -> 6.06 ms, all 262 144 of config 4 on one GPU: 35.3 -> 27.2 ms)
    // Gradient accumulators: the one-lane kernel keeps 210 per trajectory in HBM (global atomics, reduce_gacc_kernel); the pair
    // kernel sums over the 32 trajectories of a batch with FP64 MFMAs and writes one row per batch.
    // AutoTsit5(Rosenbrock23) (crnn_pyrolysis_mass.jl:29): primal launches run the composite (hychem_auto_kernel, a lane pair per
    // trajectory); gradient launches run the Rosenbrock23 adjoint with Rosenbrock23's controller constants (hychem_auto_kernel.hpp)
    const bool composite = c->cfg.solver == CRNN_SOLVER_AUTOTSIT5;
    // Rosenbrock23(autodiff = false) (crnn_ctx_set_jacobian): primal launches of either solver run hychem_auto_kernel's finite-difference
    // instantiations (the composite's, or its stiff branch alone for CRNN_SOLVER_ROSENBROCK23); gradient launches keep the analytic W
    const bool fd_primal = c->jac_mode == CRNN_JAC_FINITE_DIFF && P == 0;
    const bool auto_primal = (composite || fd_primal) && P == 0;
    const int G = (c->lanes_per_traj == 1 && !auto_primal) ? 1 : 2;
    const size_t gacc_need = G == 1 ? (size_t)((count + 63) / 64) * nth * 64 : 0;
    c->last_lanes = G;
    const int kHyBlock = 128 * G;
    KFn fn = auto_primal ? (fd_primal ? (composite ? (KFn)crnn::hychem_auto_kernel<9, 10, 256, true, false> : (KFn)crnn::hychem_auto_kernel<9, 10, 256, true, true>)
                                      : (KFn)crnn::hychem_auto_kernel<9, 10, 256>)
             : G == 2 ? (P > 0 ? (KFn)crnn::hychem2_kernel<9, 10, true, 256> : (KFn)crnn::hychem2_kernel<9, 10, false, 256>)
                      : (P > 0 ? (KFn)crnn::hychem_kernel<9, 10, true, 128> : (KFn)crnn::hychem_kernel<9, 10, false, 128>);
    struct CtlGuard {   // a gradient launch of a composite context: Rosenbrock23's PI exponents and steady band for this launch
        Ctx *c; bool on; double b1, b2, qs;
        CtlGuard(Ctx *c_, bool on_) : c(c_), on(on_), b1(c_->cfg.beta1), b2(c_->cfg.beta2), qs(c_->cfg.qsteady_max) {
            if (on) { c->cfg.beta1 = 7.0 / 20.0; c->cfg.beta2 = 2.0 / 10.0; c->cfg.qsteady_max = 1.2; c->kc_dirty = true; }
        }
        ~CtlGuard() { if (on) { c->cfg.beta1 = b1; c->cfg.beta2 = b2; c->cfg.qsteady_max = qs; c->kc_dirty = true; } }
    } ctl_guard(c, composite && P > 0);
    int occ = 0;
    HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)fn, kHyBlock, 0));
    if (occ < 1) occ = 1;
    const int64_t need_blocks = (count + 127) / 128;
    int nblk = (int)std::max<int64_t>(1, std::min<int64_t>(need_blocks, (int64_t)c->num_cu * occ));
    if (max_blocks == 0 && P > 0 && c->hy_block_cap > 0 && c->hy_cap_first == first && c->hy_cap_count == count) {
        // the remembered width is a property of the parameters that overflowed, not of the ensemble: training moves on, so every
        // kHyCapReprobe-th launch tries four times the width again (one repeated launch in kHyCapReprobe if it still does not fit)
        constexpr int kHyCapReprobe = 16;
        max_blocks = c->hy_block_cap;
        if (++c->hy_cap_uses % kHyCapReprobe == 0) {
            max_blocks = (int)std::min<int64_t>((int64_t)max_blocks * 4, (int64_t)nblk);
            if (max_blocks >= nblk) { max_blocks = 0; c->hy_block_cap = 0; c->hy_cap_uses = 0; }     // back at full width: forget the cap unless it overflows again
        }
    }
    if (max_blocks > 0) nblk = std::min(nblk, max_blocks);
    const size_t lanes = (size_t)nblk * 128;      // resident trajectories = tape slots
    const size_t recw = (size_t)c->cfg.ns + 2;
    int64_t cap = c->cfg.tape_steps;
    if (cap <= 0) {
        if (c->tape_budget == 0) {
            size_t fr = 0, tot = 0;
            HIP_TRY(c, hipMemGetInfo(&fr, &tot));
            c->tape_budget = std::min<size_t>(fr / 4, (size_t)16 << 30);
            if (const char *e = getenv("CRNN_TAPE_BUDGET_BYTES")) {   // test override: a budget small enough to exercise the overflow path
                const long long v = atoll(e);
                if (v > 0) c->tape_budget = (size_t)v;
            }
        }
        cap = std::max<int64_t>((int64_t)(c->tape_budget / (lanes * recw * sizeof(double))), 64);
    }
    cap = std::min<int64_t>(cap, c->cfg.maxiters);
    if (c->tape_doubles < lanes * (size_t)cap * recw && ensure(c, &c->d_tape, &c->tape_doubles, lanes * (size_t)cap * recw)) return -1;
    const int rblk = (int)((count + 255) / 256);
    const int wrows = (G == 2 && P > 0) ? (int)((count + 31) / 32) : 0;   // the pair kernel's rows of the partial-sum table: one per batch of 32
    if (ensure(c, &c->d_partials, &c->partials_cap, (size_t)(wrows + rblk) * std::max(npart_th, npart))) return -1;
    if (P > 0 && G == 1 && ensure(c, &c->d_gacc, &c->gacc_cap, gacc_need)) return -1;
    if (c->npart_max < npart) {
        if (c->d_red) HIP_TRY(c, hipFree(c->d_red));
        c->d_red = nullptr;
        HIP_TRY(c, hipMalloc((void **)&c->d_red, sizeof(double) * npart));
        c->npart_max = npart;
    }
    if (want_pred && ensure_pred(c)) return -1;
    crnn::SolveParams prm{};
    fill_params(c, prm, P, first, count, n_save_active, want_pred);
    crnn::HyParams hp{};
    hp.tabs = c->d_tabs; hp.tape = c->d_tape; hp.tape_cap = (int32_t)cap; hp.overflow = c->d_overflow;
    hp.gacc = G == 2 ? c->d_partials : c->d_gacc;
    hp.n_save_total = c->cfg.n_save; hp.inv_R = c->cfg.inv_R;
    if (queue_by_steps(c, lanes, first, count, &hp.perm)) return -1;
#ifdef HY_PROF
    static unsigned long long *d_prof = nullptr;
    if (!d_prof) HIP_TRY(c, hipMalloc((void **)&d_prof, 16 * sizeof(unsigned long long)));
    HIP_TRY(c, hipMemsetAsync(d_prof, 0, 16 * sizeof(unsigned long long), c->stream));
    hp.prof = d_prof;
#endif
    if (upload_consts(c)) return -1;
    if (!c->flags_zeroed) {
        HIP_TRY(c, hipMemsetAsync(c->d_queue, 0, sizeof(unsigned long long), c->stream));
        HIP_TRY(c, hipMemsetAsync(c->d_overflow, 0, sizeof(unsigned int), c->stream));
    }
    c->flags_zeroed = false;
    if (P > 0 && G == 1) HIP_TRY(c, hipMemsetAsync(c->d_gacc, 0, sizeof(double) * gacc_need, c->stream));
    c->ev0 = c->ring0[c->n_launch % Ctx::kRing];
    c->ev1 = c->ring1[c->n_launch % Ctx::kRing];
    ++c->n_launch;
    HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    hipLaunchKernelGGL(fn, dim3(nblk), dim3(kHyBlock), 0, c->stream, prm, d_theta, hp);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    if (P > 0) {
        if (G == 2)
            hipLaunchKernelGGL(crnn::hy2_extras_kernel, dim3(rblk), dim3(256), 0, c->stream, c->d_partials + (size_t)wrows * npart_th, nth,
                               c->d_loss, c->d_ret, c->d_nacc, c->d_nrej, first, count);
        else
            hipLaunchKernelGGL(crnn::reduce_gacc_kernel, dim3(rblk), dim3(256), 0, c->stream, c->d_gacc, nth, c->n_obs, c->d_loss,
                               c->d_ret, c->d_nsaved, c->d_nacc, c->d_nrej, first, count, hp.perm, c->d_partials);
        HIP_TRY(c, hipGetLastError());
        hipLaunchKernelGGL(crnn::reduce_project_kernel, dim3(1), dim3(1024), 0, c->stream, c->d_partials, wrows + rblk, d_dtheta, nth, P,
                           c->d_red_theta, c->d_red, (const unsigned int *)nullptr);
        HIP_TRY(c, hipGetLastError());
    } else {
        hipLaunchKernelGGL(crnn::reduce_traj_kernel, dim3(rblk), dim3(256), 0, c->stream, c->d_gtraj, 0, c->d_loss, c->d_ret,
                           c->d_nacc, c->d_nrej, first, count, 256, c->d_partials);
        HIP_TRY(c, hipGetLastError());
        hipLaunchKernelGGL(crnn::reduce_partials_kernel, dim3(crnn::kExtra), dim3(256), 0, c->stream, c->d_partials, rblk, 0, 0, c->d_red);
        HIP_TRY(c, hipGetLastError());
    }
    unsigned int ovf = 0;
    HIP_TRY(c, hipMemcpyAsync(&ovf, c->d_overflow, sizeof(ovf), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->last_npart = npart;
    c->last_P = P;
    c->steps_first = first; c->steps_count = count;
#ifdef HY_PROF
    {
        unsigned long long hprof[16];
        HIP_TRY(c, hipMemcpy(hprof, d_prof, sizeof(hprof), hipMemcpyDeviceToHost));
        unsigned long long tot = 0;
        for (int k = 0; k < 16; ++k) tot += hprof[k];
        fprintf(stderr, "[hy_prof] P=%d ticks(100MHz):", P);
        for (int k = 0; k < 16; ++k) fprintf(stderr, " %d:%.1f%%", k, 100.0 * (double)hprof[k] / (double)(tot ? tot : 1));
        fprintf(stderr, " total %.3f ms\n", (double)tot / 1e5);
    }
#endif
    if (ovf) {
        if (c->cfg.tape_steps > 0 || cap >= c->cfg.maxiters || nblk <= 1)
            return fail(c, "crnn_solve: a trajectory accepted more steps than the adjoint tape holds; raise crnn_config.tape_steps");
        c->hy_tape_retries++;
        return launch_hychem(c, d_theta, d_dtheta, P, first, count, n_save_active, want_pred, std::max(1, nblk / 4));
    }
    if (max_blocks > 0 && P > 0) {     // what fitted
        if (c->hy_block_cap != nblk) c->hy_cap_uses = 0;
        c->hy_block_cap = nblk; c->hy_cap_first = first; c->hy_cap_count = count;
    }
    return 0;
}

int32_t launch_solve(Ctx *c, const double *d_theta, const double *d_dtheta, int P, int64_t first, int64_t count,
                     int n_save_active, bool want_pred, bool want_percase);

// One ForwardDiff chunk with the dual-inclusive error norm: primal + P <= C*L tangent columns through every attempt
// (ros23_sens_kernel.hpp), fixed-order reduction into c->d_red in the common layout.
// n_chunks > 1 (ros23_sens_kernel with drows > P): d_dtheta holds all P directions, the launch runs the n_chunks chunks of
// chunk_size partials each (the last one short), per-trajectory gradient rows come out compact [count][P]; the per-trajectory
// losses and statistics are not written (launch_sens follows with the plain solve).
int32_t launch_sens_chunk(Ctx *c, const KernelEntry *k, const double *d_theta, const double *d_dtheta, int P, int64_t first,
                          int64_t count, int n_save_active, bool want_pred, int dual_partials, int n_chunks = 1, int chunk_size = 0) {
    const int blk = kSensBlock;
    const int C = k->C, L = k->L, gpw = 64 / L, waves = blk / 64;
    const int ppad = n_chunks > 1 ? P : L * C, npart_pad = ppad + crnn::kExtra, npart = P + crnn::kTail;
    int occ = 0;
    HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)k->fn, blk, 0));
    if (occ < 1) occ = 1;
    const int64_t nbatch = ((count + gpw - 1) / gpw) * std::max(1, n_chunks);
    const int nblk = (int)std::max<int64_t>(1, std::min<int64_t>((nbatch + waves - 1) / waves, (int64_t)c->num_cu * occ));
    const int rblk = (int)((count + 255) / 256);
    if (ensure(c, &c->d_partials, &c->partials_cap, (size_t)rblk * npart_pad)) return -1;
    if (ensure(c, &c->d_gtraj, &c->gtraj_cap, (size_t)count * ppad)) return -1;
    if (c->npart_max < npart) {
        if (c->d_red) HIP_TRY(c, hipFree(c->d_red));
        c->d_red = nullptr;
        HIP_TRY(c, hipMalloc((void **)&c->d_red, sizeof(double) * npart));
        c->npart_max = npart;
    }
    if (want_pred && ensure_pred(c)) return -1;
    crnn::SolveParams prm{};
    fill_params(c, prm, P, first, count, n_save_active, want_pred);
    prm.norm_cols = c->cfg.errnorm_sens == 2 ? dual_partials : 0;
    // batches in the order of the last plain solve's step counts (the chunk's own counts differ little from them); the order
    // stays with the context for the plain solve that ends this gradient call (queue_by_steps)
    prm.perm = (c->queue_order == CRNN_QUEUE_AUTO && c->perm_ready && c->perm_first == first && c->perm_count == count) ? c->d_perm : nullptr;
    prm.n_chunks = n_chunks; prm.chunk_size = chunk_size;
    if (upload_consts(c)) return -1;
    if (!c->flags_zeroed) HIP_TRY(c, hipMemsetAsync(c->d_queue, 0, sizeof(unsigned long long), c->stream));
    c->flags_zeroed = false;
    c->ev0 = c->ring0[c->n_launch % Ctx::kRing];
    c->ev1 = c->ring1[c->n_launch % Ctx::kRing];
    ++c->n_launch;
    HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    hipLaunchKernelGGL(k->fn, dim3(nblk), dim3(blk), 0, c->stream, prm, d_theta, d_dtheta);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
#ifdef CRNN_SENS_PROF
    if (k->solver == CRNN_SOLVER_ROSENBROCK23) {
        unsigned long long hp_[16];
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipMemcpyFromSymbol(hp_, HIP_SYMBOL(crnn::g_sens_prof), sizeof(hp_)));
        double tot = 0;
        for (int q = 0; q < 16; ++q) tot += (double)hp_[q];
        fprintf(stderr, "[sens_prof] wave 0 ticks %.0f:", tot);
        for (int q = 0; q < 12; ++q) fprintf(stderr, " %d:%.1f%%", q, 100.0 * (double)hp_[q] / (tot > 0 ? tot : 1));
        fprintf(stderr, "\n");
    }
#endif
    hipLaunchKernelGGL(crnn::reduce_traj_kernel, dim3(rblk), dim3(256), 0, c->stream, c->d_gtraj, ppad, c->d_loss, c->d_ret,
                       c->d_nacc, c->d_nrej, first, count, 256, c->d_partials);
    HIP_TRY(c, hipGetLastError());
    hipLaunchKernelGGL(crnn::reduce_partials_kernel, dim3(npart_pad), dim3(256), 0, c->stream, c->d_partials, rblk, ppad, P, c->d_red);
    HIP_TRY(c, hipGetLastError());
    c->last_npart = npart;
    c->last_P = P;
    c->steps_first = first; c->steps_count = count;
    return 0;
}

// The rows of d theta / d p that p2vec_kernel writes: for the HyChem map (crnn_pyrolysis_mass.jl:78-90) and the identity every row has at
// most one entry in w_in's species / log T rows and one in w_out (hychem_sens2_kernel.hpp's description of a direction)
static bool pmap_dirs_sparse(const Ctx *c) {
    return c->hychem && (c->cfg.param_map == CRNN_PMAP_HYCHEM || c->cfg.param_map == CRNN_PMAP_IDENTITY);
}

// HyChem: one ForwardDiff chunk of <= 12 directions, a group of twelve lanes per trajectory (hychem_sens_kernel.hpp); the same
// per-trajectory gradient rows and fixed-order reduction as launch_sens_chunk.
int32_t launch_hychem_sens_chunk(Ctx *c, const double *d_theta, const double *d_dtheta, int P, int64_t first, int64_t count,
                                 int n_save_active, bool want_pred, int dual_partials, int n_chunks = 1) {
    if (c->cfg.ns != 9 || c->cfg.nr != 10) return fail(c, "crnn_solve: the HyChem kernel is instantiated for ns = 9, nr = 10");
    if (!c->d_tabs || c->tabs_B != c->B) return fail(c, "crnn_solve: HyChem needs T/P tables (crnn_ctx_set_tables after crnn_ctx_set_data)");
    // two kernels, one result: hychem_sens_kernel (dense directions through nested duals, twelve lanes per trajectory) -- the default: the one
    // a device has executed (round 4) -- and hychem_sens2_kernel (sparse directions, closed-form tangents; every direction of the launch has to
    // fit its description -- the rows of p2vec's Jacobian always do): CRNN_HY_SENS_KERNEL=2, and the only one the composite has.  The sparse
    // kernel is parity-green under SIMT emulation only; it becomes the default once tests/test_hychem.py's sparse / errnorm tests have passed
    // on an MI355X
    const bool comp = c->cfg.solver == CRNN_SOLVER_AUTOTSIT5;      // the reference's composite inside the gradient: the sparse kernel only
    if (comp && !c->hy_dirs_sparse)
        return fail(c, "crnn_solve: the dual-norm gradient through AutoTsit5 takes the rows of p2vec's Jacobian (one entry of w_in's species / log T rows and one of w_out per direction); arbitrary directions run with solver = ROSENBROCK23");
    const bool sparse = comp || (c->hy_dirs_sparse && c->hy_sens_kernel == 2);
    constexpr int kC = 12;
    constexpr int kL2 = 12, kBlk2 = 256, kGroups2 = (kBlk2 / 64) * (64 / kL2);   // (L = 6: two columns per lane, 1.45x fewer issue slots per trajectory by the static count, but 1.2 KB of scratch per lane -- tools/ubench/hy_sens2_probe.hip)
    constexpr int kBlk1 = 128, kGroups1 = (kBlk1 / 64) * (64 / kC);
    const int kBlk = sparse ? kBlk2 : kBlk1, kGroups = sparse ? kGroups2 : kGroups1;
    const int ppad = n_chunks > 1 ? P : kC;      // gradient row: the chunk's 12 columns | all chunks in one launch: compact [P]
    const int npart_pad = ppad + crnn::kExtra, npart = P + crnn::kTail;
    using SFn = void (*)(const crnn::SolveParams, const double *, const crnn::HyParams, const crnn::HySensParams);
    const SFn fn = comp ? (SFn)crnn::hychem_sens2_kernel<9, 10, kL2, kBlk2, true>
                        : sparse ? (SFn)crnn::hychem_sens2_kernel<9, 10, kL2, kBlk2, false> : (SFn)crnn::hychem_sens_kernel<9, 10, kBlk1>;
    int &occ = comp ? c->hysens2c_occ : sparse ? c->hysens2_occ : c->hysens_occ;
    if (occ < 1) {
        HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)fn, kBlk, 0));
        if (occ < 1) occ = 1;
    }
    const int nch = std::max(1, n_chunks);
    const int nblk = nch * (int)std::max<int64_t>(1, std::min<int64_t>((count + kGroups - 1) / kGroups, std::max<int64_t>(1, (int64_t)c->num_cu * occ / nch)));
    const int rblk = (int)((count + 255) / 256);
    if (ensure(c, &c->d_partials, &c->partials_cap, (size_t)rblk * npart_pad)) return -1;
    if (ensure(c, &c->d_gtraj, &c->gtraj_cap, (size_t)count * ppad)) return -1;
    if (c->npart_max < npart) {
        if (c->d_red) HIP_TRY(c, hipFree(c->d_red));
        c->d_red = nullptr;
        HIP_TRY(c, hipMalloc((void **)&c->d_red, sizeof(double) * npart));
        c->npart_max = npart;
    }
    if (want_pred && ensure_pred(c)) return -1;
    crnn::SolveParams prm{};
    fill_params(c, prm, P, first, count, n_save_active, want_pred);
    crnn::HyParams hp{};
    hp.tabs = c->d_tabs; hp.n_save_total = c->cfg.n_save; hp.inv_R = c->cfg.inv_R;
    // batches in the order of the last plain solve's step counts (launch_sens_chunk does the same; only hychem_sens2_kernel reads it)
    hp.perm = (c->queue_order == CRNN_QUEUE_AUTO && c->perm_ready && c->perm_first == first && c->perm_count == count) ? c->d_perm : nullptr;
    crnn::HySensParams sp{};
    sp.dth = d_dtheta; sp.n_dir = P; sp.mode = c->cfg.errnorm_sens; sp.dual_partials = dual_partials;
    sp.n_chunks = n_chunks; sp.n_total = P;
    if (upload_consts(c)) return -1;
    c->flags_zeroed = false;
    c->ev0 = c->ring0[c->n_launch % Ctx::kRing];
    c->ev1 = c->ring1[c->n_launch % Ctx::kRing];
    ++c->n_launch;
    HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    hipLaunchKernelGGL(fn, dim3(nblk), dim3(kBlk), 0, c->stream, prm, d_theta, hp, sp);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    hipLaunchKernelGGL(crnn::reduce_traj_kernel, dim3(rblk), dim3(256), 0, c->stream, c->d_gtraj, ppad, c->d_loss, c->d_ret,
                       c->d_nacc, c->d_nrej, first, count, 256, c->d_partials);
    HIP_TRY(c, hipGetLastError());
    hipLaunchKernelGGL(crnn::reduce_partials_kernel, dim3(npart_pad), dim3(256), 0, c->stream, c->d_partials, rblk, ppad, P, c->d_red);
    HIP_TRY(c, hipGetLastError());
    c->last_npart = npart;
    c->last_P = P;
    c->steps_first = first; c->steps_count = count;
    return 0;
}

// errnorm_sens = 1.  P <= C*L (crnn_solve: the caller's directions are ONE chunk): a single launch.  Otherwise
// (crnn_loss_grad, crnn_train_step: the P parameters) ForwardDiff's chunking: consecutive chunks of fd_chunk_size(P)
// partials, each its own adaptive solve, the gradient pieces concatenated; per-trajectory losses, return codes and the
// step statistics are those of a final plain solve -- what loss_neuralode(p) evaluates (case2/case2.jl:199-201).
int32_t launch_sens(Ctx *c, const double *d_theta, const double *d_dtheta, int P, int64_t first, int64_t count,
                    int n_save_active, bool want_pred, bool single_chunk) {
    const KernelEntry *k = c->hychem ? nullptr : find_sens(c);
    if (!k && !c->hychem) return fail(c, "crnn_solve: errnorm_sens = 1 has no kernel for this (solver, ns, nr, has_temp)");
    const int cap = c->hychem ? 12 : k->C * k->L;
    auto chunk_launch = [&](const double *dth, int Pc, bool pred, int partials) -> int32_t {
        return c->hychem ? launch_hychem_sens_chunk(c, d_theta, dth, Pc, first, count, n_save_active, pred, partials)
                         : launch_sens_chunk(c, k, d_theta, dth, Pc, first, count, n_save_active, pred, partials);
    };
    if (single_chunk) {
        if (P > cap) return fail(c, "crnn_solve: with errnorm_sens = 1 the directions of one call are one ForwardDiff chunk: n_dir <= " + std::to_string(cap));
        return chunk_launch(d_dtheta, P, want_pred, P);
    }
    const int chunk = fd_chunk_size(P);
    if (chunk > cap) return fail(c, "crnn_loss_grad: errnorm_sens = 1 chunk size exceeds the instantiated kernel");
    const int npart = P + crnn::kTail;
    if (c->red_asm_len < npart) {
        if (c->d_red_asm) HIP_TRY(c, hipFree(c->d_red_asm));
        c->d_red_asm = nullptr;
        HIP_TRY(c, hipMalloc((void **)&c->d_red_asm, sizeof(double) * npart));
        c->red_asm_len = npart;
    }
    if (c->hychem && c->sens_one_launch && P <= 255 && chunk == 12) {
        // HyChem: the 18 chunks of 12 as the blocks of one launch (hychem_sens_kernel, n_chunks > 1)
        if (launch_hychem_sens_chunk(c, d_theta, d_dtheta, P, first, count, n_save_active, false, chunk, (P + chunk - 1) / chunk)) return -1;
        HIP_TRY(c, hipMemcpyAsync(c->d_red_asm, c->d_red, sizeof(double) * P, hipMemcpyDeviceToDevice, c->stream));
    } else if (k && k->drows > P && c->sens_one_launch) {
        // all chunks in one launch (ros23_sens_kernel, n_chunks > 1): the same independent adaptive solves, one tail
        if (launch_sens_chunk(c, k, d_theta, d_dtheta, P, first, count, n_save_active, false, chunk, (P + chunk - 1) / chunk, chunk)) return -1;
        HIP_TRY(c, hipMemcpyAsync(c->d_red_asm, c->d_red, sizeof(double) * P, hipMemcpyDeviceToDevice, c->stream));
    } else {
        for (int k0 = 0; k0 < P; k0 += chunk) {
            const int Pc = std::min(chunk, P - k0);
            // every Dual of a chunked ForwardDiff.gradient carries `chunk` partials, the last chunk's surplus ones are zero
            if (chunk_launch(d_dtheta + (size_t)k0 * c->n_theta, Pc, false, chunk)) return -1;
            HIP_TRY(c, hipMemcpyAsync(c->d_red_asm + k0, c->d_red, sizeof(double) * Pc, hipMemcpyDeviceToDevice, c->stream));
        }
    }
    const int es = c->cfg.errnorm_sens;
    c->cfg.errnorm_sens = 0;      // the plain solve
    const int32_t rc = launch_solve(c, d_theta, d_dtheta, 0, first, count, n_save_active, want_pred, false);
    c->cfg.errnorm_sens = es;
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->d_red_asm + P, c->d_red, sizeof(double) * crnn::kTail, hipMemcpyDeviceToDevice, c->stream));
    if (c->npart_max < npart) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (c->d_red) HIP_TRY(c, hipFree(c->d_red));
        c->d_red = nullptr;
        HIP_TRY(c, hipMalloc((void **)&c->d_red, sizeof(double) * npart));
        c->npart_max = npart;
    }
    HIP_TRY(c, hipMemcpyAsync(c->d_red, c->d_red_asm, sizeof(double) * npart, hipMemcpyDeviceToDevice, c->stream));
    c->last_npart = npart;
    c->last_P = P;
    c->steps_first = first; c->steps_count = count;
    return 0;
}

// Launch solve (+ fixed-order reduction into c->d_red).  theta/dtheta already on device.
int32_t launch_solve(Ctx *c, const double *d_theta, const double *d_dtheta, int P, int64_t first, int64_t count,
                     int n_save_active, bool want_pred, bool want_percase) {
    if (c->B <= 0) return fail(c, "crnn_solve: no ensemble uploaded (crnn_ctx_set_data)");
    if (first < 0 || count <= 0 || first + count > c->B) return fail(c, "crnn_solve: [first, first+count) outside the ensemble");
    if (n_save_active <= 0 || n_save_active > c->cfg.n_save) return fail(c, "crnn_solve: n_save_active out of range");
    if (c->hychem) {
        if (c->cfg.errnorm_sens && P > 0) return launch_sens(c, d_theta, d_dtheta, P, first, count, n_save_active, want_pred, want_percase);
        if (P > 0 && c->cfg.grad_mode == CRNN_GRAD_FORWARD)
            return fail(c, "crnn_solve: HyChem's forward tangents exist with the dual-inclusive error norm only (errnorm_sens = 1 / 2); the primal-norm gradient is the discrete adjoint (grad_mode AUTO or ADJOINT)");
        return launch_hychem(c, d_theta, d_dtheta, P, first, count, n_save_active, want_pred);
    }
    c->last_deferred = false;
    if (c->cfg.errnorm_sens && P > 0) return launch_sens(c, d_theta, d_dtheta, P, first, count, n_save_active, want_pred, want_percase);
    if (c->cfg.solver == CRNN_SOLVER_AUTOTSIT5) {
        // the composite exists as a tape kernel only: primal calls run it with P = 0, and there is no forward-tangent fallback
        if (P > 0 && c->cfg.grad_mode == CRNN_GRAD_FORWARD)
            return fail(c, "crnn_solve: AutoTsit5 gradients exist as discrete adjoint only (grad_mode AUTO or ADJOINT)");
        const AdjEntry *ka = find_adjoint(c);
        if (!ka) return fail(c, "crnn_solve: no AutoTsit5 kernel instantiated for this (ns, nr, has_temp)");
        const int32_t r = launch_adjoint(c, ka, d_theta, d_dtheta, P, first, count, n_save_active, want_pred, false);
        if (r > 0) return fail(c, "crnn_solve: a trajectory accepted more steps than the adjoint tape holds; raise crnn_config.tape_steps");
        return r;
    }
    // Tiny ensembles (the reference's own schedule is one experiment per update): with most lanes idle anyway, one tangent
    // column per lane -- a whole lane group per trajectory, a single sweep -- has the shorter critical path than the
    // adjoint's forward + reverse sweeps (case2, B = 1: 0.28 vs 0.33 ms per launch, 0.33 vs 0.42 ms per call; the break-even
    // is where the lane groups fill the chip).  grad_mode AUTO only; the gradient is the same derivative either way.
    // (Round 3: where a two-lanes-per-trajectory adjoint kernel exists it is faster still on tiny ensembles -- case2, one
    //  trajectory: 0.196 ms per launch against 0.279 for one column per lane and 0.307 for the one-lane adjoint; 32: 0.209 /
    //  0.312 / 0.358 -- so those shapes stay on the adjoint; robertson keeps the column-per-lane path.)
    const KernelEntry *k_small = nullptr;
    const bool two_lane_adjoint = c->cfg.solver == CRNN_SOLVER_ROSENBROCK23 && c->lanes_per_traj != 1 && find_adjoint2(c) != nullptr;
    if (P > 0 && c->cfg.grad_mode == CRNN_GRAD_AUTO && !c->force_forward && !two_lane_adjoint) {
        for (const auto &ke : kKernels)
            if (shape_match(c, ke) && ke.C == 1 && ke.L >= P && count <= (int64_t)c->num_cu * 4 * (64 / ke.L)) k_small = &ke;
    }
    if (P == 0 && c->cfg.errnorm_sens == 0) {   // primal solve: the adjoint kernel's forward sweep (wave-synchronous batches, sorted queue)
        const AdjEntry *ka = find_adjoint(c);
        if (ka && ka->fn_primal) {
            c->last_deferred = false;
            const int32_t r = launch_adjoint(c, ka, d_theta, d_dtheta, 0, first, count, n_save_active, want_pred, false);
            return r < 0 ? r : 0;
        }
    }
    if (P > 0 && c->cfg.grad_mode != CRNN_GRAD_FORWARD && !c->force_forward && !k_small) {
        const AdjEntry *ka = find_adjoint(c);
        if (ka) {
            const bool defer = c->defer_next;
            const int32_t r = launch_adjoint(c, ka, d_theta, d_dtheta, P, first, count, n_save_active, want_pred, defer);
            c->last_deferred = defer && r == 0;
            if (r <= 0) return r;
            ++c->n_fallback;  // tape overflow: same call, forward tangents (results are overwritten)
        } else if (c->cfg.grad_mode == CRNN_GRAD_ADJOINT) {
            return fail(c, "crnn_solve: grad_mode = ADJOINT but no adjoint kernel exists for this (solver, ns, nr, has_temp)");
        }
    }
    const KernelEntry *k = k_small ? k_small : pick_kernel(c, P);
    if (!k) return fail(c, "crnn_solve: no gfx950 kernel instantiated for this (ns, nr, has_temp, n_dir) shape");
    const int C = k->C, L = k->L;
    const int gpw = 64 / L;
    const int waves = kBlock / 64;
    const int ppad = C > 0 ? L * C : 0;
    const int npart_pad = ppad + crnn::kExtra;   // per-block partial rows (padded tangent columns)
    const int npart = P + crnn::kTail;           // the reduced vector: the common layout of every gradient path

    int occ = 0;
    HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)k->fn, kBlock, 0));
    if (occ < 1) occ = 1;
    int64_t groups_per_block = (int64_t)waves * gpw;
    int64_t need_blocks = (count + groups_per_block - 1) / groups_per_block;
    int64_t resident = (int64_t)c->num_cu * occ;
    int nblk = (int)std::max<int64_t>(1, std::min<int64_t>(need_blocks, resident));

    // fixed-order ensemble reduction geometry: depends on count only
    const int rows_per_block = 256;
    const int rblk = (int)((count + rows_per_block - 1) / rows_per_block);
    if (ensure(c, &c->d_partials, &c->partials_cap, (size_t)rblk * npart_pad)) return -1;
    if (C > 0 && ensure(c, &c->d_gtraj, &c->gtraj_cap, (size_t)count * ppad)) return -1;
    if (c->npart_max < npart) {
        if (c->d_red) HIP_TRY(c, hipFree(c->d_red));
        HIP_TRY(c, hipMalloc((void **)&c->d_red, sizeof(double) * npart));
        c->npart_max = npart;
    }
    if (want_pred && ensure_pred(c)) return -1;
    (void)want_percase;

    crnn::SolveParams prm{};
    fill_params(c, prm, P, first, count, n_save_active, want_pred);
    if (upload_consts(c)) return -1;
    HIP_TRY(c, hipMemsetAsync(c->d_queue, 0, sizeof(unsigned long long), c->stream));
    c->flags_zeroed = false;

    c->ev0 = c->ring0[c->n_launch % Ctx::kRing];
    c->ev1 = c->ring1[c->n_launch % Ctx::kRing];
    ++c->n_launch;
    HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    hipLaunchKernelGGL(k->fn, dim3(nblk), dim3(kBlock), 0, c->stream, prm, d_theta, d_dtheta);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    hipLaunchKernelGGL(crnn::reduce_traj_kernel, dim3(rblk), dim3(256), 0, c->stream, c->d_gtraj, ppad, c->d_loss, c->d_ret,
                       c->d_nacc, c->d_nrej, first, count, rows_per_block, c->d_partials);
    HIP_TRY(c, hipGetLastError());
    hipLaunchKernelGGL(crnn::reduce_partials_kernel, dim3(npart_pad), dim3(256), 0, c->stream, c->d_partials, rblk, ppad, P,
                       c->d_red);
    HIP_TRY(c, hipGetLastError());
    c->last_npart = npart;
    c->last_P = P;
    c->steps_first = first; c->steps_count = count;
    return 0;
}

int32_t fill_stats(Ctx *c, const double *red_host, int npart, crnn_stats *st) {
    if (!st) return 0;
    st->n_traj = (int64_t)llround(red_host[npart - 1]);
    st->n_ok = (int64_t)llround(red_host[npart - 4]);
    st->n_accept = (int64_t)llround(red_host[npart - 3]);
    st->n_reject = (int64_t)llround(red_host[npart - 2]);
    float ms = 0.f;
    HIP_TRY(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    st->kernel_ms = ms;
    return 0;
}

}  // namespace

// ---- cathode-UQ context (entry points at the end of the file) ----
namespace {
// Workspace of an SVGD move (device buffers sized for N x dim), kept between calls: crnn_svgd_update caches one per device,
// a cathode ctx owns one for its device-resident loop.  Nothing is allocated in steady state.
struct SvgdWs {
    int64_t N = 0;
    int dim = 0, nchunk = 0;
    double *d_p = nullptr, *d_g = nullptr, *d_new = nullptr, *d_dt = nullptr, *d_rep = nullptr, *d_part = nullptr;
    unsigned int *d_hist = nullptr;
    crnn::SvgdSel *d_sel = nullptr;
    double *d_dist = nullptr;              // the N (N - 1) / 2 pair distances (null: too many to store -- recomputed per pass)
    unsigned long long *d_cnt = nullptr;   // svgd_next_kernel's { #less, #equal, min above }
    void release() {
        for (void *q : {(void *)d_p, (void *)d_g, (void *)d_new, (void *)d_dt, (void *)d_rep, (void *)d_part, (void *)d_hist, (void *)d_sel,
                        (void *)d_dist, (void *)d_cnt})
            if (q) (void)hipFree(q);
        *this = SvgdWs{};
    }
};

hipError_t svgd_ws_reserve(SvgdWs &w, int64_t N, int dim, bool io_buffers);
hipError_t svgd_enqueue(SvgdWs &w, hipStream_t stream, const double *d_p, const double *d_g, int64_t N, int dim, double stepsize,
                        double h, double *d_new, double *d_dt, double *d_rep);
#ifndef CRNN_CATH_TAPE_EVERY
#define CRNN_CATH_TAPE_EVERY 1
#endif
struct CathCtx {
    crnn_cathode_config cfg{};
    std::string err;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int num_cu = 256;
    int n_sets = 0, Dmax = 0;
    double *d_ts = nullptr, *d_dbar = nullptr, *d_d2bar = nullptr, *d_beta = nullptr;
    int32_t *d_D = nullptr;
    unsigned long long *d_queue = nullptr;
    // per-call buffers
    double *d_theta = nullptr, *d_loss = nullptr, *d_grad = nullptr, *d_hrr = nullptr;
    int32_t *d_ret = nullptr, *d_nsv = nullptr, *d_nacc = nullptr, *d_nrej = nullptr;
    size_t cap_part = 0, cap_traj = 0, cap_hrr = 0;
    double *d_tape = nullptr;      // adjoint step tape
    size_t tape_doubles = 0, tape_budget = 0;
    unsigned int *d_overflow = nullptr;
    // particle-shard exchange (crnn_cathode_allgather)
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    double *d_ag_send = nullptr, *d_ag_recv = nullptr;
    size_t ag_send_cap = 0, ag_recv_cap = 0;
    int adj_occ = 0, fwd_occ = 0, prim_occ = 0;
    int solver = CRNN_CATH_SOLVER_ROSENBROCK23, auto_occ[2] = {0, 0};   // stepper of primal launches (crnn_cathode_set_solver)
    int errnorm_sens = 0, sens_occ[2] = {0, 0}, sensc_occ[2] = {0, 0};   // gradient launches as ForwardDiff evaluates them (crnn_cathode_set_errnorm_sens)
    double *d_dirscale = nullptr;                 // [17] d theta / d p of the chunked dual-norm gradient
    int64_t chunk_stats[4] = {0, 0, 0, 0};        // accepted / rejected steps of the two chunk launches of the last gradient call
    long long *d_chunk_stats = nullptr;           // the same on the device (summed there behind each chunk launch, read on request)
    bool chunk_stats_stale = false;
    double h_dirscale[CRNN_CATHODE_NP] = {};      // the p_scales of crnn_cathode_set_errnorm_sens (checked against crnn_cathode_set_particles)
    int tape_every = CRNN_CATH_TAPE_EVERY;   // adjoint tape: 1 = every step in full, 4 / 8 = checkpoint every 4th / 8th step
    // device-resident SVGD loop (crnn_cathode_set_particles / crnn_cathode_svgd_step)
    double *d_pn = nullptr, *d_pn2 = nullptr, *d_lnp = nullptr, *d_pscales = nullptr;   // particles (current / moved), lnpgrad, [p_scales(17) | mean loss, n_failed]
    size_t cap_pn = 0;
    int64_t n_particles = 0;
    double h_norm2[CRNN_CATHODE_NP] = {};
    hipEvent_t ev2 = nullptr, ev3 = nullptr;
    SvgdWs svgd;
};
// sum of the accepted / rejected step counts of a chunk launch into out[0], out[1] (wavefront sums, one atomic per wavefront)
__global__ __launch_bounds__(256) void cath_sum_counts_kernel(const int32_t *nacc, const int32_t *nrej, int64_t n, long long *out) {
    long long a = 0, r = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) { a += nacc[i]; r += nrej[i]; }
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); r += __shfl_down(r, off); }
    if ((threadIdx.x & 63) == 0) { atomicAdd((unsigned long long *)out, (unsigned long long)a); atomicAdd((unsigned long long *)out + 1, (unsigned long long)r); }
}
int32_t cfail(CathCtx *c, const std::string &msg) {
    g_last_error = msg;
    if (c) c->err = msg;
    return -1;
}
#define CHIP(c, expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return cfail(c, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
template <class T>
int32_t cgrow(CathCtx *c, T **p, size_t n) {
    if (*p) CHIP(c, hipFree(*p));
    *p = nullptr;
    CHIP(c, hipMalloc((void **)p, n * sizeof(T)));
    return 0;
}
}  // namespace

extern "C" {

int32_t crnn_abi_version(void) { return CRNN_ABI_VERSION; }

#ifndef CRNN_SRC_HASH
#define CRNN_SRC_HASH "unknown"
#endif
// "src=<first 16 hex digits of sha256 over the sorted csrc sources + include/crnn_hip.h> arch=gfx950": lets a host
// check that the loaded binary was built from the sources next to it (crnn_amd/_lib.py does, and rebuilds otherwise).
const char *crnn_build_info(void) { return "src=" CRNN_SRC_HASH " arch=gfx950"; }

int32_t crnn_debug_bounds(uint32_t *violations, uint32_t *first_site) {
#ifdef CRNN_BOUNDS_CHECK
    unsigned int v[2] = {0u, 0u}, zero[2] = {0u, 0u};
    if (hipDeviceSynchronize() != hipSuccess) return fail(nullptr, "crnn_debug_bounds: hipDeviceSynchronize failed (a kernel faulted?)");
    if (hipMemcpyFromSymbol(v, HIP_SYMBOL(crnn::g_bounds), sizeof(v)) != hipSuccess ||
        hipMemcpyToSymbol(HIP_SYMBOL(crnn::g_bounds), zero, sizeof(zero)) != hipSuccess)
        return fail(nullptr, "crnn_debug_bounds: cannot read the violation counters");
    if (violations) *violations = v[0];
    if (first_site) *first_site = v[1];
    return 0;
#else
    (void)violations; (void)first_site;
    return -1;
#endif
}

int32_t crnn_sizeof(int32_t which) {
    switch (which) {
    case 0: return (int32_t)sizeof(crnn_config);
    case 1: return (int32_t)sizeof(crnn_stats);
    case 2: return (int32_t)sizeof(crnn_opt_config);
    case 3: return (int32_t)sizeof(crnn_cathode_config);
    default: return -1;
    }
}

const char *crnn_last_error(const crnn_ctx *ctx) {
    const Ctx *c = reinterpret_cast<const Ctx *>(ctx);
    return c ? c->err.c_str() : g_last_error.c_str();
}

int32_t crnn_config_preset(crnn_config *cfg, int32_t preset) {
    if (!cfg) return fail(nullptr, "crnn_config_preset: null cfg");
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->abi_version = CRNN_ABI_VERSION;
    for (int i = 0; i < CRNN_MAX_N; ++i) { cfg->atol[i] = 1e-6; cfg->rtol[i] = 1e-3; cfg->rate_scale[i] = 1.0; }
    cfg->loss_kind = CRNN_LOSS_MAE;
    cfg->maxiters = 100000;  // DiffEq default when the script passes none (case2)
    cfg->ub = INFINITY;
    // OrdinaryDiffEq PIController defaults for Rosenbrock23 (order 2, implicit)
    cfg->gamma = 0.9; cfg->qmin = 0.2; cfg->qmax = 10.0;
    cfg->beta1 = 7.0 / 20.0; cfg->beta2 = 2.0 / 10.0;
    cfg->qsteady_min = 1.0; cfg->qsteady_max = 1.2; cfg->qoldinit = 1e-4; cfg->dtmin = 0.0;
    switch (preset) {
    case CRNN_PRESET_CASE1:  // case1/case1.jl:19-35
        cfg->ns = 5; cfg->nr = 4; cfg->has_temp = 0; cfg->param_map = CRNN_PMAP_CASE1;
        cfg->n_save = 100; cfg->clamp_pred = 1; cfg->maxiters = 10000;
        cfg->lb = (double)1e-5f; cfg->ub = 10.0;   // `lb = 1.f-5`: a Float32 literal, promoted by clamp (case1.jl:34,81)
        for (int i = 0; i < CRNN_MAX_N; ++i) { cfg->atol[i] = 1e-5; cfg->rtol[i] = 1e-2; }
        crnn_config_set_solver(cfg, CRNN_SOLVER_TSIT5);  // alg = Tsit5(), case1/case1.jl:28
        break;
    case CRNN_PRESET_CASE2:  // case2/case2.jl:18-35,113
        cfg->ns = 6; cfg->nr = 3; cfg->has_temp = 1; cfg->param_map = CRNN_PMAP_CASE2;
        cfg->n_save = 50; cfg->clamp_pred = 1;
        cfg->lb = (double)1e-6f; cfg->ub = 10.0;   // `lb = 1.f-6`: a Float32 literal, promoted by clamp (case2.jl:34,115)
        cfg->inv_R = (double)(-1.0f / 1.98720425864083e-3f);   // `- 1 / 1.98720425864083f-3`: a Float32 literal, the quotient is Float32 (case2.jl:113)
        break;
    case CRNN_PRESET_ROBER:  // robertson/rober_crnn.jl:20-37
        cfg->ns = 3; cfg->nr = 6; cfg->has_temp = 0; cfg->param_map = CRNN_PMAP_ROBER;
        cfg->n_save = 40; cfg->clamp_pred = 0; cfg->maxiters = 10000;
        cfg->lb = 1e-8; cfg->ub = INFINITY;
        cfg->atol[0] = 1e-6; cfg->atol[1] = 1e-8; cfg->atol[2] = 1e-6;
        break;
    case CRNN_PRESET_HYCHEM: {  // HyChem/crnn_pyrolysis_mass.jl:15-30,57-58,106-108
        cfg->ns = 9; cfg->nr = 10; cfg->has_temp = 0; cfg->param_map = CRNN_PMAP_HYCHEM; cfg->rhs_kind = CRNN_RHS_HYCHEM;
        cfg->n_save = 40; cfg->clamp_pred = 0; cfg->maxiters = 10000;
        cfg->lb = 1e-8; cfg->ub = 10.0;
        for (int i = 0; i < CRNN_MAX_N; ++i) { cfg->atol[i] = 1e-8; cfg->rtol[i] = 1e-3; }
        cfg->inv_R = (double)(-1.0f / 1.98720425864083e-3f);   // R is a Float32 literal; -1/R is formed in Float32 (:106,128)
        cfg->gas_const = 8.31446261815324e3;
        const double mw[9] = {136.238, 2.016, 16.043, 26.038, 28.054, 28.014, 56.108, 1.008, 15.035};
        for (int i = 0; i < 9; ++i) cfg->mw[i] = mw[i];
        break;
    }
    default:
        return fail(nullptr, "crnn_config_preset: unknown preset");
    }
    return 0;
}

int32_t crnn_config_set_solver(crnn_config *cfg, int32_t solver) {
    if (!cfg) return fail(nullptr, "crnn_config_set_solver: null cfg");
    if (solver == CRNN_SOLVER_ROSENBROCK23) { cfg->beta1 = 7.0 / 20.0; cfg->beta2 = 2.0 / 10.0; cfg->qsteady_max = 1.2; }
    else if (solver == CRNN_SOLVER_TSIT5 || solver == CRNN_SOLVER_AUTOTSIT5) {
        // the composite starts on Tsit5; its steady band is that of a non-implicit algorithm type (qsteady_max_default = 1)
        cfg->beta1 = 7.0 / 50.0; cfg->beta2 = 2.0 / 25.0; cfg->qsteady_max = 1.0;
    }
    else return fail(nullptr, "crnn_config_set_solver: unknown solver");
    cfg->solver = solver;
    return 0;
}

int32_t crnn_opt_preset(crnn_opt_config *o, int32_t preset) {
    if (!o) return fail(nullptr, "crnn_opt_preset: null");
    std::memset(o, 0, sizeof(*o));
    o->beta1 = 0.9; o->beta2 = 0.999;
    // The reference writes the decay as a Float32 literal (`1.f-6`, `1.f-8`) and Flux's WeightDecay keeps it as one: the
    // factor that multiplies the Float64 p is the Float32 value (9.999999974752427e-07, not 1e-6).  The checkpoints hold it
    // that way (case2/checkpoint/mymodel.bson: WeightDecay(Float32); tests/golden/fixtures_ckpt_opt.json).
    const double wd6 = (double)1e-6f, wd8 = (double)1e-8f;
    switch (preset) {
    case CRNN_PRESET_CASE1: o->eta = 0.001; o->wd = wd8; break;                                   // case1.jl:18
    case CRNN_PRESET_CASE2:                                                                        // case2.jl:31-32
        o->eta = 0.005; o->wd = wd6; o->use_expdecay = 1; o->ed_eta0 = 5e-3; o->ed_decay = 0.5;
        o->decay_step = 500 * 20; o->ed_clip = 1e-4; break;
    case CRNN_PRESET_ROBER: o->eta = 0.005; o->wd = wd6; o->grad_clip_norm = 10.0; break;          // rober_crnn.jl:19,29
    case CRNN_PRESET_HYCHEM: o->eta = 0.005; o->wd = wd6; o->grad_clip_norm = 10.0; break;         // crnn_pyrolysis_mass.jl:20,24
    default: return fail(nullptr, "crnn_opt_preset: unknown preset");
    }
    return 0;
}

static int extra_rows_of_pmap(int32_t pmap) { return pmap == CRNN_PMAP_CASE2 ? 1 : (pmap == CRNN_PMAP_HYCHEM ? 2 : 0); }
int32_t crnn_n_params(int32_t pmap, int32_t ns, int32_t nr) { return crnn::n_params_of(pmap, ns, nr, extra_rows_of_pmap(pmap)); }
int32_t crnn_config_n_theta(const crnn_config *cfg) {
    if (!cfg) return fail(nullptr, "crnn_config_n_theta: null cfg");
    return crnn::n_theta_of(cfg->ns, cfg->nr, cfg->rhs_kind == CRNN_RHS_HYCHEM ? 2 : cfg->has_temp);
}
int32_t crnn_n_theta(int32_t ns, int32_t nr, int32_t has_temp) { return crnn::n_theta_of(ns, nr, has_temp); }

int32_t crnn_p2vec(int32_t pmap, int32_t ns, int32_t nr, const double *p, double *theta, double *dtheta) {
    if (!p || !theta) return fail(nullptr, "crnn_p2vec: null pointer");
    if (pmap == CRNN_PMAP_IDENTITY) return fail(nullptr, "crnn_p2vec: identity map needs no p2vec");
    int has_temp = extra_rows_of_pmap(pmap);
    int nth = crnn::n_theta_of(ns, nr, has_temp), P = crnn::n_params_of(pmap, ns, nr, has_temp);
    if (P < 0) return fail(nullptr, "crnn_p2vec: unknown param_map");
    if (dtheta) std::memset(dtheta, 0, sizeof(double) * (size_t)nth * P);
    if (crnn::p2vec_eval(pmap, ns, nr, has_temp, p, theta, dtheta) != 0) return fail(nullptr, "crnn_p2vec: bad arguments");
    return 0;
}

int32_t crnn_ctx_create(const crnn_config *cfg, crnn_ctx **out) {
    if (!cfg || !out) return fail(nullptr, "crnn_ctx_create: null pointer");
    *out = nullptr;
    if (cfg->abi_version != CRNN_ABI_VERSION) return fail(nullptr, "crnn_ctx_create: abi_version mismatch");
    if (cfg->ns < 1 || cfg->nr < 1 || cfg->ns + cfg->has_temp > CRNN_MAX_N || cfg->nr > CRNN_MAX_NR)
        return fail(nullptr, "crnn_ctx_create: ns/nr out of range");
    if (cfg->errnorm_sens < 0 || cfg->errnorm_sens > 2) return fail(nullptr, "crnn_ctx_create: errnorm_sens must be 0, 1 or 2");
    // (HyChem: also through the reference's composite, AutoTsit5(Rosenbrock23) -- hychem_sens2_kernel<..., COMPOSITE>)
    if (cfg->errnorm_sens != 0 && ((cfg->solver == CRNN_SOLVER_AUTOTSIT5 && cfg->rhs_kind != CRNN_RHS_HYCHEM && !composite_is_tsit5(*cfg)) || cfg->grad_mode == CRNN_GRAD_ADJOINT ||
                                   (cfg->rhs_kind == CRNN_RHS_HYCHEM && cfg->solver == CRNN_SOLVER_TSIT5)))
        return fail(nullptr, "crnn_ctx_create: errnorm_sens = 1 / 2 exists for Rosenbrock23 and Tsit5 (HyChem: Rosenbrock23 and AutoTsit5; AutoTsit5 on a shape with a temperature state, which never leaves Tsit5) with forward tangents (grad_mode AUTO or FORWARD)");
    if (cfg->solver != CRNN_SOLVER_ROSENBROCK23 && cfg->solver != CRNN_SOLVER_TSIT5 && cfg->solver != CRNN_SOLVER_AUTOTSIT5)
        return fail(nullptr, "crnn_ctx_create: unknown solver");
    if (cfg->rhs_kind != CRNN_RHS_CRNN && cfg->rhs_kind != CRNN_RHS_HYCHEM) return fail(nullptr, "crnn_ctx_create: unknown rhs_kind");
    if (cfg->grad_mode < CRNN_GRAD_AUTO || cfg->grad_mode > CRNN_GRAD_ADJOINT || cfg->tape_steps < 0)
        return fail(nullptr, "crnn_ctx_create: bad grad_mode / tape_steps");
    if (cfg->n_save < 1 || cfg->n_save > crnn::kMaxSave) return fail(nullptr, "crnn_ctx_create: n_save must be in [1, 256]");
    Ctx *c = new Ctx();
    c->cfg = *cfg;
    c->n = cfg->ns + cfg->has_temp;
    c->hychem = cfg->rhs_kind == CRNN_RHS_HYCHEM;
    c->nfx = c->hychem ? 2 : cfg->has_temp;
    c->n_theta = crnn::n_theta_of(cfg->ns, cfg->nr, c->nfx);
    c->n_params = crnn::n_params_of(cfg->param_map, cfg->ns, cfg->nr, c->nfx);
    if (c->hychem && (cfg->has_temp != 0 || (cfg->solver != CRNN_SOLVER_ROSENBROCK23 && cfg->solver != CRNN_SOLVER_AUTOTSIT5) || !(cfg->gas_const > 0))) {
        delete c;
        return fail(nullptr, "crnn_ctx_create: HyChem needs has_temp = 0, Rosenbrock23 or AutoTsit5(Rosenbrock23), and gas_const > 0");
    }
    if (c->n_params < 0) { delete c; return fail(nullptr, "crnn_ctx_create: unknown param_map"); }
    c->use_scale = false;
    for (int i = 0; i < cfg->ns; ++i) if (cfg->rate_scale[i] != 1.0) c->use_scale = true;
    if (const char *e = getenv("CRNN_SENS_ONE_LAUNCH")) { if (*e) c->sens_one_launch = atoi(e) != 0; }   // measurement override
    if (const char *e = getenv("CRNN_HY_SENS_KERNEL")) { if (*e) c->hy_sens_kernel = atoi(e); }          // 2: hychem_sens2_kernel where the directions are sparse
    // robertson-shaped problems always take the scaled kernel (one instantiation per shape)
    auto has_kernel = [&]() { return c->cfg.solver == CRNN_SOLVER_AUTOTSIT5 ? find_adjoint(c) != nullptr : find_primal(c) != nullptr; };
    if (!c->hychem && !has_kernel()) { c->use_scale = !c->use_scale; if (!has_kernel()) c->use_scale = !c->use_scale; }
    if (!c->hychem && !has_kernel()) {
        delete c;
        return fail(nullptr, "crnn_ctx_create: no gfx950 kernel instantiated for this (solver, ns, nr, has_temp)");
    }
    if (cfg->errnorm_sens != 0 && !c->hychem && !find_sens(c)) {
        delete c;
        return fail(nullptr, "crnn_ctx_create: errnorm_sens = 1 has no kernel for this (ns, nr, has_temp)");
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1) {
        delete c;
        return fail(nullptr, std::string("crnn_ctx_create: no HIP device (") + hipGetErrorString(e) + ")");
    }
    if (cfg->device < 0 || cfg->device >= ndev) {
        delete c;
        return fail(nullptr, "crnn_ctx_create: device ordinal out of range");
    }
    auto bail = [&](const std::string &m) { std::string mm = m; crnn_ctx_destroy((crnn_ctx *)c); return fail(nullptr, mm); };
    if (hipSetDevice(cfg->device) != hipSuccess) return bail("crnn_ctx_create: hipSetDevice failed");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return bail("crnn_ctx_create: hipGetDeviceProperties failed");
    c->num_cu = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return bail("hipStreamCreate failed");
    c->own_stream = true;
    for (int i = 0; i < Ctx::kRing; ++i)
        if (hipEventCreate(&c->ring0[i]) != hipSuccess || hipEventCreate(&c->ring1[i]) != hipSuccess)
            return bail("hipEventCreate failed");
    c->max_dir = std::max(c->n_params, c->n_theta);
    if (hipMalloc((void **)&c->d_theta, sizeof(double) * c->n_theta) != hipSuccess ||
        hipMalloc((void **)&c->d_dtheta, sizeof(double) * (size_t)c->n_theta * c->max_dir) != hipSuccess ||
        hipMalloc((void **)&c->d_tsave, sizeof(double) * cfg->n_save) != hipSuccess ||
        hipMalloc((void **)&c->d_kc, sizeof(crnn::KConst)) != hipSuccess ||
        hipMalloc((void **)&c->d_queue, sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc((void **)&c->d_overflow, sizeof(unsigned int)) != hipSuccess ||
        hipMalloc((void **)&c->d_poison, 2 * sizeof(double)) != hipSuccess ||
        hipMemset(c->d_poison, 0, 2 * sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&c->d_red_theta, sizeof(double) * (c->n_theta + crnn::kExtra)) != hipSuccess ||
        hipMalloc((void **)&c->d_p, sizeof(double) * c->n_params) != hipSuccess ||
        hipMalloc((void **)&c->d_p_eval, sizeof(double) * c->n_params) != hipSuccess ||
        hipMalloc((void **)&c->d_opt, sizeof(double) * (2 * c->n_params + 4)) != hipSuccess)
        return bail("crnn_ctx_create: hipMalloc failed");
    *out = reinterpret_cast<crnn_ctx *>(c);
    return 0;
}

void crnn_ctx_destroy(crnn_ctx *ctx) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    if (c->comm) { ncclCommDestroy(c->comm); c->comm = nullptr; }
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->own_u0 && c->d_u0) (void)hipFree(c->d_u0);
    if (c->own_data && c->d_data) (void)hipFree(c->d_data);
    void *ptrs[] = {c->d_perm, c->d_red_asm, c->d_poison, c->d_tabs, c->d_gacc, c->d_tape, c->d_overflow, c->d_red_theta, c->d_queue, c->d_nacc, c->d_nrej, c->d_gtraj, c->d_kc, c->d_tsave, c->d_pred, c->d_loss, c->d_ret, c->d_nsaved, c->d_theta, c->d_dtheta,
                    c->d_partials, c->d_red, c->d_p, c->d_p_eval, c->d_opt, c->d_comm_buf};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (c->h_spread) (void)hipHostFree(c->h_spread);     // (d_spread is its device alias)
    for (int i = 0; i < Ctx::kSpreadRing; ++i) if (c->ev_spread[i]) (void)hipEventDestroy(c->ev_spread[i]);
    for (int i = 0; i < Ctx::kRing; ++i) {
        if (c->ring0[i]) (void)hipEventDestroy(c->ring0[i]);
        if (c->ring1[i]) (void)hipEventDestroy(c->ring1[i]);
    }
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int32_t crnn_ctx_set_stream(crnn_ctx *ctx, void *hip_stream) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c) return fail(nullptr, "null ctx");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    if (c->stream) HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->own_stream && c->stream) HIP_TRY(c, hipStreamDestroy(c->stream));
    c->stream = reinterpret_cast<hipStream_t>(hip_stream);
    c->own_stream = false;
    return 0;
}

// data (IC-fastest, on device) -> c->d_data (trajectory-major, owned)
static int32_t transpose_into(Ctx *c, const double *d_src, int64_t B) {
    const int rows = c->cfg.n_save * c->n_obs;
    dim3 grid((unsigned)((B + 31) / 32), (unsigned)((rows + 31) / 32));
    hipLaunchKernelGGL(crnn::transpose_data_kernel, grid, dim3(256), 0, c->stream, d_src, c->d_data, B, rows);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}

static int32_t set_data_common(Ctx *c, const double *tsteps, const double *yscale, const int32_t *i_obs, int32_t n_obs,
                               int64_t B) {
    if (!tsteps) return fail(c, "crnn_ctx_set_data: null tsteps");
    if (B <= 0) return fail(c, "crnn_ctx_set_data: B must be positive");
    const int ns = c->cfg.ns;
    if (!i_obs) n_obs = ns;
    if (n_obs < 1 || n_obs > ns) return fail(c, "crnn_ctx_set_data: n_obs out of range");
    for (int i = 0; i < CRNN_MAX_N; ++i) { c->drow[i] = -1; c->inv_yscale[i] = 0.0; }
    for (int k = 0; k < n_obs; ++k) {
        int i = i_obs ? i_obs[k] : k;
        if (i < 0 || i >= ns || c->drow[i] >= 0) return fail(c, "crnn_ctx_set_data: bad i_obs");
        c->drow[i] = k;
        double ys = yscale ? yscale[k] : 1.0;
        if (!(ys > 0)) return fail(c, "crnn_ctx_set_data: yscale must be positive");
        c->inv_yscale[i] = 1.0 / ys;
    }
    c->n_obs = n_obs;
    c->kc_dirty = true;
    c->steps_first = 0; c->steps_count = 0;     // a new ensemble: no step counts known yet
    c->hy_block_cap = 0; c->hy_cap_uses = 0; c->hy_cap_first = c->hy_cap_count = -1;
    c->perm_ready = false;
    for (int j = 1; j < c->cfg.n_save; ++j)
        if (!(tsteps[j] > tsteps[j - 1])) return fail(c, "crnn_ctx_set_data: tsteps must be strictly increasing");
    if (tsteps[0] < c->cfg.t0) return fail(c, "crnn_ctx_set_data: tsteps[0] < t0");
    c->tsave.assign(tsteps, tsteps + c->cfg.n_save);
    HIP_TRY(c, hipMemcpy(c->d_tsave, tsteps, sizeof(double) * c->cfg.n_save, hipMemcpyHostToDevice));
    if (B != c->B || !c->d_loss) {
        if (c->d_loss) HIP_TRY(c, hipFree(c->d_loss));
        if (c->d_ret) HIP_TRY(c, hipFree(c->d_ret));
        if (c->d_nsaved) HIP_TRY(c, hipFree(c->d_nsaved));
        if (c->d_nacc) HIP_TRY(c, hipFree(c->d_nacc));
        if (c->d_nrej) HIP_TRY(c, hipFree(c->d_nrej));
        c->d_loss = nullptr; c->d_ret = nullptr; c->d_nsaved = nullptr; c->d_nacc = nullptr; c->d_nrej = nullptr;
        HIP_TRY(c, hipMalloc((void **)&c->d_loss, sizeof(double) * B));
        HIP_TRY(c, hipMalloc((void **)&c->d_ret, sizeof(int32_t) * B));
        HIP_TRY(c, hipMalloc((void **)&c->d_nsaved, sizeof(int32_t) * B));
        HIP_TRY(c, hipMalloc((void **)&c->d_nacc, sizeof(int32_t) * B));
        HIP_TRY(c, hipMalloc((void **)&c->d_nrej, sizeof(int32_t) * B));
    }
    c->B = B;
    return 0;
}

int32_t crnn_ctx_set_data(crnn_ctx *ctx, const double *u0, const double *data, const double *tsteps, const double *yscale,
                          const int32_t *i_obs, int32_t n_obs, int64_t B) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c) return fail(nullptr, "null ctx");
    if (!u0 || !data) return fail(c, "crnn_ctx_set_data: null u0/data");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    if (check_pending(c, nullptr)) return -1;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->own_u0 && c->d_u0) HIP_TRY(c, hipFree(c->d_u0));
    if (c->own_data && c->d_data) HIP_TRY(c, hipFree(c->d_data));
    c->d_u0 = nullptr; c->d_data = nullptr; c->own_u0 = c->own_data = false;
    if (set_data_common(c, tsteps, yscale, i_obs, n_obs, B)) return -1;
    size_t nu = (size_t)c->n * B, nd = (size_t)c->cfg.n_save * c->n_obs * B;
    HIP_TRY(c, hipMalloc((void **)&c->d_u0, sizeof(double) * nu));
    c->own_u0 = true;
    HIP_TRY(c, hipMalloc((void **)&c->d_data, sizeof(double) * nd));
    c->own_data = true;
    double *tmp = nullptr;
    HIP_TRY(c, hipMalloc((void **)&tmp, sizeof(double) * nd));
    HIP_TRY(c, hipMemcpy(c->d_u0, u0, sizeof(double) * nu, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(tmp, data, sizeof(double) * nd, hipMemcpyHostToDevice));
    int32_t rc = transpose_into(c, tmp, B);
    (void)hipFree(tmp);
    return rc;
}

int32_t crnn_ctx_set_data_device(crnn_ctx *ctx, const void *d_u0, const void *d_data, const double *tsteps,
                                 const double *yscale, const int32_t *i_obs, int32_t n_obs, int64_t B) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c) return fail(nullptr, "null ctx");
    if (!d_u0 || !d_data) return fail(c, "crnn_ctx_set_data_device: null u0/data");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    if (check_pending(c, nullptr)) return -1;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->own_u0 && c->d_u0) HIP_TRY(c, hipFree(c->d_u0));
    if (c->own_data && c->d_data) HIP_TRY(c, hipFree(c->d_data));
    c->own_u0 = c->own_data = false;
    c->d_u0 = (double *)d_u0;   // u0 is used in place
    c->d_data = nullptr;
    if (set_data_common(c, tsteps, yscale, i_obs, n_obs, B)) return -1;
    size_t nd = (size_t)c->cfg.n_save * c->n_obs * B;
    HIP_TRY(c, hipMalloc((void **)&c->d_data, sizeof(double) * nd));   // trajectory-major working copy
    c->own_data = true;
    return transpose_into(c, (const double *)d_data, B);
}

int32_t crnn_ctx_set_tables(crnn_ctx *ctx, const double *T, const double *P) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c) return fail(nullptr, "null ctx");
    if (!c->hychem) return fail(c, "crnn_ctx_set_tables: only the HyChem right-hand side takes T/P tables");
    if (!T || !P) return fail(c, "crnn_ctx_set_tables: null pointer");
    if (c->B <= 0) return fail(c, "crnn_ctx_set_tables: call crnn_ctx_set_data first");
    const int D = c->cfg.n_save;
    const int64_t B = c->B;
    std::vector<double> h((size_t)B * 2 * D);
    for (int64_t b = 0; b < B; ++b)
        for (int j = 0; j < D; ++j) {
            const double t_ = T[(size_t)j * B + b], p_ = P[(size_t)j * B + b];
            if (!(t_ > 0) || !(p_ > 0)) return fail(c, "crnn_ctx_set_tables: temperatures and pressures must be positive");
            h[((size_t)b * 2) * D + j] = t_;
            h[((size_t)b * 2 + 1) * D + j] = p_;
        }
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->d_tabs) HIP_TRY(c, hipFree(c->d_tabs));
    c->d_tabs = nullptr;
    HIP_TRY(c, hipMalloc((void **)&c->d_tabs, sizeof(double) * h.size()));
    HIP_TRY(c, hipMemcpy(c->d_tabs, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
    c->tabs_B = B;
    return 0;
}

int32_t crnn_solve(crnn_ctx *ctx, const double *theta, const double *dtheta, int32_t n_dir, int64_t first, int64_t count,
                   int32_t n_save_active, double *pred, double *loss, double *grad, int32_t *retcode, int32_t *n_saved,
                   crnn_stats *stats) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c) return fail(nullptr, "null ctx");
    if (!theta) return fail(c, "crnn_solve: null theta");
    if (n_dir < 0 || (n_dir > 0 && !dtheta)) return fail(c, "crnn_solve: n_dir > 0 needs dtheta");
    if (n_dir > c->max_dir) return fail(c, "crnn_solve: n_dir exceeds max(n_params, n_theta)");
    if (grad && n_dir == 0) return fail(c, "crnn_solve: grad requested with n_dir = 0");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    if (check_pending(c, nullptr)) return -1;
    c->theta_current = false;
    HIP_TRY(c, hipMemcpyAsync(c->d_theta, theta, sizeof(double) * c->n_theta, hipMemcpyHostToDevice, c->stream));
    if (n_dir > 0)
        HIP_TRY(c, hipMemcpyAsync(c->d_dtheta, dtheta, sizeof(double) * (size_t)c->n_theta * n_dir, hipMemcpyHostToDevice,
                                  c->stream));
    c->hy_dirs_sparse = false;
    if (c->hychem && c->cfg.errnorm_sens && n_dir > 0 && c->cfg.ns == 9 && c->cfg.nr == 10) {   // the caller's rows: do they fit the sparse description?
        bool fits = true;
        for (int k = 0; k < n_dir && fits; ++k) fits = crnn::hy_dir_fits<9, 10>(dtheta + (size_t)k * c->n_theta);
        c->hy_dirs_sparse = fits;
    }
    if (launch_solve(c, c->d_theta, c->d_dtheta, n_dir, first, count, n_save_active, pred != nullptr, true)) return -1;
    std::vector<double> red(c->last_npart);
    HIP_TRY(c, hipMemcpyAsync(red.data(), c->d_red, sizeof(double) * c->last_npart, hipMemcpyDeviceToHost, c->stream));
    if (loss) HIP_TRY(c, hipMemcpyAsync(loss + first, c->d_loss + first, sizeof(double) * count, hipMemcpyDeviceToHost, c->stream));
    if (retcode)
        HIP_TRY(c, hipMemcpyAsync(retcode + first, c->d_ret + first, sizeof(int32_t) * count, hipMemcpyDeviceToHost, c->stream));
    if (n_saved)
        HIP_TRY(c, hipMemcpyAsync(n_saved + first, c->d_nsaved + first, sizeof(int32_t) * count, hipMemcpyDeviceToHost, c->stream));
    if (pred) {
        // rows (j, i) are B-contiguous; copy the [first, first+count) slice of each row
        HIP_TRY(c, hipMemcpy2DAsync(pred + first, sizeof(double) * c->B, c->d_pred + first, sizeof(double) * c->B,
                                    sizeof(double) * count, (size_t)n_save_active * c->n, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (grad) for (int k = 0; k < n_dir; ++k) grad[k] = red[k];
    return fill_stats(c, red.data(), c->last_npart, stats);
}

int32_t crnn_loss_grad(crnn_ctx *ctx, const double *p, int64_t first, int64_t count, int32_t n_save_active,
                       double *loss_mean, double *grad_p, crnn_stats *stats) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c) return fail(nullptr, "null ctx");
    if (!p) return fail(c, "crnn_loss_grad: null p");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    if (check_pending(c, nullptr)) return -1;
    const int P = grad_p ? c->n_params : 0;
    c->theta_current = false;
    HIP_TRY(c, hipMemcpyAsync(c->d_p_eval, p, sizeof(double) * c->n_params, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(p2vec_kernel, dim3(1), dim3(256), 0, c->stream, c->cfg.param_map, c->cfg.ns, c->cfg.nr,
                       c->nfx, c->d_p_eval, c->d_theta, c->d_dtheta, c->n_theta, c->n_params);
    HIP_TRY(c, hipGetLastError());
    c->hy_dirs_sparse = pmap_dirs_sparse(c);
    if (launch_solve(c, c->d_theta, c->d_dtheta, P, first, count, n_save_active, false, false)) return -1;
    std::vector<double> red(c->last_npart);
    HIP_TRY(c, hipMemcpyAsync(red.data(), c->d_red, sizeof(double) * c->last_npart, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const int np_ = c->last_npart;
    double ntraj = red[np_ - 1];
    if (loss_mean) *loss_mean = ntraj > 0 ? red[np_ - 5] / ntraj : 0.0;
    if (grad_p) for (int k = 0; k < P; ++k) grad_p[k] = ntraj > 0 ? red[k] / ntraj : 0.0;
    return fill_stats(c, red.data(), np_, stats);
}

int32_t crnn_opt_state_len(int32_t n_params) { return 2 * n_params + 4; }

static crnn::OptCfg to_optcfg(const crnn_opt_config *o) {
    crnn::OptCfg r{};
    r.use_expdecay = o->use_expdecay; r.decay_step = o->decay_step;
    r.ed_eta0 = o->ed_eta0; r.ed_decay = o->ed_decay; r.ed_clip = o->ed_clip;
    r.eta = o->eta; r.beta1 = o->beta1; r.beta2 = o->beta2; r.wd = o->wd; r.grad_clip_norm = o->grad_clip_norm;
    return r;
}

int32_t crnn_opt_init(const crnn_opt_config *o, int32_t n_params, double *state) {
    if (!o || !state || n_params < 1) return fail(nullptr, "crnn_opt_init: bad arguments");
    if (o->use_expdecay && o->decay_step < 1) return fail(nullptr, "crnn_opt_init: decay_step must be >= 1");
    crnn::opt_init(to_optcfg(o), n_params, state);
    return 0;
}

int32_t crnn_opt_update(const crnn_opt_config *o, int32_t n_params, double *p, const double *grad, double *state) {
    if (!o || !p || !grad || !state || n_params < 1) return fail(nullptr, "crnn_opt_update: bad arguments");
    crnn::opt_update(to_optcfg(o), n_params, p, grad, 1.0, state);
    return 0;
}

int32_t crnn_train_init(crnn_ctx *ctx, const crnn_opt_config *o, const double *p0) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c) return fail(nullptr, "null ctx");
    if (!o || !p0) return fail(c, "crnn_train_init: null pointer");
    if (o->use_expdecay && o->decay_step < 1) return fail(c, "crnn_train_init: decay_step must be >= 1");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->pending.clear();             // a new training run: whatever was in flight belongs to the old one
    HIP_TRY(c, hipMemsetAsync(c->d_poison, 0, 2 * sizeof(double), c->stream));
    c->opt = to_optcfg(o);
    std::vector<double> st(2 * c->n_params + 4);
    crnn::opt_init(c->opt, c->n_params, st.data());
    c->theta_current = false;
    HIP_TRY(c, hipMemcpyAsync(c->d_p, p0, sizeof(double) * c->n_params, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_opt, st.data(), sizeof(double) * st.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->train_ready = true;
    return 0;
}

static int32_t train_begin_impl(Ctx *c, int64_t first, int64_t count, int32_t n_save_active, bool defer) {
    if (!c->train_ready) return fail(c, "crnn_train_step: call crnn_train_init first");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    if (!c->theta_current) {
        hipLaunchKernelGGL(p2vec_kernel, dim3(1), dim3(256), 0, c->stream, c->cfg.param_map, c->cfg.ns, c->cfg.nr,
                           c->nfx, c->d_p, c->d_theta, c->d_dtheta, c->n_theta, c->n_params);
        HIP_TRY(c, hipGetLastError());
    }
    c->theta_current = false;   // consumed: anything else that touches d_theta / d_p must not find a stale flag
    c->defer_next = defer;
    c->hy_dirs_sparse = pmap_dirs_sparse(c);
    const int32_t rc = launch_solve(c, c->d_theta, c->d_dtheta, c->n_params, first, count, n_save_active, false, false);
    c->defer_next = false;
    return rc;
}

static int32_t train_end_impl(Ctx *c, double *loss_mean) {
    if (!c->train_ready || c->last_npart == 0) return fail(c, "crnn_train_step_end: no step in flight");
    if (!c->opt_fused)
    hipLaunchKernelGGL(opt_kernel, dim3(1), dim3(256), 0, c->stream, c->opt, c->n_params, c->last_npart, c->d_p, c->d_red,
                       c->d_opt, c->cfg.param_map, c->cfg.ns, c->cfg.nr, c->nfx, c->d_theta, c->d_dtheta, c->n_theta,
                       c->d_queue, c->d_overflow, c->d_poison);
    HIP_TRY(c, hipGetLastError());
    c->opt_fused = false;
    c->theta_current = true;    // (a skipped step leaves p and theta as they were: still consistent)
    c->flags_zeroed = true;
    if (loss_mean) {
        double tail[5];
        HIP_TRY(c, hipMemcpyAsync(tail, c->d_red + c->last_npart - 5, sizeof(tail), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        *loss_mean = tail[4] > 0 ? tail[0] / tail[4] : 0.0;
    }
    return 0;
}

// The device-resident training loop (crnn_train_step) enqueues adjoint steps without looking at their tape-overflow
// flag: a poisoned step is skipped on the device by every rank alike, and so is everything after it (sticky flag).
// Here the host looks: if steps were skipped they are repeated, in order, with forward tangents.  Called before anything
// that exposes training state (loss, parameters, statistics, synchronize) and every kMaxPending steps.
}  // extern "C" (interrupted: the following helpers have C++ linkage)
namespace {
int32_t check_pending(Ctx *c, double *loss_mean) {
    if (c->pending.empty()) return 0;
    double poison[2] = {0.0, 0.0};
    HIP_TRY(c, hipMemcpyAsync(poison, c->d_poison, sizeof(poison), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    std::vector<Ctx::StepArgs> steps;
    steps.swap(c->pending);
    if (poison[0] == 0.0) return 0;
    const size_t nskip = std::min<size_t>((size_t)llround(poison[1]), steps.size());
    HIP_TRY(c, hipMemsetAsync(c->d_poison, 0, 2 * sizeof(double), c->stream));
    c->force_forward = true;
    int32_t rc = 0;
    for (size_t i = steps.size() - nskip; i < steps.size() && rc == 0; ++i) {
        ++c->n_fallback;
        rc = train_begin_impl(c, steps[i].first, steps[i].count, steps[i].n_save, false);
        if (rc == 0) rc = allreduce_red(c);
        if (rc == 0) rc = train_end_impl(c, (i + 1 == steps.size()) ? loss_mean : nullptr);
    }
    c->force_forward = false;
    if (rc == 0) {   // the replay runs on forward tangents, which have no tape: the flag cannot legitimately be set again
        HIP_TRY(c, hipMemcpyAsync(poison, c->d_poison, sizeof(poison), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (poison[0] != 0.0) return fail(c, "crnn_train_step: a replayed step was skipped again (ranks disagree on the overflow count?)");
    }
    return rc;
}
}  // namespace
extern "C" {
constexpr size_t kMaxPending = 64;

int32_t crnn_train_step_begin(crnn_ctx *ctx, int64_t first, int64_t count, int32_t n_save_active) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c) return fail(nullptr, "null ctx");
    if (check_pending(c, nullptr)) return -1;
    c->opt_fused = false;
    return train_begin_impl(c, first, count, n_save_active, false);   // split API: outcome checked before returning
}

int32_t crnn_train_step_end(crnn_ctx *ctx, double *loss_mean) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c) return fail(nullptr, "null ctx");
    return train_end_impl(c, loss_mean);
}

int32_t crnn_train_step(crnn_ctx *ctx, int64_t first, int64_t count, int32_t n_save_active, double *loss_mean) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c) return fail(nullptr, "null ctx");
    if (c->pending.size() >= kMaxPending && check_pending(c, nullptr)) return -1;
    // nothing between the reduction and the update (no communicator of more than one rank, no callback), the parameters the launch differentiates
    // with respect to are the optimiser's own: one launch for both (launch_adjoint)
    c->fuse_opt = !(c->comm && c->world > 1) && !c->host_ar;
    c->opt_fused = false;
    const int32_t rc_begin = train_begin_impl(c, first, count, n_save_active, true);
    c->fuse_opt = false;
    if (rc_begin) { c->opt_fused = false; return -1; }   // (a failed launch must not leave "the update already ran" behind)
    // The replay bookkeeping must be the same on every rank.  With a collective attached the skip decision is taken on the
    // SUMMED overflow count, so a rank that itself ran forward tangents (grad_mode AUTO picks them from the LOCAL count:
    // shards of 2049 and 2048 trajectories straddle case2's threshold) is skipped too when another rank's tape overflowed:
    // it has to look at d_poison and replay like everybody else, or its next all-reduce pairs with the others' replay.
    const bool collective = (c->comm && c->world > 1) || c->host_ar;
    if (c->last_deferred || collective) c->pending.push_back({first, count, n_save_active});
    if (allreduce_red(c)) { c->opt_fused = false; return -1; }
    if (train_end_impl(c, loss_mean)) return -1;
    if (loss_mean) return check_pending(c, loss_mean);   // the caller wants this step's loss: look now
    return 0;
}

int32_t crnn_grad_buffer(crnn_ctx *ctx, void **d_ptr, int32_t *n_doubles) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c) return fail(nullptr, "null ctx");
    if (!c->d_red || c->last_npart == 0) return fail(c, "crnn_grad_buffer: no solve has run yet");
    if (d_ptr) *d_ptr = c->d_red;
    if (n_doubles) *n_doubles = c->last_npart;
    return 0;
}

int32_t crnn_get_params(crnn_ctx *ctx, double *p) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c || !p) return fail(c, "crnn_get_params: null");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    if (check_pending(c, nullptr)) return -1;
    HIP_TRY(c, hipMemcpyAsync(p, c->d_p, sizeof(double) * c->n_params, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}

int32_t crnn_set_params(crnn_ctx *ctx, const double *p) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c || !p) return fail(c, "crnn_set_params: null");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    if (check_pending(c, nullptr)) return -1;
    c->theta_current = false;
    HIP_TRY(c, hipMemcpyAsync(c->d_p, p, sizeof(double) * c->n_params, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}

int32_t crnn_get_opt_state(crnn_ctx *ctx, double *state) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c || !state) return fail(c, "crnn_get_opt_state: null");
    if (!c->train_ready) return fail(c, "crnn_get_opt_state: call crnn_train_init first");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    if (check_pending(c, nullptr)) return -1;
    HIP_TRY(c, hipMemcpyAsync(state, c->d_opt, sizeof(double) * (2 * c->n_params + 4), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}

int32_t crnn_set_opt_state(crnn_ctx *ctx, const double *state) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c || !state) return fail(c, "crnn_set_opt_state: null");
    if (!c->train_ready) return fail(c, "crnn_set_opt_state: call crnn_train_init first");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    if (check_pending(c, nullptr)) return -1;
    HIP_TRY(c, hipMemcpyAsync(c->d_opt, state, sizeof(double) * (2 * c->n_params + 4), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}

int32_t crnn_train_update(crnn_ctx *ctx, const double *grad) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c || !grad) return fail(c, "crnn_train_update: null");
    if (!c->train_ready) return fail(c, "crnn_train_update: call crnn_train_init first");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    if (check_pending(c, nullptr)) return -1;
    c->opt_fused = false;
    const int P = c->n_params, npart = P + crnn::kTail;
    if (c->npart_max < npart) {
        if (c->d_red) HIP_TRY(c, hipFree(c->d_red));
        c->d_red = nullptr;
        HIP_TRY(c, hipMalloc((void **)&c->d_red, sizeof(double) * npart));
        c->npart_max = npart;
    }
    std::vector<double> red(npart, 0.0);
    for (int k = 0; k < P; ++k) red[k] = grad[k];
    red[npart - 1] = 1.0;   // n_traj = 1: the gradient is applied as given
    HIP_TRY(c, hipMemcpyAsync(c->d_red, red.data(), sizeof(double) * npart, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));   // red is a stack-lifetime vector
    c->last_npart = npart;
    c->last_P = P;
    return train_end_impl(c, nullptr);
}

int32_t crnn_last_stats(crnn_ctx *ctx, crnn_stats *stats) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c || !stats) return fail(c, "crnn_last_stats: null");
    if (c->last_npart == 0) return fail(c, "crnn_last_stats: no solve has run yet");
    if (check_pending(c, nullptr)) return -1;
    std::vector<double> red(c->last_npart);
    HIP_TRY(c, hipMemcpyAsync(red.data(), c->d_red, sizeof(double) * c->last_npart, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return fill_stats(c, red.data(), c->last_npart, stats);
}

int32_t crnn_ctx_set_queue_order(crnn_ctx *ctx, int32_t order) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c) return fail(c, "crnn_ctx_set_queue_order: null");
    if (order != CRNN_QUEUE_AUTO && order != CRNN_QUEUE_INDEX) return fail(c, "crnn_ctx_set_queue_order: order must be CRNN_QUEUE_AUTO or CRNN_QUEUE_INDEX");
    c->queue_order = order;
    return 0;
}

int32_t crnn_last_lanes_per_traj(const crnn_ctx *ctx) {
    const Ctx *c = reinterpret_cast<const Ctx *>(ctx);
    return c ? c->last_lanes : -1;
}

int64_t crnn_tape_retries(const crnn_ctx *ctx) {
    const Ctx *c = reinterpret_cast<const Ctx *>(ctx);
    return c ? c->hy_tape_retries : -1;
}

int32_t crnn_hychem_block_cap(const crnn_ctx *ctx) {
    const Ctx *c = reinterpret_cast<const Ctx *>(ctx);
    return c ? c->hy_block_cap : -1;
}

int32_t crnn_ctx_set_lanes_per_traj(crnn_ctx *ctx, int32_t lanes) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c) return fail(c, "crnn_ctx_set_lanes_per_traj: null");
    if (lanes < 0 || lanes > 2) return fail(c, "crnn_ctx_set_lanes_per_traj: lanes must be 0 (auto), 1 or 2");
    if (lanes == 2 && !c->hychem && (c->cfg.solver != CRNN_SOLVER_ROSENBROCK23 || !find_adjoint2(c)))
        return fail(c, "crnn_ctx_set_lanes_per_traj: no two-lane kernel for this problem (Rosenbrock23, nr < ns, no rate scaling)");
    c->lanes_per_traj = lanes;
    return 0;
}

int32_t crnn_ctx_set_jacobian(crnn_ctx *ctx, int32_t mode) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c) return fail(c, "crnn_ctx_set_jacobian: null");
    if (mode != CRNN_JAC_ANALYTIC && mode != CRNN_JAC_FINITE_DIFF) return fail(c, "crnn_ctx_set_jacobian: mode must be CRNN_JAC_ANALYTIC or CRNN_JAC_FINITE_DIFF");
    if (mode == CRNN_JAC_FINITE_DIFF && c->hychem) {
        // HyChem (crnn_pyrolysis_mass.jl:29): J and dT by forward differences in the primal launches of Rosenbrock23 and of the
        // AutoTsit5(Rosenbrock23) composite (hychem_auto_kernel<..., JFD>)
        if (c->cfg.solver != CRNN_SOLVER_ROSENBROCK23 && c->cfg.solver != CRNN_SOLVER_AUTOTSIT5)
            return fail(c, "crnn_ctx_set_jacobian: HyChem's finite-difference J exists for Rosenbrock23 and AutoTsit5(Rosenbrock23)");
    } else if (mode == CRNN_JAC_FINITE_DIFF) {
        const AdjEntry *ka = (c->cfg.solver != CRNN_SOLVER_ROSENBROCK23) ? nullptr : find_adjoint(c);
        if (!ka || !ka->fn_primal_fd)
            return fail(c, "crnn_ctx_set_jacobian: the finite-difference W exists for the Rosenbrock23 primal launches of the CRNN right-hand side (case1, case2, robertson shapes) and of HyChem");
    }
    c->jac_mode = mode;
    return 0;
}

int32_t crnn_last_step_counts(crnn_ctx *ctx, int64_t first, int64_t count, int32_t *n_accept, int32_t *n_reject) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c) return fail(c, "crnn_last_step_counts: null");
    if (c->last_npart == 0) return fail(c, "crnn_last_step_counts: no solve has run yet");
    if (check_pending(c, nullptr)) return -1;
    if (count < 0 || first < c->steps_first || first + count > c->steps_first + c->steps_count)
        return fail(c, "crnn_last_step_counts: [first, first+count) is not inside the range of the most recent solve");
    if (n_accept) HIP_TRY(c, hipMemcpyAsync(n_accept, c->d_nacc + first, sizeof(int32_t) * count, hipMemcpyDeviceToHost, c->stream));
    if (n_reject) HIP_TRY(c, hipMemcpyAsync(n_reject, c->d_nrej + first, sizeof(int32_t) * count, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}

int32_t crnn_kernel_times(crnn_ctx *ctx, double *ms, int32_t n) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c || !ms || n < 1) return fail(c, "crnn_kernel_times: bad arguments");
    if (n > Ctx::kRing || n > c->n_launch) return fail(c, "crnn_kernel_times: fewer launches recorded than requested");
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < n; ++i) {  // ms[0] = oldest of the last n launches
        int64_t idx = (c->n_launch - n + i) % Ctx::kRing;
        float t = 0.f;
        HIP_TRY(c, hipEventElapsedTime(&t, c->ring0[idx], c->ring1[idx]));
        ms[i] = t;
    }
    return 0;
}

int32_t crnn_synchronize(crnn_ctx *ctx) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c) return fail(nullptr, "null ctx");
    if (check_pending(c, nullptr)) return -1;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}

int32_t crnn_comm_get_unique_id(char id[CRNN_UNIQUE_ID_BYTES]) {
    static_assert(sizeof(ncclUniqueId) <= CRNN_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId uid;
    NCCL_TRY(nullptr, ncclGetUniqueId(&uid));
    std::memset(id, 0, CRNN_UNIQUE_ID_BYTES);
    std::memcpy(id, &uid, sizeof(uid));
    return 0;
}

int32_t crnn_comm_init(crnn_ctx *ctx, const char id[CRNN_UNIQUE_ID_BYTES], int32_t rank, int32_t world) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c) return fail(nullptr, "null ctx");
    if (world < 1 || rank < 0 || rank >= world) return fail(c, "crnn_comm_init: bad rank/world");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    if (c->comm) { ncclCommDestroy(c->comm); c->comm = nullptr; }
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    NCCL_TRY(c, ncclCommInitRank(&c->comm, world, uid, rank));
    c->rank = rank;
    c->world = world;
    return 0;
}

int32_t crnn_comm_set_allreduce(crnn_ctx *ctx, crnn_allreduce_fn fn, void *user) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c) return fail(nullptr, "null ctx");
    if (check_pending(c, nullptr)) return -1;
    c->host_ar = fn;
    c->host_ar_user = user;
    return 0;
}

int64_t crnn_comm_collectives(crnn_ctx *ctx) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    return c ? c->n_collectives : -1;
}

int32_t crnn_comm_destroy(crnn_ctx *ctx) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c) return fail(nullptr, "null ctx");
    if (c->comm) { NCCL_TRY(c, ncclCommDestroy(c->comm)); c->comm = nullptr; }
    c->world = 1; c->rank = 0;
    return 0;
}

int32_t crnn_allreduce_grad(crnn_ctx *ctx, double *buf, int32_t n) {
    Ctx *c = reinterpret_cast<Ctx *>(ctx);
    if (!c) return fail(nullptr, "null ctx");
    if (!buf || n < 1) return fail(c, "crnn_allreduce_grad: bad arguments");
    if (!c->comm) return 0;  // no communicator attached: single process
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    if (c->comm_buf_len < n) {
        if (c->d_comm_buf) HIP_TRY(c, hipFree(c->d_comm_buf));
        HIP_TRY(c, hipMalloc((void **)&c->d_comm_buf, sizeof(double) * n));
        c->comm_buf_len = n;
    }
    HIP_TRY(c, hipMemcpyAsync(c->d_comm_buf, buf, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    NCCL_TRY(c, ncclAllReduce(c->d_comm_buf, c->d_comm_buf, n, ncclDouble, ncclSum, c->comm, c->stream));
    HIP_TRY(c, hipMemcpyAsync(buf, c->d_comm_buf, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}


// ============================================================================ cathode-UQ entry points

int32_t crnn_cathode_config_default(crnn_cathode_config *cfg) {
    if (!cfg) return cfail(nullptr, "crnn_cathode_config_default: null cfg");
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->abi_version = CRNN_ABI_VERSION;
    cfg->maxiters = 2500000;   // config.yaml:10 asks for 2.5e9: more than int32; never reached in practice
    cfg->lb_clamp = 1e-16; cfg->T0 = 373.15; cfg->atol = 1e-12; cfg->rtol = 1e-3;
    cfg->gamma = 0.9; cfg->qmin = 0.2; cfg->qmax = 10.0; cfg->beta1 = 7.0 / 20.0; cfg->beta2 = 2.0 / 10.0;
    cfg->qsteady_min = 1.0; cfg->qsteady_max = 1.2; cfg->qoldinit = 1e-4;
    return 0;
}

int32_t crnn_cathode_create(const crnn_cathode_config *cfg, crnn_cathode_ctx **out) {
    if (!cfg || !out) return cfail(nullptr, "crnn_cathode_create: null pointer");
    *out = nullptr;
    if (cfg->abi_version != CRNN_ABI_VERSION) return cfail(nullptr, "crnn_cathode_create: abi_version mismatch");
    if (!(cfg->atol > 0) || !(cfg->rtol > 0) || cfg->maxiters < 1) return cfail(nullptr, "crnn_cathode_create: bad tolerances/maxiters");
    if (cfg->grad_mode < CRNN_GRAD_AUTO || cfg->grad_mode > CRNN_GRAD_ADJOINT) return cfail(nullptr, "crnn_cathode_create: bad grad_mode");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1) return cfail(nullptr, std::string("crnn_cathode_create: no HIP device (") + hipGetErrorString(e) + ")");
    if (cfg->device < 0 || cfg->device >= ndev) return cfail(nullptr, "crnn_cathode_create: device ordinal out of range");
    CathCtx *c = new CathCtx();
    c->cfg = *cfg;
    hipDeviceProp_t prop;
    if (hipSetDevice(cfg->device) != hipSuccess || hipGetDeviceProperties(&prop, cfg->device) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&c->ev0) != hipSuccess ||
        hipEventCreate(&c->ev1) != hipSuccess || hipMalloc((void **)&c->d_queue, sizeof(unsigned long long)) != hipSuccess) {
        crnn_cathode_destroy((crnn_cathode_ctx *)c);
        return cfail(nullptr, "crnn_cathode_create: HIP initialisation failed");
    }
    c->num_cu = prop.multiProcessorCount;
    if (const char *e = getenv("CRNN_CATH_TAPE_EVERY")) {   // measurement override (tools/cathode_bench.py)
        const int v = atoi(e);
        if (v == 1 || v == 4 || v == 8) c->tape_every = v;
    }
    *out = reinterpret_cast<crnn_cathode_ctx *>(c);
    return 0;
}

void crnn_cathode_destroy(crnn_cathode_ctx *ctx) {
    CathCtx *c = reinterpret_cast<CathCtx *>(ctx);
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) { ncclCommDestroy(c->comm); c->comm = nullptr; }
    void *ptrs[] = {c->d_ts, c->d_dbar, c->d_d2bar, c->d_beta, c->d_D, c->d_queue, c->d_theta, c->d_loss, c->d_grad,
                    c->d_hrr, c->d_ret, c->d_nsv, c->d_nacc, c->d_nrej, c->d_tape, c->d_overflow, c->d_ag_send, c->d_ag_recv,
                    c->d_pn, c->d_pn2, c->d_lnp, c->d_pscales, c->d_dirscale, c->d_chunk_stats};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    c->svgd.release();
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev2) (void)hipEventDestroy(c->ev2);
    if (c->ev3) (void)hipEventDestroy(c->ev3);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *crnn_cathode_last_error(const crnn_cathode_ctx *ctx) {
    const CathCtx *c = reinterpret_cast<const CathCtx *>(ctx);
    return c ? c->err.c_str() : g_last_error.c_str();
}

int32_t crnn_cathode_set_obs(crnn_cathode_ctx *ctx, int32_t n_sets, int32_t Dmax, const int32_t *D, const double *ts,
                             const double *dbar, const double *d2bar, const double *beta) {
    CathCtx *c = reinterpret_cast<CathCtx *>(ctx);
    if (!c) return cfail(nullptr, "null ctx");
    if (!D || !ts || !dbar || !d2bar || !beta) return cfail(c, "crnn_cathode_set_obs: null pointer");
    if (n_sets < 1 || n_sets > CRNN_CATHODE_MAX_SETS) return cfail(c, "crnn_cathode_set_obs: n_sets must be in [1, 4096]");
    if (Dmax < 2 || Dmax > CRNN_CATHODE_MAX_D) return cfail(c, "crnn_cathode_set_obs: Dmax must be in [2, 128]");
    for (int s = 0; s < n_sets; ++s) {
        if (D[s] < 2 || D[s] > Dmax) return cfail(c, "crnn_cathode_set_obs: D[s] out of range");
        if (!(beta[s] > 0)) return cfail(c, "crnn_cathode_set_obs: heating rates must be positive");
        for (int i = 1; i < D[s]; ++i)
            if (!(ts[(size_t)s * Dmax + i] > ts[(size_t)s * Dmax + i - 1])) return cfail(c, "crnn_cathode_set_obs: ts must be strictly increasing");
    }
    CHIP(c, hipSetDevice(c->cfg.device));
    CHIP(c, hipStreamSynchronize(c->stream));
    size_t n = (size_t)n_sets * Dmax;
    if (cgrow(c, &c->d_ts, n) || cgrow(c, &c->d_dbar, n) || cgrow(c, &c->d_d2bar, n) || cgrow(c, &c->d_beta, (size_t)n_sets) ||
        cgrow(c, &c->d_D, (size_t)n_sets))
        return -1;
    CHIP(c, hipMemcpy(c->d_ts, ts, n * sizeof(double), hipMemcpyHostToDevice));
    CHIP(c, hipMemcpy(c->d_dbar, dbar, n * sizeof(double), hipMemcpyHostToDevice));
    CHIP(c, hipMemcpy(c->d_d2bar, d2bar, n * sizeof(double), hipMemcpyHostToDevice));
    CHIP(c, hipMemcpy(c->d_beta, beta, n_sets * sizeof(double), hipMemcpyHostToDevice));
    CHIP(c, hipMemcpy(c->d_D, D, n_sets * sizeof(int32_t), hipMemcpyHostToDevice));
    c->n_sets = n_sets;
    c->Dmax = Dmax;
    c->cap_hrr = 0;
    return 0;
}

}  // extern "C" (interrupted)
namespace {
// Device-level run: theta [n_part][17] already in c->d_theta; integrates every particle for the observation sets
// [set_first, set_first + set_count) (trajectory tr = particle * set_count + (set - set_first)) and leaves loss / grad / hrr /
// retcode / n_saved / step counts in the ctx's device buffers.  Adjoint first (unless grad_mode says forward); a trajectory
// that outruns the tape makes the call repeat with forward tangents (one 4-byte read-back decides).
int32_t cath_run(CathCtx *c, int64_t n_part, int set_first, int set_count, bool want_grad, bool want_hrr) {
    const bool grad_requested = want_grad;
    (void)grad_requested;
    const int64_t ntraj = n_part * set_count;
    if ((size_t)ntraj > c->cap_traj) {
        if (cgrow(c, &c->d_loss, (size_t)ntraj) || cgrow(c, &c->d_grad, (size_t)ntraj * CRNN_CATHODE_NP) ||
            cgrow(c, &c->d_ret, (size_t)ntraj) || cgrow(c, &c->d_nsv, (size_t)ntraj) || cgrow(c, &c->d_nacc, (size_t)ntraj) ||
            cgrow(c, &c->d_nrej, (size_t)ntraj))
            return -1;
        c->cap_traj = (size_t)ntraj;
    }
    if (want_hrr && (size_t)ntraj * c->Dmax > c->cap_hrr) {
        if (cgrow(c, &c->d_hrr, (size_t)ntraj * c->Dmax)) return -1;
        c->cap_hrr = (size_t)ntraj * c->Dmax;
    }
    CHIP(c, hipMemsetAsync(c->d_queue, 0, sizeof(unsigned long long), c->stream));
    if (want_hrr) CHIP(c, hipMemsetAsync(c->d_hrr, 0, sizeof(double) * (size_t)ntraj * c->Dmax, c->stream));
    crnn::CathodeParams prm{};
    prm.theta = c->d_theta;
    prm.ts = c->d_ts + (size_t)set_first * c->Dmax; prm.dbar = c->d_dbar + (size_t)set_first * c->Dmax;
    prm.d2bar = c->d_d2bar + (size_t)set_first * c->Dmax; prm.beta = c->d_beta + set_first; prm.D = c->d_D + set_first;
    prm.loss = c->d_loss; prm.grad = c->d_grad; prm.hrr = want_hrr ? c->d_hrr : nullptr;
    prm.retcode = c->d_ret; prm.n_saved = c->d_nsv; prm.n_accept = c->d_nacc; prm.n_reject = c->d_nrej;
    prm.queue = c->d_queue;
    prm.n_traj = ntraj; prm.n_sets = set_count; prm.Dmax = c->Dmax; prm.maxiters = c->cfg.maxiters;
    prm.want_grad = want_grad ? 1 : 0;
    prm.lb = c->cfg.lb_clamp; prm.T0 = c->cfg.T0; prm.atol = c->cfg.atol; prm.rtol = c->cfg.rtol;
    prm.gamma = c->cfg.gamma; prm.qmin = c->cfg.qmin; prm.qmax = c->cfg.qmax; prm.beta1 = c->cfg.beta1; prm.beta2 = c->cfg.beta2;
    prm.qsteady_min = c->cfg.qsteady_min; prm.qsteady_max = c->cfg.qsteady_max; prm.qoldinit = c->cfg.qoldinit;
    constexpr int kB = 256;
    bool done = false;
    if (want_grad && c->errnorm_sens != 0) {
        // The gradient as the reference evaluates it (network.jl:232): ForwardDiff's two chunks of p, each its own adaptive solve whose
        // error norm weighs the chunk's partials (cathode_sens_kernel.hpp); loss, heat-release curves, return codes and step statistics
        // are those of the plain solve that follows.
        crnn::CathSensParams sp{};
        sp.dir_scale = c->d_dirscale; sp.mode = c->errnorm_sens; sp.dual_partials = 9;
        crnn::CathodeParams prs = prm;
        prs.hrr = nullptr; prs.want_grad = 1;
        if (!c->d_chunk_stats) CHIP(c, hipMalloc((void **)&c->d_chunk_stats, sizeof(long long) * 4));
        CHIP(c, hipMemsetAsync(c->d_chunk_stats, 0, sizeof(long long) * 4, c->stream));
        // ... through Rosenbrock23 (cathode_sens_kernel: one lane per trajectory) or, with the reference's own stepper selected
        // (crnn_cathode_set_solver: AutoTsit5(TRBDF2), network.jl:195), through that composite (cathode_sens_auto_kernel: nine lanes per trajectory)
        const bool comp = c->solver == CRNN_CATH_SOLVER_AUTOTSIT5_TRBDF2;
        if (comp) { prs.qsteady_min = 1.0; prs.qsteady_max = 1.0; }   // qsteady_max_default of a composite
        for (int ch = 0; ch < 2; ++ch) {
            using SensFn = void (*)(const crnn::CathodeParams, const crnn::CathSensParams);
            const SensFn fn = comp ? (ch == 0 ? (SensFn)crnn::cathode_sens_auto_kernel<kB, 0> : (SensFn)crnn::cathode_sens_auto_kernel<kB, 1>)
                                   : (ch == 0 ? (SensFn)crnn::cathode_sens_kernel<kB, 0> : (SensFn)crnn::cathode_sens_kernel<kB, 1>);
            int &occ_s = comp ? c->sensc_occ[ch] : c->sens_occ[ch];
            if (occ_s < 1) {
                CHIP(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_s, (const void *)fn, kB, 0));
                if (occ_s < 1) occ_s = 1;
            }
            const int64_t per_blk = comp ? (kB / 64) * 7 : kB;        // trajectories a block holds at a time
            const int nblk = (int)std::max<int64_t>(1, std::min<int64_t>((ntraj + per_blk - 1) / per_blk, (int64_t)c->num_cu * occ_s));
            CHIP(c, hipMemsetAsync(c->d_queue, 0, sizeof(unsigned long long), c->stream));
            hipLaunchKernelGGL(fn, dim3(nblk), dim3(kB), 0, c->stream, prs, sp);
            CHIP(c, hipGetLastError());
            // the chunk's step counts, summed where they are (no copy, no wait: crnn_cathode_last_chunk_stats reads four numbers on request)
            hipLaunchKernelGGL(cath_sum_counts_kernel, dim3(256), dim3(256), 0, c->stream, c->d_nacc, c->d_nrej, ntraj, c->d_chunk_stats + 2 * ch);
            CHIP(c, hipGetLastError());
        }
        c->chunk_stats_stale = true;
        CHIP(c, hipMemsetAsync(c->d_queue, 0, sizeof(unsigned long long), c->stream));
        prm.grad = nullptr; prm.want_grad = 0;      // the gradient rows are complete; what follows is the plain solve
        want_grad = false;
    }
    const bool primal = !want_grad;      // primal calls: the adjoint kernel's forward sweep alone (cathode_adj_kernel<..., PRIMAL>)
    if (primal && c->solver != CRNN_CATH_SOLVER_ROSENBROCK23) {
        // the reference's composite (network.jl:195): cathode_auto_kernel, a wavefront takes 64 particles of one heating rate
        const bool tr = c->solver == CRNN_CATH_SOLVER_AUTOTSIT5_TRBDF2;
        using AutoFn = void (*)(const crnn::CathodeParams, const crnn::CathAdjParams);
        const AutoFn fn = tr ? (AutoFn)crnn::cathode_auto_kernel<kB, true> : (AutoFn)crnn::cathode_auto_kernel<kB, false>;
        int &occ_ = c->auto_occ[tr ? 0 : 1];
        if (occ_ < 1) {
            CHIP(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_, (const void *)fn, kB, 0));
            if (occ_ < 1) occ_ = 1;
        }
        prm.qsteady_min = 1.0; prm.qsteady_max = 1.0;   // qsteady_max_default of a composite (not an implicit algorithm type)
        const int64_t n_batches = ((n_part + 63) / 64) * set_count;
        const int nblk = (int)std::max<int64_t>(1, std::min<int64_t>((n_batches + 3) / 4, (int64_t)c->num_cu * occ_));
        crnn::CathAdjParams adj{};
        adj.n_part = n_part;
        CHIP(c, hipEventRecord(c->ev0, c->stream));
        hipLaunchKernelGGL(fn, dim3(nblk), dim3(kB), 0, c->stream, prm, adj);
        CHIP(c, hipGetLastError());
        CHIP(c, hipEventRecord(c->ev1, c->stream));
        return 0;
    }
    if (primal || c->cfg.grad_mode != CRNN_GRAD_FORWARD) {
        // discrete adjoint (cathode_adj_kernel): a wavefront takes 64 particles of one heating rate
        // tape layout: every step in full (40 B) or checkpointed every kcp-th step (8 + 32 / kcp B per step; cathode_kernel.hpp)
        const int kcp = primal ? 1 : c->tape_every;
        using AdjFn = void (*)(const crnn::CathodeParams, const crnn::CathAdjParams);
        const AdjFn adj_fn = primal ? (AdjFn)crnn::cathode_adj_kernel<kB, 1, true>
                           : kcp == 2 ? (AdjFn)crnn::cathode_adj_kernel<kB, 2>
                           : kcp == 4 ? (AdjFn)crnn::cathode_adj_kernel<kB, 4> : kcp == 8 ? (AdjFn)crnn::cathode_adj_kernel<kB, 8>
                                                                                          : (AdjFn)crnn::cathode_adj_kernel<kB, 1>;
        int &occ_ = primal ? c->prim_occ : c->adj_occ;      // (the primal instantiation fits two wavefronts per SIMD)
        if (occ_ < 1) {
            CHIP(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_, (const void *)adj_fn, kB, 0));
            if (occ_ < 1) occ_ = 1;
        }
        const int64_t n_batches = ((n_part + 63) / 64) * set_count;
        const int nblk = (int)std::max<int64_t>(1, std::min<int64_t>((n_batches + 3) / 4, (int64_t)c->num_cu * occ_));
        const size_t lanes = (size_t)nblk * kB;
        if (c->tape_budget == 0) {
            size_t fr = 0, tot = 0;
            CHIP(c, hipMemGetInfo(&fr, &tot));
            c->tape_budget = std::min<size_t>(fr / 4, (size_t)16 << 30);
        }
        const double per_step = kcp > 1 ? 1.0 + 4.0 / kcp : 5.0;     // doubles per recorded step
        int64_t cap = std::max<int64_t>((int64_t)((double)c->tape_budget / ((double)lanes * per_step * sizeof(double))) - kcp, 64);
        cap = std::min<int64_t>(cap, c->cfg.maxiters);
        const size_t lane_doubles = kcp > 1 ? (size_t)cap + 4 * (((size_t)cap + kcp - 1) / kcp) : (size_t)cap * 5;
        if (!primal && c->tape_doubles < lanes * lane_doubles) {
            if (cgrow(c, &c->d_tape, lanes * lane_doubles)) return -1;
            c->tape_doubles = lanes * lane_doubles;
        }
        if (!c->d_overflow) CHIP(c, hipMalloc((void **)&c->d_overflow, sizeof(unsigned int)));
        CHIP(c, hipMemsetAsync(c->d_overflow, 0, sizeof(unsigned int), c->stream));
        crnn::CathAdjParams adj{};
        adj.tape = c->d_tape; adj.tape_cap = (int32_t)cap; adj.overflow = c->d_overflow; adj.n_part = n_part;
        CHIP(c, hipEventRecord(c->ev0, c->stream));
        hipLaunchKernelGGL(adj_fn, dim3(nblk), dim3(kB), 0, c->stream, prm, adj);
        CHIP(c, hipGetLastError());
        CHIP(c, hipEventRecord(c->ev1, c->stream));
        unsigned int ovf = 0;
        CHIP(c, hipMemcpyAsync(&ovf, c->d_overflow, sizeof(ovf), hipMemcpyDeviceToHost, c->stream));
        CHIP(c, hipStreamSynchronize(c->stream));
        done = (ovf == 0);   // otherwise some trajectory outran the tape: repeat the call with forward tangents
        if (!done) CHIP(c, hipMemsetAsync(c->d_queue, 0, sizeof(unsigned long long), c->stream));
    }
    if (!done) {
        if (c->fwd_occ < 1) {
            CHIP(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&c->fwd_occ, (const void *)crnn::cathode_kernel<kB>, kB, 0));
            if (c->fwd_occ < 1) c->fwd_occ = 1;
        }
        int nblk = (int)std::max<int64_t>(1, std::min<int64_t>((ntraj + kB - 1) / kB, (int64_t)c->num_cu * c->fwd_occ));
        CHIP(c, hipEventRecord(c->ev0, c->stream));
        hipLaunchKernelGGL(crnn::cathode_kernel<kB>, dim3(nblk), dim3(kB), 0, c->stream, prm);
        CHIP(c, hipGetLastError());
        CHIP(c, hipEventRecord(c->ev1, c->stream));
    }
    return 0;
}

// theta = p .* p_scales (network.jl:152-157: parameters pre-multiplied by p_scales) for the device-resident particles
__global__ void cath_theta_kernel(const double *__restrict__ p, const double *__restrict__ p_scales, int64_t n, double *__restrict__ theta) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx < n) theta[idx] = p[idx] * p_scales[idx % CRNN_CATHODE_NP];
}
// lnpgrad[:, k] = -(d loss / d p_k) / normalizer2[k] = -(d loss / d theta_k) p_scales[k] / normalizer2[k]  (dlnprob, network.jl:234-250);
// block 0 also forms the mean loss and the number of failed solves in fixed order: out2 = { mean loss, n_failed }
__global__ __launch_bounds__(256) void cath_lnpgrad_kernel(const double *__restrict__ grad, const double *__restrict__ p_scales, int64_t n_part,
                                                           const double *__restrict__ norm2, double *__restrict__ lnp, const double *__restrict__ loss,
                                                           const int32_t *__restrict__ ret, double *out2) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx < n_part * CRNN_CATHODE_NP) lnp[idx] = -(grad[idx] * p_scales[idx % CRNN_CATHODE_NP]) / norm2[idx % CRNN_CATHODE_NP];
    if (blockIdx.x == 0) {
        __shared__ double sh[256], shf[256];
        double a = 0.0, f = 0.0;
        for (int64_t i = threadIdx.x; i < n_part; i += 256) { a += loss[i]; f += ret[i] != 0 ? 1.0 : 0.0; }
        sh[threadIdx.x] = a; shf[threadIdx.x] = f;
        __syncthreads();
        for (int s_ = 128; s_ > 0; s_ >>= 1) {
            if ((int)threadIdx.x < s_) { sh[threadIdx.x] += sh[threadIdx.x + s_]; shf[threadIdx.x] += shf[threadIdx.x + s_]; }
            __syncthreads();
        }
        if (threadIdx.x == 0) { out2[0] = sh[0] / (double)n_part; out2[1] = shf[0]; }
    }
}
}  // namespace
extern "C" {

int32_t crnn_cathode_solve(crnn_cathode_ctx *ctx, const double *theta, int64_t n_part, double *loss, double *grad,
                           double *hrr, int32_t *retcode, int32_t *n_saved, crnn_stats *stats) {
    CathCtx *c = reinterpret_cast<CathCtx *>(ctx);
    if (!c) return cfail(nullptr, "null ctx");
    if (!theta || n_part < 1) return cfail(c, "crnn_cathode_solve: bad theta / n_part");
    if (c->n_sets < 1) return cfail(c, "crnn_cathode_solve: no observation sets (crnn_cathode_set_obs)");
    CHIP(c, hipSetDevice(c->cfg.device));
    const int64_t ntraj = n_part * c->n_sets;
    if ((size_t)n_part > c->cap_part) {
        if (cgrow(c, &c->d_theta, (size_t)n_part * CRNN_CATHODE_NP)) return -1;
        c->cap_part = (size_t)n_part;
    }
    CHIP(c, hipMemcpyAsync(c->d_theta, theta, sizeof(double) * (size_t)n_part * CRNN_CATHODE_NP, hipMemcpyHostToDevice, c->stream));
    if (cath_run(c, n_part, 0, c->n_sets, grad != nullptr, hrr != nullptr)) return -1;
    std::vector<int32_t> h_ret((size_t)ntraj), h_nacc((size_t)ntraj), h_nrej((size_t)ntraj);
    if (loss) CHIP(c, hipMemcpyAsync(loss, c->d_loss, sizeof(double) * ntraj, hipMemcpyDeviceToHost, c->stream));
    if (grad) CHIP(c, hipMemcpyAsync(grad, c->d_grad, sizeof(double) * ntraj * CRNN_CATHODE_NP, hipMemcpyDeviceToHost, c->stream));
    if (hrr) CHIP(c, hipMemcpyAsync(hrr, c->d_hrr, sizeof(double) * ntraj * c->Dmax, hipMemcpyDeviceToHost, c->stream));
    if (n_saved) CHIP(c, hipMemcpyAsync(n_saved, c->d_nsv, sizeof(int32_t) * ntraj, hipMemcpyDeviceToHost, c->stream));
    CHIP(c, hipMemcpyAsync(h_ret.data(), c->d_ret, sizeof(int32_t) * ntraj, hipMemcpyDeviceToHost, c->stream));
    CHIP(c, hipMemcpyAsync(h_nacc.data(), c->d_nacc, sizeof(int32_t) * ntraj, hipMemcpyDeviceToHost, c->stream));
    CHIP(c, hipMemcpyAsync(h_nrej.data(), c->d_nrej, sizeof(int32_t) * ntraj, hipMemcpyDeviceToHost, c->stream));
    CHIP(c, hipStreamSynchronize(c->stream));
    if (retcode) std::memcpy(retcode, h_ret.data(), sizeof(int32_t) * ntraj);
    if (stats) {
        stats->n_traj = ntraj; stats->n_ok = 0; stats->n_accept = 0; stats->n_reject = 0;
        for (int64_t i = 0; i < ntraj; ++i) { stats->n_ok += h_ret[i] == 0; stats->n_accept += h_nacc[i]; stats->n_reject += h_nrej[i]; }
        float ms = 0.f;
        CHIP(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
        stats->kernel_ms = ms;
    }
    return 0;
}

// ---- device-resident SVGD loop of the Bayesian ensemble (crnn_cathode.jl:36-50): particles, gradients and the move stay on
// the device; per iteration one solve launch over the particles of ONE heating rate (the reference draws i_exp at random)
// and the SVGD move, enqueued back to back.
int32_t crnn_cathode_set_tape_every(crnn_cathode_ctx *ctx, int32_t every) {
    CathCtx *c = reinterpret_cast<CathCtx *>(ctx);
    if (!c) return cfail(nullptr, "null ctx");
    if (every != 1 && every != 2 && every != 4 && every != 8) return cfail(c, "crnn_cathode_set_tape_every: every must be 1, 2, 4 or 8");
    if (every != c->tape_every) { c->tape_every = every; c->adj_occ = 0; }
    return 0;
}

int32_t crnn_cathode_set_solver(crnn_cathode_ctx *ctx, int32_t solver) {
    CathCtx *c = reinterpret_cast<CathCtx *>(ctx);
    if (!c) return cfail(nullptr, "null ctx");
    if (solver != CRNN_CATH_SOLVER_ROSENBROCK23 && solver != CRNN_CATH_SOLVER_AUTOTSIT5_TRBDF2 && solver != CRNN_CATH_SOLVER_AUTOTSIT5_ROS23)
        return cfail(c, "crnn_cathode_set_solver: solver must be CRNN_CATH_SOLVER_ROSENBROCK23, _AUTOTSIT5_TRBDF2 or _AUTOTSIT5_ROS23");
    c->solver = solver;
    return 0;
}

int32_t crnn_cathode_set_errnorm_sens(crnn_cathode_ctx *ctx, int32_t mode, const double *p_scales) {
    CathCtx *c = reinterpret_cast<CathCtx *>(ctx);
    if (!c) return cfail(nullptr, "null ctx");
    if (mode < 0 || mode > 2) return cfail(c, "crnn_cathode_set_errnorm_sens: mode must be 0 (off), 1 or 2");
    if (mode != 0 && !p_scales) return cfail(c, "crnn_cathode_set_errnorm_sens: p_scales (d theta / d p, 17 entries) is required");
    CHIP(c, hipSetDevice(c->cfg.device));
    if (mode != 0) {
        for (int k = 0; k < CRNN_CATHODE_NP; ++k)
            if (!std::isfinite(p_scales[k])) return cfail(c, "crnn_cathode_set_errnorm_sens: p_scales must be finite");
        if (!c->d_dirscale) CHIP(c, hipMalloc((void **)&c->d_dirscale, sizeof(double) * CRNN_CATHODE_NP));
        CHIP(c, hipStreamSynchronize(c->stream));
        CHIP(c, hipMemcpy(c->d_dirscale, p_scales, sizeof(double) * CRNN_CATHODE_NP, hipMemcpyHostToDevice));
        for (int k = 0; k < CRNN_CATHODE_NP; ++k) c->h_dirscale[k] = p_scales[k];
    }
    c->errnorm_sens = mode;
    return 0;
}

int32_t crnn_cathode_last_chunk_stats(crnn_cathode_ctx *ctx, int64_t *out /* [4] */) {
    CathCtx *c = reinterpret_cast<CathCtx *>(ctx);
    if (!c || !out) return cfail(c, "crnn_cathode_last_chunk_stats: null");
    if (c->chunk_stats_stale) {
        long long h[4];
        CHIP(c, hipSetDevice(c->cfg.device));
        CHIP(c, hipMemcpyAsync(h, c->d_chunk_stats, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        CHIP(c, hipStreamSynchronize(c->stream));
        for (int i = 0; i < 4; ++i) c->chunk_stats[i] = h[i];
        c->chunk_stats_stale = false;
    }
    for (int i = 0; i < 4; ++i) out[i] = c->chunk_stats[i];
    return 0;
}

int32_t crnn_cathode_set_particles(crnn_cathode_ctx *ctx, const double *p, const double *p_scales, int64_t n_part) {
    CathCtx *c = reinterpret_cast<CathCtx *>(ctx);
    if (!c) return cfail(nullptr, "null ctx");
    if (!p || !p_scales || n_part < 2) return cfail(c, "crnn_cathode_set_particles: bad arguments (n_part >= 2)");
    if (c->errnorm_sens != 0)      // the dual-norm gradient weighs partials with d theta / d p: one p_scales for both
        for (int k = 0; k < CRNN_CATHODE_NP; ++k)
            if (p_scales[k] != c->h_dirscale[k]) return cfail(c, "crnn_cathode_set_particles: p_scales differ from those given to crnn_cathode_set_errnorm_sens");
    CHIP(c, hipSetDevice(c->cfg.device));
    CHIP(c, hipStreamSynchronize(c->stream));
    if ((size_t)n_part > c->cap_part) {
        if (cgrow(c, &c->d_theta, (size_t)n_part * CRNN_CATHODE_NP)) return -1;
        c->cap_part = (size_t)n_part;
    }
    if ((size_t)n_part > c->cap_pn) {
        if (cgrow(c, &c->d_pn, (size_t)n_part * CRNN_CATHODE_NP) || cgrow(c, &c->d_pn2, (size_t)n_part * CRNN_CATHODE_NP) ||
            cgrow(c, &c->d_lnp, (size_t)n_part * CRNN_CATHODE_NP))
            return -1;
        c->cap_pn = (size_t)n_part;
    }
    if (!c->d_pscales) CHIP(c, hipMalloc((void **)&c->d_pscales, sizeof(double) * (2 * CRNN_CATHODE_NP + 2)));   // [p_scales | mean loss, n_failed | normalizer2]
    CHIP(c, svgd_ws_reserve(c->svgd, n_part, CRNN_CATHODE_NP, false));
    CHIP(c, hipMemsetAsync(c->svgd.d_sel, 0, sizeof(crnn::SvgdSel), c->stream));   // new particles: the sticky `bad` flag is cleared
    CHIP(c, hipMemcpyAsync(c->d_pn, p, sizeof(double) * (size_t)n_part * CRNN_CATHODE_NP, hipMemcpyHostToDevice, c->stream));
    CHIP(c, hipMemcpyAsync(c->d_pscales, p_scales, sizeof(double) * CRNN_CATHODE_NP, hipMemcpyHostToDevice, c->stream));
    CHIP(c, hipStreamSynchronize(c->stream));
    c->n_particles = n_part;
    return 0;
}

int32_t crnn_cathode_get_particles(crnn_cathode_ctx *ctx, double *p) {
    CathCtx *c = reinterpret_cast<CathCtx *>(ctx);
    if (!c) return cfail(nullptr, "null ctx");
    if (!p || c->n_particles < 2) return cfail(c, "crnn_cathode_get_particles: no particles on the device (crnn_cathode_set_particles)");
    CHIP(c, hipSetDevice(c->cfg.device));
    crnn::SvgdSel sel{};
    CHIP(c, hipMemcpyAsync(p, c->d_pn, sizeof(double) * (size_t)c->n_particles * CRNN_CATHODE_NP, hipMemcpyDeviceToHost, c->stream));
    CHIP(c, hipMemcpyAsync(&sel, c->svgd.d_sel, sizeof(sel), hipMemcpyDeviceToHost, c->stream));
    CHIP(c, hipStreamSynchronize(c->stream));
    if (sel.bad) return cfail(c, "crnn_cathode_get_particles: an earlier crnn_cathode_svgd_step found a bandwidth that was not finite and "
                                 "positive and left the particles unmoved from there on (coincident particles, or NaN gradients of failed solves)");
    return 0;
}

int32_t crnn_cathode_svgd_step(crnn_cathode_ctx *ctx, int32_t i_set, const double *normalizer2 /*[17]*/, double stepsize, double h,
                               double *loss_mean, double *h_out, double *ms /* [2]: solve kernel, SVGD move; may be NULL */) {
    CathCtx *c = reinterpret_cast<CathCtx *>(ctx);
    if (!c) return cfail(nullptr, "null ctx");
    if (c->n_particles < 2) return cfail(c, "crnn_cathode_svgd_step: no particles on the device (crnn_cathode_set_particles)");
    if (i_set < 0 || i_set >= c->n_sets) return cfail(c, "crnn_cathode_svgd_step: i_set out of range");
    if (!normalizer2) return cfail(c, "crnn_cathode_svgd_step: normalizer2 is null");
    for (int k = 0; k < CRNN_CATHODE_NP; ++k)
        if (!(normalizer2[k] > 0)) return cfail(c, "crnn_cathode_svgd_step: normalizer2 must be positive");
    CHIP(c, hipSetDevice(c->cfg.device));
    const int64_t N = c->n_particles, nd = N * CRNN_CATHODE_NP;
    std::memcpy(c->h_norm2, normalizer2, sizeof(c->h_norm2));   // (stable host copy for the asynchronous upload)
    CHIP(c, hipMemcpyAsync(c->d_pscales + CRNN_CATHODE_NP + 2, c->h_norm2, sizeof(c->h_norm2), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(cath_theta_kernel, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, c->stream, c->d_pn, c->d_pscales, nd, c->d_theta);
    CHIP(c, hipGetLastError());
    if (cath_run(c, N, i_set, 1, true, false)) return -1;
    if (!c->ev2) { CHIP(c, hipEventCreate(&c->ev2)); CHIP(c, hipEventCreate(&c->ev3)); }
    hipLaunchKernelGGL(cath_lnpgrad_kernel, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, c->stream, c->d_grad, c->d_pscales, N,
                       c->d_pscales + CRNN_CATHODE_NP + 2, c->d_lnp, c->d_loss, c->d_ret, c->d_pscales + CRNN_CATHODE_NP);
    CHIP(c, hipGetLastError());
    CHIP(c, hipEventRecord(c->ev2, c->stream));
    CHIP(c, svgd_enqueue(c->svgd, c->stream, c->d_pn, c->d_lnp, N, CRNN_CATHODE_NP, stepsize, h, c->d_pn2, nullptr, nullptr));
    CHIP(c, hipEventRecord(c->ev3, c->stream));
    std::swap(c->d_pn, c->d_pn2);      // the moved particles are the current ones
    if (loss_mean || h_out || ms) {    // the caller looks: read back (otherwise the step stays enqueued)
        double out2[2] = {0.0, 0.0};
        crnn::SvgdSel sel{};
        CHIP(c, hipMemcpyAsync(out2, c->d_pscales + CRNN_CATHODE_NP, sizeof(out2), hipMemcpyDeviceToHost, c->stream));
        CHIP(c, hipMemcpyAsync(&sel, c->svgd.d_sel, sizeof(sel), hipMemcpyDeviceToHost, c->stream));
        CHIP(c, hipStreamSynchronize(c->stream));
        if (out2[1] != 0.0) printf("ode solver failed\n");     // network.jl:214
        if (sel.bad || !(sel.h > 0) || !(sel.h < INFINITY))
            return cfail(c, "crnn_cathode_svgd_step: bandwidth h is not finite and positive (all particles coincide, or a failed solve "
                            "gave NaN gradients); the particles were left where they were");
        if (loss_mean) *loss_mean = out2[0];
        if (h_out) *h_out = sel.h;
        if (ms) {
            float a = 0.f, b2 = 0.f;
            CHIP(c, hipEventElapsedTime(&a, c->ev0, c->ev1));
            CHIP(c, hipEventElapsedTime(&b2, c->ev2, c->ev3));
            ms[0] = a; ms[1] = b2;
        }
    }
    return 0;
}


// ---- particle-shard exchange of the Bayesian ensemble: the particles are sharded over the ranks, every
// rank needs all rows of [loss | lnpgrad] for the replicated SVGD move that follows dlnprob (crnn_cathode.jl:31,36-50).
int32_t crnn_cathode_comm_init(crnn_cathode_ctx *ctx, const char id[CRNN_UNIQUE_ID_BYTES], int32_t rank, int32_t world) {
    CathCtx *c = reinterpret_cast<CathCtx *>(ctx);
    if (!c) return cfail(nullptr, "null ctx");
    if (!id || world < 1 || rank < 0 || rank >= world) return cfail(c, "crnn_cathode_comm_init: bad id/rank/world");
    CHIP(c, hipSetDevice(c->cfg.device));
    if (c->comm) { ncclCommDestroy(c->comm); c->comm = nullptr; }
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    ncclResult_t r = ncclCommInitRank(&c->comm, world, uid, rank);
    if (r != ncclSuccess) return cfail(c, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
    c->rank = rank;
    c->world = world;
    return 0;
}

int32_t crnn_cathode_comm_destroy(crnn_cathode_ctx *ctx) {
    CathCtx *c = reinterpret_cast<CathCtx *>(ctx);
    if (!c) return cfail(nullptr, "null ctx");
    if (c->comm) { ncclCommDestroy(c->comm); c->comm = nullptr; }
    c->rank = 0; c->world = 1;
    return 0;
}

int32_t crnn_cathode_allgather(crnn_cathode_ctx *ctx, const double *local, int64_t n_local, int32_t width, int64_t n_total,
                               double *full) {
    CathCtx *c = reinterpret_cast<CathCtx *>(ctx);
    if (!c) return cfail(nullptr, "null ctx");
    if (!local || !full || width < 1 || n_total < 1) return cfail(c, "crnn_cathode_allgather: bad arguments");
    const int world = c->comm ? c->world : 1, rank = c->comm ? c->rank : 0;
    const int64_t base = n_total / world, rem = n_total % world;
    auto first_of = [&](int r) { return (int64_t)r * base + std::min<int64_t>(r, rem); };
    auto count_of = [&](int r) { return base + (r < rem ? 1 : 0); };
    if (n_local != count_of(rank)) return cfail(c, "crnn_cathode_allgather: this rank's block must hold rows [first, first+count) of the contiguous partition");
    if (world == 1) { std::memcpy(full, local, sizeof(double) * (size_t)n_total * width); return 0; }
    CHIP(c, hipSetDevice(c->cfg.device));
    const size_t blk = (size_t)(base + (rem ? 1 : 0)) * width;    // equal, padded block per rank (ncclAllGather wants equal counts)
    if (c->ag_send_cap < blk) { if (cgrow(c, &c->d_ag_send, blk)) return -1; c->ag_send_cap = blk; }
    if (c->ag_recv_cap < blk * world) { if (cgrow(c, &c->d_ag_recv, blk * world)) return -1; c->ag_recv_cap = blk * world; }
    CHIP(c, hipMemsetAsync(c->d_ag_send, 0, sizeof(double) * blk, c->stream));
    CHIP(c, hipMemcpyAsync(c->d_ag_send, local, sizeof(double) * (size_t)n_local * width, hipMemcpyHostToDevice, c->stream));
    ncclResult_t r = ncclAllGather(c->d_ag_send, c->d_ag_recv, blk, ncclDouble, c->comm, c->stream);
    if (r != ncclSuccess) return cfail(c, std::string("ncclAllGather: ") + ncclGetErrorString(r));
    for (int q = 0; q < world; ++q)
        CHIP(c, hipMemcpyAsync(full + (size_t)first_of(q) * width, c->d_ag_recv + (size_t)q * blk,
                               sizeof(double) * (size_t)count_of(q) * width, hipMemcpyDeviceToHost, c->stream));
    CHIP(c, hipStreamSynchronize(c->stream));
    return 0;
}

// ============================================================================ SVGD move
// Workspace of a move (device buffers sized for N x dim), kept between calls: crnn_svgd_update caches one per device, a
// cathode ctx owns one for its device-resident loop.  Nothing is allocated in steady state.
}  // extern "C" (interrupted: helpers with C++ linkage)
namespace {
hipError_t svgd_ws_reserve(SvgdWs &w, int64_t N, int dim, bool io_buffers) {
    const int nchunk = (int)std::max<int64_t>(1, std::min<int64_t>(64, (256 * 256 * 4) / std::max<int64_t>(N, 1)));
    if (w.N == N && w.dim == dim && w.d_part && (!io_buffers || w.d_p)) return hipSuccess;
    w.release();
    const size_t nd = (size_t)N * dim;
    hipError_t e = hipSuccess;
    auto get = [&](void **q, size_t bytes) { if (e == hipSuccess) e = hipMalloc(q, bytes); };
    if (io_buffers) {
        get((void **)&w.d_p, nd * sizeof(double)); get((void **)&w.d_g, nd * sizeof(double)); get((void **)&w.d_new, nd * sizeof(double));
        get((void **)&w.d_dt, nd * sizeof(double)); get((void **)&w.d_rep, nd * sizeof(double));
    }
    get((void **)&w.d_part, (size_t)N * nchunk * (1 + 2 * dim) * sizeof(double));
    get((void **)&w.d_hist, crnn::kSvgdBins * sizeof(unsigned int));
    get((void **)&w.d_sel, sizeof(crnn::SvgdSel));
    get((void **)&w.d_cnt, 3 * sizeof(unsigned long long));
    const size_t npairs = (size_t)N * (size_t)(N - 1) / 2;
    // the stored pair distances are an optimisation (4 096 particles: 67 MB): if they do not fit, or the allocation fails, the
    // select recomputes them per pass
    if (e == hipSuccess && npairs * sizeof(double) <= ((size_t)8 << 30) &&
        hipMalloc((void **)&w.d_dist, npairs * sizeof(double)) != hipSuccess) { w.d_dist = nullptr; (void)hipGetLastError(); }
    if (e == hipSuccess) e = hipMemset(w.d_hist, 0, crnn::kSvgdBins * sizeof(unsigned int));   // the pick kernel keeps it zeroed from here on
    if (e == hipSuccess) e = hipMemset(w.d_sel, 0, sizeof(crnn::SvgdSel));                      // (the sticky `bad` flag starts clear)
    if (e != hipSuccess) { w.release(); return e; }
    w.N = N; w.dim = dim; w.nchunk = nchunk;
    return hipSuccess;
}

// The whole move on `stream`, device buffers in and out, no host round trip: two order statistics of the pair distances by
// radix select (12 + 4 x 13 bits, state in device memory), bandwidth, row sums, update.  h >= 0: given bandwidth.
hipError_t svgd_enqueue(SvgdWs &w, hipStream_t stream, const double *d_p, const double *d_g, int64_t N, int dim, double stepsize,
                        double h, double *d_new, double *d_dt, double *d_rep) {
    if (h < 0 && w.d_dist) {
        // distances once, lower middle order statistic by radix select over the stored values, upper one by one more pass
        const int64_t npairs = N * (N - 1) / 2;
        const int hblocks = (int)std::max<int64_t>(1, std::min<int64_t>(2048, (npairs + 255) / 256));
        const int digits[5] = {12, 13, 13, 13, 13};
        const int64_t T = (N + 63) / 64;
        if (dim == CRNN_CATHODE_NP)
            hipLaunchKernelGGL(crnn::svgd_pairdist_kernel<CRNN_CATHODE_NP>, dim3((unsigned)(T * (T + 1) / 2)), dim3(256), 0, stream, d_p, N, dim, w.d_dist);
        else
            hipLaunchKernelGGL(crnn::svgd_pairdist_kernel<0>, dim3((unsigned)(T * (T + 1) / 2)), dim3(256), 0, stream, d_p, N, dim, w.d_dist);
        hipLaunchKernelGGL(crnn::svgd_sel_init_kernel, dim3(1), dim3(1), 0, stream, w.d_sel, (long long)((npairs - 1) / 2));
        int shift = 64;
        for (int pass = 0; pass < 5; ++pass) {
            shift -= digits[pass];
            hipLaunchKernelGGL(crnn::svgd_hist_buf_kernel, dim3(hblocks), dim3(256), 0, stream, (const double *)w.d_dist, npairs, shift,
                               digits[pass], (const crnn::SvgdSel *)w.d_sel, w.d_hist);
            hipLaunchKernelGGL(crnn::svgd_pick_kernel, dim3(1), dim3(1024), 0, stream, w.d_hist, digits[pass], shift, w.d_sel, pass == 4 ? 0 : -1);
        }
        hipLaunchKernelGGL(crnn::svgd_next_init_kernel, dim3(1), dim3(1), 0, stream, w.d_cnt);
        hipLaunchKernelGGL(crnn::svgd_next_kernel, dim3(hblocks), dim3(256), 0, stream, (const double *)w.d_dist, npairs,
                           (const crnn::SvgdSel *)w.d_sel, w.d_cnt);
        hipLaunchKernelGGL(crnn::svgd_next_pick_kernel, dim3(1), dim3(1), 0, stream, w.d_sel, (const unsigned long long *)w.d_cnt, (long long)(npairs / 2));
    } else if (h < 0) {
        const int64_t npairs = N * (N - 1) / 2;
        const int hblocks = (int)std::max<int64_t>(1, std::min<int64_t>(2048, (npairs + 255) / 256));
        const int digits[5] = {12, 13, 13, 13, 13};
        const int64_t ranks[2] = {(npairs - 1) / 2, npairs / 2};
        for (int which = 0; which < 2; ++which) {
            hipLaunchKernelGGL(crnn::svgd_sel_init_kernel, dim3(1), dim3(1), 0, stream, w.d_sel, (long long)ranks[which]);
            int shift = 64;
            for (int pass = 0; pass < 5; ++pass) {
                shift -= digits[pass];
                hipLaunchKernelGGL(crnn::svgd_hist_dev_kernel, dim3(hblocks), dim3(256), 0, stream, d_p, N, dim, shift, digits[pass],
                                   (const crnn::SvgdSel *)w.d_sel, w.d_hist);
                hipLaunchKernelGGL(crnn::svgd_pick_kernel, dim3(1), dim3(1024), 0, stream, w.d_hist, digits[pass], shift, w.d_sel,
                                   pass == 4 ? which : -1);
            }
        }
    }
    hipLaunchKernelGGL(crnn::svgd_bandwidth_kernel, dim3(1), dim3(1), 0, stream, w.d_sel, std::log((double)N + 1.0), h);
    dim3 grid((unsigned)((N + 255) / 256), (unsigned)w.nchunk);
    if (dim == CRNN_CATHODE_NP)
        hipLaunchKernelGGL(crnn::svgd_rows_dim_kernel<CRNN_CATHODE_NP>, grid, dim3(256), 0, stream, d_p, d_g, N,
                           (const crnn::SvgdSel *)w.d_sel, w.nchunk, w.d_part);
    else
        hipLaunchKernelGGL(crnn::svgd_rows_dev_kernel, grid, dim3(256), 0, stream, d_p, d_g, N, dim, (const crnn::SvgdSel *)w.d_sel,
                           w.nchunk, w.d_part);
    hipLaunchKernelGGL(crnn::svgd_update_dev_kernel, dim3((unsigned)(((size_t)N * dim + 255) / 256)), dim3(256), 0, stream, d_p,
                       (const double *)w.d_part, N, dim, w.nchunk, (crnn::SvgdSel *)w.d_sel, stepsize / (double)N, d_new, d_dt, d_rep);
    return hipGetLastError();
}

std::mutex g_svgd_mutex;
SvgdWs g_svgd_ws[16];   // one cached workspace per device ordinal (crnn_svgd_update)
}  // namespace
extern "C" {

// stateless entry point: host arrays in and out; the device workspace of the previous call on this device is reused
int32_t crnn_svgd_update(int32_t device, const double *p, const double *lnpgrad, int64_t N, int32_t dim, double stepsize,
                         double h, double *p_new, double *h_out, double *data_term, double *repulsion) {
    if (!p || !lnpgrad || !p_new) return cfail(nullptr, "crnn_svgd_update: null pointer");
    if (N < 2 || dim < 1 || dim > crnn::kSvgdMaxDim) return cfail(nullptr, "crnn_svgd_update: need N >= 2 and 1 <= dim <= 32");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1) return cfail(nullptr, std::string("crnn_svgd_update: no HIP device (") + hipGetErrorString(e) + ")");
    if (device < 0 || device >= ndev || device >= 16) return cfail(nullptr, "crnn_svgd_update: device ordinal out of range");
    CHIP(nullptr, hipSetDevice(device));
    std::lock_guard<std::mutex> lock(g_svgd_mutex);
    SvgdWs &w = g_svgd_ws[device];
    CHIP(nullptr, svgd_ws_reserve(w, N, dim, true));
    const size_t nd = (size_t)N * dim;
    CHIP(nullptr, hipMemcpyAsync(w.d_p, p, nd * sizeof(double), hipMemcpyHostToDevice, nullptr));
    CHIP(nullptr, hipMemcpyAsync(w.d_g, lnpgrad, nd * sizeof(double), hipMemcpyHostToDevice, nullptr));
    CHIP(nullptr, svgd_enqueue(w, nullptr, w.d_p, w.d_g, N, dim, stepsize, h, w.d_new, w.d_dt, w.d_rep));
    crnn::SvgdSel sel{};
    CHIP(nullptr, hipMemcpyAsync(&sel, w.d_sel, sizeof(sel), hipMemcpyDeviceToHost, nullptr));
    CHIP(nullptr, hipMemcpyAsync(p_new, w.d_new, nd * sizeof(double), hipMemcpyDeviceToHost, nullptr));
    if (data_term) CHIP(nullptr, hipMemcpyAsync(data_term, w.d_dt, nd * sizeof(double), hipMemcpyDeviceToHost, nullptr));
    if (repulsion) CHIP(nullptr, hipMemcpyAsync(repulsion, w.d_rep, nd * sizeof(double), hipMemcpyDeviceToHost, nullptr));
    CHIP(nullptr, hipStreamSynchronize(nullptr));
    if (!(sel.h > 0)) return cfail(nullptr, "crnn_svgd_update: bandwidth h is not positive (all particles coincide?)");
    if (h_out) *h_out = sel.h;
    return 0;
}

}

// tests/simt/hip/hip_runtime.h -- TEST INFRASTRUCTURE, NOT PRODUCT.
//
// A SIMT emulator for the GPU-less build container: the kernel sources under crnn_amd/csrc (unchanged, not a line of them
// conditional on this file) are compiled as host C++ against this header in place of <hip/hip_runtime.h>, every GPU thread
// becomes a fibre, every wavefront a group of 64 fibres that meet at the cross-lane operations (DPP moves, __shfl*, ballot,
// readfirstlane, the FP64 MFMA, wave barriers) and every block meets at __syncthreads().  The result, tests/simt/libcrnn_simt.so,
// exports the same C ABI as libcrnn_hip.so, so the `-m gpu` parity tests can execute the kernels' own arithmetic and control flow
// against the oracle where no MI355X is reachable.  It proves what the source computes; it says nothing about time, registers,
// LDS banks or the ISA, and it is never loaded by crnn_amd unless a test points CRNN_HIP_LIB at it (crnn_build_info() says
// "SIMT-EMULATION"; bench.py and smoke() refuse such a library).
//
// Execution model
//   * a block runs on ONE OS thread; its lanes are cooperative fibres (hand-rolled x86-64 context switch), run in lane order
//     until each is blocked at a cross-lane operation or has returned.  Only then are operations resolved:
//       pull operations (DPP quad_perm, __shfl, __shfl_xor, __shfl_down): a lane is released when its source lane is blocked
//           at the SAME site with the same per-site sequence number (or has exited: "inactive lane" semantics);
//       wave operations (ballot, readfirstlane, mfma): released when every live lane of the wavefront is at that site;
//       soft barriers (wave_barrier, fences): released when nothing else in the wavefront can move;
//       __syncthreads: released when every live lane of the block is there.
//     If nothing can be released the run aborts with the lanes' positions (a divergent collective -- on the GPU that is a
//     partial EXEC mask, which this emulator deliberately does not guess at).
//   * blocks run one after the other (SIMT_THREADS=n: n OS threads take blocks from a counter; global atomics are atomic).
//   * __shared__ is `static thread_local`: one copy per OS thread = per running block.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <math.h>
#include <stdint.h>
#include <vector>
#include <sys/mman.h>

// ---------------------------------------------------------------- language surface
#define __global__
#define __device__
#define __host__
#define __constant__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define __align__(n) __attribute__((aligned(n)))
#define HIP_SYMBOL(x) x
// kernel-only attributes the host compiler rejects outright: `__attribute__((amdgpu_waves_per_eu(1, 2)))` -> `__attribute__(())`
#define amdgpu_waves_per_eu(...)
#define amdgpu_flat_work_group_size(...)
// the kernels' inline assembly is empty optimisation fences with AMDGPU register constraints: `asm volatile("" : "+v"(x))`
// -> `asm` disappears, `volatile( ... )` is swallowed by a function-like macro (the qualifier `volatile T` is not followed by
// a parenthesis and stays what it is)
#define asm
#define volatile(...)

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace simt {

extern "C" void simt_switch(void **save_sp, void *load_sp);
__asm__(R"(
.text
.globl simt_switch
.type simt_switch,@function
simt_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size simt_switch,.-simt_switch
)");

enum OpKind { OP_NONE = 0, OP_PULL, OP_BALLOT, OP_RFL, OP_MFMA, OP_SOFT, OP_SYNC };
enum LaneState { L_RUNNABLE = 0, L_BLOCKED, L_DONE };

struct Op {
    int kind = OP_NONE;
    int site = 0;
    int src = -1;              // pull: wave-local source lane
    unsigned seq = 0;          // per-site sequence number of this lane
    uint64_t payload = 0;      // pull value / ballot predicate / readfirstlane value
    uint64_t result = 0;
    int src_active = 0;        // pull: 1 if the source lane took part
    double a = 0, b = 0, c[4] = {0, 0, 0, 0}, d[4] = {0, 0, 0, 0};   // mfma
};

struct Lane {
    void *sp = nullptr;
    char *stack = nullptr;
    dim3 tid;
    int state = L_RUNNABLE;
    int index = 0;             // linear thread index in the block
    Op op;
    int prev_site = -1;        // site of the last operation this lane completed (a lane whose next one is the same site span a loop idle)
    long long round = 0;       // scheduler round in which the lane blocked
    std::vector<unsigned> seq; // per-site sequence numbers
};

struct BlockCtx {
    dim3 grid, bdim, bid;
    int nthreads = 0, nwaves = 0;
    std::vector<Lane> lanes;
    void *sched_sp = nullptr;
    long long round = 0;
    const std::function<void()> *body = nullptr;
};

inline thread_local BlockCtx *g_blk = nullptr;
inline thread_local Lane *g_lane = nullptr;
inline int g_nsites = 0;        // sites are numbered by __COUNTER__ at compile time; sized generously
constexpr int kMaxSites = 4096;
constexpr size_t kStack = 1u << 20;   // 1 MiB of address space per lane, committed on touch

inline bool trace_on() { static int t = getenv("SIMT_TRACE") ? atoi(getenv("SIMT_TRACE")) : 0; return t != 0; }

[[noreturn]] inline void die(const char *what) {
    BlockCtx *b = g_blk;
    fprintf(stderr, "[simt] FATAL: %s\n", what);
    if (b) {
        fprintf(stderr, "[simt] block (%u,%u,%u) of (%u,%u,%u), %d threads\n", b->bid.x, b->bid.y, b->bid.z, b->grid.x, b->grid.y, b->grid.z, b->nthreads);
        for (int w = 0; w < b->nwaves; ++w) {
            fprintf(stderr, "[simt]  wave %d:", w);
            for (int l = w * 64; l < std::min(b->nthreads, w * 64 + 64); ++l) {
                Lane &L = b->lanes[l];
                if (L.state == L_DONE) fprintf(stderr, " %d:done", l & 63);
                else fprintf(stderr, " %d:k%d@%d#%u>%d", l & 63, L.op.kind, L.op.site, L.op.seq, L.op.src);
            }
            fprintf(stderr, "\n");
        }
    }
    fflush(stderr);
    abort();
}

// ---- a lane blocks on an operation and yields to the block's scheduler
inline void block_on(int kind, int site) {
    Lane *L = g_lane;
    L->op.kind = kind;
    L->op.site = site;
    if (site >= 0 && site < kMaxSites) {
        if (L->seq.empty()) L->seq.assign(kMaxSites, 0u);
        L->op.seq = ++L->seq[site];
    }
    L->state = L_BLOCKED;
    L->round = g_blk->round;
    simt_switch(&L->sp, g_blk->sched_sp);
    g_lane->prev_site = site;
}

inline void fiber_entry() {
    Lane *L = g_lane;
    (*g_blk->body)();
    L = g_lane;
    L->state = L_DONE;
    simt_switch(&L->sp, g_blk->sched_sp);
    abort();   // a finished lane is never resumed
}

struct StackPool {
    std::vector<char *> stacks;
    char *get(size_t i) {
        while (stacks.size() <= i) {
            void *p = mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            if (p == MAP_FAILED) { perror("[simt] mmap"); abort(); }
            stacks.push_back((char *)p);
        }
        return stacks[i];
    }
    ~StackPool() { for (char *p : stacks) munmap(p, kStack); }
};
inline thread_local StackPool g_stacks;

// one wavefront-wide operation over the lanes `mem` (all blocked at the same kind and site): results into their ops
inline void wave_op_results(BlockCtx &b, int lo, const std::vector<int> &mem, int kind) {
    if (kind == OP_BALLOT) {
        uint64_t m = 0;
        for (int l : mem) if (b.lanes[l].op.payload) m |= 1ull << (l - lo);
        for (int l : mem) b.lanes[l].op.result = m;
    } else if (kind == OP_RFL) {
        const uint64_t v = b.lanes[mem[0]].op.payload;          // lowest-numbered participating lane
        for (int l : mem) b.lanes[l].op.result = v;
    } else if (kind == OP_MFMA) {
        // v_mfma_f64_16x16x4_f64: A[i][k] in lane 16 k + i, B[k][j] in lane 16 k + j, D[i][j] = register i / 4 of
        // lane 16 (i % 4) + j  (tools/ubench/mfma_f64_layout.hip measured this on the device).  The matrix unit ignores EXEC;
        // lanes that are not at the instruction contribute zeros here
        double A[16][4] = {}, B[4][16] = {};
        for (int l : mem) {
            const int q = l - lo;
            A[q % 16][q / 16] = b.lanes[l].op.a;
            B[q / 16][q % 16] = b.lanes[l].op.b;
        }
        for (int l : mem) {
            Lane &L = b.lanes[l];
            const int q = l - lo, jj = q % 16;
            for (int r = 0; r < 4; ++r) {
                const int ii = 4 * r + q / 16;
                double acc = L.op.c[r];
                for (int k = 0; k < 4; ++k) acc = std::fma(A[ii][k], B[k][jj], acc);
                L.op.d[r] = acc;
            }
        }
    }
}

inline std::atomic<long long> g_partial_releases{0};

// resolve what can be resolved; returns false if nothing moved
inline bool resolve(BlockCtx &b) {
    bool any = false;
    int live = 0, at_sync = 0;
    for (Lane &L : b.lanes) if (L.state != L_DONE) { ++live; if (L.op.kind == OP_SYNC) ++at_sync; }
    if (live > 0 && at_sync == live) {
        for (Lane &L : b.lanes) if (L.state == L_BLOCKED) { L.state = L_RUNNABLE; L.op.kind = OP_NONE; }
        return true;
    }
    for (int w = 0; w < b.nwaves; ++w) {
        const int lo = w * 64, hi = std::min(b.nthreads, lo + 64);
        int wlive = 0;
        for (int l = lo; l < hi; ++l) if (b.lanes[l].state != L_DONE) ++wlive;
        if (!wlive) continue;
        // 1. pull operations: two passes (values first, then release) so that a released source's payload is still read
        bool rel[64] = {}, moved = false;
        for (int l = lo; l < hi; ++l) {
            Lane &L = b.lanes[l];
            if (L.state != L_BLOCKED || L.op.kind != OP_PULL) continue;
            const int s = lo + L.op.src;
            if (L.op.src < 0 || s >= hi || b.lanes[s].state == L_DONE) { L.op.src_active = 0; L.op.result = 0; rel[l - lo] = true; continue; }
            Lane &S = b.lanes[s];
            if (S.state == L_BLOCKED && S.op.kind == OP_PULL && S.op.site == L.op.site) {
                if (S.op.seq != L.op.seq) die("pull operation: source lane is at another dynamic instance of the site (lanes of a group out of step)");
                L.op.src_active = 1; L.op.result = S.op.payload; rel[l - lo] = true;
            }
        }
        for (int l = lo; l < hi; ++l) if (rel[l - lo]) { b.lanes[l].state = L_RUNNABLE; b.lanes[l].op.kind = OP_NONE; moved = true; }
        if (moved) { any = true; continue; }
        // 2. wavefront-wide operations and soft barriers, grouped by (kind, site).  A group holding every live lane is the
        //    converged case.  Otherwise the wavefront is divergent and ONE group runs under its partial mask, as on the device:
        //    lanes that came round a loop without taking part in anything (their previous operation is the one they wait at
        //    again: the idle lanes of `while (ballot(busy)) { if (busy) {...} }`) wait for the others; then the group whose
        //    longest-waiting member blocked last (the lanes the device is executing right now, not those parked at a
        //    reconvergence point); then source order.
        struct Grp { int kind, site; std::vector<int> mem; bool spin; long long oldest; };
        std::vector<Grp> grps;
        for (int l = lo; l < hi; ++l) {
            Lane &L = b.lanes[l];
            if (L.state != L_BLOCKED) continue;
            const int k = L.op.kind;
            if (k != OP_BALLOT && k != OP_RFL && k != OP_MFMA && k != OP_SOFT) continue;
            Grp *g = nullptr;
            for (Grp &x : grps) if (x.kind == k && x.site == L.op.site) { g = &x; break; }
            if (!g) { grps.push_back(Grp{k, L.op.site, {}, true, L.round}); g = &grps.back(); }
            g->mem.push_back(l);
            g->spin = g->spin && (L.prev_site == L.op.site);
            g->oldest = std::min(g->oldest, L.round);
        }
        if (grps.empty()) continue;      // only unresolvable pulls and/or __syncthreads in this wavefront
        Grp *pick = nullptr;
        for (Grp &g : grps) if ((int)g.mem.size() == wlive) pick = &g;
        if (!pick) {
            // lanes of this wavefront parked at __syncthreads or at unresolvable pulls do not take part either way
            for (Grp &g : grps) {
                if (!pick) { pick = &g; continue; }
                if (g.spin != pick->spin) { if (!g.spin) pick = &g; continue; }
                if (g.oldest != pick->oldest) { if (g.oldest > pick->oldest) pick = &g; continue; }
                if (g.site < pick->site) pick = &g;
            }
            ++g_partial_releases;
            if (trace_on()) fprintf(stderr, "[simt] partial release: wave %d kind %d site %d (%zu of %d live lanes; %zu groups)\n", w, pick->kind, pick->site, pick->mem.size(), wlive, grps.size());
        }
        wave_op_results(b, lo, pick->mem, pick->kind);
        for (int l : pick->mem) { b.lanes[l].state = L_RUNNABLE; b.lanes[l].op.kind = OP_NONE; }
        any = true;
    }
    return any;
}

inline void run_block(BlockCtx &b) {
    g_blk = &b;
    for (int t = 0; t < b.nthreads; ++t) {
        Lane &L = b.lanes[t];
        L.index = t;
        L.tid = dim3(t % b.bdim.x, (t / b.bdim.x) % b.bdim.y, t / (b.bdim.x * b.bdim.y));
        L.state = L_RUNNABLE;
        L.op = Op();
        L.prev_site = -1;
        L.round = 0;
        if (!L.seq.empty()) std::fill(L.seq.begin(), L.seq.end(), 0u);
        L.stack = g_stacks.get(t);
        uint64_t *s = (uint64_t *)(L.stack + kStack);   // 16-aligned top
        *--s = 0;                                       // alignment slot: the entry's frame sees rsp % 16 == 8 as after a call
        *--s = (uint64_t)(void *)&fiber_entry;
        for (int i = 0; i < 6; ++i) *--s = 0;           // rbp rbx r12 r13 r14 r15
        L.sp = s;
    }
    for (;;) {
        int done = 0;
        for (int t = 0; t < b.nthreads; ++t) {
            Lane &L = b.lanes[t];
            if (L.state == L_RUNNABLE) { g_lane = &L; simt_switch(&b.sched_sp, L.sp); }
            if (L.state == L_DONE) ++done;
        }
        if (done == b.nthreads) break;
        ++b.round;
        if (!resolve(b)) die("deadlock: no cross-lane operation can be resolved");
    }
    g_lane = nullptr;
    g_blk = nullptr;
}

inline int env_threads() { static int n = getenv("SIMT_THREADS") ? std::max(1, atoi(getenv("SIMT_THREADS"))) : 1; return n; }
inline std::atomic<long long> g_launches{0}, g_blocks{0};

inline void launch(dim3 grid, dim3 block, const std::function<void()> &body) {
    const long long nblk = (long long)grid.x * grid.y * grid.z;
    const int nthreads = (int)(block.x * block.y * block.z);
    ++g_launches;
    g_blocks += nblk;
    std::atomic<long long> next{0};
    auto worker = [&]() {
        BlockCtx b;
        b.grid = grid; b.bdim = block; b.nthreads = nthreads; b.nwaves = (nthreads + 63) / 64; b.body = &body;
        b.lanes.resize(nthreads);
        for (;;) {
            const long long i = next.fetch_add(1);
            if (i >= nblk) break;
            b.bid = dim3((unsigned)(i % grid.x), (unsigned)((i / grid.x) % grid.y), (unsigned)(i / ((long long)grid.x * grid.y)));
            run_block(b);
        }
    };
    const int nt = (int)std::min<long long>(env_threads(), nblk);
    if (nt <= 1) { worker(); return; }
    std::vector<std::thread> th;
    for (int i = 0; i < nt; ++i) th.emplace_back(worker);
    for (auto &t : th) t.join();
}

// ---- cross-lane operations as the kernels call them
inline int lane_id() { return g_lane->index & 63; }

__attribute__((noinline)) inline uint64_t pull(int site, uint64_t v, int src, int *active) {
    Lane *L = g_lane;
    L->op.payload = v;
    L->op.src = src;
    block_on(OP_PULL, site);
    L = g_lane;
    if (active) *active = L->op.src_active;
    return L->op.result;
}
template <class T> inline T shfl_to(int site, T v, int src) {
    static_assert(sizeof(T) <= 8, "shuffle payload");
    uint64_t u = 0; memcpy(&u, &v, sizeof(T));
    int act; const uint64_t r = pull(site, u, src, &act);
    T out; memcpy(&out, &r, sizeof(T));
    return act ? out : T(0);     // ds_bpermute from a lane that did not take part reads zero
}
template <class T> inline T shfl(int site, T v, int src, int width = 64) {
    const int self = lane_id();
    return shfl_to(site, v, (src & (width - 1)) + (self & ~(width - 1)));
}
template <class T> inline T shfl_xor(int site, T v, int mask, int width = 64) {
    const int self = lane_id(); int idx = self ^ mask;
    if (idx >= ((self + width) & ~(width - 1))) idx = self;
    return shfl_to(site, v, idx);
}
template <class T> inline T shfl_down(int site, T v, unsigned delta, int width = 64) {
    const int self = lane_id(); int idx = self + (int)delta;
    if ((int)((self & (width - 1)) + delta) >= width) idx = self;
    return shfl_to(site, v, idx);
}
inline int update_dpp(int site, int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    if (ctrl < 0 || ctrl > 0xFF) die("update_dpp: only quad_perm controls are emulated");
    if ((row_mask & 0xF) != 0xF || (bank_mask & 0xF) != 0xF) die("update_dpp: row/bank masks are not emulated");
    const int self = lane_id(), from = (self & ~3) | ((ctrl >> (2 * (self & 3))) & 3);
    uint64_t u = (uint32_t)src; int act;
    const uint64_t r = pull(site, u, from, &act);
    return act ? (int)(uint32_t)r : (bound_ctrl ? 0 : old);
}
__attribute__((noinline)) inline uint64_t ballot(int site, bool p) {
    g_lane->op.payload = p ? 1 : 0;
    block_on(OP_BALLOT, site);
    return g_lane->op.result;
}
__attribute__((noinline)) inline uint64_t readfirstlane64(int site, uint64_t v) {
    g_lane->op.payload = v;
    block_on(OP_RFL, site);
    return g_lane->op.result;
}
template <class T> inline T readfirstlane(int site, T v) {
    static_assert(sizeof(T) <= 8, "readfirstlane payload");
    uint64_t u = 0; memcpy(&u, &v, sizeof(T));
    const uint64_t r = readfirstlane64(site, u);
    T out; memcpy(&out, &r, sizeof(T));
    return out;
}
inline void soft_barrier(int site) { block_on(OP_SOFT, site); }
inline void syncthreads(int site) { block_on(OP_SYNC, site); }

typedef double v4f64 __attribute__((ext_vector_type(4)));
__attribute__((noinline)) inline v4f64 mfma_f64_16x16x4(int site, double a, double b, v4f64 c) {
    Lane *L = g_lane;
    L->op.a = a; L->op.b = b;
    for (int r = 0; r < 4; ++r) L->op.c[r] = c[r];
    block_on(OP_MFMA, site);
    L = g_lane;
    v4f64 d;
    for (int r = 0; r < 4; ++r) d[r] = L->op.d[r];
    return d;
}

}   // namespace simt

#define threadIdx (simt::g_lane->tid)
#define blockIdx (simt::g_blk->bid)
#define blockDim (simt::g_blk->bdim)
#define gridDim (simt::g_blk->grid)
static constexpr int warpSize = 64;

#define SIMT_SITE (__COUNTER__)
#define __syncthreads() simt::syncthreads(SIMT_SITE)
#define __threadfence_block() simt::soft_barrier(SIMT_SITE)
#define __threadfence() simt::soft_barrier(SIMT_SITE)
#define __shfl(...) simt::shfl(SIMT_SITE, __VA_ARGS__)
#define __shfl_xor(...) simt::shfl_xor(SIMT_SITE, __VA_ARGS__)
#define __shfl_down(...) simt::shfl_down(SIMT_SITE, __VA_ARGS__)
#define __builtin_amdgcn_update_dpp(...) simt::update_dpp(SIMT_SITE, __VA_ARGS__)
#define __builtin_amdgcn_ballot_w64(p) simt::ballot(SIMT_SITE, (p))
#define __builtin_amdgcn_readfirstlane(v) simt::readfirstlane(SIMT_SITE, (v))
#define __builtin_amdgcn_wave_barrier() simt::soft_barrier(SIMT_SITE)
#define __builtin_amdgcn_fence(order, scope) simt::soft_barrier(SIMT_SITE)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, x, y, z) simt::mfma_f64_16x16x4(SIMT_SITE, (a), (b), (c))

// ---------------------------------------------------------------- device math the kernels call
inline double __builtin_amdgcn_frexp_mant(double x) { int e; return std::frexp(x, &e); }
inline int __builtin_amdgcn_frexp_exp(double x) { int e; std::frexp(x, &e); return e; }
inline double __builtin_amdgcn_ldexp(double x, int e) { return std::ldexp(x, e); }
#ifdef SIMT_ULP_NOISE
// SIMT_ULP_NOISE (SIMT_NOISE=1 tools/simt_suite.sh): what the device rounds differently from the host.  v_rcp_f64 / v_rsq_f64 are 1-ulp approximations and
// ocml's exp / log / pow are other implementations than glibc's: here their results move by one unit in the last place for two thirds of the
// arguments (a fixed function of the result's bits: +1, -1 or 0).  A parity bar that holds in the plain emulation only because host and oracle
// share a libm fails under this build -- before it fails at first contact with the device.  exp / log / pow reach this through the linker
// (-Wl,--wrap=...: every call inside the emulation library, and only those; the oracle keeps the host's libm).
namespace simt_noise {
inline double jig(double r) {
    if (!std::isfinite(r) || r == 0.0) return r;
    uint64_t b; memcpy(&b, &r, 8);
    static const uint64_t seed = getenv("SIMT_NOISE_SEED") ? strtoull(getenv("SIMT_NOISE_SEED"), nullptr, 10) * 0xD6E8FEB86659FD93ull : 0;   // another realisation of the noise
    uint64_t h = (b ^ seed) * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    const int m = (int)(h % 3);
    if (m == 1) b += 1; else if (m == 2) b -= 1;
    memcpy(&r, &b, 8); return r;
}
}
extern "C" double __real_exp(double); extern "C" double __real_log(double); extern "C" double __real_pow(double, double);
extern "C" __attribute__((used, visibility("default"))) inline double __wrap_exp(double x) { return simt_noise::jig(__real_exp(x)); }
extern "C" __attribute__((used, visibility("default"))) inline double __wrap_log(double x) { return simt_noise::jig(__real_log(x)); }
extern "C" __attribute__((used, visibility("default"))) inline double __wrap_pow(double x, double y) { return simt_noise::jig(__real_pow(x, y)); }
inline double __builtin_amdgcn_rcp(double x) { return simt_noise::jig(1.0 / x); }
inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
inline double __builtin_amdgcn_rsq(double x) { return simt_noise::jig(1.0 / std::sqrt(x)); }
#else
inline double __builtin_amdgcn_rcp(double x) { return 1.0 / x; }
inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
inline double __builtin_amdgcn_rsq(double x) { return 1.0 / std::sqrt(x); }
#endif
inline double __ocml_exp_f64(double x) { return std::exp(x); }
inline double __ocml_log_f64(double x) { return std::log(x); }
inline long long __double_as_longlong(double x) { long long r; memcpy(&r, &x, 8); return r; }
inline double __longlong_as_double(long long x) { double r; memcpy(&r, &x, 8); return r; }
inline int __double2hiint(double x) { return (int)(uint32_t)((uint64_t)__double_as_longlong(x) >> 32); }
inline int __double2loint(double x) { return (int)(uint32_t)((uint64_t)__double_as_longlong(x)); }
inline double __hiloint2double(int hi, int lo) { return __longlong_as_double((long long)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo)); }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
inline int __ffs(unsigned x) { return __builtin_ffs((int)x); }
inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
inline double rsqrt(double x) { return 1.0 / std::sqrt(x); }
inline double __dsqrt_rn(double x) { return std::sqrt(x); }
inline double __fma_rn(double a, double b, double c) { return std::fma(a, b, c); }
inline long long clock64() { return (long long)__builtin_readcyclecounter(); }
inline long long wall_clock64() { return (long long)__builtin_readcyclecounter(); }
// HIP's integer / floating min and max overloads in the global namespace
#define SIMT_MINMAX(T) inline T max(T a, T b) { return a > b ? a : b; } inline T min(T a, T b) { return a < b ? a : b; }
SIMT_MINMAX(int) SIMT_MINMAX(unsigned) SIMT_MINMAX(long) SIMT_MINMAX(unsigned long) SIMT_MINMAX(long long) SIMT_MINMAX(unsigned long long)
inline double max(double a, double b) { return std::fmax(a, b); }
inline double min(double a, double b) { return std::fmin(a, b); }
inline float max(float a, float b) { return std::fmax(a, b); }
inline float min(float a, float b) { return std::fmin(a, b); }
using std::abs; using std::exp; using std::fabs; using std::floor; using std::fma; using std::fmax; using std::fmin; using std::isfinite; using std::isnan;
using std::log; using std::pow; using std::sqrt; using std::ceil; using std::isinf; using std::copysign;

// ---------------------------------------------------------------- atomics (blocks may run on several OS threads)
inline int atomicAdd(int *p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline unsigned atomicAdd(unsigned *p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline long long atomicAdd(long long *p, long long v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline double atomicAdd(double *p, double v) {
    uint64_t *u = (uint64_t *)p, old = __atomic_load_n(u, __ATOMIC_RELAXED), neu;
    double o;
    do { memcpy(&o, &old, 8); const double n = o + v; memcpy(&neu, &n, 8); } while (!__atomic_compare_exchange_n(u, &old, neu, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST));
    return o;
}
inline float atomicAdd(float *p, float v) {
    uint32_t *u = (uint32_t *)p, old = __atomic_load_n(u, __ATOMIC_RELAXED), neu;
    float o;
    do { memcpy(&o, &old, 4); const float n = o + v; memcpy(&neu, &n, 4); } while (!__atomic_compare_exchange_n(u, &old, neu, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST));
    return o;
}
inline double unsafeAtomicAdd(double *p, double v) { return atomicAdd(p, v); }
inline float unsafeAtomicAdd(float *p, float v) { return atomicAdd(p, v); }
template <class T> inline T atomicMax(T *p, T v) { T old = __atomic_load_n(p, __ATOMIC_RELAXED); while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {} return old; }
template <class T> inline T atomicMin(T *p, T v) { T old = __atomic_load_n(p, __ATOMIC_RELAXED); while (old > v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {} return old; }
template <class T> inline T atomicExch(T *p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST); }
template <class T> inline T atomicOr(T *p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
template <class T> inline T atomicAnd(T *p, T v) { return __atomic_fetch_and(p, v, __ATOMIC_SEQ_CST); }
template <class T> inline T atomicCAS(T *p, T cmp, T v) { __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST); return cmp; }

// ---------------------------------------------------------------- runtime API (one synchronous "device" whose memory is the host's)
typedef enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNoDevice = 100, hipErrorUnknown = 999 } hipError_t;
typedef struct simtStream *hipStream_t;
struct simtEvent { std::chrono::steady_clock::time_point t; };
typedef simtEvent *hipEvent_t;
typedef enum { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 } hipMemcpyKind;
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocMapped = 2, hipHostMallocCoherent = 0x40000000, hipHostMallocDefault = 0 };
struct hipDeviceProp_t {
    char name[256];
    char gcnArchName[256];
    size_t totalGlobalMem;
    int multiProcessorCount, warpSize, maxThreadsPerBlock;
    size_t sharedMemPerBlock;
};
inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "simt error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidValue; }
inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
inline int simt_cus() { static int n = getenv("SIMT_CUS") ? std::max(1, atoi(getenv("SIMT_CUS"))) : 2; return n; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
    memset(p, 0, sizeof(*p));
    snprintf(p->name, sizeof(p->name), "SIMT-EMULATION of gfx950 (host fibres)");
    snprintf(p->gcnArchName, sizeof(p->gcnArchName), "gfx950:simt");
    p->totalGlobalMem = (size_t)8 << 30; p->multiProcessorCount = simt_cus(); p->warpSize = 64; p->maxThreadsPerBlock = 1024;
    p->sharedMemPerBlock = 160 * 1024;
    return hipSuccess;
}
inline hipError_t hipMemGetInfo(size_t *fr, size_t *tot) { *fr = (size_t)6 << 30; *tot = (size_t)8 << 30; return hipSuccess; }
inline hipError_t hipMalloc(void **p, size_t n) { *p = nullptr; if (posix_memalign(p, 256, n ? n : 1)) return hipErrorOutOfMemory; memset(*p, 0xCD, n); return hipSuccess; }
template <class T> inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void **)p, n); }
inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { *p = nullptr; if (posix_memalign(p, 256, n ? n : 1)) return hipErrorOutOfMemory; memset(*p, 0, n); return hipSuccess; }
template <class T> inline hipError_t hipHostMalloc(T **p, size_t n, unsigned f = 0) { return hipHostMalloc((void **)p, n, f); }
inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned) { *d = h; return hipSuccess; }
template <class T> inline hipError_t hipHostGetDevicePointer(T **d, void *h, unsigned f) { return hipHostGetDevicePointer((void **)d, h, f); }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { if (n) memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t = nullptr) {
    for (size_t r = 0; r < h; ++r) memmove((char *)d + r * dp, (const char *)s + r * sp, w);
    return hipSuccess;
}
inline hipError_t hipMemset(void *d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = nullptr) { if (n) memset(d, v, n); return hipSuccess; }
template <class S> inline hipError_t hipMemcpyFromSymbol(void *d, const S &sym, size_t n, size_t off = 0, hipMemcpyKind = hipMemcpyDeviceToHost) { memcpy(d, (const char *)&sym + off, n); return hipSuccess; }
template <class S> inline hipError_t hipMemcpyToSymbol(S &sym, const void *s, size_t n, size_t off = 0, hipMemcpyKind = hipMemcpyHostToDevice) { memcpy((char *)&sym + off, s, n); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new simtEvent(); return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new simtEvent(); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
// what a CU can hold is not modelled: the launch code sizes its grids from this, one block per CU keeps emulated grids small
inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, const void *, int, size_t) { *n = 1; return hipSuccess; }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    simt::launch(dim3(grid), dim3(block), [&]() { kernel(__VA_ARGS__); })

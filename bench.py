#!/usr/bin/env python3
"""bench.py -- headline benchmark of the CRNN neural-ODE hot path on MI355X.

Metric (BASELINE.json): stiff neural-ODE trajectories+gradients per second,
case2 (6 species + T, 3 reactions, P = 25), 65 536 random initial conditions
per GPU, fp64, Rosenbrock23 at the reference tolerances (atol 1e-6, rtol 1e-3).

One "step" = one pass of the hot path over the rank's batch, with the ensemble
already resident in HBM: solve + loss + gradient kernel (65 536 trajectories;
--grad adjoint: forward sweep + reversed accepted steps, one lane per trajectory;
--grad forward: P tangent columns on lane groups) -> fixed-order gradient
reduction + chain rule through p2vec -> all-reduce of the (P+6)-vector over ranks
(RCCL) -> Flux-style ExpDecay/ADAM/WeightDecay update of p on the device (the same
kernel forms p2vec of the new p).  --scaling weak (default): every rank owns its own 65 536 ICs; --scaling strong: 65 536
ICs in total, 65 536 / N per rank.  With N > 1 both are timed and the other one is reported under "other_scaling".

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the field meanings).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

LB_CASE1, LB_CASE2 = float(np.float32(1e-5)), float(np.float32(1e-6))   # `lb = 1.f-5` / `lb = 1.f-6`: Float32 literals (case1/case1.jl:34, case2/case2.jl:34)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Algorithmic HBM bytes per trajectory+gradient (case2, no pred output) -- SURVEY 8(d)'s per-unit figure:
#   8 * [ n (u0) + n_obs*D (data) + 1 (loss) + 1 (retcode, n_saved as 2 x int32) ] = 2472 B.
# Implementation traffic on top of it (per-trajectory gradient rows, step counters, the adjoint's step tape) is NOT
# counted here; it shows up in roofline.traffic.
BYTES_PER_TRAJ = 8 * (7 + 6 * 50 + 1 + 1)
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6     # MI355X FP64 vector peak (spec); 2 flop per FMA
# FP64 operation model of the kernel (DESIGN.md "flop model"), 2 flop per operation (upper bound: mul/add count as FMA):
FLOP_PRIMAL_STEP = 2 * 1480      # one Rosenbrock23 attempt: 12 log, 6 exp, 12 rcp, J, 6x6 LU, 3 solves, error norm, controller
FLOP_COL_STEP = 2 * 441          # one tangent column through one accepted step
# adjoint kernel: v_fma/v_mul/v_add_f64 instructions per loop body in the gfx950 ISA (tools/isa_blocks.py), x2
FLOP_ADJ_ATTEMPT = 2 * 808       # forward sweep, one Rosenbrock23 attempt
FLOP_ADJ_REVERSE = 2 * 719       # reverse sweep, one accepted step (re-formation 310 + adjoint 409)
FLOP_ADJ_SAVE = 2 * 40           # loss + seeds of one save point
# two-lanes-per-trajectory kernel (ros23_adj2_kernel<6,3,T>; tools/kisa.py): per LANE 602 / 530 / 20 such instructions per
# attempt / reverse step / save point, two lanes per trajectory (the replicated part -- LU of the 3x3 matrix, controller,
# exponentials -- is executed by both and counted for both)
FLOP_ADJ2_ATTEMPT = 2 * 2 * 602
FLOP_ADJ2_REVERSE = 2 * 2 * 530
FLOP_ADJ2_SAVE = 2 * 2 * 20


def usable_cores():
    """host cores this process may actually use: min(affinity mask, cgroup CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_reference_algorithm_rate(orc, pb_kwargs, p, u0_s, ts, data_s, cores, seconds):
    """trajectories+gradients/s of the C oracle running what the reference's CPU path executes per `ForwardDiff.gradient` call on case2
    (case2/case2.jl:26,195; which algorithm and which norm: tests/test_case2_stream_pin.py): Tsit5 -- the AutoTsit5 composite never leaves it
    on this model --, ForwardDiff's chunks of 9 + 9 + 7 partials, every chunk its own adaptive solve with the chunk's partials in the error norm
    / totallength(u).  OpenMP over trajectories (the stand-in for EnsembleThreads()); bounded to `seconds` of wall time."""
    from crnn_amd.api import fd_chunk_size
    pb = orc.make_problem(**dict(pb_kwargs, solver=1, errnorm_sens=2))
    th, dth = orc.p2vec(2, 6, 3, p)
    P = dth.shape[1]
    chunk = fd_chunk_size(P)
    cols = []
    for k0 in range(0, P, chunk):
        c = np.zeros((dth.shape[0], chunk), order="F")          # the Dual carries `chunk` partials; the last chunk's surplus ones are zero
        c[:, :min(P, k0 + chunk) - k0] = dth[:, k0:k0 + chunk]
        cols.append(c)
    ns_ = u0_s.shape[1]
    for c in cols:                                              # warm-up
        orc.solve_batch(pb, th, u0_s[:, :256].copy(), ts, data_s[:, :, :256].copy(), dtheta=c, nthreads=cores)
    tc, passes = 0.0, 0
    while tc < seconds and passes < 64:
        t1 = time.perf_counter()
        for c in cols:
            orc.solve_batch(pb, th, u0_s, ts, data_s, dtheta=c, nthreads=cores)
        tc += time.perf_counter() - t1
        passes += 1
    return ns_ * passes / tc, passes, tc


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=65536, help="initial conditions per GPU")
    ap.add_argument("--comm", choices=["rccl", "torch"], default="rccl")
    ap.add_argument("--cols", type=int, default=0, help="tangent columns per lane (0 = library default)")
    ap.add_argument("--theta0", choices=["ckpt", "init"], default="ckpt",
                    help="start from the reference's trained checkpoint p or a reference-style random init")
    ap.add_argument("--solver", choices=["rosenbrock23", "tsit5", "autotsit5"], default="rosenbrock23",
                    help="time stepper; the headline metric is quoted on the Rosenbrock23-equivalent stepper")
    ap.add_argument("--grad", choices=["auto", "forward", "adjoint"], default="auto",
                    help="gradient algorithm: discrete adjoint of the accepted steps (auto) or forward tangents")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default): --batch ICs PER RANK; strong: --batch ICs in total, split over the ranks.  Whichever is "
                         "chosen, the other mode is timed too (same K, W) and reported under 'other_scaling' when N > 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary figures (other configs / regimes, N = 1 only)")
    ap.add_argument("--secondary-seconds", type=float, default=420.0, help="wall-clock limit of the child process that measures the secondary figures")
    ap.add_argument("--cpu-sample", type=int, default=65536)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall time to spend on the CPU baseline")
    return ap.parse_args()


def secondary_in_child(cmd, seconds):
    """Runs `cmd --out <file>` with a wall-clock limit and returns the dictionary it left in <file>: complete, or -- child crashed, was killed by
    a signal, outlived the limit -- the entries it had finished plus "_incomplete": why.  Never raises: the headline line does not depend on it."""
    import subprocess
    import tempfile
    status = "ok"
    sec_path = None
    try:
        with tempfile.NamedTemporaryFile(prefix="crnn_bench_secondary_", suffix=".json", delete=False) as tf:
            sec_path = tf.name
        try:
            rc_ = subprocess.run(cmd + ["--out", sec_path], timeout=seconds, stdout=sys.stderr, stderr=sys.stderr).returncode
            if rc_ != 0:
                status = f"child exited with code {rc_}"
        except subprocess.TimeoutExpired:
            status = f"child stopped after --secondary-seconds = {seconds:.0f} s"
        except Exception as e:  # noqa: BLE001
            status = f"child could not be run: {e}"
        try:
            sec = json.load(open(sec_path))
            if not isinstance(sec, dict):
                sec = {}
        except Exception:  # noqa: BLE001
            sec = {}
    except Exception as e:  # noqa: BLE001
        sec, status = {}, f"no temporary file: {e}"
    finally:
        for pth in (sec_path, (sec_path or "") + ".tmp"):
            try:
                if pth:
                    os.unlink(pth)
            except OSError:
                pass
    if status != "ok":
        print(f"[bench] secondary: {status}; {len(sec)} entries kept", file=sys.stderr, flush=True)
        sec["_incomplete"] = status
    return sec


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    import crnn_amd
    from crnn_amd import _lib as _L
    if "SIMT-EMULATION" in _L.lib.crnn_build_info().decode():      # (tests/simt: the kernels' sources on host fibres -- a checker, never a measurement)
        raise SystemExit("bench.py: CRNN_HIP_LIB points at the SIMT emulation library; a bench line needs the gfx950 library on an MI355X")
    from crnn_amd import NeuralODE, ODEProblem, Optimiser, PRESET_CASE2, SOLVER_AUTOTSIT5, SOLVER_ROSENBROCK23, SOLVER_TSIT5, cases
    from crnn_amd._lib import check, dptr, lib
    from crnn_amd.dist import DataParallel

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); no CPU fallback exists")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    B = args.batch
    ts = cases.case2_tsteps()
    rng = np.random.Generator(np.random.PCG64([1234, rank]))
    u0 = cases.case2_u0(B, rng)

    # ---- synthetic ensemble, mirroring case2/case2.jl:62-83: true mechanism (an exact CRNN)
    #      integrated at tight tolerance by the same gfx950 stepper, 5 % multiplicative noise ----
    gen = NeuralODE(ODEProblem(PRESET_CASE2, ts, atol=1e-10, rtol=1e-8, device=local_rank))
    clean = gen.predict_theta(u0, cases.case2_true_theta())[:, :6, :]      # [B, 6, 50]
    gen.close()
    data = cases.add_noise(clean, 0.05, rng)
    yscale = cases.max_min(data, lb=LB_CASE2)
    if world > 1:  # yscale is a global statistic of the data set (case2.jl:83)
        t = torch.tensor(yscale, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        yscale = t.cpu().numpy()

    solver = {"rosenbrock23": SOLVER_ROSENBROCK23, "tsit5": SOLVER_TSIT5, "autotsit5": SOLVER_AUTOTSIT5}[args.solver]
    gmode = {"auto": 0, "forward": 1, "adjoint": 2}[args.grad]
    adjoint = gmode != 1                      # every stepper has a discrete-adjoint kernel; forward tangents on request
    ros = solver == SOLVER_ROSENBROCK23       # the flop model and the PMC traffic figure below are those of the headline kernels
    node = NeuralODE(ODEProblem(PRESET_CASE2, ts, device=local_rank, cols_per_lane=args.cols, solver=solver, grad_mode=gmode))
    node.set_ensemble(u0, data, yscale)          # one PCIe upload; resident in HBM from here on
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures.json")))
    p0 = np.array(fx["case2_ckpt"]["p"]) if args.theta0 == "ckpt" else cases.case2_init_p(np.random.Generator(np.random.PCG64(7)))
    node.train_init(Optimiser(25, PRESET_CASE2), p0)

    # The in-library ncclAllReduce (crnn_comm_init) is the default; if its self-test does not pass on this node the
    # run switches -- loudly -- to torch.distributed's all_reduce (the same RCCL) on the library's device buffer.
    comm = args.comm
    dp = None
    try:
        dp = DataParallel(node, comm=comm)
        ok = dp.selftest()
    except Exception as e:  # noqa: BLE001
        print(f"[bench] rank {rank}: comm={comm} failed to initialise: {e}", file=sys.stderr, flush=True)
        ok = False
    if world > 1:
        flag = torch.tensor([1.0 if ok else 0.0], device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = bool(flag.item() > 0.5)
    if not ok:
        if comm == "torch":
            raise SystemExit(f"rank {rank}: all-reduce self-test failed on comm=torch")
        print(f"[bench] rank {rank}: comm=rccl self-test failed, falling back to comm=torch", file=sys.stderr, flush=True)
        comm = "torch"
        dp = DataParallel(node, comm=comm)
        if not dp.selftest():
            raise SystemExit(f"rank {rank}: all-reduce self-test failed on comm=torch")

    def sync_all():
        check(lib.crnn_synchronize(node.handle), node.handle)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Per-rank share of a step.  weak: the rank's whole resident ensemble (B ICs per rank, global batch N * B);
    # strong: the GLOBAL batch is B -- BASELINE's "case2 batch 65k, 1/2/4/8 MI355X" read literally -- and every rank takes
    # B / N of it (the first B / N of its resident ICs: the ranks' ensembles are drawn from the same distribution).
    def timed(count):
        for _ in range(args.warmup):
            dp.train_step(0, count)
        sync_all()
        t0_ = time.perf_counter()
        for _ in range(args.steps):
            dp.train_step(0, count)
        sync_all()
        el = time.perf_counter() - t0_
        if world > 1:
            te = torch.tensor([el], device="cuda", dtype=torch.float64)
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            el = float(te.item())
        return el

    count_of = {"weak": B, "strong": max(1, B // world)}
    other = "strong" if args.scaling == "weak" else "weak"
    B_rank = count_of[args.scaling]
    elapsed = timed(B_rank)       # the headline mode FIRST: same p, optimiser state and queue history as an N = 1 run

    # ---- per-launch kernel durations over the timed region (HIP events on the ctx stream) ----
    nk = min(args.steps, 64)
    kms = np.zeros(nk)
    check(lib.crnn_kernel_times(node.handle, dptr(kms), nk), node.handle)
    st = node.stats()          # last step: n_traj, n_ok, n_accept, n_reject
    lanes_used = node.last_lanes_per_traj() if adjoint and ros else 0
    p_now = node.params()

    other_line = None
    if world > 1:      # the mode that is not the headline of this run, afterwards, from the same starting point (p0, fresh optimiser)
        node.train_init(Optimiser(25, PRESET_CASE2), p0)
        el_o = timed(count_of[other])
        other_line = {"scaling": other, "batch_per_gpu": count_of[other], "global_batch": count_of[other] * world,
                      "ms_per_step": el_o / args.steps * 1e3, "value": world * count_of[other] * args.steps / el_o,
                      "unit": "trajectories+grads/s"}

    out = None
    if rank == 0:
        k_ms = float(kms.mean())
        value = world * B_rank * args.steps / elapsed
        ach_gbs = BYTES_PER_TRAJ * B_rank / (k_ms * 1e-3) / 1e9
        if not ros:
            flops = None   # no instruction-count model for the Tsit5 / composite kernels
        elif adjoint:
            fa, fr, fs = ((FLOP_ADJ2_ATTEMPT, FLOP_ADJ2_REVERSE, FLOP_ADJ2_SAVE) if lanes_used == 2 else
                          (FLOP_ADJ_ATTEMPT, FLOP_ADJ_REVERSE, FLOP_ADJ_SAVE))
            flops = (st["n_accept"] + st["n_reject"]) * fa + st["n_accept"] * fr + st["n_traj"] * len(ts) * fs
        else:
            flops = (st["n_accept"] + st["n_reject"]) * FLOP_PRIMAL_STEP + st["n_accept"] * 25 * FLOP_COL_STEP
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if ros and os.path.exists(tpath):   # HBM bytes per launch from separate rocprofv3 --pmc passes (profiles/README.md):
            try:                            # only for the launch shapes the passes were taken on, else null
                tj = json.load(open(tpath))
                if B_rank == 65536 and (not adjoint or lanes_used == 1):
                    traffic = tj.get("case2_B65536_adjoint_bytes_per_launch" if adjoint else "case2_B65536_bytes_per_launch")
                elif adjoint and lanes_used == 2:
                    traffic = tj.get("secondary_bytes_per_launch", {}).get(f"case2_B{B_rank}_lane_pair")
            except Exception:
                traffic = None
        out = {
            "metric": "stiff neural-ODE trajectories+grads/sec, case2 batch 65k",
            "value": value, "unit": "trajectories+grads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "case2: 6 species + T, 3 reactions, P=25, D=50 save points on [0,50], "
                                   f"{ {'tsit5': 'Tsit5', 'autotsit5': 'AutoTsit5(Rosenbrock23)', 'rosenbrock23': 'Rosenbrock23'}[args.solver]} atol 1e-6 rtol 1e-3, MAE loss, "
                                   f"{'discrete-adjoint' if adjoint else 'forward-tangent'} gradient of the accepted Rosenbrock23 steps, dt held fixed (= ForwardDiff's derivative of THOSE steps: 6e-4 from the converged sensitivity at the reference's tolerances, where the reference's own Tsit5 sits at 3e-6 -- profiles/r04o), "
                                   "ExpDecay+ADAM+WeightDecay update",
                       "batch_per_gpu": B_rank, "global_batch": B_rank * world, "theta0": args.theta0,
                       "comm": comm if world > 1 else "none", "parallelism": f"dp{world} (ICs sharded, 1 all-reduce/step)"},
            "roofline": {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach_gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_measured_on": (None if traffic is None else
                                                 "profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an EARLIER library (round 3 / round 4, "
                                                 "sources named there); the tape records this kernel writes and reads are unchanged since, the figure is not "
                                                 "re-measured in this run"),
                         # traffic: PMC figure of the committed rocprofv3 passes (profiles/traffic.json; counters cannot be read
                         # inside this process).  traffic_model: the same quantity from THIS run's step counts -- algorithmic bytes +
                         # the adjoint's step tape, 64 B per accepted step written at 1.27x its payload (lane-strided partial
                         # lines, profiles/r02_fetch_write_calibration.txt) and read back once.
                         "traffic_model": (BYTES_PER_TRAJ * B_rank + st["n_accept"] * 64 * (1.27 + 1.0)) if (ros and adjoint) else None,
                         "kernel": (((("ros23_adj2_kernel<6,3,T> (two lanes per trajectory)" if lanes_used == 2 else "ros23_adj_kernel<6,3,T>")) if ros
                                     else "auto_adj_kernel<6,3,T>") if adjoint else
                                    ("ros23_kernel" if ros else "tsit5_kernel") + "<6,3,T,C,L>"), "kernel_ms": k_ms,
                         "algorithmic_bytes_per_launch": BYTES_PER_TRAJ * B_rank,
                         "lanes_per_traj": lanes_used,
                         "note": "state lives in VGPR/LDS for the whole integration; the path is instruction-issue bound "
                                 "(one wavefront per SIMD), see valu_fp64 (SURVEY F8)"},
            # flop count and duration of the LAST timed launch (step counts drift slightly as p is updated)
            "valu_fp64": {"achieved": None if flops is None else flops / (kms[-1] * 1e-3) / 1e12, "peak": FP64_VALU_PEAK_TFLOPS,
                          "unit": "TFLOP/s", "frac": None if flops is None else flops / (kms[-1] * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                          "steps_per_traj": st["n_accept"] / max(st["n_traj"], 1),
                          "rejects_per_traj": st["n_reject"] / max(st["n_traj"], 1),
                          "n_ok": st["n_ok"], "n_traj": st["n_traj"]},
        }
        if other_line is not None:
            out["other_scaling"] = other_line
        # ---- CPU baseline: the C oracle ("port", OpenMP over trajectories) on a bounded sample ----
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle as orc
            ns_ = min(args.cpu_sample, B)
            th, dth = orc.p2vec(2, 6, 3, p0)
            pb_kwargs = dict(ns=6, nr=3, has_temp=1, lb=LB_CASE2, ub=10.0, inv_R=cases.INV_R, atol=1e-6, rtol=1e-3, yscale=yscale, clamp_pred=1)
            pb = orc.make_problem(**pb_kwargs, solver={"rosenbrock23": 0, "tsit5": 1, "autotsit5": 2}[args.solver])
            u0_s = np.ascontiguousarray(u0[:ns_].T)
            data_s = np.ascontiguousarray(data[:ns_].transpose(2, 1, 0))
            cores = usable_cores()

            def cpu_rate(grad_adjoint, seconds):
                pb.grad_adjoint = grad_adjoint
                orc.solve_batch(pb, th, u0_s[:, :256].copy(), ts, data_s[:, :, :256].copy(), dtheta=dth, nthreads=cores)  # warm-up
                # repeat the sample until >= `seconds` of wall time have been spent (bounded CPU work, stable rate)
                tc, passes = 0.0, 0
                while tc < seconds and passes < 64:
                    t1 = time.perf_counter()
                    orc.solve_batch(pb, th, u0_s, ts, data_s, dtheta=dth, nthreads=cores)
                    tc += time.perf_counter() - t1
                    passes += 1
                return ns_ * passes / tc, passes, tc
            # the SAME algorithm as the GPU (discrete adjoint of the accepted steps; Rosenbrock23 only) and ForwardDiff's
            # arithmetic (forward tangents, 1 + 25 columns: what the reference's CPU path executes), half of the time each
            have_adj = args.solver == "rosenbrock23"
            share = args.cpu_seconds / (3 if have_adj else 2)
            v_fwd, n_fwd, t_fwd = cpu_rate(0, share)
            v_adj, n_adj, t_adj = cpu_rate(1, share) if have_adj else (None, 0, 0.0)
            try:      # a third share: the algorithm the reference's own run evaluated (a sample of 8 192 ICs: three chunk solves each)
                nr_ = min(ns_, 8192)
                v_ref, n_ref, t_ref = cpu_reference_algorithm_rate(orc, pb_kwargs, p0, u0_s[:, :nr_].copy(), ts, data_s[:, :, :nr_].copy(), cores, share)
            except Exception as e:  # noqa: BLE001  (an extra figure must not take the baseline along)
                v_ref, n_ref, t_ref = None, 0, 0.0
                print(f"[bench] cpu_baseline: reference-algorithm figure failed: {e}", file=sys.stderr, flush=True)
            out["cpu_baseline"] = {"value": v_adj if have_adj else v_fwd, "unit": "trajectories+grads/s", "cores": cores, "kind": "port",
                                   "algorithm": ("discrete adjoint of the accepted steps -- the algorithm the GPU kernel runs" if have_adj else
                                                 "forward tangents, 1 + 25 columns per trajectory"),
                                   "forward_tangents_value": v_fwd,
                                   "reference_algorithm_value": v_ref,
                                   "reference_algorithm": "Tsit5 (case2's AutoTsit5(Rosenbrock23) never leaves it), ForwardDiff's chunks 9 + 9 + 7, every chunk its own "
                                                          "adaptive solve with the partials in the error norm / totallength(u): what the reference's recorded training "
                                                          f"history reproduces with (tests/test_case2_stream_pin.py); {n_ref} passes over 8 192 ICs, {t_ref:.1f} s wall",
                                   "sample": f"the first {ns_} ICs of the same ensemble at the same p, solve+loss+gradient, C oracle with OpenMP over "
                                             f"trajectories on {cores} cores: {n_adj} passes with the discrete adjoint ({t_adj:.1f} s wall) -> value; {n_fwd} "
                                             f"passes with forward tangents, 1 + 25 columns = ForwardDiff's arithmetic, what the reference's CPU path "
                                             f"executes ({t_fwd:.1f} s wall) -> forward_tangents_value; a C restatement, not DifferentialEquations.jl "
                                             f"(Julia absent)"}
        # ---- secondary figures (N = 1): the same hot path at FIXED parameters on the other regimes / BASELINE configs ----
        # In a CHILD process with a wall-clock limit (tools/bench_secondary.py --all; it redraws rank 0's ensemble from the same seed): the
        # headline line above is printed whatever happens there -- an entry that throws is recorded as {"error": ...} by the child, a child
        # that crashes or outlives the limit leaves the entries it had finished (it rewrites its output file after each one).
        if world == 1 and not args.no_secondary:
            out["secondary"] = secondary_in_child([sys.executable, os.path.join(ROOT, "tools", "bench_secondary.py"), "--all", "--device", str(local_rank),
                                                   "--batch", str(B)], args.secondary_seconds)
        # RCCL prints a version banner through C stdio (block-buffered on a pipe): flush it first so
        # that the JSON line is the LAST line of stdout.
        C.CDLL(None).fflush(None)
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    dp.close()
    node.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

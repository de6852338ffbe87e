"""Secondary measured figures for bench.py's JSON line (rank 0, N = 1 only): the same hot path at fixed parameters on the
other BASELINE configurations and regimes.  Every entry: kernel time = median HIP-event duration of the solve kernel over
`reps` loss+gradient calls (crnn_stats.kernel_ms), value = trajectories+gradients per second of that kernel, `roofline` =
algorithmic HBM bytes per trajectory x trajectories / kernel time against 8 TB/s, plus the step statistics
the kernel reports; `primal_kernel_ms` = the same median for the primal launch (losses only: the epoch-end evaluation loop) at the
same p.  Ensembles are synthetic and seeded; nothing here reads /root/reference.
"""
import json
import os
import time

import numpy as np

LB_CASE1, LB_CASE2 = float(np.float32(1e-5)), float(np.float32(1e-6))   # `lb = 1.f-5` / `lb = 1.f-6`: Float32 literals (case1/case1.jl:34, case2/case2.jl:34)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBS = 8000.0
# 8 * [n (u0) + n_obs * D (data) + loss + (retcode, n_saved)] bytes per trajectory (+ 8 P for per-trajectory gradients)
BYTES = {"case2": 8 * (7 + 6 * 50 + 2), "robertson": 8 * (3 + 3 * 40 + 2), "hychem": 8 * (9 + 9 * 40 + 2),
         "cathode": 8 * (3 + 17 + 64 + 2 + 17)}


def _traffic(key):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/traffic.json: secondary_bytes_per_launch), or None."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["secondary_bytes_per_launch"].get(key)
    except Exception:
        return None


def _entry(kind, B, kms, st, extra=None, wall_ms=None, traffic_key=None):
    k = float(np.median(kms))
    gbs = BYTES[kind] * B / (k * 1e-3) / 1e9
    e = {"trajectories": int(B), "kernel_ms": k, "value": B / (k * 1e-3), "unit": "trajectories+grads/s",
         "steps_per_traj": st["n_accept"] / max(st["n_traj"], 1), "rejects_per_traj": st["n_reject"] / max(st["n_traj"], 1),
         "n_ok": int(st["n_ok"]),
         "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                      "algorithmic_bytes_per_traj": BYTES[kind], "traffic": _traffic(traffic_key) if traffic_key else None}}
    if wall_ms is not None:
        e["call_ms"] = wall_ms
    if extra:
        e.update(extra)
    return e


def _med(x):
    """median, or None for an empty list (an entry whose extras failed must still serialise as strict JSON)"""
    return float(np.median(x)) if len(x) else None


def _time_calls(node, p, reps):
    kms, walls = [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        node.loss_and_grad(p)
        walls.append((time.perf_counter() - t0) * 1e3)
        kms.append(node.last_stats["kernel_ms"])
    return kms[1:], float(np.median(walls[1:])), node.last_stats


def _primal_ms(node, p, reps=4):
    """Median kernel time of the primal launch (crnn_solve with no directions: the epoch-end loss loop, case2.jl:199-203)."""
    kms = []
    for _ in range(reps):
        node.losses(p)
        kms.append(node.last_stats["kernel_ms"])
    return float(np.median(kms[1:]))


def case2_fixed(u0, data, yscale, p, label_extra=None, reps=6, device=0, lanes=None, **probkw):
    """case2 at a fixed p (no optimiser update) on a caller-supplied ensemble.  lanes: crnn_ctx_set_lanes_per_traj (None = AUTO)."""
    from crnn_amd import NeuralODE, ODEProblem, PRESET_CASE2, cases
    node = NeuralODE(ODEProblem(PRESET_CASE2, cases.case2_tsteps(), device=device, **probkw))
    node.set_ensemble(u0, data, yscale)
    if lanes is not None:
        node.set_lanes_per_traj(lanes)
    kms, wall, st = _time_calls(node, p, reps)
    e = _entry("case2", u0.shape[0], kms, st, label_extra, wall)
    e["lanes_per_traj"] = node.last_lanes_per_traj()
    if not probkw.get("errnorm_sens"):
        e["primal_kernel_ms"] = _primal_ms(node, p)
    if e["lanes_per_traj"] == 2 and u0.shape[0] in (8192, 32768):
        e["roofline"]["traffic"] = _traffic(f"case2_B{u0.shape[0]}_lane_pair")
    node.close()
    return e


def case2_ensemble(B, seed, device=0):
    from crnn_amd import NeuralODE, ODEProblem, PRESET_CASE2, cases
    rng = np.random.Generator(np.random.PCG64(seed))
    ts = cases.case2_tsteps()
    u0 = cases.case2_u0(B, rng)
    gen = NeuralODE(ODEProblem(PRESET_CASE2, ts, atol=1e-10, rtol=1e-8, device=device))
    clean = gen.predict_theta(u0, cases.case2_true_theta())[:, :6, :]
    gen.close()
    data = cases.add_noise(clean, 0.05, rng)
    return u0, data, cases.max_min(data, lb=LB_CASE2)


def robertson(B=65536, reps=6, device=0):
    """BASELINE config 3: robertson CRNN (3 species, 6 reactions, P = 43, stiffness 1e11), checkpoint p."""
    from crnn_amd import NeuralODE, ODEProblem, PRESET_ROBER, cases
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures.json")))
    rng = np.random.Generator(np.random.PCG64([1234, 3]))
    ts = cases.rober_tsteps()
    u0 = cases.rober_u0(B, rng)
    rb = fx["robertson"]
    ys, sc = np.array(rb["yscale"]), np.array(rb["dydt_scale"])
    gen = NeuralODE(ODEProblem(PRESET_ROBER, ts, rate_scale=sc, device=device))
    p = np.array(fx["rober_ckpt"]["p"])
    clean = gen.predict_neuralode(u0, p)                      # data: the checkpoint CRNN itself + the reference's 1e-4 noise
    gen.close()
    data = cases.add_noise(clean, 1e-4, rng)
    node = NeuralODE(ODEProblem(PRESET_ROBER, ts, rate_scale=sc, device=device))
    node.set_ensemble(u0, data, ys)
    kms, wall, st = _time_calls(node, p, reps)
    prim = _primal_ms(node, p)
    node.close()
    return _entry("robertson", B, kms, st, {"primal_kernel_ms": prim, "workload": "robertson CRNN, 65 536 ICs, Rosenbrock23 atol [1e-6,1e-8,1e-6] rtol 1e-3, adjoint gradient (P = 43)",
                                            "kernel": "ros23_adj_kernel<3,6,scaled>"}, wall, traffic_key="robertson_B65536" if B == 65536 else None)


def hychem(B=32768, reps=4, device=0, unproven=False):
    """BASELINE config 4, one GPU's share (262 144 / 8): HyChem pyrolysis CRNN, 9 species, 10 reactions, P = 211."""
    from crnn_amd import NeuralODE, ODEProblem, PRESET_HYCHEM, hychem as hy
    rng = np.random.Generator(np.random.PCG64([1234, 4]))
    ts, u0, Tt, Pt = hy.sample_conditions(B, rng)
    node = NeuralODE(ODEProblem(PRESET_HYCHEM, ts, rate_scale=hy.DYDT_SCALE, device=device))
    node.set_ensemble(u0, np.zeros((B, 9, len(ts))), np.ones(9))
    node.set_tables(Tt, Pt)
    clean = node.predict_n_ode(hy.true_p())
    data = clean * (1.0 + 0.01 * rng.standard_normal(clean.shape))
    ys = np.maximum((data.max(axis=2) - data.min(axis=2)).max(axis=0), hy.LB)
    node.set_ensemble(u0, data, ys)
    node.set_tables(Tt, Pt)
    p = hy.true_p() + 0.02 * np.random.Generator(np.random.PCG64(5)).standard_normal(hy.NP)
    p[-1] = 0.1
    kms, wall, st = _time_calls(node, p, reps)
    prim = _primal_ms(node, p, 3)
    node.close()
    extra4 = {}
    if B == 32768:      # round 4: the reference's composite for primal launches; the gradient as ForwardDiff evaluates it (1 024 of the ICs)
      try:
        if not unproven:
            from crnn_amd import SOLVER_AUTOTSIT5
            comp = NeuralODE(ODEProblem(PRESET_HYCHEM, ts, rate_scale=hy.DYDT_SCALE, device=device, solver=SOLVER_AUTOTSIT5))
            comp.set_ensemble(u0, data, ys); comp.set_tables(Tt, Pt)
            extra4["primal_autotsit5_kernel_ms"] = _primal_ms(comp, p, 3)
            extra4["primal_autotsit5_steps_per_traj"] = comp.last_stats["n_accept"] / B
            comp.close()
        if unproven:
                # the reference-faithful gradient (errnorm_sens = 2).  Two kernels (crnn_capi.hip launch_hychem_sens_chunk): the dense-direction
                # hychem_sens_kernel is the library's default until a device session has passed the sparse kernel's parity tests; round 5's
                # hychem_sens2_kernel (sparse directions) is asked for with CRNN_HY_SENS_KERNEL=2 (read at crnn_ctx_create).  1 024 ICs -- the size
                # round 4's nested-dual kernel was quoted on (522.8 ms, 1 959 /s: profiles/r04i) -- for both, the whole share for the sparse one
                import os
                for kern, sizes in (("dense", (1024,)), ("sparse", (1024, B))):
                    for n in sizes:
                        old_env = os.environ.get("CRNN_HY_SENS_KERNEL")
                        os.environ["CRNN_HY_SENS_KERNEL"] = "2" if kern == "sparse" else "1"
                        try:
                            sens = NeuralODE(ODEProblem(PRESET_HYCHEM, ts, rate_scale=hy.DYDT_SCALE, device=device, errnorm_sens=2))
                        finally:
                            if old_env is None:
                                os.environ.pop("CRNN_HY_SENS_KERNEL", None)
                            else:
                                os.environ["CRNN_HY_SENS_KERNEL"] = old_env
                        sens.set_ensemble(u0[:n], data[:n], ys); sens.set_tables(Tt[:n], Pt[:n])
                        sens.loss_and_grad(p)
                        t0 = time.perf_counter(); sens.loss_and_grad(p); w = (time.perf_counter() - t0) * 1e3
                        sens.close()
                        extra4[f"errnorm_sens2_{kern}_B{n}_call_ms"] = w
                        extra4[f"errnorm_sens2_{kern}_B{n}_value"] = n / (w * 1e-3)
                extra4["errnorm_sens2_value"] = extra4[f"errnorm_sens2_sparse_B{B}_value"]
                extra4["errnorm_sens_note"] = ("crnn_config.errnorm_sens = 2 on the HyChem preset: ForwardDiff's 18 chunks of 12 partials, each its own "
                                               "adaptive solve with the partials in the error norm + the plain solve; wall time of one loss+gradient call. "
                                               "dense = hychem_sens_kernel (library default), sparse = hychem_sens2_kernel (CRNN_HY_SENS_KERNEL=2: sparse "
                                               "directions, closed-form tangents, one column per lane)")
      except Exception as e:  # noqa: BLE001  (these kernels are round 4 / 5 additions: their failure must not take the adjoint figures of the entry along)
        extra4["extras_error"] = f"{type(e).__name__}: {e}"[:500]
    return _entry("hychem", B, kms, st, {**extra4, "primal_kernel_ms": prim, "workload": "HyChem pyrolysis CRNN, 32 768 ICs (one GPU's share of 262 144), T(t)/P(t) tables, "
                                                     "Rosenbrock23 atol 1e-8 rtol 1e-3, adjoint gradient (P = 211)",
                                         "kernel": "hychem2_kernel<9,10,GRAD,256> (a lane pair per trajectory, W's rows in registers, LDS frame, "
                                                   "gradient summed over each batch of 32 by v_mfma_f64_16x16x4: no accumulator in HBM)"}, wall,
                  traffic_key="hychem_B32768_lane_pair" if B == 32768 else None)


def cathode(n_part=4096, n_rates=256, reps=3, device=0, unproven=False):
    """BASELINE config 5 on one GPU: 4 096 particles x 256 heating rates, per-particle 17-parameter gradients."""
    from crnn_amd.cathode import CathodeUQ
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures_cathode.json")))
    betas = np.exp(np.linspace(np.log(2.0), np.log(20.0), n_rates))
    meas = np.array([s["beta"] for s in fx["sets"]])
    exp_data = []
    for b in betas:
        s = fx["sets"][int(np.argmin(np.abs(np.log(meas) - np.log(b))))]
        dbar, d2bar = np.array(s["dbar"]), np.array(s["d2bar"])
        sd = np.sqrt(np.maximum(d2bar - dbar ** 2, 0.0))
        exp_data.append(np.stack([np.array(s["ts"]) * s["beta"] / b, dbar + sd, dbar - sd], axis=1))
    uq = CathodeUQ(exp_data, betas, fx["theta"], normalizer=np.ones((n_rates, 3)), device=device, errnorm_sens=0)      # the primal-norm adjoint (opt-in since round 5)
    rng = np.random.default_rng(0)
    p = 1 + 1e-3 * rng.standard_normal((n_part, 17))
    p[:, 6:9] = 0.0
    kms = []
    for _ in range(reps):
        uq.solve(p)
        kms.append(uq.last_stats["kernel_ms"])
    st = uq.last_stats
    pk = []
    for _ in range(3):
        uq.solve(p, want_grad=False)
        pk.append(uq.last_stats["kernel_ms"])
    # round 4: primal launches through the reference's composite (network.jl:195); the gradient as ForwardDiff evaluates it
    comp_ms = {"autotsit5_trbdf2": (None, None), "autotsit5_rosenbrock23": (None, None)}
    sens_wall = adj_wall = None
    sens_chunks = ((0, 0), (0, 0))
    sv, so = [], []
    extras_error = None
    try:
        for name in (() if unproven else ("autotsit5_trbdf2", "autotsit5_rosenbrock23")):
            uq.set_solver(name)
            ck = []
            for _ in range(3):
                uq.solve(p, want_grad=False)
                ck.append(uq.last_stats["kernel_ms"])
            comp_ms[name] = (float(np.median(ck[1:])), uq.last_stats["n_accept"] / uq.last_stats["n_traj"])
        uq.set_solver("rosenbrock23")
        t0 = time.perf_counter(); uq.solve(p); adj_wall = (time.perf_counter() - t0) * 1e3
        if unproven:      # cathode_sens_kernel in its round-5 form (stage tangents parked in 135 KB of LDS): no device has launched it
            sens = CathodeUQ(exp_data, betas, fx["theta"], normalizer=np.ones((n_rates, 3)), device=device, errnorm_sens=2)
            sens.solve(p)
            t0 = time.perf_counter(); sens.solve(p); sens_wall = (time.perf_counter() - t0) * 1e3
            sens_chunks = sens.last_chunk_stats()
            sens.close()
            raise StopIteration        # (the device-resident SVGD figures belong to the proven pass)
        # the reference's own iteration (crnn_cathode.jl:36-50): ONE heating rate per SVGD move, particles resident on the device --
        # solve of n_part trajectories + chain rule + exact-median select + kernel sums + update, enqueued back to back
        uq.set_particles(p)
        sv, so = [], []
        for it in range(8):
            _, _, ms = uq.svgd_step((37 * it) % n_rates, 1e-3)
            sv.append(ms["svgd_ms"]); so.append(ms["solve_ms"])
        uq.close()
    except StopIteration:
        uq.close()
    except Exception as e:  # noqa: BLE001  (composites, dual-norm chunks, device-resident SVGD: must not take the adjoint figures of the entry along)
        extras_error = f"{type(e).__name__}: {e}"[:500]
        try:
            uq.close()
        except Exception:  # noqa: BLE001
            pass
    return _entry("cathode", n_part * n_rates, kms[1:], st,
                  {"workload": "Cathode-UQ: 4 096 particles x 256 heating rates, non-autonomous Rosenbrock23 atol 1e-12 rtol 1e-3, "
                               "per-particle adjoint gradients (17 parameters each)", "kernel": "cathode_adj_kernel<256,1> (full tape, two wavefronts per SIMD, accumulators in LDS)",
                   "primal_kernel_ms": float(np.median(pk[1:])),
                   "primal_autotsit5_trbdf2_kernel_ms": comp_ms["autotsit5_trbdf2"][0], "primal_autotsit5_trbdf2_steps_per_traj": comp_ms["autotsit5_trbdf2"][1],
                   "primal_autotsit5_rosenbrock23_kernel_ms": comp_ms["autotsit5_rosenbrock23"][0],
                   "composite_note": "crnn_cathode_set_solver: primal launches through AutoTsit5(TRBDF2) (the reference's alg, network.jl:195) / "
                                     "AutoTsit5(Rosenbrock23): a third of Rosenbrock23's accepted steps, six right-hand sides per step instead of two",
                   "errnorm_sens2_call_ms": sens_wall, "adjoint_call_ms": adj_wall,
                   "errnorm_sens2_steps_per_traj_chunks": [sens_chunks[0][0] / (n_part * n_rates), sens_chunks[1][0] / (n_part * n_rates)],
                   "errnorm_sens_note": "crnn_cathode_set_errnorm_sens(2): ForwardDiff's chunks 9 + 8, each its own adaptive solve with the partials in "
                                        "the error norm (cathode_sens_kernel) + the plain solve; wall time of one gradient call incl. the read-back, next "
                                        "to the adjoint call's",
                   "extras_error": extras_error,
                   "svgd_move_ms": _med(sv[2:]), "svgd_iteration_solve_ms": _med(so[2:]),
                   "svgd_note": "device-resident SVGD iteration (crnn_cathode_svgd_step): svgd_iteration_solve_ms = the solve kernel "
                                "over the 4 096 particles of ONE heating rate, svgd_move_ms = median select + kernel sums + move "
                                "(HIP events on the ctx stream)"}, traffic_key="cathode_4096x256_full_tape" if (n_part, n_rates) == (4096, 256) else None)

def run_all(u0, data, yscale, device=0, out_path=None):
    """Every secondary entry of bench.py's line, each on its own: an entry that throws becomes {"error": ...} and the next one runs; the
    dictionary is rewritten to out_path after each entry, so a caller that has to stop this process keeps what was finished."""
    import sys
    import traceback
    from crnn_amd import NeuralODE, ODEProblem, Optimiser, PRESET_CASE2, SOLVER_TSIT5, cases
    sec = {}
    B = u0.shape[0]
    local_rank = device
    ts = cases.case2_tsteps()
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures.json")))

    def progress(msg):
        print(f"[bench] secondary: {msg}", file=sys.stderr, flush=True)

    def flush():
        if out_path:
            with open(out_path + ".tmp", "w") as f:
                json.dump(sec, f)
            os.replace(out_path + ".tmp", out_path)

    def put(key, fn):
        try:
            sec[key] = fn()
        except Exception as e:  # noqa: BLE001
            sec[key] = {"error": f"{type(e).__name__}: {e}"[:500]}
            print(f"[bench] secondary: {key} FAILED: {e}", file=sys.stderr, flush=True)
            traceback.print_exc(file=sys.stderr)
        flush()

    ck = np.array(fx["case2_ckpt"]["p"])
    p_init = cases.case2_init_p(np.random.Generator(np.random.PCG64(7)))
    p_hard = np.array(json.load(open(os.path.join(ROOT, "tests", "golden", "case2_hard_p.json")))["p"])
    note = "case2, 65 536 ICs of the headline ensemble, Rosenbrock23 atol 1e-6 rtol 1e-3, adjoint gradient, fixed p: "
    progress("case2 at fixed p (init / early training / diverged / errnorm_sens)")
    put("case2_reference_init_p", lambda: case2_fixed(u0, data, yscale, p_init, {"workload": note + "the reference's random initialiser (case2.jl:85-89)"}))
    put("case2_early_training_p", lambda: case2_fixed(u0, data, yscale, p_hard, {
        "workload": note + "p after epoch 2 of a reference-schedule training run from that initialiser -- the hardest state a healthy run "
                           "visits (tests/golden/case2_hard_p.json, tools/train_case2_converge.py)"}))
    # 30 FULL-BATCH ADAM steps from the initialiser (what --theta0 init times): after four such updates the
    # network sits in a sliding mode on the kink of log(clamp(u, lb, ub)) -- about 10 % of the trajectories alternate
    # accepted and rejected steps thousands of times and their tangents overflow (1e175), ADAM's second moment swallows
    # the update and training stalls at loss 0.237.  The CPU restatement reproduces all of it step for step; it is a
    # diverged training state of this schedule (the reference updates per experiment), timed here for the record.
    def diverged_p():
        nd = NeuralODE(ODEProblem(PRESET_CASE2, ts, device=local_rank))
        nd.set_ensemble(u0, data, yscale)
        nd.train_init(Optimiser(25, PRESET_CASE2), p_init)
        for _ in range(30):
            nd.train_step(want_loss=False)
        p_ = nd.params()
        nd.close()
        return p_
    try:
        p_deg = diverged_p()
    except Exception as e:  # noqa: BLE001
        p_deg = None
        sec["case2_after_30_full_batch_adam_steps_from_init"] = {"error": f"{type(e).__name__}: {e}"[:500]}
        flush()
    if p_deg is not None:
      put("case2_after_30_full_batch_adam_steps_from_init", lambda: case2_fixed(u0, data, yscale, p_deg, {
        "workload": note + "p after 30 full-batch ADAM steps from the initialiser: a diverged (sliding-mode) state, launch time = the longest "
                           "trajectory's thousands of attempts; see DESIGN.md"}, reps=3))
    put("case2_errnorm_sens1", lambda: case2_fixed(u0, data, yscale, ck, {
        "workload": note + "errnorm_sens = 1 (ForwardDiff's dual-inclusive error norm, chunks 9 + 9 + 7, forward tangents through every "
                           "attempt) + the plain solve for the loss: the reference-faithful gradient mode; kernel_ms is the LAST launch only, "
                           "call_ms the whole loss+gradient call"}, reps=3, errnorm_sens=1))
    if "error" not in sec.get("case2_errnorm_sens1", {"error": 1}):
        sec["case2_errnorm_sens1"]["value"] = B / (sec["case2_errnorm_sens1"]["call_ms"] * 1e-3)
    progress("case2 strong-scaling shares (8 192 / 16 384 / 32 768 of the 65 536)")
    for nb in (8192, 16384, 32768):
        put(f"case2_B{nb}_share", lambda: case2_fixed(u0[:nb], data[:nb], yscale, ck, {
            "workload": f"case2, {nb} ICs = one GPU's share of the 65 536 batch on {65536 // nb} GPUs (strong scaling), checkpoint p, adjoint gradient; "
                        "AUTO takes the lane-pair kernel (ros23_adj2_kernel) below 32 769 trajectories"}))
        put(f"case2_B{nb}_share_one_lane", lambda: case2_fixed(u0[:nb], data[:nb], yscale, ck, {
            "workload": f"the same with one lane per trajectory (crnn_ctx_set_lanes_per_traj(1): round 2's kernel)"}, lanes=1))
    put("case2_B65536_two_lanes", lambda: case2_fixed(u0, data, yscale, ck, {
        "workload": note + "checkpoint p, TWO lanes per trajectory forced (two generations of pairs, longest first)"}, lanes=2))
    progress("case2 B = 131072 / 262144")
    try:
        ub_, db_, yb_ = case2_ensemble(262144, [1234, 99], device=local_rank)
    except Exception as e:  # noqa: BLE001
        ub_ = None
        sec["case2_B262144"] = {"error": f"{type(e).__name__}: {e}"[:500]}
        flush()
    if ub_ is not None:
        for nb in (131072, 262144):
            put(f"case2_B{nb}", lambda: case2_fixed(ub_[:nb], db_[:nb], yb_, ck, {
                "workload": f"case2, {nb} ICs on one GPU (more than the 65 536 resident lanes: queued by the previous launch's step counts), "
                            "checkpoint p, adjoint gradient"}))
        del ub_, db_
    progress("robertson")
    put("robertson_B65536", lambda: robertson(device=local_rank))
    progress("hychem 32768")
    put("hychem_B32768", lambda: hychem(device=local_rank))
    progress("hychem 262144")
    put("hychem_B262144_one_gpu", lambda: hychem(B=262144, reps=3, device=local_rank))
    if "error" not in sec["hychem_B262144_one_gpu"]:
      sec["hychem_B262144_one_gpu"]["workload"] = ("HyChem pyrolysis CRNN, ALL 262 144 ICs of BASELINE config 4 on ONE GPU (eight generations of "
                                                 "wavefronts, queued by the previous launch's step counts), adjoint gradient (P = 211)")
    progress("cathode 4096 x 256")
    put("cathode_4096x256", lambda: cathode(device=local_rank))
    # LAST: kernels no device has executed in their shipping form (rounds 5 / 6 had none).  Each in an entry of its own, after everything a driver has
    # seen before: if one of them hangs, the parent's wall-clock limit ends this process and every figure above is already on file
    progress("unproven kernels, last: tsit5_sens_kernel (79 KB build)")
    # the same through Tsit5 -- the branch of case2's AutoTsit5(Rosenbrock23) the reference stays in (tsit5_sens_kernel; round 5: 79 KB of
    # LDS per block instead of 100, two blocks per CU)
    put("case2_errnorm_sens1_tsit5", lambda: case2_fixed(u0, data, yscale, ck, {
        "workload": note.replace("Rosenbrock23", "Tsit5") + "errnorm_sens = 1 as above, explicit Tsit5 (case2's reference algorithm while it stays "
                                                            "non-stiff; case1's Tsit5()): tsit5_sens_kernel"}, reps=3, errnorm_sens=1, solver=SOLVER_TSIT5))
    if "error" not in sec.get("case2_errnorm_sens1_tsit5", {"error": 1}):
        sec["case2_errnorm_sens1_tsit5"]["value"] = B / (sec["case2_errnorm_sens1_tsit5"]["call_ms"] * 1e-3)
    # what the reference's recorded case2 history was computed with (tests/test_case2_stream_pin.py): Tsit5 -- its AutoTsit5(Rosenbrock23) never
    # switches -- and the dual norm divided by totallength(u)
    put("case2_reference_gradient_tsit5_errnorm_sens2", lambda: case2_fixed(u0, data, yscale, ck, {
        "workload": note.replace("Rosenbrock23", "Tsit5") + "the gradient exactly as the reference's training run evaluated it: ForwardDiff's chunks 9 + 9 + 7, "
                    "every chunk its own adaptive Tsit5 solve with the partials in the error norm / totallength(u) (errnorm_sens = 2; the mode that "
                    "reproduces the recorded training history), + the plain solve for the loss: tsit5_sens_kernel; call_ms is the whole call"},
        reps=3, errnorm_sens=2, solver=SOLVER_TSIT5))
    if "error" not in sec.get("case2_reference_gradient_tsit5_errnorm_sens2", {"error": 1}):
        sec["case2_reference_gradient_tsit5_errnorm_sens2"]["value"] = B / (sec["case2_reference_gradient_tsit5_errnorm_sens2"]["call_ms"] * 1e-3)
    progress("unproven kernels: HyChem dual-norm gradient (dense default / sparse opt-in)")
    put("hychem_B32768_dual_norm", lambda: hychem(reps=2, device=local_rank, unproven=True))
    progress("unproven kernels: cathode dual-norm gradient (135 KB-LDS build)")
    put("cathode_4096x256_dual_norm", lambda: cathode(reps=2, device=local_rank, unproven=True))
    progress("done")
    flush()
    return sec


if __name__ == "__main__":
    import argparse
    import sys
    sys.path.insert(0, ROOT)
    ap = argparse.ArgumentParser(description="the secondary figures of bench.py's line, as its child process (or by hand)")
    ap.add_argument("--all", action="store_true")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from crnn_amd import _lib as _L
    if "SIMT-EMULATION" in _L.lib.crnn_build_info().decode():
        raise SystemExit("bench_secondary.py: CRNN_HIP_LIB points at the SIMT emulation library; timings need the gfx950 library on an MI355X")
    u0_, data_, ys_ = case2_ensemble(a.batch, [1234, 0], device=a.device)      # rank 0's ensemble of bench.py, from the same seed
    res = run_all(u0_, data_, ys_, device=a.device, out_path=a.out)
    if not a.out:
        print(json.dumps(res))

#!/usr/bin/env python3
"""Cross-build bit-identity check of the gradient kernels.

A kernel whose results move when the optimisation level, an unrelated macro or a printf changes has undefined behaviour
somewhere (an uninitialised value read on a divergent path, an out-of-range LDS / tape index on an inactive lane, a
convergent operation after a divergent loop tail) -- or sits on a miscompile.  The library is compiled with
`-ffp-contract=on` (crnn_amd/csrc/Makefile): a*b+c is fused where the SOURCE says so (one expression / an explicit fma),
never across statements at the optimiser's discretion, so correct code gives the same bits at every optimisation level.
This tool builds the library several ways and asserts exactly that:

    python tools/crossbuild.py build           # here or on the GPU box: crnn_amd/csrc/dbg/libcrnn_xb_<variant>.so
    python tools/crossbuild.py run [--json f]  # GPU: every problem through every build, all outputs compared bit for bit
    python tools/crossbuild.py worker          # (internal: one build, prints digests)

Variants: O3 (the shipped flags), O2, O1, O3 + -DCRNN_ADJ_PROF (phase timers in the adjoint kernels: different register
allocation and schedule, same arithmetic), O3 + -DCRNN_BOUNDS_CHECK (every indexed access of the adjoint kernels checked
against its extent -- the stand-in for a device address sanitizer, whose instrumented runtime this image lacks; the build
must also report ZERO violations).  Problems: the AutoTsit5(Rosenbrock23) composite on robertson (switches
algorithm mid-run), Tsit5 adjoint on case1 and on case2, Rosenbrock23 adjoint on case2 (one lane and two lanes per trajectory),
case1's shape (two lanes, odd species count), robertson, HyChem (one lane and a lane pair),
forward tangents on case2; round 4: the HyChem primal launches (pair kernel, AutoTsit5 composite) and the cathode kernels (adjoint with
the full and the checkpointed tapes, forward tangents, primal, both composites, the chunked dual-norm gradient) -- the bounds checks
cover those kernel families too -- per-trajectory losses, return codes, saved counts, accepted / rejected steps and the batch
gradient in index order (crnn_ctx_set_queue_order(INDEX): the batch sum is then a function of the inputs alone).
tests/test_gpu_crossbuild.py runs `run` on every GPU session.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "crnn_amd", "csrc")
DBG = os.path.join(CSRC, "dbg")
VARIANTS = {"O3": "-O3", "O2": "-O2", "O1": "-O1", "O3prof": "-O3 -DCRNN_ADJ_PROF", "O3chk": "-O3 -DCRNN_BOUNDS_CHECK"}
BASE = "-std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Wno-unused-result"
LINK = "-shared -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib"


def lib_path(name):
    return os.path.join(DBG, f"libcrnn_xb_{name}.so")


def source_hash():
    h = hashlib.sha256()
    for f in sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))):
        h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "crnn_hip.h"), "rb").read())
    h.update(open(os.path.join(CSRC, "Makefile"), "rb").read())      # as crnn_amd/_lib.py source_hash()
    return h.hexdigest()[:16]


def build(names=None, force=False):
    """Builds the variants (in parallel) unless a build of the current sources is already there."""
    os.makedirs(DBG, exist_ok=True)
    want = source_hash()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    procs = []
    for name in (names or VARIANTS):
        out, side = lib_path(name), lib_path(name) + ".srchash"
        if not force and os.path.exists(out) and os.path.exists(side) and open(side).read().strip() == want:
            continue
        cmd = f"{hipcc} {VARIANTS[name]} {BASE} -DCRNN_SRC_HASH='\"{want}\"' -o {out} {CSRC}/crnn_capi.hip {LINK} && echo {want} > {side}"
        procs.append((name, subprocess.Popen(cmd, shell=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for name, pr in procs:
        out = pr.communicate()[0]
        if pr.returncode != 0:
            raise RuntimeError(f"build of variant {name} failed:\n{out[-4000:]}")
    return [lib_path(n) for n in (names or VARIANTS)]


# ------------------------------------------------------------------------------------------------------------ worker
def _problems():
    import numpy as np
    sys.path.insert(0, ROOT)
    from crnn_amd import NeuralODE, ODEProblem, PRESET_CASE1, PRESET_CASE2, PRESET_ROBER, cases
    from crnn_amd import _lib as L
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures.json")))
    rng = np.random.Generator(np.random.PCG64([77, 1]))
    out = []

    def case2(B):
        ts = cases.case2_tsteps()
        u0 = cases.case2_u0(B, rng)
        data = np.abs(rng.standard_normal((B, 6, len(ts)))) * 0.5
        return ts, u0, data, cases.max_min(data, lb=cases.LB_CASE2), np.array(fx["case2_ckpt"]["p"])

    def rober(B):
        ts = cases.rober_tsteps()
        u0 = cases.rober_u0(B, rng)
        ys = np.array(fx["robertson"]["yscale"])
        data = np.abs(rng.standard_normal((B, 3, len(ts)))) * ys[None, :, None]
        return ts, u0, data, ys, np.array(fx["rober_ckpt"]["p"])

    def case1(B):
        ts = cases.case1_tsteps()
        u0 = cases.case1_u0(B, rng)
        data = np.abs(rng.standard_normal((B, 5, len(ts)))) * 0.5
        return ts, u0, data, cases.max_min(data, lb=cases.LB_CASE1), np.array(fx["case1"]["p"])

    B = 1024 + 37    # ragged: the last wavefront is partly empty
    c2, rb, c1 = case2(B), rober(B), case1(B)
    sc = np.array(fx["robertson"]["dydt_scale"])
    specs = [
        ("rober_autotsit5_adjoint", PRESET_ROBER, rb, dict(rate_scale=sc, solver=L.SOLVER_AUTOTSIT5, grad_mode=2)),
        ("case1_tsit5_adjoint", PRESET_CASE1, c1, dict(solver=L.SOLVER_TSIT5, grad_mode=2)),
        ("case1_autotsit5_adjoint", PRESET_CASE1, c1, dict(solver=L.SOLVER_AUTOTSIT5, grad_mode=2)),
        ("case2_tsit5_adjoint", PRESET_CASE2, c2, dict(solver=L.SOLVER_TSIT5, grad_mode=2)),
        ("case2_ros23_adjoint", PRESET_CASE2, c2, dict(grad_mode=2, lanes=1)),
        ("case2_ros23_adjoint_2lanes", PRESET_CASE2, c2, dict(grad_mode=2, lanes=2)),
        ("case1_ros23_adjoint_2lanes", PRESET_CASE1, c1, dict(solver=L.SOLVER_ROSENBROCK23, grad_mode=2, lanes=2)),
        ("rober_ros23_adjoint", PRESET_ROBER, rb, dict(rate_scale=sc, grad_mode=2)),
        ("case2_ros23_forward", PRESET_CASE2, c2, dict(grad_mode=1)),
        ("case2_ros23_primal", PRESET_CASE2, c2, dict(primal=True)),       # ros23_adj_kernel<..., PRIMAL>: predictions + losses, no directions
    ]
    # HyChem (its own kernels: one lane and a lane pair per trajectory), 300 trajectories, 211 parameters
    from crnn_amd import PRESET_HYCHEM, hychem as hy
    hrng = np.random.Generator(np.random.PCG64([77, 4]))
    hts, hu0, hT, hP = hy.sample_conditions(300, hrng)
    hdata = np.abs(hrng.standard_normal((300, 9, len(hts)))) * 0.05
    hp = hy.true_p() + 0.02 * hrng.standard_normal(hy.NP)
    hp[-1] = 0.1
    for lanes in (1, 2):
        node = NeuralODE(ODEProblem(PRESET_HYCHEM, hts, rate_scale=hy.DYDT_SCALE, grad_mode=2))
        node.set_queue_order(L.QUEUE_INDEX)
        node.set_ensemble(hu0, hdata, np.ones(9))
        node.set_tables(hT, hP)
        node.set_lanes_per_traj(lanes)
        out.append((f"hychem_adjoint_{lanes}lane", node, hp))
    for name, preset, (ts, u0, data, ys, p), kw in specs:
        lanes = kw.pop("lanes", None)
        primal = kw.pop("primal", False)
        node = NeuralODE(ODEProblem(preset, ts, **kw))
        node._xb_primal = primal
        if lanes is not None:
            node.set_lanes_per_traj(lanes)
        node.set_queue_order(L.QUEUE_INDEX)
        node.set_ensemble(u0, data, ys)
        out.append((name, node, p))
    return out


def worker():
    import numpy as np
    import ctypes as C
    sys.path.insert(0, ROOT)
    from crnn_amd import _lib as L
    from crnn_amd.api import p2vec_jac
    res = {}
    for name, node, p in _problems():
        th, dth = p2vec_jac(node.pmap, node.ns, node.nr, p)
        if getattr(node, "_xb_primal", False):
            grad, loss, _, ret, nsv = node._solve(node._ctx, node.B, th, None, 0, node.B, None, True)    # "grad" := the predictions
        else:
            _, loss, grad, ret, nsv = node._solve(node._ctx, node.B, th, dth, 0, node.B, None, False)
        na, nr = node.step_counts()
        h = hashlib.sha256()
        for a in (loss, grad, ret, nsv, na, nr):
            h.update(np.ascontiguousarray(a).tobytes())
        viol, site = C.c_uint32(0), C.c_uint32(0)
        chk = L.lib.crnn_debug_bounds(C.byref(viol), C.byref(site))     # -1: no checks compiled into this build
        res[name] = dict(digest=h.hexdigest()[:24], bounds_checked=(chk == 0), bounds_violations=int(viol.value), bounds_site=int(site.value), loss_sum=float(loss.sum()).hex(), grad0=float(np.ravel(grad)[0]).hex(),
                         gnorm=float(np.linalg.norm(grad)).hex(), n_accept=int(na.sum()), n_reject=int(nr.sum()),
                         n_fail=int((ret != 0).sum()), loss=[float(x).hex() for x in loss[:8]])
        node.close()
    # ---- round 4: the HyChem primal launches (pair kernel, and the AutoTsit5 composite), the cathode kernels
    from crnn_amd import PRESET_HYCHEM, hychem as hy

    def record(name, arrays, na, nr, nfail):
        h = hashlib.sha256()
        for a in arrays:
            h.update(np.ascontiguousarray(a).tobytes())
        viol, site = C.c_uint32(0), C.c_uint32(0)
        chk = L.lib.crnn_debug_bounds(C.byref(viol), C.byref(site))
        res[name] = dict(digest=h.hexdigest()[:24], bounds_checked=(chk == 0), bounds_violations=int(viol.value), bounds_site=int(site.value),
                         loss_sum=float(np.sum(arrays[0])).hex(), n_accept=int(na), n_reject=int(nr), n_fail=int(nfail))

    hrng = np.random.Generator(np.random.PCG64([77, 4]))
    hts, hu0, hT, hP = hy.sample_conditions(300, hrng)
    hdata = np.abs(hrng.standard_normal((300, 9, len(hts)))) * 0.05
    hp = hy.true_p() + 0.02 * hrng.standard_normal(hy.NP)
    hp[-1] = 0.1
    for name, kw in (("hychem_primal_pair", dict()), ("hychem_autotsit5_primal", dict(solver=L.SOLVER_AUTOTSIT5))):
        from crnn_amd import NeuralODE, ODEProblem
        node = NeuralODE(ODEProblem(PRESET_HYCHEM, hts, rate_scale=hy.DYDT_SCALE, **kw))
        node.set_queue_order(L.QUEUE_INDEX)
        node.set_ensemble(hu0, hdata, np.ones(9))
        node.set_tables(hT, hP)
        th, _ = p2vec_jac(node.pmap, node.ns, node.nr, hp)
        pred, loss, _, ret, nsv = node._solve(node._ctx, node.B, th, None, 0, node.B, None, True)
        na, nr = node.step_counts()
        record(name, (loss, pred, ret, nsv, na, nr), na.sum(), nr.sum(), (ret != 0).sum())
        node.close()
    from crnn_amd.cathode import CathodeUQ
    cfx = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures_cathode.json")))

    def two(s_):
        dbar, d2bar = np.array(s_["dbar"]), np.array(s_["d2bar"])
        sd = np.sqrt(np.maximum(d2bar - dbar ** 2, 0.0))
        return np.stack([np.array(s_["ts"]), dbar + sd, dbar - sd], axis=1)

    crng = np.random.default_rng(21)
    cp = 1 + 0.05 * crng.standard_normal((70, 17))
    cp[:, 6:9] = 0.0
    for name, kw, grad in (("cathode_adjoint", dict(grad_mode=2, errnorm_sens=0), True), ("cathode_adjoint_tape4", dict(grad_mode=2, tape_every=4, errnorm_sens=0), True),
                           ("cathode_adjoint_tape2", dict(grad_mode=2, tape_every=2, errnorm_sens=0), True), ("cathode_forward", dict(grad_mode=1, errnorm_sens=0), True),
                           ("cathode_primal", dict(), False), ("cathode_autotsit5_trbdf2_primal", dict(solver="autotsit5_trbdf2"), False),
                           ("cathode_autotsit5_ros23_primal", dict(solver="autotsit5_rosenbrock23"), False),
                           ("cathode_errnorm_sens2", dict(errnorm_sens=2), True)):
        uq = CathodeUQ([two(s_) for s_ in cfx["sets"]], [s_["beta"] for s_ in cfx["sets"]], cfx["theta"], **kw)
        loss, grad_, hrr = uq.solve(cp, want_grad=grad, want_hrr=True)
        st = uq.last_stats
        record(name, (loss, hrr, uq.last_retcode, uq.last_n_saved) + ((grad_,) if grad else ()), st["n_accept"], st["n_reject"], (uq.last_retcode != 0).sum())
        uq.close()
    print("XBRESULT " + json.dumps(res))


def run(json_out=None, names=None, timeout=900):
    libs = build(names)
    results = {}
    for name, lib in zip(names or VARIANTS, libs):
        env = dict(os.environ, CRNN_HIP_LIB=lib)
        pr = subprocess.run([sys.executable, os.path.abspath(__file__), "worker"], env=env, capture_output=True, text=True, timeout=timeout)
        line = [l for l in pr.stdout.splitlines() if l.startswith("XBRESULT ")]
        if pr.returncode != 0 or not line:
            results[name] = dict(error=f"rc {pr.returncode}", tail=(pr.stdout + pr.stderr)[-1500:])
        else:
            results[name] = json.loads(line[0][len("XBRESULT "):])
    ref_name = next(iter(results))
    ref = results[ref_name]
    mismatches = []
    for name, r in results.items():
        if "error" in r:
            mismatches.append((name, "*", r["error"], r.get("tail", "")))
            continue
        for prob, v in r.items():
            if "error" in ref or v["digest"] != ref[prob]["digest"] or v["bounds_violations"] != 0:
                mismatches.append((name, prob, v, ref.get(prob)))
    if json_out:
        with open(json_out, "w") as f:
            json.dump(dict(source=source_hash(), variants={k: VARIANTS[k] for k in results}, results=results,
                           identical=not mismatches), f, indent=1)
    return results, mismatches


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("cmd", choices=["build", "run", "worker"])
    ap.add_argument("--json", default=None)
    ap.add_argument("--variants", default=None, help="comma-separated subset of " + ",".join(VARIANTS))
    ap.add_argument("--force", action="store_true")
    a = ap.parse_args()
    names = a.variants.split(",") if a.variants else None
    if a.cmd == "build":
        print("\n".join(build(names, a.force)))
    elif a.cmd == "worker":
        worker()
    else:
        results, mism = run(a.json, names)
        for name, r in results.items():
            if "error" in r:
                print(f"{name:8s} ERROR {r['error']}\n{r.get('tail', '')}")
                continue
            for prob, v in r.items():
                print(f"{name:8s} {prob:28s} {v['digest']} acc {v['n_accept']} rej {v['n_reject']} fail {v['n_fail']} loss_sum {float.fromhex(v['loss_sum']):.17g}")
        print("IDENTICAL" if not mism else f"MISMATCH: {[(m[0], m[1]) for m in mism]}")
        sys.exit(1 if mism else 0)

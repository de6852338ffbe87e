"""bench.py prints its one JSON line AFTER the secondary figures: they must not be able to take it along.  Two mechanisms, both checked here
without a device: (i) tools/bench_secondary.py::run_all turns an entry that throws into {"error": ...}, goes on, and rewrites its output file
after every entry; (ii) bench.py runs it in a child process with a wall-clock limit (secondary_in_child) and keeps what the child had finished
when it crashes or outlives the limit."""
import json
import os
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_a_failing_entry_is_recorded_and_the_others_run(tmp_path, monkeypatch):
    import bench_secondary as bs
    import crnn_amd

    calls = []

    def fake_case2_fixed(u0, data, yscale, p, label_extra=None, reps=6, device=0, lanes=None, **kw):
        calls.append(("case2", u0.shape[0], lanes, tuple(sorted(kw))))
        if kw.get("errnorm_sens") and kw.get("solver") is not None:
            raise RuntimeError("tsit5_sens_kernel: launch failed")
        return {"kernel_ms": 1.0, "call_ms": 2.0, "value": u0.shape[0] / 1e-3, **(label_extra or {})}

    class FakeNode:
        def __init__(self, prob): pass
        def set_ensemble(self, *a): pass
        def train_init(self, *a): pass
        def train_step(self, **k): pass
        def params(self): return np.zeros(25)
        def close(self): pass

    monkeypatch.setattr(bs, "case2_fixed", fake_case2_fixed)
    monkeypatch.setattr(bs, "case2_ensemble", lambda B, seed, device=0: (np.zeros((B, 7)), np.zeros((B, 6, 2)), np.ones(6)))
    monkeypatch.setattr(bs, "robertson", lambda **k: {"kernel_ms": 3.0})
    monkeypatch.setattr(bs, "hychem", lambda **k: (_ for _ in ()).throw(MemoryError("hipMalloc")) if k.get("B") == 262144 else {"kernel_ms": 4.0})
    monkeypatch.setattr(bs, "cathode", lambda **k: {"kernel_ms": 5.0})
    monkeypatch.setattr(crnn_amd, "NeuralODE", FakeNode)
    monkeypatch.setattr(crnn_amd, "ODEProblem", lambda *a, **k: None)
    monkeypatch.setattr(crnn_amd, "Optimiser", lambda *a, **k: None)
    out = tmp_path / "sec.json"
    sec = bs.run_all(np.zeros((65536, 7)), np.zeros((65536, 6, 2)), np.ones(6), device=0, out_path=str(out))
    assert "error" in sec["case2_errnorm_sens1_tsit5"] and "launch failed" in sec["case2_errnorm_sens1_tsit5"]["error"]
    assert "error" in sec["hychem_B262144_one_gpu"] and "MemoryError" in sec["hychem_B262144_one_gpu"]["error"]
    for k in ("case2_reference_init_p", "case2_errnorm_sens1", "case2_B8192_share", "case2_B65536_two_lanes", "case2_B262144", "robertson_B65536",
              "hychem_B32768", "cathode_4096x256"):
        assert "error" not in sec[k], k
    assert sec["case2_errnorm_sens1"]["value"] == 65536 / 2e-3           # the post-processing of a healthy entry still happens
    assert json.load(open(out)) == json.loads(json.dumps(sec))           # and the file holds the final state, as strict JSON
    json.dumps(sec, allow_nan=False)


def test_the_child_runner_keeps_finished_entries_when_the_child_dies_or_hangs():
    import bench
    child = textwrap.dedent("""
        import json, sys, time, os
        out = sys.argv[sys.argv.index('--out') + 1]
        json.dump({'a': {'kernel_ms': 1.0}}, open(out, 'w'))
        mode = sys.argv[1]
        if mode == 'crash': os._exit(7)
        if mode == 'hang': time.sleep(60)
        json.dump({'a': {'kernel_ms': 1.0}, 'b': {'kernel_ms': 2.0}}, open(out, 'w'))
    """)
    ok = bench.secondary_in_child([sys.executable, "-c", child, "fine"], 30)
    assert ok == {"a": {"kernel_ms": 1.0}, "b": {"kernel_ms": 2.0}}
    crashed = bench.secondary_in_child([sys.executable, "-c", child, "crash"], 30)
    assert crashed["a"] == {"kernel_ms": 1.0} and "code 7" in crashed["_incomplete"]
    hung = bench.secondary_in_child([sys.executable, "-c", child, "hang"], 2)
    assert hung["a"] == {"kernel_ms": 1.0} and "stopped after" in hung["_incomplete"]
    nothing = bench.secondary_in_child(["/nonexistent/interpreter"], 5)
    assert list(nothing) == ["_incomplete"]

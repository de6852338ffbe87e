"""examples/case2_train.c: the C ABI driven from plain C99 (no Python, no torch) -- the compiled-host view of the boundary.
CPU: it builds against include/crnn_hip.h and the in-tree library and fails loudly without a GPU.  GPU: it trains."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "crnn_amd", "csrc")


def _build(tmp_path):
    from crnn_amd import _lib  # noqa: F401  (builds / locates libcrnn_hip.so)
    exe = str(tmp_path / "case2_train")
    cc = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-O2", "-I" + os.path.join(ROOT, "include"),
                         os.path.join(ROOT, "examples", "case2_train.c"), "-o", exe, "-L" + CSRC, "-lcrnn_hip", "-lm",
                         "-Wl,-rpath," + CSRC], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-2000:]
    return exe


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_c_example_builds_and_needs_a_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    exe = _build(tmp_path)
    run = subprocess.run([exe, "16", "2"], capture_output=True, text=True, timeout=120)
    assert run.returncode == 2 and "no HIP device" in run.stderr      # no CPU fallback


@pytest.mark.gpu
def test_c_example_trains_on_the_gpu(tmp_path):
    exe = _build(tmp_path)
    run = subprocess.run([exe, "2048", "12"], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout[-1500:] + run.stderr[-1500:]
    assert "final mean loss" in run.stdout and "ok 2048/2048" in run.stdout
    assert "library build: src=" in run.stdout and "restart from the mid-run checkpoint reproduces" in run.stdout


@pytest.mark.gpu
def test_reference_style_training_script_with_checkpoint_restart(tmp_path):
    """examples/case2_train.py: per-experiment updates in random order, epoch-end losses, BSON checkpoint and restart."""
    import sys
    pytest.importorskip("bson")
    ck = str(tmp_path / "mymodel.bson")
    cmd = [sys.executable, os.path.join(ROOT, "examples", "case2_train.py"), "--checkpoint", ck, "--n-plot", "4"]
    a = subprocess.run(cmd + ["--epochs", "8"], capture_output=True, text=True, timeout=600)
    assert a.returncode == 0, a.stderr[-2000:]
    lines = [ln for ln in a.stdout.splitlines() if ln.startswith("epoch")]
    assert len(lines) == 8
    first, last = float(lines[0].split()[4]), float(lines[-1].split()[4])
    assert last < first                                        # it learns
    from crnn_amd.io import load_checkpoint
    c = load_checkpoint(ck)
    assert int(c["iter"]) == 8 and c["p"].shape == (25,) and len(c["l_loss_train"]) == 8
    b = subprocess.run(cmd + ["--epochs", "10", "--restart"], capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-2000:]
    assert "restarting from" in b.stdout and len([ln for ln in b.stdout.splitlines() if ln.startswith("epoch")]) == 2
    r = load_checkpoint(ck)
    assert int(r["iter"]) == 10 and r["opt_state"].shape == (2 * 25 + 4,)
    # `@save ... p opt` / `@load`: the restart restores the optimiser state (ADAM moments, beta powers, ExpDecay counter),
    # so the resumed run ends exactly where an uninterrupted one does (case2/case2.jl:178-187,213)
    ck2 = str(tmp_path / "straight.bson")
    cmd2 = [sys.executable, os.path.join(ROOT, "examples", "case2_train.py"), "--checkpoint", ck2, "--n-plot", "100"]
    s = subprocess.run(cmd2 + ["--epochs", "10"], capture_output=True, text=True, timeout=600)
    assert s.returncode == 0, s.stderr[-2000:]
    u = load_checkpoint(ck2)
    import numpy as np
    assert np.array_equal(u["p"], r["p"]) and np.array_equal(u["opt_state"], r["opt_state"])
    assert r["opt_state"][2 * 25 + 3] == 10 * 20          # ExpDecay's call counter: one update per experiment per epoch

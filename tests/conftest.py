import json
import os
import sys

import numpy as np

LB_CASE1, LB_CASE2 = float(np.float32(1e-5)), float(np.float32(1e-6))   # `lb = 1.f-5` / `lb = 1.f-6`: Float32 literals (case1/case1.jl:34, case2/case2.jl:34)
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    config.addinivalue_line("markers", "needs_reference: reads /root/reference (build container only; never joins the -m gpu session)")
    config.addinivalue_line("markers", "cpu_only: a stand-in for the device (the SIMT emulation sample) -- pointless where the device itself runs the same tests")


def _gpu_present(config=None):
    # CRNN_TEST_ASSUME_GPU=1: collection checks only (tests/test_host.py: the order of a device session) -- honoured with --collect-only and
    # nowhere else, so a stray variable cannot mark every test `gpu` on a CPU box
    if os.environ.get("CRNN_TEST_ASSUME_GPU") == "1" and config is not None and getattr(config.option, "collectonly", False):
        return True
    try:
        import torch
        return bool(torch.cuda.is_available())
    except Exception:
        return False


@pytest.hookimpl(tryfirst=True)
def pytest_collection_modifyitems(config, items):
    """On a box WITH a GPU every test also carries the `gpu` marker, so the driver's `-m gpu` session runs the whole chain
    in one go: golden vectors -> oracle (test_oracle_golden, the CPU halves of test_cathode / test_hychem, test_host) ->
    HIP kernels.  Without a GPU nothing changes: `-m "not gpu"` runs the CPU tests, the GPU tests stay deselected.
    (tryfirst: this runs before the mark plugin evaluates -m.)"""
    # no test may hang a session: a default per-test limit wherever pytest-timeout is installed (explicit marks win).  On a box with a device the
    # limit is shorter and enforced from a watchdog THREAD (os._exit): a kernel that never returns blocks inside hipStreamSynchronize, where the
    # default SIGALRM handler never gets to run -- the session would sit there until the driver's own limit, and a wedged GPU is a strike.  Ending
    # the process ends the kernel.  (Kernels no device has executed yet are what this is for; every device test of rounds 1-4 ran in seconds.)
    on_device = _gpu_present(config) and os.environ.get("CRNN_TEST_ASSUME_GPU") != "1"
    if config.pluginmanager.hasplugin("timeout"):
        for it in items:
            if it.get_closest_marker("timeout") is None:
                it.add_marker(pytest.mark.timeout(420, method="thread") if on_device else pytest.mark.timeout(900))
    if not _gpu_present(config):
        return
    for it in items:
        if it.get_closest_marker("gpu") is None and it.get_closest_marker("needs_reference") is None and it.get_closest_marker("cpu_only") is None:
            it.add_marker(pytest.mark.gpu)
    # Order of a device session: first the test functions that existed when a driver last ran the suite on an MI355X (round 3: 184 passed,
    # tests/golden/device_history.json), then what rounds 4 and 5 added -- kernels that have only been run by the builder (round 4) or only
    # under SIMT emulation (round 5).  The driver runs `pytest -x`: a first-contact failure in a new kernel then still leaves the record of
    # everything that was green before, instead of cutting the session short in the middle of the alphabet.  Nothing is skipped or deselected.
    try:
        with open(os.path.join(ROOT, "tests", "golden", "device_history.json")) as f:
            seen = set(json.load(f)["functions"])
    except Exception:
        seen = set()
    if seen:
        def known(it):
            path = os.path.relpath(str(it.fspath), ROOT).replace(os.sep, "/")
            return f"{path}::{getattr(it, 'originalname', None) or it.name.split('[')[0]}" in seen
        items.sort(key=lambda it: 0 if known(it) else 1)       # stable: file / definition order is kept within each group


# A device session has a wall-clock budget.  The driver gives `pytest -m gpu` a fixed time (1 200 s in rounds 1-5); a session it has to kill leaves no
# record at all.  Nobody could measure this suite's wall time on a device since round 3 (159 s for 184 tests then; 270 tests now), so the session
# watches its own clock: once CRNN_SESSION_BUDGET_S (default 900) have passed, the remaining tests are SKIPPED with that reason -- visibly, not
# silently -- instead of running into the driver's limit.  The driver-proven tests run first (above), so what can fall off the end is the newest.
_SESSION_T0 = None


def pytest_sessionstart(session):
    global _SESSION_T0
    import time
    _SESSION_T0 = time.monotonic()


def pytest_runtest_setup(item):
    import time
    budget = float(os.environ.get("CRNN_SESSION_BUDGET_S", "900"))
    if _SESSION_T0 is not None and budget > 0 and item.get_closest_marker("gpu") is not None and _gpu_present(item.config) \
            and os.environ.get("CRNN_TEST_ASSUME_GPU") != "1" and time.monotonic() - _SESSION_T0 > budget:
        pytest.skip(f"device session budget of {budget:.0f} s spent (tests/conftest.py): not run, NOT passed")


def emulated():
    """True when CRNN_HIP_LIB points at the SIMT emulation library (tools/simt_suite.sh).  The BASELINE-size tests then run the same logic and the
    same assertions (minus the clock) at sizes scaled to the emulated device's two CUs instead of being left out."""
    from crnn_amd import _lib as L
    return "SIMT-EMULATION" in L.lib.crnn_build_info().decode()


@pytest.fixture(scope="session")
def fx():
    with open(os.path.join(ROOT, "tests", "golden", "fixtures.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure): builds oracle/libcrnn_oracle.so on demand."""
    from oracle import oracle as o
    o.build()
    o.lib()
    return o


INV_R = float(np.float32(-1.0) / np.float32(1.98720425864083e-3))   # Float32 literal in case2.jl:113
# ADAMW's decay is a Float32 literal too (`1.f-6` case2.jl:32, rober_crnn.jl:19; `1.f-8` case1.jl:18): the optimiser presets carry
# these values, pinned by the reference's checkpoints (tests/test_ckpt_opt_pin.py)
WD6, WD8 = float(np.float32(1e-6)), float(np.float32(1e-8))


@pytest.fixture(scope="session")
def case2_setup(fx):
    c2 = fx["case2"]
    return dict(u0=np.array(c2["u0"]), tsteps=np.array(c2["tsteps"]), data=np.array(c2["data"]),
                yscale=np.array(c2["yscale"]), pred_ckpt=np.array(c2["pred_ckpt"]),
                p_ckpt=np.array(fx["case2_ckpt"]["p"]), p_init=np.array(c2["p_init"]), grads=c2["grads"])


@pytest.fixture(scope="session")
def rober_setup(fx):
    rb = fx["robertson"]
    return dict(u0=np.array(rb["u0"]), tsteps=np.array(rb["tsteps"]), data=np.array(rb["data"]),
                yscale=np.array(rb["yscale"]), dydt_scale=np.array(rb["dydt_scale"]),
                pred_ckpt=np.array(rb["pred_ckpt"]), p_ckpt=np.array(fx["rober_ckpt"]["p"]), grads=rb["grads"],
                kat=rb["kat"])


def oracle_problem(orc, case, setup, atol=None, rtol=None, maxiters=None, **kw):
    if case == "case2":
        return orc.make_problem(ns=6, nr=3, has_temp=1, lb=LB_CASE2, ub=10.0, inv_R=INV_R,
                                atol=1e-6 if atol is None else atol, rtol=1e-3 if rtol is None else rtol,
                                yscale=setup["yscale"], clamp_pred=1, maxiters=maxiters or 100000, **kw)
    if case == "rober":
        return orc.make_problem(ns=3, nr=6, lb=1e-8, atol=[1e-6, 1e-8, 1e-6] if atol is None else atol,
                                rtol=1e-3 if rtol is None else rtol, yscale=setup["yscale"],
                                rate_scale=setup["dydt_scale"], maxiters=maxiters or 10000, **kw)
    raise ValueError(case)

"""The kernels' OWN SOURCES executed on the CPU (no `gpu` marker: this runs in the driver's `-m "not gpu"` session).

tests/simt/ compiles crnn_amd/csrc/crnn_capi.hip and every kernel header UNCHANGED as host C++ against a SIMT emulation of
<hip/hip_runtime.h> (a fibre per lane; DPP moves, shuffles, ballots, readfirstlane, the FP64 MFMA and the barriers resolved between the
fibres of a wavefront) into tests/simt/libcrnn_simt.so -- the same C ABI.  A sample of the `-m gpu` parity tests is then run against it in
a child process (CRNN_HIP_LIB selects the library at import; one library per process): every stepper family, both gradient algorithms,
the lane-pair kernel, the dual-norm kernels of all three right-hand sides -- round 5's hychem_sens2_kernel (sparse directions), its
COMPOSITE instantiation (the reference's gradient through the reference's own AutoTsit5(Rosenbrock23)), the cathode's chunked gradient on
Rosenbrock23 and through AutoTsit5(TRBDF2) (cathode_sens_auto_kernel).
It proves what the sources compute (control flow, tapes, queues, reductions included), under an interleaving of lanes more adversarial than
the device's lockstep; it says nothing about time or about the ISA.  The whole emulated suite: `bash tools/simt_suite.sh`
(profiles/r05a_simt_suite.txt).  The emulation library is never loaded by the product (crnn_amd/_lib.py loads libcrnn_hip.so unless a test
sets CRNN_HIP_LIB; bench.py and smoke() refuse a library whose build info says SIMT-EMULATION)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMT = os.path.join(ROOT, "tests", "simt")
CLANG = os.environ.get("SIMT_CXX", "/opt/rocm/lib/llvm/bin/clang++")

# cpu_only: on a box WITH a device the same parity tests run on the device itself; tests/conftest.py keeps these out of the -m gpu session
pytestmark = [pytest.mark.cpu_only,
              pytest.mark.skipif(not os.path.exists(CLANG) or os.uname().machine != "x86_64",
                                 reason="the SIMT emulation build needs the ROCm clang++ on an x86-64 host")]


@pytest.fixture(scope="module")
def simt_lib():
    subprocess.run(["bash", os.path.join(SIMT, "build.sh")], check=True, timeout=1500)
    lib = os.path.join(SIMT, "libcrnn_simt.so")
    assert os.path.exists(lib)
    return lib


SAMPLES = {      # child sessions of `-m gpu` parity tests against the emulation library; started together (they are independent processes)
    "case2_rober": ["tests/test_gpu_parity.py", "-k", "gradient_matches_oracle or adjoint_equals_forward_tangents or tsit5_adjoint_equals"],
    "lanes2": ["tests/test_gpu_lanes2.py", "-k", "not auto_beyond_one_generation"],
    "hychem_dual": ["tests/test_hychem.py", "-k",
                    "(errnorm_sens_matches_oracle_chunk_for_chunk and 2-2) or through_the_reference_composite_matches or sparse_direction_kernel"],
    "cathode": ["tests/test_cathode.py", "-k", "(errnorm_sens_matches_oracle_chunk_for_chunk and 2) or gradient_through_the_reference_composite"],
    "case2_stream": ["tests/test_case2_stream_pin.py", "-k", "loss_at_the_checkpoint"],
}


@pytest.fixture(scope="module")
def samples(simt_lib):
    """All child sessions at once, two emulation threads each: the wall time of the slowest instead of the sum (the CPU suite's budget)."""
    env = dict(os.environ, CRNN_HIP_LIB=simt_lib, SIMT_THREADS=os.environ.get("SIMT_THREADS", "2"))
    procs = {k: subprocess.Popen([sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-p", "no:cacheprovider", *a], cwd=ROOT, env=env,
                                 stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for k, a in SAMPLES.items()}
    res = {}
    for k, p in procs.items():
        try:
            out, err = p.communicate(timeout=1500)
        except subprocess.TimeoutExpired:
            p.kill()
            out, err = p.communicate()
            err += "\n[timeout]"
        res[k] = (p.returncode, out, err)
    return res


def _passed(samples, key):
    rc, out, err = samples[key]
    tail = out[-3000:] + err[-2000:]
    assert rc == 0, tail
    m = re.search(r"(\d+) passed", out)
    assert m, tail
    return int(m.group(1))


def test_emulated_library_says_what_it_is_and_is_refused_by_the_bench(simt_lib):
    code = ("import os, sys; sys.path.insert(0, %r); from crnn_amd import _lib as L; i = L.lib.crnn_build_info().decode(); "
            "assert 'SIMT-EMULATION' in i, i; print(i)") % ROOT
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CRNN_HIP_LIB=simt_lib), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], env=dict(os.environ, CRNN_HIP_LIB=simt_lib),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "SIMT emulation" in (out.stderr + out.stdout)


def test_case2_and_robertson_kernels_against_the_oracle_under_emulation(samples):
    """ros23_adj_kernel / ros23_adj2_kernel (lane pair) / ros23_kernel (forward tangents) / tsit5 / auto_adj: loss 1e-9, gradient 1e-7, step counts."""
    assert _passed(samples, "case2_rober") >= 8
    assert _passed(samples, "lanes2") >= 4


def test_hychem_dual_norm_kernels_against_the_oracle_under_emulation(samples):
    """hychem_sens2_kernel chunk for chunk against the oracle (mode 2), the same through the reference's composite, and equal to the dense kernel."""
    assert _passed(samples, "hychem_dual") == 3


# (the finite-difference-Jacobian kernels, 70 s under emulation, are part of the whole emulated suite only: tools/simt_suite.sh, profiles/r06b)


def test_cathode_chunked_gradient_against_the_oracle_under_emulation(samples):
    assert _passed(samples, "cathode") == 2


def test_case2_recorded_checkpoint_loss_under_emulation(samples):
    """The product's loss at the reference's checkpoint on the experiments re-drawn from the reference's RNG stream equals the number the
    reference recorded (tests/test_case2_stream_pin.py), Tsit5 / AutoTsit5 / Rosenbrock23 kernels."""
    assert _passed(samples, "case2_stream") == 1

"""The SIMT emulator calibrated against the hardware, once.

tests/simt/calib.hip applies every cross-lane / matrix / special-function primitive the kernels use (in the kernels' own call forms) to random
values with one wavefront.  It is built twice: as host C++ against the emulation shim (tests/simt/hip/hip_runtime.h) and with hipcc for gfx950.
  CPU test   the emulator's result against a plain numpy statement of what each primitive does (so the emulator's model is written down once
             more, independently of the shim);
  -m gpu     the device's result against the emulator's, bit for bit (rcp / rsq: the device rounds them to one ulp, compared to one ulp).
After the second has passed on an MI355X, "green under emulation" is a statement about the device as far as these primitives go; what the
emulator still cannot say is listed in tests/simt/hip/hip_runtime.h (time, occupancy, LDS banking, inline asm).  The emulator is frozen: no
further features.
"""
import ctypes as C
import os
from fractions import Fraction
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SIMT = os.path.join(HERE, "simt")
SRC = os.path.join(SIMT, "calib.hip")
SLOTS = 32
NAMES = ["dpp_B1", "dpp_F5", "dpp_A0", "dpp_B1_int", "shfl", "group9_sum", "group12_sum", "shfl_down_tree", "shfl_xor_max", "ballot", "readfirstlane",
         "ballot_divergent", "readfirstlane_divergent", "mfma0", "mfma1", "mfma2", "mfma3", "rcp", "rsq", "ldexp", "frexp_mant", "frexp_exp"]


def _inputs(seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((4, 64)) * np.exp(3.0 * rng.standard_normal((4, 64)))
    x[3] = np.abs(x[3]) + 1e-3
    idx = rng.integers(0, 64, 64).astype(np.int32)
    return np.ascontiguousarray(x), idx


def _build_emulated():
    out = os.path.join(SIMT, "libcalib_simt.so")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(SRC), os.path.getmtime(os.path.join(SIMT, "hip", "hip_runtime.h"))):
        cxx = os.environ.get("SIMT_CXX", "/opt/rocm/lib/llvm/bin/clang++")
        subprocess.check_call([cxx, "-x", "c++", "-std=c++17", "-O1", "-g1", "-fPIC", "-shared", "-ffp-contract=on", "-mfma", "-fno-strict-aliasing", "-pthread",
                               "-Wno-unknown-attributes", "-Wno-ignored-attributes", "-Wno-unused-value", "-Wno-macro-redefined", "-Wno-keyword-macro",
                               "-Wno-builtin-macro-redefined", "-I", SIMT, "-DCRNN_SIMT_EMULATION=1", "-o", out, SRC])
    return out


def _build_device():
    out = os.path.join(SIMT, "libcalib_hip.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(SRC):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-o", out, SRC])
    return out


_RUNNER = """
import ctypes as C, sys, numpy as np
lib = C.CDLL(sys.argv[1]); S = lib.crnn_calib_slots()
x = np.load(sys.argv[2]); idx = np.load(sys.argv[3]); out = np.zeros((64, S), np.uint64)
rc = lib.crnn_calib_run(x.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
assert rc == 0, rc
np.save(sys.argv[4], out)
"""


def _run(libpath, x, idx, tmp_path, tag):
    """Each library in a process of its own: the emulation shim and the real runtime both define the hip* entry points."""
    fx, fi, fo = (str(tmp_path / f"{tag}_{n}.npy") for n in ("x", "idx", "out"))
    np.save(fx, x); np.save(fi, idx)
    subprocess.check_call([sys.executable, "-c", _RUNNER, libpath, fx, fi, fo], timeout=300)
    out = np.load(fo)
    assert out.shape == (64, SLOTS)
    return out


def _f(col):
    return np.ascontiguousarray(col).view(np.float64)


def _model(x, idx):
    """What each primitive does, in numpy.  Returns {name: uint64[64]} for the exactly defined ones."""
    a, b, c, d = x
    lane = np.arange(64)
    u = lambda v: np.ascontiguousarray(v, np.float64).view(np.uint64)
    m = {}
    m["dpp_B1"] = u(a[lane ^ 1])                                   # quad_perm [1,0,3,2]
    m["dpp_F5"] = u(a[lane | 1])                                   # quad_perm [1,1,3,3]
    m["dpp_A0"] = u(a[lane & ~1])                                  # quad_perm [0,0,2,2]
    m["dpp_B1_int"] = idx[lane ^ 1].astype(np.uint64)
    m["shfl"] = u(a[idx])
    g9, g12 = lane // 9 * 9, lane // 12 * 12
    s9 = np.zeros(64); s12 = np.zeros(64)
    for q in range(9):
        src = g9 + q
        s9 = s9 + b[src & 63]                                               # __shfl takes the source lane modulo the width (HIP: srcLane & (width - 1); ds_bpermute wraps likewise)
    for q in range(12):
        src = g12 + q
        s12 = s12 + b[src & 63]
    m["group9_sum"], m["group12_sum"] = u(s9), u(s12)
    v = c.copy()
    for off in (32, 16, 8, 4, 2, 1):
        src = lane + off
        v = v + np.where(src < 64, v[np.minimum(src, 63)], v)              # __shfl_down beyond the end: own value
    m["shfl_down_tree"] = u(v)
    m["shfl_xor_max"] = u(np.full(64, c.max()))
    m["ballot"] = np.full(64, sum(1 << int(i) for i in lane if a[i] > 0), np.uint64)
    m["readfirstlane"] = np.full(64, int(idx[0]), np.uint64)
    act = (lane >= 5) & (b > 0)
    bal = sum(1 << int(i) for i in lane if act[i] and c[i] > 0)
    first = int(lane[act][0])
    m["ballot_divergent"] = np.where(act, np.uint64(bal), np.uint64(0))
    m["readfirstlane_divergent"] = np.where(act, np.uint64(first * 7 + 3), np.uint64(0))
    # v_mfma_f64_16x16x4f64: lane l supplies A[l % 16][l // 16] and B[l // 16][l % 16]; D[4 r + l // 16][l % 16] lands in accumulator r of lane l
    # (the layout tools/ubench/mfma_f64_layout.hip measured on the device in round 3); products accumulate over k = 0..3 in order, fused
    D = np.zeros((16, 16))
    for A_, B_ in ((a, b), (c, d)):
        Am = A_.reshape(4, 16).T        # [i][k]
        Bm = B_.reshape(4, 16)          # [k][j]
        for k in range(4):
            D = np.array([[float(Fraction(float(Am[i, k])) * Fraction(float(Bm[k, j])) + Fraction(float(D[i, j]))) for j in range(16)] for i in range(16)])      # one rounding: fma
    for j in range(4):
        m[f"mfma{j}"] = u(np.array([D[4 * j + l // 16, l % 16] for l in lane]))
    m["ldexp"] = u(np.ldexp(a, idx - 32))
    mant, ex = np.frexp(a)
    m["frexp_mant"] = u(mant)
    m["frexp_exp"] = ex.astype(np.int64).view(np.uint64)
    return m


@pytest.mark.parametrize("seed", [0, 1])
def test_emulator_primitives_match_their_numpy_statement(seed, tmp_path):
    x, idx = _inputs(seed)
    out = _run(_build_emulated(), x, idx, tmp_path, "emu")
    m = _model(x, idx)
    for k, name in enumerate(NAMES):
        if name in m:
            assert np.array_equal(out[:, k], m[name]), (name, out[:4, k], m[name][:4])
    # rcp / rsq: the emulator divides exactly
    assert np.array_equal(_f(out[:, NAMES.index("rcp")]), 1.0 / x[3])
    assert np.allclose(_f(out[:, NAMES.index("rsq")]), 1.0 / np.sqrt(x[3]), rtol=2e-16)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_gpu_primitives_equal_the_emulators_bit_for_bit(seed, tmp_path):
    if "simt" in os.path.basename(os.environ.get("CRNN_HIP_LIB", "")):
        pytest.skip("the emulated suite (tools/simt_suite.sh) has no device to calibrate against")
    x, idx = _inputs(seed)
    emu = _run(_build_emulated(), x, idx, tmp_path, "emu")
    dev = _run(_build_device(), x, idx, tmp_path, "dev")
    for k, name in enumerate(NAMES):
        if name in ("rcp", "rsq"):
            e, d = _f(emu[:, k]), _f(dev[:, k])
            assert np.all(np.abs(d - e) <= np.spacing(np.abs(e))), (name, d[:4], e[:4])      # v_rcp_f64 / v_rsq_f64: 1 ulp
        else:
            assert np.array_equal(dev[:, k], emu[:, k]), (name, dev[:4, k], emu[:4, k])

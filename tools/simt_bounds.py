#!/usr/bin/env python3
"""The kernels' own bounds checks (-DCRNN_BOUNDS_CHECK: every indexed access of the adjoint / HyChem / cathode / forward-tangent kernels checked against
its extent on the lanes that perform it, violations counted on the 'device') under SIMT emulation -- what tools/crossbuild.py's "O3chk" build does on an
MI355X, for a round without one.  Builds tests/simt with the checks compiled in, runs the given -m gpu tests in THIS process and reads the counter.
  python tools/simt_bounds.py [pytest args...]      default: the case2 / robertson / case1 parity files"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = "/tmp/libcrnn_simt_chk.so"
subprocess.run(["bash", os.path.join(ROOT, "tests", "simt", "build.sh")], check=True,
               env=dict(os.environ, SIMT_OUT=lib, SIMT_FLAGS="-DCRNN_BOUNDS_CHECK=1"))
os.environ["CRNN_HIP_LIB"] = lib
os.environ.setdefault("SIMT_THREADS", "6")
sys.path.insert(0, ROOT)
import pytest  # noqa: E402

args = sys.argv[1:] or ["tests/test_gpu_lanes2.py", "tests/test_gpu_parity.py", "tests/test_gpu_primal.py", "tests/test_gpu_fullsize.py",
                        "-k", "not auto_beyond_one_generation and not two_gpu and not loaded_library"]
rc = pytest.main(["-m", "gpu", "-q", "-p", "no:cacheprovider", *args])
from crnn_amd import _lib as L  # noqa: E402

viol, site = C.c_uint32(0), C.c_uint32(0)
chk = L.lib.crnn_debug_bounds(C.byref(viol), C.byref(site))
print(f"pytest exit {int(rc)}; crnn_debug_bounds: checks compiled in = {chk == 0}, violations = {viol.value}, first site code = {site.value}")
sys.exit(int(rc) or (0 if chk == 0 and viol.value == 0 else 3))

"""The CPU restatement under AddressSanitizer + UndefinedBehaviorSanitizer: oracle/sanitize_main.c
drives every solver family of oracle/crnn_oracle.c (CRNN shapes x Rosenbrock23 / Tsit5 / AutoTsit5, the OpenMP batch
driver, the optimiser chain, cathode, HyChem) on small seeded problems.  A checker of the checker: it keeps
out-of-bounds reads and undefined arithmetic in the oracle from silently shaping the parity tests."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_oracle_is_clean_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "orc_san")
    cc = subprocess.run(["gcc", "-O1", "-g", "-std=c11", "-fopenmp", "-fsanitize=address,undefined",
                         "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-o", exe,
                         os.path.join(ROOT, "oracle", "sanitize_main.c"), "-lm"], capture_output=True, text=True)
    if cc.returncode != 0 and "sanitize" in cc.stderr and "cannot find" in cc.stderr:
        pytest.skip("sanitizer runtime libraries not installed")
    assert cc.returncode == 0, cc.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", OMP_NUM_THREADS="2")
    run = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
    assert run.stdout.strip().endswith("OK")
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr

#!/bin/bash
# tools/simt_suite.sh -- the `-m gpu` parity tests executed on the CPU through the SIMT emulator (tests/simt/: the kernels' own sources
# compiled as host C++, a fibre per lane).  Everything except the tests that need the device's throughput (full BASELINE sizes), other
# builds of the device library (crossbuild) or more than one rank.  Output: gpurun-free evidence of what the kernel sources compute;
# the log is committed under profiles/ when it backs a claim.
#   bash tools/simt_suite.sh [pytest args...]        e.g.  bash tools/simt_suite.sh tests/test_hychem.py -k errnorm
#   SIMT_NOISE=1 bash tools/simt_suite.sh [...]      the same with one-ulp noise on what the device rounds differently (profiles/r05j_simt_ulp_noise.txt)
#   SIMT_ASAN=1 bash tools/simt_suite.sh [...]       the same with the emulated kernels compiled under AddressSanitizer: "device" buffers are
#       host allocations, LDS arrays and per-lane arrays are host objects, so a read or write of a kernel outside the object it indexes (beyond an allocation, an LDS array, a per-lane array) is reported with its
#       source line -- the compute-sanitizer this toolchain does not have (profiles/r05g_simt_asan_suite.txt)
R=$(cd $(dirname $0)/.. && pwd)
if [ -n "$SIMT_ASAN" ]; then
  RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
  SIMT_OUT=$R/tests/simt/libcrnn_simt_asan.so SIMT_FLAGS="-fsanitize=address -fno-omit-frame-pointer -shared-libasan" bash $R/tests/simt/build.sh || exit 1
  export CRNN_HIP_LIB=$R/tests/simt/libcrnn_simt_asan.so LD_PRELOAD=$RT
  export ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1:abort_on_error=1
elif [ -n "$SIMT_NOISE" ]; then
  # one-ulp noise on rcp / rsq / exp / log / pow (tests/simt/hip/hip_runtime.h, SIMT_ULP_NOISE): which parity bars survive last-place differences
  # between the device's arithmetic and the host's -- the bars that would otherwise be met only because emulation and oracle share a libm
  SIMT_OUT=$R/tests/simt/libcrnn_simt_noise.so SIMT_FLAGS="-DSIMT_ULP_NOISE=1 -Wl,--wrap=exp -Wl,--wrap=log -Wl,--wrap=pow" bash $R/tests/simt/build.sh || exit 1
  export CRNN_HIP_LIB=$R/tests/simt/libcrnn_simt_noise.so
else
bash $R/tests/simt/build.sh || exit 1
export CRNN_HIP_LIB=$R/tests/simt/libcrnn_simt.so
fi
export SIMT_THREADS=${SIMT_THREADS:-6}
cd $R
if [ $# -gt 0 ]; then exec python -m pytest -m gpu -p no:cacheprovider "$@"; fi
# Not run here, and why:
#   (The BASELINE-size tests -- full batches, queue / generation logic, config 4 as eight shards, config 5's tiles, HyChem's full share -- DO run, with
#   the same assertions minus the clock and the 256-CU geometry, at sizes scaled to the emulated device's two CUs: tests/conftest.py::emulated.)
#   test_two_gpu_data_parallel_bench_when_available   launches bench.py, which refuses the emulation library
#   test_gpu_crossbuild.py          builds the DEVICE library five ways; test_dist_gpu2proc.py: two ranks (the emulation's RCCL is one rank)
#   test_c_example_trains_on_the_gpu, test_loaded_library_was_built_from_these_sources     link / fingerprint libcrnn_hip.so itself
#   test_auto_beyond_one_generation_follows_the_step_count_spread   sized for 256 CUs' resident lanes (the emulated device has SIMT_CUS = 2)
#   test_gpu_cathode_matches_oracle_step_for_step[tol1]   at rtol 1e-9 the adjoint gradient sits 3.9e-7 from the oracle's against a bar of 1e-7
#       (loss and curve 1e-9): host libm / exact reciprocal against the device's transcendental and rcp rounding, amplified by 10^4 steps
#   test_kernel_resources.py, test_oracle_sanitizers.py   CPU tests of the ordinary suite
#   under SIMT_NOISE=1 only: test_gpu_device_resident_svgd_loop_matches_host_driven_loop -- it drives the OPT-IN primal-norm adjoint of the cathode over a
#       2 % particle cloud; one-ulp noise moves particle 20's gradient from 9e1 to 5e9 (the discrete map's own derivative: adjoint = forward
#       tangents to 1e-14 of it; presumably the kink of the depleted-species clamp -- profiles/r04m's census from another side), the SVGD move then throws the
#       cloud out of the solvable region.  The default dual-norm gradient (errnorm_sens = 2) stays at 90.7 under the same noise (profiles/r05j).
NOISE_DESELECT=""   # (since the last session of round 5 the test runs its primal-norm half on a well-conditioned cloud and nothing is deselected)
true
python -m pytest tests -m gpu -p no:cacheprovider -v --timeout=${SIMT_TIMEOUT:-2400} --durations=20 $NOISE_DESELECT \
  --ignore=tests/test_gpu_crossbuild.py --ignore=tests/test_dist_gpu2proc.py \
  --ignore=tests/test_kernel_resources.py --ignore=tests/test_oracle_sanitizers.py \
  --deselect "tests/test_cathode.py::test_gpu_cathode_matches_oracle_step_for_step[tol1]" \
  -k "not two_gpu and not 65536 and not test_c_example_trains and not loaded_library_was_built and not auto_beyond_one_generation"

#!/bin/bash
# tests/simt/build.sh -- TEST INFRASTRUCTURE: compiles crnn_amd/csrc/crnn_capi.hip (the product's one translation unit, unchanged) as host
# C++ against the SIMT emulation shim in this directory -> tests/simt/libcrnn_simt.so (same C ABI as libcrnn_hip.so).
#   -ffp-contract=on -mfma: a*b+c contracts where the device build contracts it (crnn_amd/csrc/Makefile)
set -e
D=$(cd $(dirname $0) && pwd); R=$(cd $D/../.. && pwd)
CXX=${SIMT_CXX:-/opt/rocm/lib/llvm/bin/clang++}
OUT=${SIMT_OUT:-$D/libcrnn_simt.so}      # SIMT_OUT + SIMT_FLAGS="-fsanitize=address": the sanitizer build of tools/simt_asan.sh
HASH=$(cat $R/crnn_amd/csrc/*.hip $R/crnn_amd/csrc/*.hpp $R/include/crnn_hip.h $D/hip/hip_runtime.h $D/rccl/rccl.h $D/build.sh | sha256sum | cut -c1-16)
if [ -f $OUT ] && [ "$(cat $OUT.srchash 2>/dev/null)" = "$HASH" ] && [ -z "$SIMT_FORCE" ]; then exit 0; fi
$CXX -x c++ -std=c++17 ${SIMT_OPT:--O1} -g1 -fPIC -shared -ffp-contract=on -mfma -fno-strict-aliasing -pthread \
  -Wno-unknown-attributes -Wno-ignored-attributes -Wno-unused-value -Wno-macro-redefined -Wno-keyword-macro -Wno-builtin-macro-redefined \
  -I $D -DCRNN_SRC_HASH="\"SIMT-EMULATION-$HASH\"" -DCRNN_SIMT_EMULATION=1 $SIMT_FLAGS \
  -o $OUT.tmp $R/crnn_amd/csrc/crnn_capi.hip
mv $OUT.tmp $OUT
echo $HASH > $OUT.srchash

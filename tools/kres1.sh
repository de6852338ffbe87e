#!/bin/bash
# usage: kres1.sh <pattern> [extra flags]  -> resource lines for matching kernels (compiles crnn_capi.hip)
pat=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on "$@" -Rpass-analysis=kernel-resource-usage -c crnn_capi.hip -o /tmp/x_$$.o 2>&1 | grep -E "remark|error" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | awk '/Function Name/{if(line)print line; line=$0; next}{line=line" | "$0}END{print line}' | sed 's/Dynamic Stack: False | //; s/TotalSGPRs/SGPR/; s/ScratchSize \[bytes\/lane\]/scratch/; s/Occupancy \[waves\/SIMD\]/occ/; s/LDS Size \[bytes\/block\]/lds/; s/Function Name: //; s/_ZN4crnn//; s/EvNS_11SolveParams.*AdjParamsE//' | grep -E "$pat" | cut -c1-220
rm -f /tmp/x_$$.o

#!/bin/bash
# usage: tools/kres_one.sh <file.hip> [extra flags]  -- register / scratch / LDS report of the kernels a small translation unit
# instantiates (seconds, against ~100 s for crnn_capi.hip): include one kernel header, explicitly instantiate what is of interest
f=$1; shift
cd $(dirname $0)/../crnn_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -I. "$@" -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/kres_one_$$.o 2>&1 | grep -E "error|Function Name|VGPRs:|AGPRs|Scratch|Occupancy|SGPRs Spill|LDS" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//; s/Function Name: _ZN4crnn[0-9]*//; s/ScratchSize \[bytes\/lane\]/scratch/; s/Occupancy \[waves\/SIMD\]/occ/; s/LDS Size \[bytes\/block\]/lds/' | paste - - - - - - - | cut -c1-200
rm -f /tmp/kres_one_$$.o

#!/bin/bash
# usage: tools/kloop.sh <file.hip> <min loop depth> [flags] -- compiles a small translation unit to gfx950 assembly and prints the
# instruction mix of its loop blocks at or below that depth (tools/isa_blocks.py): a static per-iteration instruction count
f=$1; d=$2; shift 2
cd $(dirname $0)/../crnn_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -I. "$@" -S --cuda-device-only -o /tmp/kloop_$$.s $f 2>&1 | grep -E "error" 
python ../../tools/isa_blocks.py /tmp/kloop_$$.s $d | awk '$2=="depth"{n+=$5; fp+=$7; ds+=$9; rl+=$13; acc+=$15; sm+=$17; print} END{print "SUM n",n,"fp64",fp,"ds",ds,"rdlane",rl,"acc",acc,"smem",sm}' | tail -${KLOOP_TAIL:-12}
grep -E "scratch_(load|store)" /tmp/kloop_$$.s | wc -l
rm -f /tmp/kloop_$$.s

#!/usr/bin/env python3
"""Extract one kernel's gfx950 assembly from a device-only -S dump and print its per-loop instruction mix.
   hipcc -O3 ... -S --cuda-device-only -o /tmp/capi.s crnn_amd/csrc/crnn_capi.hip
   python tools/kisa.py /tmp/capi.s ros23_adj2_kernelILi6 [min_depth]   (writes /tmp/k_<pattern>.s)"""
import re, subprocess, sys, os
s = open(sys.argv[1]).read()
pat = sys.argv[2]
m = re.search(r'^(_ZN[^\n:]*' + re.escape(pat) + r'[^\n:]*):', s, re.M)
if not m:
    sys.exit("kernel not found")
a = m.start(1)
e = s.find('.Lfunc_end', a)
out = f"/tmp/k_{pat}.s"
open(out, 'w').write(s[a:e])
print(m.group(1), s[a:e].count('\n'), "lines ->", out)
subprocess.call([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "isa_blocks.py"), out, sys.argv[3] if len(sys.argv) > 3 else "1"])

#!/usr/bin/env python3
"""Where a kernel's scratch traffic is: every scratch_load / scratch_store of the gfx950 assembly (compiled with -gline-tables-only) attributed
to the source line chain of the instruction it feeds, with the loop depth it sits at.  Constants of polynomial chains that the compiler
materialised once per kernel and reloads at each use show up as reloads inside math helpers (cure: ros23_kernel.hpp sconst()); arrays that
did not fit show up at their own lines.
  python tools/isa_scratch_sites.py k.s [kernel substring]"""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else ""
i = 0
while i < len(lines):
    l = lines[i]
    mk = re.match(r"^(_Z\S+):", l)
    if mk and want in mk.group(1):
        name = mk.group(1)
        end = next(k for k in range(i, len(lines)) if lines[k].strip().startswith(".Lfunc_end"))
        depth = 0; loc = ""; sites = Counter(); n = 0
        for k in range(i, end):
            s = lines[k]
            m = re.search(r"Loop Header: Depth=(\d+)", s) or re.search(r"in Loop: Header=BB\d+_\d+ Depth=(\d+)", s)
            if re.match(r"^(\.LBB\d+_\d+):", s) or re.match(r"^; %bb\.\d+:", s): depth = int(m.group(1)) if m else 0
            elif m: depth = max(depth, int(m.group(1)))
            t = s.strip()
            if t.startswith(".loc"):
                loc = t.split(";", 1)[1].strip() if ";" in t else t
                continue
            if t.startswith("scratch_"):
                n += 1
                short = re.sub(r"/opt/rocm[^ ]*/", "", loc); short = re.sub(r":\d+( |$)", r"\1", short)
                sites[(depth, t.split()[0].replace("_dwordx2", "").replace("_dword", ""), short)] += 1
        if n:
            print(f"== {name[:110]}  scratch instructions: {n}")
            for (d, op, where), c in sorted(sites.items(), key=lambda x: (-x[0][0], -x[1]))[:14]:
                print(f"   depth {d} {op:14s} x{c:3d}  {where[:170]}")
        i = end
    i += 1

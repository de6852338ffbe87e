#!/usr/bin/env python3
"""Static instruction table of a kernel's loops by source phase, from its gfx950 assembly compiled with -gline-tables-only: every instruction at
loop depth >= D is attributed to the OUTERMOST line of the kernel's source file in its inlined-at chain (a helper inlined at a call site counts for
the call site's phase) and the lines are cut into phases at the given boundaries.
  python tools/isa_phase_table.py k.s <kernel substring> <source file> --depth 2 --phase 337="loop control" --phase 347="factor + solve" ...
(phase NAME covers the lines after the previous boundary up to and including its own).  Issue slots per pass through the loops, both sides of
every branch, no trip counts: a table of WHERE the instructions are, not of time."""
import argparse, re
from collections import Counter, OrderedDict
ap = argparse.ArgumentParser()
ap.add_argument("asm"); ap.add_argument("kernel"); ap.add_argument("source")
ap.add_argument("--depth", type=int, default=2)
ap.add_argument("--phase", action="append", default=[])
a = ap.parse_args()
bounds = []
for ph in a.phase:
    ln, name = ph.split("=", 1)
    bounds.append((int(ln), name))
bounds.sort()
lines = open(a.asm).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and a.kernel in l and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
def kind(op):
    if op.startswith("v_") and "f64" in op: return "fp64"
    if op.startswith("ds_"): return "lds"
    if op.startswith("scratch_"): return "scratch"
    if "accvgpr" in op: return "acc"
    if op.startswith(("global_", "flat_", "buffer_")): return "vmem"
    if op.startswith("s_"): return "salu"
    return "valu"
tab = OrderedDict((n, Counter()) for _, n in bounds)
tab["(elsewhere)"] = Counter()
depth = 0; cur = None
pat = re.compile(re.escape(a.source) + r":(\d+)")
for l in lines[start:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", l) or re.match(r"^; (%bb\.\d+):", l)
    if m:
        depth = 0
        mi = re.search(r"Depth=(\d+)", l)
        if mi: depth = int(mi.group(1))
    mi = re.search(r"Loop Header: Depth=(\d+)", l) or re.search(r"in Loop: Header=BB\d+_\d+ Depth=(\d+)", l)
    if mi: depth = max(depth, int(mi.group(1)))
    ls = l.strip()
    if ls.startswith(".loc"):
        mm = pat.findall(l)
        if mm: cur = int(mm[-1])
        continue
    m2 = re.match(r"^\s+([a-z_0-9]+)", l)
    if not m2 or ls.startswith((".", ";")) or depth < a.depth or cur is None: continue
    name = "(elsewhere)"
    lo = 0
    for b, n in bounds:
        if lo < cur <= b: name = n; break
        lo = b
    if cur <= (bounds[0][0] - 60 if bounds else 0): name = "(elsewhere)"
    tab[name][kind(m2.group(1))] += 1
tot = Counter()
print(f"{'phase':52s} {'all':>6s} {'fp64':>6s} {'valu':>6s} {'lds':>5s} {'acc':>5s} {'vmem':>5s} {'salu':>6s} {'scratch':>7s}")
for n, c in tab.items():
    al = sum(c.values()); tot += c
    print(f"{n:52s} {al:6d} {c['fp64']:6d} {c['valu']:6d} {c['lds']:5d} {c['acc']:5d} {c['vmem']:5d} {c['salu']:6d} {c['scratch']:7d}")
al = sum(tot.values())
print(f"{'total':52s} {al:6d} {tot['fp64']:6d} {tot['valu']:6d} {tot['lds']:5d} {tot['acc']:5d} {tot['vmem']:5d} {tot['salu']:6d} {tot['scratch']:7d}")

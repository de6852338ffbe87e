#!/usr/bin/env python3
"""Static cost of one pass through a loop of a kernel, from its gfx950 assembly compiled with -gline-tables-only: the instructions of
the loop at depth D (issue slots per wavefront; FP64 arithmetic, LDS, register-file copies and scratch among them), inner loops weighted
by trip counts given per source-line range of their header.
  python tools/isa_attempt_cost.py k.s <kernel name substring> <source file> [--depth 2] [--trip 300-320=10 ...] [--list]
An estimate of ISSUE, not of time: no latencies, both sides of every branch counted, FP64 VALU operations take four cycles each on a
SIMD (the `cycles` column charges them so, everything else one)."""
import argparse
import re
from collections import Counter

ap = argparse.ArgumentParser()
ap.add_argument("asm"); ap.add_argument("kernel"); ap.add_argument("source")
ap.add_argument("--depth", type=int, default=2)
ap.add_argument("--trip", action="append", default=[])
ap.add_argument("--list", action="store_true")
a = ap.parse_args()
trips = []
for t in a.trip:
    rng, n = t.split("=")
    lo, hi = rng.split("-")
    trips.append((int(lo), int(hi), float(n)))
lines = open(a.asm).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and a.kernel in l and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
fileno = None
for l in lines[:end]:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m and (m.group(3) or m.group(2)).endswith(a.source): fileno = m.group(1)


def kind(op):
    if op.startswith("v_") and "f64" in op: return "fp64"
    if op.startswith("ds_"): return "lds"
    if op.startswith("scratch_"): return "scratch"
    if "accvgpr" in op: return "acc"
    if op.startswith(("global_", "flat_", "buffer_")): return "vmem"
    if op.startswith("s_"): return "salu"
    return "valu"


# blocks: label, depth, header label of the innermost loop they belong to
blocks = []
cur = None
loc = None
for l in lines[start:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", l) or re.match(r"^; (%bb\.\d+):", l)      # (fall-through blocks carry no label, only this comment)
    if m:
        cur = dict(label=m.group(1).lstrip(".L"), depth=0, hdr="", ops=Counter(), line=None)
        mi = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", l)
        if mi: cur["depth"] = int(mi.group(2)); cur["hdr"] = mi.group(1)
        blocks.append(cur)
        mh = re.search(r"Loop Header: Depth=(\d+)", l)
        if mh: cur["depth"] = int(mh.group(1)); cur["hdr"] = cur["label"]
        continue
    if cur is None:
        cur = dict(label="entry", depth=0, hdr="", ops=Counter(), line=None); blocks.append(cur)
    mh = re.search(r";\s+=>.*Loop Header: Depth=(\d+)", l)
    if mh and int(mh.group(1)) >= cur["depth"]: cur["depth"] = int(mh.group(1)); cur["hdr"] = cur["label"]
    m = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", l)
    if m and int(m.group(2)) > cur["depth"]: cur["depth"] = int(m.group(2)); cur["hdr"] = m.group(1)
    ls = l.strip()
    if ls.startswith(".loc"):
        p = ls.split()
        if p[1] == fileno:
            loc = int(p[2])
            if cur["line"] is None: cur["line"] = loc
        continue
    m2 = re.match(r"^\s+([a-z_0-9]+)", l)
    if m2 and not ls.startswith((".", ";")): cur["ops"][kind(m2.group(1))] += 1
# innermost-loop groups
groups = {}
for b in blocks:
    if b["depth"] < a.depth: continue
    g = groups.setdefault((b["depth"], b["hdr"]), dict(ops=Counter(), line=None))
    g["ops"] += b["ops"]
    if g["line"] is None: g["line"] = b["line"]
tot = Counter()
for (d, h), g in sorted(groups.items(), key=lambda x: (x[0][0], x[1]["line"] or 0)):
    w = 1.0
    if d > a.depth:
        w = None
        for lo, hi, n in trips:
            if g["line"] is not None and lo <= g["line"] <= hi: w = n
        if w is None: w = 1.0; note = "  (no --trip given: counted once)"
        else: note = ""
    else: note = ""
    n = sum(g["ops"].values())
    if a.list and n > 8:
        print(f"depth {d} header {h:10s} line {g['line']}  n {n:5d} fp64 {g['ops']['fp64']:4d} lds {g['ops']['lds']:4d} acc {g['ops']['acc']:4d} scratch {g['ops']['scratch']:3d}  x {w:g}{note}")
    for k, v in g["ops"].items(): tot[k] += w * v
allc = sum(tot.values())
cyc = allc + 3 * tot["fp64"]
print(" ".join(f"{k} {tot[k]:.0f}" for k in ("fp64", "valu", "lds", "acc", "scratch", "vmem", "salu")), f"| all {allc:.0f} cycles>= {cyc:.0f}")

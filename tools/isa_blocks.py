#!/usr/bin/env python3
"""per-basic-block instruction mix of one kernel's gfx950 assembly: python tools/isa_blocks.py k.s [min_depth]"""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split('\n')
mind = int(sys.argv[2]) if len(sys.argv) > 2 else 2
blocks = []; cur = None
for l in lines:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        cur = {'name': m.group(1), 'ins': [], 'depth': 0, 'hdr': ''}; blocks.append(cur); continue
    if cur is None: continue
    m = re.search(r'in Loop: Header=(BB\d+_\d+) Depth=(\d+)', l)
    if m: cur['depth'] = max(cur['depth'], int(m.group(2))); cur['hdr'] = m.group(1)
    m2 = re.match(r'^\s+([a-z_0-9]+)', l)
    if m2 and not l.strip().startswith(('.', ';')): cur['ins'].append(m2.group(1))
tot = Counter()
for b in blocks:
    if b['depth'] >= mind and len(b['ins']) > 20:
        c = Counter(b['ins'])
        fp = sum(v for k, v in c.items() if k.startswith(('v_fma', 'v_add_f64', 'v_mul_f64', 'v_fmac')))
        print(b['name'], 'depth', b['depth'], 'n', len(b['ins']), 'fp64', fp, 'ds', sum(v for k, v in c.items() if k.startswith('ds_')),
              'cnd', sum(v for k, v in c.items() if 'cndmask' in k), 'rdlane', c.get('v_readlane_b32', 0),
              'acc', c.get('v_accvgpr_read_b32', 0) + c.get('v_accvgpr_write_b32', 0), 'smem', sum(v for k, v in c.items() if k.startswith('s_load')),
              'wait', c.get('s_waitcnt', 0), 'nop', c.get('s_nop', 0))
        tot += c
print('total', sum(tot.values()))
for k, v in tot.most_common(22): print(v, k)

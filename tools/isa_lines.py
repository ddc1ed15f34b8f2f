#!/usr/bin/env python3
"""Static instruction mix per source line of one kernel (hipcc -gline-tables-only -S output).
   python tools/isa_lines.py k4g.s '_Z18burst_pull4_kernelILb0ELb0E' """
import re, sys, collections
path, prefix = sys.argv[1], sys.argv[2]
on = False
cur = (0, 0)
cnt = collections.defaultdict(lambda: collections.Counter())
for ln in open(path):
    if not on:
        if ln.startswith(prefix) and ':' in ln: on = True
        continue
    t = ln.strip()
    if t.startswith('.loc'):
        p = t.split()
        cur = (int(p[1]), int(p[2]))
        continue
    if t.startswith('s_endpgm'): break
    if not t or t.startswith(('.', ';')) or t.endswith(':'): continue
    op = t.split()[0]
    if op.startswith('v_'): k = 'valu'
    elif op.startswith('ds_'): k = 'lds'
    elif op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')): k = 'vmem'
    elif op.startswith('s_waitcnt'): k = 'wait'
    elif op.startswith('s_cbranch') or op.startswith('s_branch'): k = 'br'
    elif op.startswith('s_'): k = 'salu'
    else: k = 'other'
    cnt[cur][k] += 1
tot = collections.Counter()
for key in sorted(cnt):
    c = cnt[key]
    tot.update(c)
    print(f"f{key[0]}:{key[1]:4d}  valu {c['valu']:4d} salu {c['salu']:4d} lds {c['lds']:3d} vmem {c['vmem']:3d} wait {c['wait']:3d} br {c['br']:3d}")
print("total", dict(tot))

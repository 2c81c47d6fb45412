#!/usr/bin/env python
"""Instruction mix of the largest loop of one kernel in a gfx950 assembly listing (hipcc --cuda-device-only -S):
   python tools/isa_loop_count.py file.s <mangled-name-prefix>"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
i = s.index(sys.argv[2]); i = s.index(':', i)
j = s.index('s_endpgm', i)
body = s[i:j].splitlines()
labels = {}
for n, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = n
best = None
for n, l in enumerate(body):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < n:
        span = (labels[m.group(1)], n)
        if best is None or span[1] - span[0] > best[1] - best[0]: best = span
print('loop lines', best, 'of', len(body))
c = Counter()
for l in body[best[0]:best[1]]:
    l = l.strip()
    if not l or l.startswith(('.', ';', '//')) or l.endswith(':'): continue
    op = l.split()[0]
    if op.startswith('v_mfma'): c['mfma'] += 1
    elif op.startswith('v_'): c['valu'] += 1; c['v:' + re.sub(r'_e(32|64)|_dpp|_sdwa', '', op)] += 1
    elif op.startswith('ds_'): c['lds'] += 1; c['d:' + op] += 1
    elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): c['vmem'] += 1; c['m:' + op] += 1
    elif op.startswith('s_waitcnt'): c['waitcnt'] += 1
    elif op.startswith('s_'): c['salu'] += 1
    else: c['other:' + op] += 1
print({k: c[k] for k in ['valu', 'mfma', 'lds', 'vmem', 'salu', 'waitcnt']})
for p in 'vdm':
    print(sorted([(v, k[2:]) for k, v in c.items() if k.startswith(p + ':')], reverse=True)[:24])

#!/usr/bin/env python
"""LDS waits of a kernel's main loop: every `s_waitcnt lgkmcnt(N)` with the number of instructions since the last LDS read was
issued -- a wait of 0-1 outstanding a few instructions behind its read is an exposed LDS round trip (~64-128 cycles); the fix
is to issue the reads earlier (registers across the loop for loop-invariant operands, or a batch of reads behind
__builtin_amdgcn_sched_barrier(0): the scheduler otherwise sinks each read to its use).
Usage: python tools/isa_lds_waits.py file.s <mangled-name-prefix>     (file.s: hipcc --cuda-device-only -S)"""
import re, sys
s = open(sys.argv[1]).read()
i = s.index(sys.argv[2]); i = s.index(':', i)
j = s.index('s_endpgm', i)
body = [l.strip().split(';')[0].rstrip() for l in s[i:j].splitlines()]
body = [b for b in body if b]
labels = {m.group(1): n for n, l in enumerate(body) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
best = None
for n, l in enumerate(body):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < n:
        sp = (labels[m.group(1)], n)
        if best is None or sp[1] - sp[0] > best[1] - best[0]: best = sp
print('loop', best)
if best is None:
    sys.exit(0)  # (no loop in this kernel)
last = None
for n in range(best[0], best[1]):
    l = body[n]
    if l.startswith(('ds_read', 'ds_bpermute', 'ds_swizzle')): last = n
    m = re.search(r'lgkmcnt\((\d+)\)', l)
    if m and last is not None and int(m.group(1)) <= 1 and n - last <= 8:
        nxt = next((body[k] for k in range(n + 1, min(n + 4, len(body))) if not body[k].startswith('s_')), '')
        print('%5d  %-28s %2d instr after %-40s then %s' % (n, l, n - last, body[last][:40], nxt[:60]))

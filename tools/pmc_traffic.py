#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), per launch.
Units and corrections as guides/MI355X_MICROARCH.md (HBM section): counters are in KiB; on gfx950 FETCH_SIZE
reports half of the bytes of wide coalesced reads, so the read side is doubled (upper bound for narrow reads).
The first output line is {"_build": ptv2_build_info()} of the library in this tree: bench.py accepts the profile for
its roofline.traffic only when that source digest equals the running build's.
usage: tools/pmc_traffic.py <dir_fetch> <dir_write> [name substrings...]"""
import csv, glob, sys, collections, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ao_amd import _lib
print(json.dumps({"_build": _lib.lib().ptv2_build_info().decode()}))

def load(d, counter):
    f = (glob.glob(d + '/*/*_counter_collection.csv') + glob.glob(d + '/*_counter_collection.csv'))[0]
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter: continue
        a = acc[r['Kernel_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
    return acc

fetch, write = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
pats = sys.argv[3:]
out = {}
for name in fetch:
    if pats and not any(p in name for p in pats): continue
    f, nf = fetch[name]; w, nw = write.get(name, [0.0, 1])
    short = name.split('(')[0].replace('void ', '')
    out[short] = dict(launches=nf, fetch_MB_per_launch=round(f * 1024 / nf / 1e6, 3),
                      fetch_x2_MB_per_launch=round(2 * f * 1024 / nf / 1e6, 3),
                      write_MB_per_launch=round(w * 1024 / max(nw, 1) / 1e6, 3))
for k, v in sorted(out.items(), key=lambda kv: -(kv[1]['fetch_x2_MB_per_launch'] + kv[1]['write_MB_per_launch']) * kv[1]['launches'])[:30]:
    print(json.dumps({k: v}))

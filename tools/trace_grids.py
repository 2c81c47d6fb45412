#!/usr/bin/env python
"""Per (kernel, grid) durations of a rocprofv3 --kernel-trace run: tools/trace_grids.py <dir> <steps> [skip_frac]
Separates the latency-bound deep-stage launches of a kernel from its bandwidth-bound S0 launches."""
import csv, glob, re, sys, collections
d, steps = sys.argv[1], float(sys.argv[2])
skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
f = (glob.glob(d + '/*/*_kernel_trace.csv') + glob.glob(d + '/*_kernel_trace.csv'))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[int(len(rows) * skip):]
steps *= (1 - skip)
per = collections.defaultdict(list)
for r in rows:
    g = tuple(int(r['Grid_Size_' + a]) // max(1, int(r['Workgroup_Size_' + a])) for a in 'XYZ')
    name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')
    per[(name[:60], g)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in per.values())
print("kernel time per step %.3f ms over %.1f steps" % (tot / steps / 1e3, steps))
print("%-60s %-18s %6s %8s %9s" % ("kernel", "grid", "n/step", "avg us", "ms/step"))
for (k, g), v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[4]) if len(sys.argv) > 4 else 70]:
    print("%-60s %-18s %6.1f %8.1f %9.3f" % (k, "x".join(map(str, g)), len(v) / steps, sum(v) / len(v), sum(v) / steps / 1e3))

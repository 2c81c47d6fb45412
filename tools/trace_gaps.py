#!/usr/bin/env python
"""GPU busy / idle analysis of a rocprofv3 --kernel-trace run: tools/trace_gaps.py <dir> [skip_frac]
Prints the busy time, the idle gaps and, per kernel name, launches and time split by workgroup count
(small launches are latency-bound, big ones bandwidth-bound)."""
import csv, glob, sys, collections
d = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
f = (glob.glob(d + '/*/*_kernel_trace.csv') + glob.glob(d + '/*_kernel_trace.csv'))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[int(len(rows) * skip):]  # drop warm-up part
t0, t1 = int(rows[0]['Start_Timestamp']), max(int(r['End_Timestamp']) for r in rows)
busy = 0; gaps = []; cur_end = t0
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s > cur_end:
        gaps.append(s - cur_end); busy += e - s
    else:
        busy += max(0, e - max(s, cur_end))
    cur_end = max(cur_end, e)
span = t1 - t0
print("window %.2f ms, %d launches, busy %.2f ms (%.1f%%), idle %.2f ms" % (span / 1e6, len(rows), busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6))
g = sorted(gaps)
if g:
    print("gaps: n=%d median %.1f us p90 %.1f us max %.1f us; gaps>50us: %d totalling %.2f ms" % (
        len(g), g[len(g) // 2] / 1e3, g[int(len(g) * .9)] / 1e3, g[-1] / 1e3, sum(1 for x in g if x > 50000), sum(x for x in g if x > 50000) / 1e6))
per = collections.defaultdict(lambda: [0, 0.0, 0, 0.0])
for r in rows:
    nwg = (int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))) * max(1, int(r.get('Grid_Size_Y', 1)) // max(1, int(r.get('Workgroup_Size_Y', 1)))) * max(1, int(r.get('Grid_Size_Z', 1)) // max(1, int(r.get('Workgroup_Size_Z', 1))))
    dur = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    p = per[r['Kernel_Name'][:70]]
    if nwg < 512:
        p[0] += 1; p[1] += dur
    else:
        p[2] += 1; p[3] += dur
print("%-70s %7s %9s %7s %9s" % ("kernel", "n<512wg", "ms", "n>=512", "ms"))
for k, p in sorted(per.items(), key=lambda kv: -(kv[1][1] + kv[1][3]))[:45]:
    print("%-70s %7d %9.3f %7d %9.3f" % (k, p[0], p[1] / 1e6, p[2], p[3] / 1e6))

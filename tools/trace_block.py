#!/usr/bin/env python
"""Print the kernel sequence (duration, gap to the previous kernel) around the n-th launch of a marker kernel in a
rocprofv3 kernel trace: tools/trace_block.py <dir> <marker regex> [occurrence] [before] [after] [queue]
(queue: only the kernels of the marker's queue when "same")"""
import csv, glob, re, sys
d, marker = sys.argv[1], sys.argv[2]
occ = int(sys.argv[3]) if len(sys.argv) > 3 else 10
before = int(sys.argv[4]) if len(sys.argv) > 4 else 45
after = int(sys.argv[5]) if len(sys.argv) > 5 else 40
f = (glob.glob(d + '/*/*_kernel_trace.csv') + glob.glob(d + '/*_kernel_trace.csv'))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
same = len(sys.argv) > 6 and sys.argv[6] == "same"
idx = [i for i, r in enumerate(rows) if re.search(marker, r['Kernel_Name'])]
if not idx:
    sys.exit("no kernel matches %r" % marker)
i0 = idx[min(occ, len(idx) - 1)]
if same:  # the marker's queue only (the geometry stream's kernels interleave by time otherwise)
    qid = rows[i0]['Queue_Id']
    rows = [r for r in rows if r['Queue_Id'] == qid]
    idx = [i for i, r in enumerate(rows) if re.search(marker, r['Kernel_Name'])]
    i0 = idx[min(occ, len(idx) - 1)]
prev_end = None
tot = 0
for r in rows[max(0, i0 - before):i0 + after]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    prev_end = e
    tot += (e - s)
    nwg = int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))
    print("%7.1f us  gap %6.1f  wg %6d  %s" % ((e - s) / 1e3, gap, nwg, r['Kernel_Name'][:90]))
print("sum of durations %.1f us" % (tot / 1e3))

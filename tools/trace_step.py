#!/usr/bin/env python
"""Kernel sequence of ONE optimizer step from a rocprofv3 --kernel-trace run (between two launches of the marker kernel):
tools/trace_step.py <dir> [step index] [marker]  ->  start offset us, duration us, queue, workgroups, kernel"""
import csv, glob, re, sys
d = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 8
marker = sys.argv[3] if len(sys.argv) > 3 else "adamw_flat_kernel"
f = (glob.glob(d + '/*/*_kernel_trace.csv') + glob.glob(d + '/*_kernel_trace.csv'))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
a, b = marks[which], marks[which + 1]
t0 = int(rows[a]['End_Timestamp'])
queues = {}
busy = {}
for r in rows[a + 1:b + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    q = queues.setdefault(r['Queue_Id'], len(queues))
    busy[q] = busy.get(q, 0) + (e - s)
    wg = 1
    for ax in 'XYZ':
        wg *= int(r['Grid_Size_' + ax]) // max(1, int(r['Workgroup_Size_' + ax]))
    name = re.sub(r'^void ', '', r['Kernel_Name'])
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'\((int|long|float|double|unsigned|HIP_vector|gva::|dense::|gemm::|Map|at::|char|bool|std::).*', '', name)
    print("%9.1f %7.1f q%d %6d %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, wg, name[:100]))
print("# step %.1f us; launches %d; busy per queue (us): %s" % ((int(rows[b]['End_Timestamp']) - t0) / 1e3, b - a,
                                                              {q: round(v / 1e3, 1) for q, v in busy.items()}))

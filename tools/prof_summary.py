#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats run: tools/prof_summary.py <dir> [steps] [top]"""
import csv, glob, sys
d = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1; top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
f = (glob.glob(d + '/*/*_kernel_stats.csv') + glob.glob(d + '/*_kernel_stats.csv') + glob.glob(d + '/**/*_kernel_stats.csv', recursive=True))[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("kernel time total %.2f ms, per step %.2f ms over %d steps; %d distinct kernels, %d launches/step" % (
    tot / 1e6, tot / 1e6 / steps, steps, len(rows), sum(int(r['Calls']) for r in rows) / steps))
for r in rows[:top]:
    print("%-92s n/step=%6.1f ms/step=%7.3f avg_us=%8.1f %5.1f%%" % (r['Name'][:92], int(r['Calls']) / steps,
          float(r['TotalDurationNs']) / 1e6 / steps, float(r['AverageNs']) / 1e3, float(r['Percentage'])))

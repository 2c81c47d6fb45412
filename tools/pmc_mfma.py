#!/usr/bin/env python
"""MFMA pipe utilisation per kernel from one rocprofv3 pass
   rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d <dir> -- python3 bench.py ...
usage: tools/pmc_mfma.py <dir>
GRBM_GUI_ACTIVE is summed over the 8 XCDs by rocprofv3 (MI355X_MICROARCH.md, DVFS note); SQ_VALU_MFMA_BUSY_CYCLES counts
cycles over all SIMDs, so utilisation = MFMA busy / (GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)."""
import csv, glob, sys, collections
d = sys.argv[1]
f = (glob.glob(d + '/*/*_counter_collection.csv') + glob.glob(d + '/*_counter_collection.csv'))[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'GRBM_GUI_ACTIVE': cnt[k] += 1
rows = []
for k, v in acc.items():
    gui, mf = v.get('GRBM_GUI_ACTIVE', 0), v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0)
    if mf > 0 and gui > 0:
        rows.append((mf, k, cnt[k], mf / (gui / 8 * 256 * 4)))
for mf, k, n, util in sorted(rows, reverse=True)[:20]:
    print("%-58s launches %4d  MFMA busy / (active cycles x 1024 SIMDs) = %5.1f %%" % (k[:58], n, 100 * util))

#!/usr/bin/env python
"""Per-kernel SQ counter table from one rocprofv3 --pmc pass (averages per launch, plus the shares the microarchitecture
guide defines: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES, all in quad-cycles).
   rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \\
             SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d <dir> -- python3 bench.py ...
usage: tools/pmc_sq.py <dir> [kernel name substrings...]"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ao_amd import _lib

d, pats = sys.argv[1], sys.argv[2:]
f = (glob.glob(d + '/*/*_counter_collection.csv') + glob.glob(d + '/*_counter_collection.csv'))[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
    if pats and not any(p in k for p in pats):
        continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[k][r['Counter_Name']] += 1
print(json.dumps({"_build": _lib.lib().ptv2_build_info().decode()}))
for k in sorted(acc, key=lambda k: -acc[k].get('SQ_WAVE_CYCLES', 0)):
    v, n = acc[k], max(cnt[k].values())
    wc = max(v.get('SQ_WAVE_CYCLES', 0), 1)
    row = {"kernel": k, "launches": n}
    for name in sorted(v):
        row[name + "_per_launch"] = round(v[name] / n, 1)
    row["share_wait_any"] = round(v.get('SQ_WAIT_ANY', 0) / wc, 3)
    row["share_wait_inst_any"] = round(v.get('SQ_WAIT_INST_ANY', 0) / wc, 3)
    row["share_active_inst_any"] = round(v.get('SQ_ACTIVE_INST_ANY', 0) / wc, 3)
    row["share_active_inst_valu"] = round(v.get('SQ_ACTIVE_INST_VALU', 0) / wc, 3)
    if v.get('SQ_LDS_IDX_ACTIVE'):
        row["lds_conflict_over_active"] = round(v.get('SQ_LDS_BANK_CONFLICT', 0) / v['SQ_LDS_IDX_ACTIVE'], 3)
    print(json.dumps(row))

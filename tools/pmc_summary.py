#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc counter_collection CSVs by kernel: mean counter value per dispatch.
usage: pmc_summary.py <dir> [substring filter]"""
import csv, glob, sys, collections
d = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:70]
        if flt and flt not in k: continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:44s} n={len(v):3d} mean={sum(v)/len(v):.4g}")

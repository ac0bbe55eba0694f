#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel name."""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void aqc::", "").replace("aqc::", "")
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name in sorted(agg):
    if "stage_kernel" not in name:
        continue
    print(name)
    for c, v in sorted(agg[name].items()):
        print(f"   {c:28s} n={len(v):3d} mean={sum(v) / len(v):16.1f}")

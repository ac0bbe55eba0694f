#!/usr/bin/env python3
"""Per-launch split of a rocprofv3 --kernel-trace CSV: mean duration of the k-th launch of each kernel inside a step
(python tools/trace_split.py <kernel_trace.csv> [kernel substring]); the sweep and V^H kernels run once per stage, so
launches alternate between the stages of the plan."""
import csv
import sys
from collections import defaultdict

path, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "mfma_kernel")
rows = defaultdict(list)
with open(path) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        if pat in name:
            rows[name.split("(")[0]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for name, v in sorted(rows.items()):
    v.sort()
    nst = 2   # stages per plan at the headline shape
    for k in range(nst):
        d = [(e - s) / 1e3 for i, (s, e) in enumerate(v) if i % nst == k]
        d = d[len(d) // 4:]   # drop the warm-up quarter
        print(f"{name} launch {k} of {nst}: n={len(d)} mean {sum(d) / len(d):.1f} us  min {min(d):.1f} us")

#!/usr/bin/env python3
"""Clock a kernel really ran at, from one rocprofv3 pass with --pmc GRBM_GUI_ACTIVE --kernel-trace:
python tools/clock_probe.py <dir with *_counter_collection.csv and *_kernel_trace.csv> [kernel substring].
GRBM_GUI_ACTIVE counts on each of the 8 XCDs; cycles / duration = the shader clock during the launch."""
import csv
import glob
import os
import sys
from collections import defaultdict

d, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "mfma_kernel")
cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
dur = {}
with open(kt) as f:
    for r in csv.DictReader(f):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
cyc = defaultdict(float)
with open(cc) as f:
    for r in csv.DictReader(f):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cyc[r["Dispatch_Id"]] += float(r["Counter_Value"])
rows = defaultdict(list)
for k, (name, s, e) in dur.items():
    if pat in name and k in cyc:
        rows[name.split("(")[0]].append((s, (e - s) / 1e3, cyc[k] / 8.0))
for name, v in sorted(rows.items()):
    v.sort()
    for par in range(2):
        w = [x for i, x in enumerate(v) if i % 2 == par]
        w = w[len(w) // 4:]
        us = sum(x[1] for x in w) / len(w); c = sum(x[2] for x in w) / len(w)
        print(f"{name} launch {par}: n={len(w)} duration {us:.1f} us, cycles per XCD {c:.0f}, clock {c / us / 1e3:.3f} GHz")

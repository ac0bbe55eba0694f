#!/usr/bin/env python3
"""Timeline of the last evaluations in a rocprofv3 --kernel-trace CSV: per kernel of one evaluation its start relative to the
evaluation's first kernel, duration and the gap to the previous kernel's end (python tools/trace_timeline.py <csv> <kernels per eval>)."""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]))
rows.sort()
per = int(sys.argv[2])
tail = rows[-per * 50:]
names = [x[2] for x in tail[:per]]
acc = [[0.0, 0.0, 0.0] for _ in range(per)]
cnt = 0
for e in range(0, len(tail) - per + 1, per):
    ev = tail[e:e + per]
    if [x[2] for x in ev] != names:
        continue
    cnt += 1
    for i, (s, t, nm) in enumerate(ev):
        acc[i][0] += (s - ev[0][0]) / 1e3
        acc[i][1] += (t - s) / 1e3
        acc[i][2] += ((s - ev[i - 1][1]) / 1e3) if i else 0.0
print(f"{cnt} evaluations averaged")
for i, nm in enumerate(names):
    print(f"{nm:42s} start {acc[i][0] / cnt:8.2f} us  duration {acc[i][1] / cnt:7.2f} us  gap before {acc[i][2] / cnt:6.2f} us")
last = tail[-per:]
print(f"first start -> last end: {sum((tail[e + per - 1][1] - tail[e][0]) for e in range(0, len(tail) - per + 1, per)) / (len(tail) // per) / 1e3:.2f} us")

#!/usr/bin/env python3
"""Mean per-launch value of every counter of rocprofv3 --pmc CSVs, per kernel (and per distinct grid when asked)."""
import collections
import csv
import sys


def main(paths):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for path in paths:
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k in sorted(acc):
        if "aqc::" not in k:
            continue
        print(f"{k}   (mean duration under the profiler {sum(dur[k]) / len(dur[k]):.1f} us)")
        for c in sorted(acc[k]):
            v = acc[k][c]
            print(f"   {c:28s} n={len(v):3d} mean={sum(v) / len(v):16.1f}")


if __name__ == "__main__":
    main(sys.argv[1:])

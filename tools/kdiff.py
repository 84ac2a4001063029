#!/usr/bin/env python3
"""Per-kernel difference of two rocprofv3 kernel_stats.csv runs (total time per kernel name, us):
    python tools/kdiff.py <dirA> <dirB> [min_abs_diff_us]"""
import csv
import glob
import os
import re
import sys


def load(path):
    files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True)
    out = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            m = re.search(r"mvster::(\w+)(<[^>]*>)?", r["Name"])
            name = (m.group(1) + (m.group(2) or "")) if m else r["Name"].split("(")[0][:90]
            c, t = out.get(name, (0, 0.0))
            out[name] = (c + int(r["Calls"]), t + float(r["TotalDurationNs"]) / 1e3)
    return out


a, b = load(sys.argv[1]), load(sys.argv[2])
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 50.0
rows = []
for k in set(a) | set(b):
    ca, ta = a.get(k, (0, 0.0))
    cb, tb = b.get(k, (0, 0.0))
    if abs(ta - tb) >= thr:
        rows.append((tb - ta, k, ca, ta, cb, tb))
for d, k, ca, ta, cb, tb in sorted(rows):
    print("%-70s A %5d x %10.1f us   B %5d x %10.1f us   B-A %+10.1f" % (k[:70], ca, ta, cb, tb, d))
print("total A %.1f us, B %.1f us" % (sum(t for _, t in a.values()), sum(t for _, t in b.values())))

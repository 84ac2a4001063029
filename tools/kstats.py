#!/usr/bin/env python3
"""Condense a rocprofv3 kernel_stats.csv to short mvster kernel names:  python tools/kstats.py <dir-or-csv> [min_us]"""
import csv
import glob
import os
import re
import sys

path = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True)
for f in files:
    for r in csv.DictReader(open(f)):
        avg = float(r["AverageNs"]) / 1e3
        if avg < min_us:
            continue
        m = re.search(r"mvster::(\w+)(<[^>]*>)?", r["Name"])
        name = (m.group(1) + (m.group(2) or "")) if m else r["Name"].split("(")[0][:70]
        print("%-56s calls %4s  avg %9.1f us  total %9.1f us" % (name[:56], r["Calls"], avg, float(r["TotalDurationNs"]) / 1e3))

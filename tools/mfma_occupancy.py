#!/usr/bin/env python3
"""Matrix-core occupancy per kernel from one rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA ... --kernel-trace pass:
MFMA instructions x their measured busy cycles, over 1024 SIMDs x the kernel's duration at 2.4 GHz.
python tools/mfma_occupancy.py gpurun_out/pmc_<tag>"""
import collections
import csv
import glob
import re
import sys

d = sys.argv[1].rstrip("/")
ct = glob.glob(d + "/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: [0.0, 0.0, 0.0, 0])          # MFMA instructions, ns, MFMA busy cycles, launches
for r in csv.DictReader(open(ct)):
    m = re.search(r"mvster::(\w+(<[^>]*>)?)", r["Kernel_Name"])
    a = agg[m.group(1) if m else r["Kernel_Name"][:48]]
    if r["Counter_Name"] == "SQ_INSTS_MFMA":
        a[0] += float(r["Counter_Value"])
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a[3] += 1
    elif r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
        a[2] += float(r["Counter_Value"])
print("%-46s %8s %14s %16s %10s" % ("kernel", "launches", "MFMA / launch", "busy cyc / MFMA", "occupancy"))
for k, a in sorted(agg.items()):
    if a[0] > 0:
        print("%-46s %8d %14.3g %16.1f %10.2f" % (k, a[3], a[0] / a[3], a[2] / a[0], a[2] / (1024 * a[1] * 2.4)))
print("occupancy = MFMA busy cycles / (1024 SIMDs x kernel duration x 2.4 GHz); kernels without MFMA instructions omitted")

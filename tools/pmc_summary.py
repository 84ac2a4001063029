#!/usr/bin/env python3
"""Condense rocprofv3 --pmc counter CSVs (one counter per run, MI355X_MICROARCH.md "rocprofv3 PMC slots") into a
small JSON keyed by kernel name:  python tools/pmc_summary.py gpurun_out/pmc_* -o profiles/r1_pmc.json

gfx950 corrections applied when deriving bytes (MI355X_MICROARCH.md "HBM"): FETCH_SIZE and WRITE_SIZE are in KiB;
FETCH_SIZE reports exactly half of the bytes of a coalesced stream (checked here on our own layout-conversion and
softmax kernels, whose byte counts are known: ratio 2.00); WRITE_SIZE is exact for streaming stores and counts the
dwords of float atomics."""
import argparse
import collections
import csv
import glob
import json
import os
import re


def short(name):
    m = re.search(r"mvster::(\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name.split("(")[0][:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("-o", "--out", required=True)
    args = ap.parse_args()
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in args.dirs:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                per[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, counters in per.items():
        e = {"launches": max(len(v) for v in counters.values())}
        for c, v in counters.items():
            e[c + "_avg"] = sum(v) / len(v)
        if "FETCH_SIZE_avg" in e and "WRITE_SIZE_avg" in e:
            e["hbm_read_bytes_per_launch"] = 2.0 * e["FETCH_SIZE_avg"] * 1024
            e["hbm_write_bytes_per_launch"] = e["WRITE_SIZE_avg"] * 1024
            e["hbm_bytes_per_launch"] = e["hbm_read_bytes_per_launch"] + e["hbm_write_bytes_per_launch"]
        out[k] = e
    # tie the summary to the kernel sources it was measured on: bench.py reports roofline.traffic only while they match
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    try:
        head = subprocess.run(["git", "-C", root, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
    except OSError:
        head = ""
    out["_meta"] = {"csrc_sha": bench.source_fingerprint(), "head_at_collection": head or os.environ.get("MVSTER_HEAD", "")}
    json.dump(out, open(args.out, "w"), indent=1, sort_keys=True)
    print("wrote", args.out, len(out), "kernels")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel over consecutive GROUPS of dispatches (a probe that launches the same kernel
N times per case):  python tools/pmc_groups.py gpurun_out/pmc_x --kernel fwd_nhwc --group 12 --names noise4,noise1,..."""
import argparse
import collections
import csv
import glob
import os


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--kernel", required=True)
    ap.add_argument("--group", type=int, required=True)
    ap.add_argument("--names", default="")
    args = ap.parse_args()
    names = args.names.split(",") if args.names else []
    table = collections.defaultdict(dict)
    for d in args.dirs:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            per = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if args.kernel in r["Kernel_Name"]:
                    per[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
            for c, vals in per.items():
                vals.sort()
                for gi in range(0, len(vals) // args.group):
                    chunk = [v for _, v in vals[gi * args.group:(gi + 1) * args.group]]
                    table[names[gi] if gi < len(names) else "g%d" % gi][c] = sum(chunk) / len(chunk)
    ctrs = sorted({c for v in table.values() for c in v})
    print("%-10s" % "case" + "".join("%24s" % c[-23:] for c in ctrs))
    for k, v in table.items():
        print("%-10s" % k + "".join("%24.4g" % v.get(c, float("nan")) for c in ctrs))


if __name__ == "__main__":
    main()

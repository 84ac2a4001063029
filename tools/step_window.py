#!/usr/bin/env python3
"""Kernel split of the LAST step of a rocprofv3 --kernel-trace run (the warm-up steps hold MIOpen's solver search):
python tools/step_window.py <kernel_trace.csv> [--ms 56] [--top 40]  -> kernel time, busy fraction, per-kernel totals."""
import argparse
import collections
import csv

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--ms", type=float, default=56.0, help="length of the window before the last kernel's end")
ap.add_argument("--top", type=int, default=40)
args = ap.parse_args()
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(args.trace)))
end = max(e for _, e, _ in ev)
lo = end - args.ms * 1e6
agg, cnt = collections.Counter(), collections.Counter()
busy, cursor = 0, lo
for s, e, n in ev:
    if s < lo:
        continue
    agg[n[:100]] += e - s
    cnt[n[:100]] += 1
    if e > cursor:
        busy += e - max(s, cursor)
        cursor = e
tot = sum(agg.values())
print("window %.1f ms: %d launches, kernel time %.2f ms, GPU busy %.2f ms (%.0f %%)"
      % (args.ms, sum(cnt.values()), tot / 1e6, busy / 1e6, 100.0 * busy / (args.ms * 1e6)))
for n, t in agg.most_common(args.top):
    print("%8.3f ms %5d  %s" % (t / 1e6, cnt[n], n))

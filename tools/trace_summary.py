#!/usr/bin/env python3
"""rocprofv3 --kernel-trace of a harness run: per kernel name the launches, total and average duration, and the idle gap in front of
each launch (host round trip + launch latency: what a latency-bound round really costs).
usage: python tools/trace_summary.py <dir> [last_ms]   (last_ms: only the dispatches of the final N milliseconds = the last pass)"""
import collections
import csv
import glob
import sys

fs = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r["Start_Timestamp"]))
if len(sys.argv) > 2:
    tend = int(rows[-1]["End_Timestamp"]); rows = [r for r in rows if int(r["Start_Timestamp"]) >= tend - float(sys.argv[2]) * 1e6]
agg = collections.OrderedDict()
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sbn::", "")[:44]
    a = agg.setdefault(name, [0, 0.0, 0.0])
    a[0] += 1; a[1] += (e - s) / 1e3
    if prev_end is not None:
        a[2] += max(0.0, (s - prev_end) / 1e3)
    prev_end = max(prev_end or 0, e)
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
print(f"{'kernel':44s} {'launches':>8s} {'total_us':>10s} {'avg_us':>8s} {'gap_before_total_us':>20s} {'avg_gap_us':>10s}")
for k, (n, t, g) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print(f"{k:44s} {n:8d} {t:10.1f} {t / n:8.1f} {g:20.1f} {g / n:10.1f}")
tk = sum(v[1] for v in agg.values()); tg = sum(v[2] for v in agg.values())
print(f"span {span / 1e3:.2f} ms: kernels {tk / 1e3:.2f} ms, gaps {tg / 1e3:.2f} ms, {sum(v[0] for v in agg.values())} launches")

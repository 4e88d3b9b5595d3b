#!/usr/bin/env python3
"""rocprofv3 --kernel-trace of a harness run: per kernel name the launches, total and average duration, and the idle gap in front of
each launch (host round trip + launch latency: what a latency-bound round really costs).
usage: python tools/trace_summary.py <dir> [last_ms | pass | stages]
  last_ms  only the dispatches of the final N milliseconds
  pass     the LAST pass of the harness, cut at its pass-start marker (tools/trace_harness.py runs with marker launches: k_scalars_synthetic
           with 2 / 3 / 4 blocks = pass start / timed stage begins / timed stage ends; the markers themselves are left out)
  stages   (default when markers are present) the last pass, and only what lies INSIDE its timed stages — what prove_stages' total adds up:
           the synthetic-input generation between the stages is not part of a prove"""
import collections
import csv
import glob
import sys

fs = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r["Start_Timestamp"]))
mode = sys.argv[2] if len(sys.argv) > 2 else "stages"


def marker(r):
    if "k_scalars_synthetic" not in r["Kernel_Name"]:
        return 0
    try:
        g = int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0); w = int(r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or 256)
    except ValueError:
        return 0
    blocks = g // max(w, 1) if g >= w else g
    return blocks if blocks in (2, 3, 4) else 0


marks = [marker(r) for r in rows]
have = any(m == 2 for m in marks)
windows = None
if mode not in ("pass", "stages"):
    tend = int(rows[-1]["End_Timestamp"]); rows = [r for r in rows if int(r["Start_Timestamp"]) >= tend - float(mode) * 1e6]
    marks = [marker(r) for r in rows]
elif have:
    start = max(i for i, m in enumerate(marks) if m == 2)
    rows, marks = rows[start:], marks[start:]
    if mode == "stages":
        windows = []; depth = 0
        for r, m in zip(rows, marks):
            if m == 3:
                if depth == 0:
                    t0 = int(r["End_Timestamp"])
                depth += 1
            elif m == 4 and depth:
                depth -= 1
                if depth == 0:
                    windows.append((t0, int(r["Start_Timestamp"])))
else:
    print("(no marker launches in this trace: whole trace)")
agg = collections.OrderedDict()
prev_end = None
span = 0.0
nl = 0


def account(rs):
    global prev_end, nl
    for r in rs:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sbn::", "")[:44]
        a = agg.setdefault(name, [0, 0.0, 0.0])
        a[0] += 1; a[1] += (e - s) / 1e3; nl += 1
        if prev_end is not None:
            a[2] += max(0.0, (s - prev_end) / 1e3)
        prev_end = max(prev_end or 0, e)


body = [r for r, m in zip(rows, marks) if not m]
if windows:
    for (w0, w1) in windows:
        prev_end = w0
        account([r for r in body if w0 <= int(r["Start_Timestamp"]) < w1])
        span += (w1 - w0) / 1e3
    what = f"the {len(windows)} timed stages of the last pass"
else:
    account(body)
    span = (int(body[-1]["End_Timestamp"]) - int(body[0]["Start_Timestamp"])) / 1e3 if body else 0.0
    what = "last pass" if (have and mode == "pass") else "selected dispatches"
print(f"{'kernel':44s} {'launches':>8s} {'total_us':>10s} {'avg_us':>8s} {'gap_before_total_us':>20s} {'avg_gap_us':>10s}")
for k, (n, t, g) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print(f"{k:44s} {n:8d} {t:10.1f} {t / n:8.1f} {g:20.1f} {g / n:10.1f}")
tk = sum(v[1] for v in agg.values()); tg = sum(v[2] for v in agg.values())
print(f"{what}: span {span / 1e3:.2f} ms: kernels {tk / 1e3:.2f} ms, gaps {tg / 1e3:.2f} ms, {nl} launches  (kernel-trace profiling adds a few microseconds to every launch: the untraced prove is ~20 % shorter)")

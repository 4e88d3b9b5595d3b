#!/usr/bin/env python3
"""print the kernel sequence of a rocprofv3 --kernel-trace run: start offset, duration and the gap to the previous kernel's end
usage: python tools/trace_rounds.py <dir with *_kernel_trace.csv> [first_row] [rows]"""
import csv, glob, sys
fs = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else max(0, len(rows) - 80)
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 80
prev_end = None; t0 = int(rows[first]["Start_Timestamp"])
for r in rows[first:first + cnt]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sbn::", "")[:34]
    print(f"{(s - t0) / 1e3:9.1f} us  {name:34s} grid {r['Grid_Size_X']:>7s}x{r['Grid_Size_Y']:>3s} wg {r['Workgroup_Size_X']:>4s} vgpr {r['VGPR_Count']:>3s}  dur {(e - s) / 1e3:7.1f}  gap {gap:6.1f}")
    prev_end = e

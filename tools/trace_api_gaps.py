#!/usr/bin/env python3
"""What the host did between two kernels: joins rocprofv3's kernel trace and HIP runtime-API trace of a harness run and prints, for
the last N occurrences of a kernel name, the API calls (name, duration) issued between the end of the previous kernel and its start.
usage: trace_api_gaps.py <dir> <kernel substring> [count]"""
import csv
import glob
import sys

d, name = sys.argv[1], sys.argv[2]
count = int(sys.argv[3]) if len(sys.argv) > 3 else 3
kt = sorted(csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])), key=lambda r: int(r["Start_Timestamp"]))
api = sorted(csv.DictReader(open(glob.glob(d + "/**/*hip_api_trace.csv", recursive=True)[0])), key=lambda r: int(r["Start_Timestamp"]))
hits = [i for i, r in enumerate(kt) if name in r["Kernel_Name"]][-count:]
for i in hits:
    prev_end = int(kt[i - 1]["End_Timestamp"]); start = int(kt[i]["Start_Timestamp"])
    print(f"--- {kt[i]['Kernel_Name'].split('(')[0][-30:]}: gap {(start - prev_end) / 1e3:.1f} us after {kt[i - 1]['Kernel_Name'].split('(')[0][-30:]}")
    for a in api:
        s, e = int(a["Start_Timestamp"]), int(a["End_Timestamp"])
        if e >= prev_end - 20000 and s <= start:
            print(f"   {(s - prev_end) / 1e3:8.1f} us  {a['Function']:32s} {(e - s) / 1e3:7.1f} us")

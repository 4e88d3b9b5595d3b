#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof_*/) into the tracked summaries under profiles/.

  python tools/summarize_prof.py <prof_dir> <tag> <workload> <size>

Reads <prof_dir>/trace*/**/_kernel_stats.csv (rocprofv3 --kernel-trace --stats) and the two separate PMC passes
<prof_dir>/pmc_fetch, <prof_dir>/pmc_write (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE; they cannot share a pass: TCC has 4
slots, FETCH_SIZE takes 3 and WRITE_SIZE 2 — MI355X_MICROARCH.md).  Units and corrections as that guide prescribes:
counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes actually fetched, so the read side is doubled.
Writes profiles/<tag>[_serial]_kernel_stats.csv, profiles/<tag>_pmc.csv and updates profiles/pmc_traffic.json: one entry per
(workload, size, kernel) — bench.py only quotes an entry whose workload AND size match the run.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("sbn::", "")
    return n


def main():
    prof_dir, tag, workload, size = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_dir = os.path.join(root, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    for sub, suffix in (("trace", ""), ("trace_serial", "_serial")):
        # gpurun merges every run's files into the same local directory: take the NEWEST run's file, never the first the glob returns
        stats = sorted(glob.glob(os.path.join(prof_dir, sub, "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)
        if stats:
            rows = list(csv.DictReader(open(stats[-1])))
            with open(os.path.join(out_dir, f"{tag}{suffix}_kernel_stats.csv"), "w", newline="") as f:
                w = csv.writer(f)
                w.writerow(["kernel", "calls", "total_ms", "avg_ms", "percent", "min_ms", "max_ms"])
                for r in rows:
                    if short(r["Name"]).startswith(("at::", "__amd_rocclr")):
                        continue                      # torch's fill / copy kernels of the bench script
                    w.writerow([short(r["Name"]), r["Calls"], f"{float(r['TotalDurationNs']) / 1e6:.4f}", f"{float(r['AverageNs']) / 1e6:.4f}", r["Percentage"],
                                f"{float(r['MinNs']) / 1e6:.4f}", f"{float(r['MaxNs']) / 1e6:.4f}"])
            print("wrote", f"profiles/{tag}{suffix}_kernel_stats.csv")
    sums = defaultdict(lambda: defaultdict(float)); cnts = defaultdict(lambda: defaultdict(int))
    for sub, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        for path in sorted(glob.glob(os.path.join(prof_dir, sub, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]:
            for r in csv.DictReader(open(path)):
                if r["Counter_Name"] == ctr:
                    k = short(r["Kernel_Name"])
                    sums[k][ctr] += float(r["Counter_Value"]); cnts[k][ctr] += 1
    if not sums:
        return
    traffic = {}
    with open(os.path.join(out_dir, f"{tag}_pmc.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "FETCH_SIZE_KiB_per_launch_raw", "WRITE_SIZE_KiB_per_launch", "hbm_bytes_per_launch (2*FETCH + WRITE) * 1024"])
        for k in sorted(sums):
            if k.startswith(("at::", "__amd_rocclr")):
                continue
            fe = sums[k]["FETCH_SIZE"] / max(cnts[k]["FETCH_SIZE"], 1)
            wr = sums[k]["WRITE_SIZE"] / max(cnts[k]["WRITE_SIZE"], 1)
            b = (2.0 * fe + wr) * 1024.0
            traffic[k] = round(b)
            w.writerow([k, cnts[k]["FETCH_SIZE"], f"{fe:.2f}", f"{wr:.2f}", f"{b:.0f}"])
    print("wrote", f"profiles/{tag}_pmc.csv")
    tj = os.path.join(out_dir, "pmc_traffic.json")
    allt = json.load(open(tj)) if os.path.exists(tj) else {}
    if "entries" not in allt:
        allt = {"entries": []}
    allt["entries"] = [e for e in allt["entries"] if not (e["workload"] == workload and e["size"] == size)]
    for k, b in sorted(traffic.items()):
        base = k.split("<")[0]
        allt["entries"].append({"workload": workload, "size": size, "kernel": base if base != k and not any(e for e in allt["entries"] if e["workload"] == workload and e["size"] == size and e["kernel"] == base) else k,
                                "kernel_full": k, "hbm_bytes_per_launch": b, "source": f"profiles/{tag}_pmc.csv"})
    # launch-weighted average over the template variants of one kernel (e.g. k_sc_round_mixed<true> = the first, scaled launch of a
    # sumcheck and <false> = the later rounds): the figure that pairs with a per-launch average over ALL launches of the kernel
    variants = defaultdict(list)
    for k in traffic:
        if "<" in k:
            variants[k.split("<")[0]].append(k)
    for base, ks in sorted(variants.items()):
        if len(ks) > 1:
            nl = sum(cnts[k]["FETCH_SIZE"] for k in ks)
            b = sum(traffic[k] * cnts[k]["FETCH_SIZE"] for k in ks) / max(nl, 1)
            allt["entries"].append({"workload": workload, "size": size, "kernel": base + "<*>", "kernel_full": " + ".join(sorted(ks)), "launches": nl,
                                    "hbm_bytes_per_launch": round(b), "source": f"profiles/{tag}_pmc.csv"})
    allt["note"] = "bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 from separate rocprofv3 --pmc passes (gfx950: FETCH_SIZE counts 64 B per 128-B request); averaged over every launch of the kernel in the run, setup launches included"
    json.dump(allt, open(tj, "w"), indent=1)
    print("updated profiles/pmc_traffic.json")


if __name__ == "__main__":
    main()

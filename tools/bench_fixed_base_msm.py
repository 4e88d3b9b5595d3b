#!/usr/bin/env python3
"""One large MSM over RESIDENT bases two ways: sbn_msm_bases_dev (variable-base pipeline: W bucket sets) against the shared-bucket-set
path with the window table 2^(c w) P_j (sbn_commit_rows_dev with L = 1: one bucket set, one reduction).  Same scalars, same bases,
results compared.  usage: bench_fixed_base_msm.py [log_n] [reps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg  # noqa: E402

sbn = load_pkg()
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n = 1 << logn
ctx = sbn.Context(0)
s0 = (12345).to_bytes(32, "little"); d = (67891).to_bytes(32, "little")
bases = ctx.bases_synthetic(n, 0, s0, d)
dz = ctx.dev_alloc(n * 32)
ctx.scalars_synthetic(0x5BA27A2B4E254, 0, n, dz)
res = {"log_n": logn}


def timed(fn, label):
    out = fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    ctx.sync()
    ms = (time.perf_counter() - t0) / reps * 1e3
    ctx.prof_enable(True); ctx.prof_reset(); fn(); ctx.sync()
    ks = {k: round(v[0], 4) for k, v in ctx.prof_get().items()}
    job = ctx.prof_last_job(); ctx.prof_enable(False)
    res[label] = {"ms": round(ms, 4), "points_per_s": round(n / ms * 1e3), "job": job, "kernels_ms": ks}
    return out


a = timed(lambda: ctx.msm_bases_dev(bases, dz, n), "variable_base")
b = timed(lambda: ctx.commit_rows_dev(bases, dz, 0, 1, n), "shared_bucket_set")
res["same_result"] = (a[0] == b[0][:64])
print(json.dumps(res))

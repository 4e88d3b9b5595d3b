#!/usr/bin/env python3
"""Where the B1 boundary (`msm_affine`, group.rs:171-175) should hand an MSM to the GPU: per-call latency of `sbn_msm` from HOST pointers
(the shim's call: upload + MSM + 64 bytes back) and of `sbn_msm_bases` (resident bases, host scalars) for 2^6 .. 2^18 terms, next to the CPU
oracle's arkworks-style Pippenger on 1 thread and on all host threads (measurement only: the checker is not part of any product path).
The shim's `MIN_GPU_MSM` is read off this table."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
from __graft_entry__ import load_pkg
import bench
import oracle_lib as ol

sbn = load_pkg(); ctx = sbn.Context(0)
ncpu = os.cpu_count() or 1
rows = []
nmax = 1 << 18
b_all = ctx.bases_synthetic(nmax, 0, bench.S0.to_bytes(32, "little"), bench.DSTEP.to_bytes(32, "little"))
pts_all = np.frombuffer(ctx.bases_download(b_all, 0, nmax), dtype=np.uint8).copy()
scal_all = np.frombuffer(bench.splitmix_scalars(nmax, 3), dtype=np.uint8).copy()
for logn in range(6, 19):
    n = 1 << logn
    scal, pts = scal_all[:32 * n].copy(), pts_all[:64 * n].copy()
    b = ctx.bases_upload(bytes(pts), None)
    want = ol.msm_pippenger(bytes(scal), bytes(pts), min(ncpu, 16))

    def timed(fn, reps):
        fn(); t0 = time.perf_counter()
        for _ in range(reps): r = fn()
        return (time.perf_counter() - t0) / reps * 1e6, r
    reps = 50 if n <= 1 << 14 else 10
    t_host, r1 = timed(lambda: ctx.msm(scal, pts), reps)
    t_res, r2 = timed(lambda: ctx.msm_bases(b, scal), reps)
    assert r1[0] == want and r2[0] == want, n
    creps = 20 if n <= 1 << 12 else 3
    t_c1, _ = timed(lambda: ol.msm_pippenger(bytes(scal), bytes(pts), 1), creps)
    t_cn, _ = timed(lambda: ol.msm_pippenger(bytes(scal), bytes(pts), min(ncpu, 16)), creps)
    rows.append({"n": n, "gpu_host_pointers_us": round(t_host, 1), "gpu_resident_bases_us": round(t_res, 1), "cpu_1_thread_us": round(t_c1, 1), f"cpu_{min(ncpu, 16)}_threads_us": round(t_cn, 1)})
    print(rows[-1], flush=True)
    b.free()
print(json.dumps({"small_msm_latency": rows, "host_threads": min(ncpu, 16)}))
# per-kernel view (in-library HIP-event profiler; synchronises after every job, so only the kernel times mean anything)
ctx.prof_enable(True)
for logn in (8, 10, 12, 14, 16):
    n = 1 << logn
    scal = scal_all[:32 * n].copy(); b = ctx.bases_upload(bytes(pts_all[:64 * n]), None)
    ctx.msm_bases(b, scal); ctx.prof_reset()
    for _ in range(5): ctx.msm_bases(b, scal)
    k = {name: round(ms / cnt * 1e3, 1) for name, (ms, cnt) in ctx.prof_get().items() if cnt}
    print("n=2^%d" % logn, ctx.prof_last_job(), "kernel us:", k, "sum", round(sum(ms / 5 for ms, cnt in ctx.prof_get().values()) * 1e3, 1), flush=True)
    b.free()

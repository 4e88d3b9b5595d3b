#!/usr/bin/env python3
"""window bits of small single MSMs (2^7 .. 2^17 terms, resident bases): per-call latency under SBN_MSM_C = 7 .. 16 against the automatic choice"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
from __graft_entry__ import load_pkg
import bench
sbn = load_pkg(); ctx = sbn.Context(0)
nmax = 1 << 17
b_all = ctx.bases_synthetic(nmax, 0, bench.S0.to_bytes(32, "little"), bench.DSTEP.to_bytes(32, "little"))
pts_all = np.frombuffer(ctx.bases_download(b_all, 0, nmax), dtype=np.uint8).copy()
scal_all = np.frombuffer(bench.splitmix_scalars(nmax, 3), dtype=np.uint8).copy()
for logn in range(7, 18):
    n = 1 << logn
    scal = scal_all[:32 * n].copy(); b = ctx.bases_upload(bytes(pts_all[:64 * n]), None)
    res = {}; ref = None
    for c in ["auto"] + list(range(7, 17)):
        if c == "auto": os.environ.pop("SBN_MSM_C", None)
        else: os.environ["SBN_MSM_C"] = str(c)
        try:
            r = ctx.msm_bases(b, scal)
        except Exception as e:
            res[c] = "err"; continue
        if ref is None: ref = r
        assert r == ref, (n, c)
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps): ctx.msm_bases(b, scal)
        res[c] = round((time.perf_counter() - t0) / reps * 1e6)
        if c == "auto": res["auto_c"] = ctx.prof_last_job()["c"]
    os.environ.pop("SBN_MSM_C", None)
    print("n=2^%d" % logn, res, flush=True)
    b.free()

#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-pointer entry points (DESIGN.md §4 note; bench.py's `value` is the resident rate):
   sbn_msm(host scalars + host points, 2^20)   and   sbn_commit_rows(host Z, 4096 x 8192 = 1 GiB) with resident generators."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa
from __graft_entry__ import load_pkg
import bench

sbn = load_pkg(); ctx = sbn.Context(0)
n = 1 << 20
scal = np.frombuffer(bench.splitmix_scalars(n, 1), dtype=np.uint8).copy()
b = ctx.bases_synthetic(n, 0, bench.S0.to_bytes(32, "little"), bench.DSTEP.to_bytes(32, "little"))
pts = np.frombuffer(ctx.bases_download(b, 0, n), dtype=np.uint8).copy()
res = {}
for name, fn in (("sbn_msm host scalars+points 2^20 (96 MB H2D)", lambda: ctx.msm(scal, pts)),
                 ("sbn_msm_bases host scalars 2^20, resident bases (32 MB H2D)", lambda: ctx.msm_bases(b, scal))):
    fn(); t0 = time.perf_counter(); reps = 5
    for _ in range(reps): fn()
    dt = (time.perf_counter() - t0) / reps
    res[name] = {"ms": round(dt * 1e3, 2), "points_per_s": round(n / dt)}
b.free()
L, R = 4096, 8192
bases, _ = ctx.gens_new(R, b"gens_r1cs_eval", want_points=False)
rng = np.random.default_rng(1)
Z = rng.integers(0, 2**32, size=(L * R, 8), dtype=np.uint32); Z[:, 7] &= 0x0fffffff; Z[(3 * L // 4) * R:] = 0
ctx.commit_rows(bases, Z, None, L, R); t0 = time.perf_counter(); reps = 2
for _ in range(reps): ctx.commit_rows(bases, Z, None, L, R)
dt = (time.perf_counter() - t0) / reps
res["sbn_commit_rows host Z 4096x8192 (1 GiB H2D), resident generators"] = {"ms": round(dt * 1e3, 2), "points_per_s": round(L * R / dt)}
ctx.bases_precompute(bases, 100 << 30)      # with the generator set's lookup table the commit hides behind the upload
ctx.commit_rows(bases, Z, None, L, R); t0 = time.perf_counter()
for _ in range(reps): ctx.commit_rows(bases, Z, None, L, R)
dt = (time.perf_counter() - t0) / reps
res["same, generator set with its 94 GB lookup table"] = {"ms": round(dt * 1e3, 2), "points_per_s": round(L * R / dt)}
print(json.dumps(res))

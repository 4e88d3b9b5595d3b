#!/usr/bin/env python3
"""what a bullet round would cost on MATERIALISED (folded) generators once the vectors are short, against the shipped
round over the original generators (one sbn_bullet_fold_cross: ~160 us whatever the current length).
A round on m materialised generators needs (bullet.rs:72-91): two MSMs of m/2 variable points with full-width scalars (L and R) and the fold
G' = u_inv * G_L + u * G_R — m/2 double-scalar multiplications, no table can be precomputed for points that change every round.
Measured with the library's own entry points: sbn_msm on m/2 random points, and sbn_bases_scale on m/2 points (ONE 254-bit scalar
multiplication per point and lane: half of the fold's work, the same dependent chain)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg, rand_scalars  # noqa: E402
import oracle_lib as ol  # noqa: E402   (only to make valid curve points for the inputs)

sbn = load_pkg()
ctx = sbn.Context(0)
for m in (256, 64, 16):
    h = m // 2
    pts = ol.g1_mul_gen_batch(rand_scalars(h, 7 + m), 8)
    sc = rand_scalars(h, 9 + m)
    ctx.msm(sc, pts)
    t0 = time.perf_counter(); reps = 20
    for _ in range(reps):
        ctx.msm(sc, pts)
    t_msm = (time.perf_counter() - t0) / reps
    b = ctx.bases_upload(pts, None)
    s1 = rand_scalars(1, 3)
    x = ctx.bases_scale(b, s1); x.free()
    t0 = time.perf_counter()
    for _ in range(reps):
        x = ctx.bases_scale(b, s1); x.free()
    t_scale = (time.perf_counter() - t0) / reps
    b.free()
    print(f"m = {m:4d}: MSM of {h} variable points {t_msm * 1e6:7.1f} us (x2 per round: L and R); {h} scalar multiplications {t_scale * 1e6:7.1f} us (the fold needs two per point)"
          f"  -> a round on materialised generators >= {(2 * t_msm + 2 * t_scale) * 1e6:7.1f} us against ~160 us for the shipped round")
ctx.close()

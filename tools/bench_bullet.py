"""Time BulletReductionProof::prove's device work (nizk/bullet.rs:63-108) at n = 8192 (the derefs opening's right-hand vector):
13 rounds, challenges fixed: sbn_bullet_cross once, then one sbn_bullet_fold_cross per challenge (sbn_bullet_fold for the last); BULLET_SEPARATE=1 times
the round-3 sequence (cross + fold per round) instead.  Usage: python tools/bench_bullet.py [log_n]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg, rand_scalars  # noqa: E402

sbn = load_pkg()
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 13
n = 1 << lg
ctx = sbn.Context(0)
gens, pts = ctx.gens_new(n, b"gens_r1cs_eval")
G_xy, H_xy = pts[:64 * n], pts[64 * n:]
Q_xy = (1).to_bytes(32, "little") + (2).to_bytes(32, "little")
a, b = rand_scalars(n, 1), rand_scalars(n, 2)
bl = rand_scalars(2 * lg, 3); us = rand_scalars(lg, 4)
R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
uis = b"".join(pow(int.from_bytes(us[32 * i:32 * i + 32], "little"), R - 2, R).to_bytes(32, "little") for i in range(lg))
G = ctx.bases_upload(G_xy, H_xy)
for rep in range(3):
    ta, tb = ctx.table_upload(a), ctx.table_upload(b)
    ctx.sync(); t0 = time.perf_counter(); tc = tf = 0.0
    st, Gamma = ctx.bullet_begin(G, Q_xy, ta, tb, us[:32])
    tb0 = time.perf_counter() - t0
    if os.environ.get("BULLET_SEPARATE"):
        for r in range(lg):
            t1 = time.perf_counter()
            ctx.bullet_cross(st, bl[64 * r:64 * r + 32], bl[64 * r + 32:64 * r + 64])
            t2 = time.perf_counter()
            ctx.bullet_fold(st, us[32 * r:32 * r + 32], uis[32 * r:32 * r + 32])
            t3 = time.perf_counter()
            tc += t2 - t1; tf += t3 - t2
    else:
        t1 = time.perf_counter()
        ctx.bullet_cross(st, bl[:32], bl[32:64])
        tc += time.perf_counter() - t1
        for r in range(lg):
            t2 = time.perf_counter()
            if r + 1 < lg:
                ctx.bullet_fold_cross(st, us[32 * r:32 * r + 32], uis[32 * r:32 * r + 32], bl[64 * (r + 1):64 * (r + 1) + 32], bl[64 * (r + 1) + 32:64 * (r + 1) + 64])
            else:
                ctx.bullet_fold(st, us[32 * r:32 * r + 32], uis[32 * r:32 * r + 32])
            tf += time.perf_counter() - t2
    t4 = time.perf_counter()
    ctx.bullet_finish(st)
    te = time.perf_counter() - t4
    dt = time.perf_counter() - t0
    print(f"n=2^{lg}: {lg} rounds {dt * 1e3:.2f} ms  (begin+Gamma {tb0 * 1e3:.2f}, first cross terms {tc * 1e3:.2f}, fold+cross calls {tf * 1e3:.2f}, finish {te * 1e3:.2f} ms)")
    st.free(); ta.free(); tb.free()
G.free()
# per-kernel view of one cross round (HIP events)
G = ctx.bases_upload(G_xy, H_xy)
ta, tb = ctx.table_upload(a), ctx.table_upload(b)
st, _ = ctx.bullet_begin(G, Q_xy, ta, tb, us[:32])
ctx.bullet_cross(st, bl[:32], bl[32:64])
ctx.prof_enable(True); ctx.prof_reset()
t0 = time.perf_counter()
ctx.bullet_cross(st, bl[:32], bl[32:64])
dt = time.perf_counter() - t0
prof = ctx.prof_get(); ctx.prof_enable(False)
print(f"one cross call {dt * 1e3:.3f} ms (with event timing on); kernels ms:", {k: round(v[0], 4) for k, v in prof.items()})
st.free(); ta.free(); tb.free(); G.free()

"""Quick on-GPU correctness + timing probe for the MSM path (dev tool; the real tests live in tests/)."""
import importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("spartan_bn254_amd", os.path.join(ROOT, "spartan-bn254_amd", "__init__.py"), submodule_search_locations=[os.path.join(ROOT, "spartan-bn254_amd")])
sbn = importlib.util.module_from_spec(spec); sys.modules["spartan_bn254_amd"] = sbn; spec.loader.exec_module(sbn)
import oracle_lib as ol, pyref as pr
import numpy as np

def synth(n, seed=1):
    rng = np.random.default_rng(seed)
    # scalars: random 254-bit values reduced mod r
    k = [int.from_bytes(rng.bytes(32), "little") % pr.R for _ in range(n)]
    kb = b"".join(x.to_bytes(32, "little") for x in k)
    # bases with known dlogs s_i = s0 + i*d
    s0, d = 0x1234567890abcdef1234567890abcdef, 0x0fedcba987654321
    dl = [(s0 + i * d) % pr.R for i in range(n)]
    dlb = b"".join(x.to_bytes(32, "little") for x in dl)
    pts = ol.g1_mul_gen_batch(dlb, 8)
    exp_s = sum(a * b for a, b in zip(k, dl)) % pr.R
    G = pr.point_to_xy(pr.G)
    exp = ol.g1_mul(G, exp_s.to_bytes(32, "little"))
    return kb, pts, exp

ctx = sbn.Context(0)
ok = True
for n in [1, 2, 3, 33, 100, 1000, 8193, 1 << 16, 1 << 18]:
    kb, pts, exp = synth(n)
    t0 = time.time(); out, inf = ctx.msm(kb, pts); dt = time.time() - t0
    good = out == exp
    ok &= good
    print(f"n={n:8d} match={good} inf={inf} host-ptr call {dt*1e3:.2f} ms", flush=True)
    if n <= 1000:
        assert ol.msm_pippenger(kb, pts, 4) == exp
# edge cases
G = pr.point_to_xy(pr.G)
z = bytes(32)
out, inf = ctx.msm(z * 5, G * 5); print("all-zero scalars -> inf:", inf, out == bytes(64)); ok &= inf and out == bytes(64)
k2 = (2).to_bytes(32, "little"); k3 = (3).to_bytes(32, "little")
out, inf = ctx.msm(k2 + k3, G + G); e = ol.g1_mul(G, (5).to_bytes(32, "little")); print("2G+3G=5G:", out == e); ok &= out == e
rm1 = (pr.R - 1).to_bytes(32, "little")
out, inf = ctx.msm(rm1 + (1).to_bytes(32, "little"), G + G); print("(r-1)G+G=inf:", inf); ok &= inf
nG = ol.g1_neg(G)
out, inf = ctx.msm(k3 + k3, G + nG); print("3G-3G=inf:", inf); ok &= inf
out, inf = ctx.msm(k3 * 40, G * 40); e = ol.g1_mul(G, (120).to_bytes(32, "little")); print("40 x 3G:", out == e); ok &= out == e
# skewed scalars: every scalar equal -> one bucket per window holds all n points (segment splitting path)
n = 5000
kb1, pts, _ = synth(n, 11)
same = kb1[:32] * n
out, inf = ctx.msm(same, pts); e = ol.msm_pippenger(same, pts, 8); print("all-equal scalars n=5000:", out == e); ok &= out == e
# Hyrax row commits (MODE_ROWS): L x R with and without blinds, zero rows, constant rows
for (L, R) in [(1, 1), (3, 5), (8, 64), (16, 300), (4, 1025)]:
    gx, gdl = ol.gens_new(R, b"gens_r1cs_eval")
    Gxy, hxy = gx[: 64 * R], gx[64 * R:]
    rng = np.random.default_rng(L * 1000 + R)
    Z = bytearray(b"".join((int.from_bytes(rng.bytes(32), "little") % pr.R).to_bytes(32, "little") for _ in range(L * R)))
    if L >= 3:
        Z[32 * R: 64 * R] = bytes(32 * R)                         # an all-zero row -> identity
        Z[64 * R: 96 * R] = Z[64 * R: 64 * R + 32] * R            # a constant row
    Z = bytes(Z)
    bl = b"".join((int.from_bytes(rng.bytes(32), "little") % pr.R).to_bytes(32, "little") for _ in range(L))
    b = ctx.bases_upload(Gxy, hxy)
    out, infs = ctx.commit_rows(b, Z, bl, L, R); e = ol.commit_rows(Z, bl, L, R, Gxy, hxy, 8); print(f"commit_rows {L}x{R} blinds:", out == e); ok &= out == e
    out, infs = ctx.commit_rows(b, Z, None, L, R); e = ol.commit_rows(Z, None, L, R, Gxy, hxy, 8); print(f"commit_rows {L}x{R} no blinds:", out == e, list(infs)[:4]); ok &= out == e
    b.free()
# timing with resident bases
for n in [1 << 16, 1 << 18, 1 << 20, 1 << 22]:
    kb, pts, exp = synth(n, 7) if n <= (1 << 18) else (None, None, None)
    if kb is None:
        rng = np.random.default_rng(3); k = rng.integers(0, 2**32, size=(n, 8), dtype=np.uint32); k[:, 7] &= 0x0fffffff; kb = k.tobytes()
        base_k, base_p, _ = synth(1 << 16, 9); pts = base_p * (n >> 16)
    b = ctx.bases_upload(pts)
    dptr = ctx.dev_alloc(n * 32); ctx.dev_upload(dptr, kb)
    ctx.prof_enable(True); ctx.prof_reset()
    out, inf = ctx.msm_bases_dev(b, dptr, n)
    t0 = time.time(); reps = 3
    for _ in range(reps): out, inf = ctx.msm_bases_dev(b, dptr, n)
    dt = (time.time() - t0) / reps
    print(f"resident n={n}: {dt*1e3:.3f} ms/msm  {n/dt:.3e} pts/s  match={out == exp if exp else 'n/a'}")
    for name, (ms, cnt) in ctx.prof_get().items(): print(f"    {name:24s} {ms/cnt:9.3f} ms avg x{cnt}")
    ctx.prof_enable(False); ctx.dev_free(dptr); b.free()
# Hyrax shape timing: rows x 8192 over the reference generator set
R = 8192
t0 = time.time(); bg, gxy = ctx.gens_new(R, b"gens_r1cs_eval", want_points=False); print(f"gens_new(8192): {(time.time()-t0)*1e3:.1f} ms")
for L in [64, 512]:
    rng = np.random.default_rng(5); Zk = rng.integers(0, 2**32, size=(L * R, 8), dtype=np.uint32); Zk[:, 7] &= 0x0fffffff
    dptr = ctx.dev_alloc(L * R * 32); ctx.dev_upload(dptr, Zk.tobytes())
    ctx.prof_enable(True); ctx.prof_reset()
    out, infs = ctx.commit_rows_dev(bg, dptr, 0, L, R)
    ctx.prof_reset()
    t0 = time.time(); out, infs = ctx.commit_rows_dev(bg, dptr, 0, L, R); dt = time.time() - t0
    print(f"commit_rows {L}x{R}: {dt*1e3:.2f} ms  {L*R/dt:.3e} pairs/s")
    for name, (ms, cnt) in ctx.prof_get().items(): print(f"    {name:24s} {ms/cnt:9.3f} ms avg x{cnt}")
    ctx.prof_enable(False); ctx.dev_free(dptr)
print("MSM CHECK", "OK" if ok else "FAIL")
sys.exit(0 if ok else 1)

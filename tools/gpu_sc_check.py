"""Quick on-GPU probe of the sumcheck / eq / gens entry points against the oracle (dev tool)."""
import importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("spartan_bn254_amd", os.path.join(ROOT, "spartan-bn254_amd", "__init__.py"), submodule_search_locations=[os.path.join(ROOT, "spartan-bn254_amd")])
sbn = importlib.util.module_from_spec(spec); sys.modules["spartan_bn254_amd"] = sbn; spec.loader.exec_module(sbn)
import oracle_lib as ol, pyref as pr
import numpy as np
ctx = sbn.Context(0)
ok = True
def rnd_table(n, seed):
    rng = np.random.default_rng(seed)
    return b"".join((int.from_bytes(rng.bytes(32), "little") % pr.R).to_bytes(32, "little") for _ in range(n))
for n in [2, 4, 64, 1024, 1 << 14]:
    A, B, Cc, T = (rnd_table(n, s) for s in (1, 2, 3, 4))
    tA, tB, tC, tT = (ctx.table_upload(x) for x in (A, B, Cc, T))
    g = ctx.sc_eval_cubic(tA, tB, tC); e = ol.sc_eval_cubic(A, B, Cc); ok &= g == e; print(n, "cubic", g == e)
    g = ctx.sc_eval_r1cs(tT, tA, tB, tC); e = ol.sc_eval_r1cs(T, A, B, Cc); ok &= g == e; print(n, "r1cs", g == e)
    g = ctx.sc_eval_quad(tA, tB); e = ol.sc_eval_quad(A, B); ok &= g == e; print(n, "quad", g == e)
    g = ctx.sc_eval_cubic_batched([tA, tB, tT], [tB, tC, tA], [tC, tC, tC]); e = ol.sc_eval_cubic(A, B, Cc) + ol.sc_eval_cubic(B, Cc, Cc) + ol.sc_eval_cubic(T, A, Cc); ok &= g == e; print(n, "batched", g == e)
    r = rnd_table(1, 99)
    ctx.bind_top_many([tA, tB], r); ctx.bind_top(tC, r)
    for t, h in ((tA, A), (tB, B), (tC, Cc)):
        g = ctx.table_download(t); e = ol.bind_top(h, r); ok &= g == e; print(n, "bind", g == e, len(t))
    ok &= ctx.table_read0(tA) == ol.bind_top(A, r)[:32]
for ell in [0, 1, 2, 5, 12]:
    r = rnd_table(max(ell, 1), 7)[: 32 * ell]
    t = ctx.eq_evals(r); g = ctx.table_download(t); e = ol.eq_evals(r) if ell else (1).to_bytes(32, "little"); ok &= g == e; print("eq", ell, g == e)
for n, label in [(20, b"gens_r1cs_eval"), (1025, b"gens_r1cs_sat")]:
    t0 = time.time(); b, xy = ctx.gens_new(n, label); dt = time.time() - t0
    exy, edl = ol.gens_new(n, label); ok &= xy == exy; print("gens", n, xy == exy, f"{dt*1e3:.1f} ms")
    # commit through the resident table == oracle commit
    sc = rnd_table(n, 5); blind = rnd_table(1, 6)
    out, inf = ctx.commit_rows(b, sc, blind, 1, n); e = ol.commit(sc, blind, exy[: 64 * n], exy[64 * n:]); ok &= out == e; print("commit", n, out == e)
    out, inf = ctx.commit_rows(b, sc, None, 1, n); e = ol.commit(sc, bytes(32), exy[: 64 * n], exy[64 * n:]); ok &= out == e; print("commit(no blind)", n, out == e)
# bandwidth probe: cubic eval + bind at 2^22
n = 1 << 22
import torch
dev = torch.device("cuda:0")
ts = []
for s in range(3):
    x = torch.randint(0, 2**31 - 1, (n, 8), dtype=torch.int32, device=dev); x[:, 7] &= 0x0fffffff
    ts.append(ctx.table_from_dev(x.data_ptr(), n, sbn.SBN_SCALARS_MONT)); del x
ctx.prof_enable(True); ctx.prof_reset()
for _ in range(3): ctx.sc_eval_cubic(*ts)
r = rnd_table(1, 11)
ctx.bind_top_many(ts, r)
for name, (ms, cnt) in ctx.prof_get().items():
    per = ms / cnt
    if name == "k_sc_eval_cubic": print(f"{name}: {per:.3f} ms  -> {3*n*32/per/1e9*1e3/1e3:.1f} GB/s")
    elif name == "k_bind_top": print(f"{name}: {per:.3f} ms  -> {(3*n*32 + 3*n*16)/per/1e6:.1f} GB/s")
    else: print(f"{name}: {per:.3f} ms")
print("SC CHECK", "OK" if ok else "FAIL")
sys.exit(0 if ok else 1)

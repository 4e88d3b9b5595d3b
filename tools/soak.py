#!/usr/bin/env python3
"""Randomised soak of the C ABI against the oracle (test infrastructure, like tests/): MSMs, row commits (bucket and lookup paths),
sumcheck rounds and bullet reductions of random shapes until the time budget is spent.  Usage: python tools/soak.py [seconds] [seed]"""
import hashlib
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg, rand_scalars  # noqa: E402
import oracle_lib as ol  # noqa: E402
import pyref as pr  # noqa: E402

sbn = load_pkg()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = sbn.Context(0)
t_end = time.time() + budget
counts = {"msm": 0, "commit": 0, "sumcheck": 0, "stateful": 0, "bullet": 0}


def special(b, n):
    """sprinkle edge scalars into a scalar blob"""
    b = bytearray(b)
    for _ in range(min(n, 6)):
        i = rng.randrange(n); v = rng.choice([0, 1, pr.R - 1, pr.R - 2, 1 << 253, (1 << 128) - 1, 1 << 15, (1 << 16) - 1, (1 << 32) - 1])
        b[32 * i:32 * i + 32] = pr.scalar_to_bytes(v % pr.R)
    return bytes(b)


case = 0
t_note = time.time() + 60
while time.time() < t_end:
    case += 1
    if time.time() > t_note:          # a line a minute: a silent GPU command is taken to be hung
        print("soak: %d cases so far" % case, counts, flush=True); t_note = time.time() + 60
    kind = rng.choice(["msm", "commit", "commit", "sumcheck", "stateful", "stateful", "bullet"])
    sd = rng.randrange(1 << 30)
    if kind == "msm":
        n = rng.choice([1, 2, 3, 31, 64, 257, 1000, 4097, 20000, 70000, 150000])
        k = special(rand_scalars(n, sd), n)
        pts = bytearray(ol.g1_mul_gen_batch(rand_scalars(n, sd + 1), 8))
        if n > 3 and rng.random() < 0.5:
            pts[64:128] = pts[0:64]; pts[128:192] = bytes(64)
        pts = bytes(pts)
        got = ctx.msm(k, pts)[0]
        assert got == ol.msm_pippenger(k, pts, 8), ("msm", n, sd)
    elif kind == "commit":
        R = rng.choice([1, 5, 64, 100, 1024, 3000]); L = rng.choice([1, 2, 7, 33, 130])
        if R <= 100 and rng.random() < 0.3: L = rng.choice([256, 300])       # many rows: the window rule of the throughput regime (choose_shape)
        label = rng.choice([b"gens_r1cs_eval", b"gens_r1cs_sat", b"x%d" % sd])
        gx, _ = ol.gens_new(R, label, 8)
        Z = bytearray(special(rand_scalars(L * R, sd), L * R))
        for i in range(L):
            m = rng.randrange(6)
            if m == 0: Z[32 * R * i:32 * R * (i + 1)] = bytes(32 * R)
            elif m == 1: Z[32 * R * i:32 * R * (i + 1)] = Z[32 * R * i:32 * R * i + 32] * R
        Z = bytes(Z); bl = rand_scalars(L, sd + 2) if rng.random() < 0.5 else None
        b = ctx.bases_upload(gx[:64 * R], gx[64 * R:])
        want = ol.commit_rows(Z, bl, L, R, gx[:64 * R], gx[64 * R:], 8)
        assert ctx.commit_rows(b, Z, bl, L, R)[0] == want, ("commit bucket", L, R, sd)
        if rng.random() < 0.6:
            ctx.bases_precompute(b, rng.choice([256, 1024, 4096]) << 20)
            assert ctx.commit_rows(b, Z, bl, L, R)[0] == want, ("commit lookup", L, R, sd)
        b.free()
    elif kind == "sumcheck":
        logn = rng.choice([1, 2, 5, 9, 13]); n = 1 << logn; cnt = rng.choice([1, 3, 18, 30])
        tabs = [rand_scalars(n, sd + i) for i in range(cnt + 2)]
        dev = [ctx.table_upload(x) for x in tabs]
        idx = [(i, i + 1, i + 2) for i in range(cnt)]
        As, Bs, Cs = ([dev[t[j]] for t in idx] for j in range(3))
        ev = ctx.sc_eval_cubic_batched(As, Bs, Cs)
        host = list(tabs)
        while True:
            assert ev == b"".join(ol.sc_eval_cubic(host[a], host[b], host[c]) for a, b, c in idx), ("sumcheck", logn, cnt, sd)
            r = pr.scalar_to_bytes(int.from_bytes(hashlib.sha3_256(ev).digest(), "little") % pr.R)
            host = [ol.bind_top(x, r) for x in host]
            if len(dev[0]) >= 4 and rng.random() < 0.8:
                ev = ctx.sc_bind_eval_cubic_batched(As, Bs, Cs, r)
            else:
                ctx.bind_top_many(dev, r)
                if len(dev[0]) < 2: break
                ev = ctx.sc_eval_cubic_batched(As, Bs, Cs)
        assert [ctx.table_read0(t) for t in dev] == host, ("sumcheck final", logn, cnt, sd)
        for t in dev: t.free()
    elif kind == "stateful":
        # sbn_sumcheck_begin / round / finish against the oracle's whole prove_cubic_batched loop: every round's combined triple and
        # the final claims; random instance counts, sizes on both sides of the combined kernels' threshold, edge coefficients
        logn = rng.choice([1, 2, 4, 7, 10, 13, 15, 16, 17]); n = 1 << logn
        n_par = rng.choice([0, 1, 2, 5, 12, 17]); n_seq = rng.choice([0, 0, 1, 3, 6])
        if n_par + n_seq == 0: n_par = 1
        if n_par + n_seq > 24: n_seq = 24 - n_par
        ntab = 2 * n_par + (1 if n_par else 0) + 3 * n_seq
        host = [rand_scalars(n, sd + i) for i in range(ntab)]
        dev = [ctx.table_upload(x) for x in host]
        o = 2 * n_par + (1 if n_par else 0)
        co = bytearray(rand_scalars(n_par + n_seq, sd + 100))
        for _ in range(rng.choice([0, 0, 1, 2])):
            i = rng.randrange(n_par + n_seq); co[32 * i:32 * i + 32] = pr.scalar_to_bytes(rng.choice([0, 1, 2, pr.R - 1]))
        co = bytes(co); ch = rand_scalars(logn, sd + 101)
        if n_par and rng.random() < 0.4:
            # round 4: poly_C_par = eq(rand) built inside the call (sbn_sumcheck_begin_eq); the oracle gets the same table from the device's sbn_eq_evals
            rand = rand_scalars(logn, sd + 102)
            eqt = ctx.eq_evals(rand); host[2 * n_par] = ctx.table_download(eqt); eqt.free()
            st, ev = ctx.sumcheck_begin_eq(dev[:n_par], dev[n_par:2 * n_par], rand, dev[o:o + n_seq], dev[o + n_seq:o + 2 * n_seq], dev[o + 2 * n_seq:], co)
            counts["begin_eq"] = counts.get("begin_eq", 0) + 1
            host_c_replaced = True
        else:
            host_c_replaced = False
            st, ev = ctx.sumcheck_begin(dev[:n_par], dev[n_par:2 * n_par], dev[2 * n_par] if n_par else None, dev[o:o + n_seq], dev[o + n_seq:o + 2 * n_seq], dev[o + 2 * n_seq:], co)
        got = [ev]
        for j in range(logn):
            ev = st.round(ch[32 * j:32 * j + 32])
            if j + 1 < logn: got.append(ev)
        fin = st.finish(); st.free()
        import numpy as np
        hn = [np.frombuffer(x, dtype=np.uint8) for x in host]
        _, want_comb, want_fin = ol.sc_prove_cubic_batched(hn[:n_par], hn[n_par:2 * n_par], hn[2 * n_par] if n_par else None, hn[o:o + n_seq], hn[o + n_seq:o + 2 * n_seq], hn[o + 2 * n_seq:], co, ch, 8)
        assert got == list(want_comb[:logn]), ("stateful rounds", logn, n_par, n_seq, sd)
        assert fin == want_fin, ("stateful finals", logn, n_par, n_seq, sd)
        if not host_c_replaced:
            assert ctx.table_read0_many(dev) == [x[:32] for x in host], ("stateful left the caller's tables alone", sd)
        for t in dev: t.free()
    else:
        lg = rng.choice([1, 2, 3, 4, 5, 6, 7, 7, 10, 11]); n = 1 << lg
        pts, _ = ol.gens_new(n, rng.choice([b"gens_r1cs_eval", b"b%d" % sd]), 8)
        G_xy, H_xy = pts[:64 * n], pts[64 * n:]
        Q_xy = pr.point_to_xy(pr.mul((1, 2), 1 + sd))
        a, b = special(rand_scalars(n, sd), n), special(rand_scalars(n, sd + 1), n)
        blind = rand_scalars(1, sd + 2); bv = rand_scalars(2 * lg, sd + 3); us = rand_scalars(lg, sd + 4)
        if rng.random() < 0.5:
            # round 4: Q = q_scale * Q_base (nizk/mod.rs:478-494) — the oracle is handed the product
            qs = 1 + rng.randrange(pr.R - 1); Qb_xy = Q_xy
            Q_xy = pr.point_to_xy(pr.mul((1, 2), (1 + sd) * qs % pr.R))
            want = ol.bullet_prove(G_xy, Q_xy, H_xy, a, b, blind, bv, us)
            G = ctx.bases_upload(G_xy, H_xy); ta, tb = ctx.table_upload(a), ctx.table_upload(b)
            st, Gamma = ctx.bullet_begin_scaled(G, Qb_xy, pr.scalar_to_bytes(qs), ta, tb, blind)
            counts["begin_scaled"] = counts.get("begin_scaled", 0) + 1
        else:
            want = ol.bullet_prove(G_xy, Q_xy, H_xy, a, b, blind, bv, us)
            G = ctx.bases_upload(G_xy, H_xy); ta, tb = ctx.table_upload(a), ctx.table_upload(b)
            st, Gamma = ctx.bullet_begin(G, Q_xy, ta, tb, blind)
        assert Gamma == want["Gamma"], ("bullet gamma", n, sd)
        fused = rng.random() < 0.6
        inv = lambda u: pr.scalar_to_bytes(pow(pr.scalar_from_bytes(u), pr.R - 2, pr.R))
        for rnd in range(lg):
            bl, br = bv[64 * rnd:64 * rnd + 32], bv[64 * rnd + 32:64 * rnd + 64]
            if fused and rnd > 0:       # one call per challenge: fold with the previous u, return this round's L, R (sbn_bullet_fold_cross)
                up = us[32 * (rnd - 1):32 * rnd]
                L, _, Rp, _, _, _ = ctx.bullet_fold_cross(st, up, inv(up), bl, br)
            else:
                L, _, Rp, _, _, _ = ctx.bullet_cross(st, bl, br)
            assert L == want["L"][64 * rnd:64 * rnd + 64] and Rp == want["R"][64 * rnd:64 * rnd + 64], ("bullet round", n, rnd, sd)
            u = us[32 * rnd:32 * rnd + 32]
            if not fused or rnd == lg - 1: ctx.bullet_fold(st, u, inv(u))
        ah, bh, gh = ctx.bullet_finish(st)
        assert (ah, bh, gh) == (want["a_hat"], want["b_hat"], want["g_hat"]), ("bullet finish", n, sd)
        st.free(); ta.free(); tb.free(); G.free()
    counts[kind] += 1
print("soak OK:", counts, "cases in %.0f s" % budget)

"""BASELINE config 5 in the small: the device-side stages of one keyless-SHAPED prove (the stage list of
examples/keyless_benchmark.rs:171-238 as tools/bench_prove_stages.py runs it), at 2^-14 of the keyless sizes, with EVERY stage's
outputs compared with the CPU oracle — and with the tables flowing from stage to stage on the device exactly as in the full-size
tool (the lazily represented table values of one kernel are the inputs of the next: this is the test that covers those hand-offs).
Fiat-Shamir challenges are SHA3 of the round values (a stand-in for the Merlin transcript, which stays in Rust).  Integer work:
bit-exact."""
import hashlib

import numpy as np
import pytest
from conftest import rand_scalars

pytestmark = pytest.mark.gpu
LOG_OPS, LOG_MEM, LOG_CONS = 8, 7, 6


def chal(pr, data):
    return pr.scalar_to_bytes(int.from_bytes(hashlib.sha3_256(data).digest(), "little") % pr.R)


def inv(pr, u):
    return pr.scalar_to_bytes(pow(pr.scalar_from_bytes(u), pr.R - 2, pr.R))


def opening(ctx, ol, pr, sbn, Z_dev, Z_host, log_l, log_r, label, seed):
    """PolyEvalProof::prove (hyrax.rs:83-160): L*Z (bound), then the bullet reduction over the right vector"""
    n = 1 << log_r
    rl, rr = rand_scalars(log_l, seed), rand_scalars(log_r, seed + 1)
    Lv, Rv = ctx.eq_evals(rl), ctx.eq_evals(rr)
    hL, hR = ol.eq_evals(rl), ol.eq_evals(rr)
    LZ = ctx.table_bound(Z_dev, Lv)
    hLZ = ol.bound(Z_host, hL, 1 << log_l, n)
    assert ctx.table_download(LZ) == hLZ, "L*Z"
    assert ctx.table_evaluate(Z_dev, rl + rr) == ol.fr_dot(hLZ, hR), "evaluate == <LZ, R>"          # hyrax.rs:217-222 / :311-324
    pts, _ = ol.gens_new(n, label)
    G_xy, H_xy = pts[:64 * n], pts[64 * n:]
    Q_xy = pr.point_to_xy((1, 2))
    G = ctx.bases_upload(G_xy, H_xy)
    blind = rand_scalars(1, seed + 2); bvec = rand_scalars(2 * log_r, seed + 3)
    st, Gamma = ctx.bullet_begin(G, Q_xy, LZ, Rv, blind)
    us, Ls, Rs = b"", [], []
    for rnd in range(log_r):
        L, _, Rp, _, cL, cR = ctx.bullet_cross(st, bvec[64 * rnd:64 * rnd + 32], bvec[64 * rnd + 32:64 * rnd + 64])
        u = chal(pr, L + Rp); us += u
        ctx.bullet_fold(st, u, inv(pr, u))
        Ls.append(L); Rs.append(Rp)
    a_hat, b_hat, g_hat = ctx.bullet_finish(st)
    want = ol.bullet_prove(G_xy, Q_xy, H_xy, hLZ, hR, blind, bvec, us)
    assert (b"".join(Ls), b"".join(Rs), Gamma, a_hat, b_hat, g_hat) == (want["L"], want["R"], want["Gamma"], want["a_hat"], want["b_hat"], want["g_hat"]), "bullet reduction"
    st.free(); G.free()
    for t in (Lv, Rv, LZ):
        t.free()


def test_keyless_shaped_prove_stages_vs_oracle(ctx, ol, pr, sbn):
    # ---------------------------------------------------------------- R1CS satisfiability proof (r1csproof.rs:241-420)
    wl = LOG_CONS // 2; wr = LOG_CONS - wl
    hz = rand_scalars(1 << LOG_CONS, 1); z = ctx.table_upload(hz)
    gens_w, gw = ctx.gens_new(1 << wr, b"gens_r1cs_sat")
    wbl = rand_scalars(1 << wl, 2)
    out, _ = ctx.commit_table(gens_w, z, wbl, 1 << wl, 1 << wr)
    assert out == ol.commit_rows(hz, wbl, 1 << wl, 1 << wr, gw[:64 << wr], gw[64 << wr:], 4), "witness commitment"
    # phase 1: tau * (Az * Bz - Cz), four tables (sumcheck.rs:502-530)
    rtau = rand_scalars(LOG_CONS, 3)
    tau = ctx.eq_evals(rtau); htau = ol.eq_evals(rtau)
    hAz, hBz, hCz = (rand_scalars(1 << LOG_CONS, s) for s in (4, 5, 6))
    Az, Bz, Cz = (ctx.table_upload(x) for x in (hAz, hBz, hCz))
    host = [htau, hAz, hBz, hCz]; dev = [tau, Az, Bz, Cz]
    ev = ctx.sc_eval_r1cs(*dev); n = 1 << LOG_CONS
    while n >= 2:
        assert ev == ol.sc_eval_r1cs(*host), ("phase 1", n)
        r = chal(pr, ev)
        host = [ol.bind_top(h, r) for h in host]
        if n >= 4:
            ev = ctx.sc_bind_eval_r1cs(*dev, r)
        else:
            ctx.bind_top_many(dev, r)
        n //= 2
    assert [ctx.table_read0(t) for t in dev] == [h[:32] for h in host], "phase 1 final claims"
    for t in dev:
        t.free()
    # phase 2: z * ABC, two tables of twice the size (sumcheck.rs:691-699)
    hZp, hABC = rand_scalars(2 << LOG_CONS, 7), rand_scalars(2 << LOG_CONS, 8)
    dev = [ctx.table_upload(hZp), ctx.table_upload(hABC)]; host = [hZp, hABC]
    ev = ctx.sc_eval_quad(*dev); n = 2 << LOG_CONS
    while n >= 2:
        assert ev == ol.sc_eval_quad(*host), ("phase 2", n)
        r = chal(pr, ev)
        host = [ol.bind_top(h, r) for h in host]
        if n >= 4:
            ev = ctx.sc_bind_eval_quad(*dev, r)
        else:
            ctx.bind_top_many(dev, r)
        n //= 2
    assert [ctx.table_read0(t) for t in dev] == [h[:32] for h in host], "phase 2 final claims"
    for t in dev:
        t.free()
    opening(ctx, ol, pr, sbn, z, hz, wl, wr, b"gens_r1cs_sat", 20)
    z.free(); gens_w.free()

    # ---------------------------------------------------------------- eq tables, derefs, derefs commitment (sparse_mlpoly_full.rs:1713-1727)
    nops = 1 << LOG_OPS
    rng = np.random.default_rng(9)
    addr = [rng.integers(0, 1 << (LOG_MEM - 1), size=nops, dtype=np.uint32) for _ in range(6)]
    for a in addr:
        a[int(nops * 0.57):] = 0                                   # ~43 % padded ops read cell 0
    read_ts = [rng.integers(0, 64, size=nops, dtype=np.uint32) for _ in range(6)]
    audit_ts = [rng.integers(0, 64, size=1 << LOG_MEM, dtype=np.uint32) for _ in range(2)]

    def up(a):
        p = ctx.dev_alloc(a.nbytes); ctx.dev_upload(p, a.tobytes()); return p
    d_addr, d_rts, d_ats = [up(a) for a in addr], [up(a) for a in read_ts], [up(a) for a in audit_ts]
    rx, ry = rand_scalars(LOG_MEM, 30), rand_scalars(LOG_MEM, 31)
    mem = [ctx.eq_evals(rx), ctx.eq_evals(ry)]; hmem = [ol.eq_evals(rx), ol.eq_evals(ry)]
    assert [ctx.table_download(t) for t in mem] == hmem, "eq tables"
    comb = ctx.gather_merge([mem[0]] * 3 + [mem[1]] * 3, d_addr, nops)
    hval = [b"".join(hmem[k // 3][32 * int(i):32 * int(i) + 32] for i in addr[k]) for k in range(6)]
    hcomb = b"".join(hval) + bytes(32 * (8 * nops - 6 * nops))
    assert ctx.table_download(comb) == hcomb, "derefs gather + merge"
    dl = (LOG_OPS + 3) // 2; dr = LOG_OPS + 3 - dl
    gens_d, gd = ctx.gens_new(1 << dr, b"gens_r1cs_eval")
    out, infs = ctx.commit_table(gens_d, comb, None, 1 << dl, 1 << dr)
    want = ol.commit_rows(hcomb, None, 1 << dl, 1 << dr, gd[:64 << dr], gd[64 << dr:], 4)
    assert out == want and sbn.g1_compress(out) == b"".join(ol.g1_compress(want[64 * i:64 * i + 64]) for i in range(1 << dl)), "derefs commitment"
    ctx.bases_precompute(gens_d, 64 << 20)                         # the fixed-base lookup path gives the same commitment
    assert ctx.commit_table(gens_d, comb, None, 1 << dl, 1 << dr)[0] == want
    gens_d.free()

    # ---------------------------------------------------------------- network construction (sparse_mlpoly_full.rs:745-796, product_tree.rs:39-57)
    g_, tau_ = rand_scalars(1, 40), rand_scalars(1, 41)
    ops_circ, mem_circ, h_ops, h_mem = [], [], [], []
    for side in range(2):
        mem_circ.append([ctx.hash_layer(None, mem[side], None, 0, g_, tau_)]); h_mem.append([ol.hash_layer(None, hmem[side], None, 0, g_, tau_)])
        mem_circ.append([ctx.hash_layer(None, mem[side], d_ats[side], 0, g_, tau_)]); h_mem.append([ol.hash_layer(None, hmem[side], audit_ts[side], 0, g_, tau_)])
        for i in range(3):
            k = 3 * side + i
            val = ctx.gather_merge([mem[side]], [d_addr[k]], nops)
            for ts_add in (0, 1):                                  # read set, write set
                ops_circ.append([ctx.hash_layer(d_addr[k], val, d_rts[k], ts_add, g_, tau_)]); h_ops.append([ol.hash_layer(addr[k], hval[k], read_ts[k], ts_add, g_, tau_)])
            val.free()
    for circ, hc in zip(ops_circ + mem_circ, h_ops + h_mem):
        assert ctx.table_download(circ[0]) == hc[0], "hash layer"
        layers = ctx.product_circuit(circ[0])
        w = hc[0]
        for x in layers:
            w = ol.product_layer(w); hc.append(w)
            assert ctx.table_download(x) == w, "product layer"
        circ.extend(layers[:-1]); layers[-1].free(); hc.pop()

    # ---------------------------------------------------------------- network proof: batched cubic sumchecks, layer by layer (product_tree.rs:261-340)
    def layered(circs, hcs, extra, hextra, seed):
        nl = len(circs[0]); rounds = 0
        for lay in range(nl - 1, -1, -1):
            halves = [ctx.table_halves(c[lay]) for c in circs]
            As, Bs = [h[0] for h in halves], [h[1] for h in halves]
            half = len(hcs[0][lay]) // 2
            hA, hB = [h[lay][:half] for h in hcs], [h[lay][half:] for h in hcs]
            k = (half // 32).bit_length() - 1
            rC = rand_scalars(max(k, 1), seed + lay)[:32 * k]
            Cpar = ctx.eq_evals(rC) if k else ctx.table_upload((1).to_bytes(32, "little"))
            hCpar = ol.eq_evals(rC) if k else (1).to_bytes(32, "little")
            Cs, hC = [Cpar] * len(As), [hCpar] * len(As)
            own = [Cpar]
            if lay == 0 and extra:
                As, Bs, Cs = As + extra[0], Bs + extra[1], Cs + extra[2]
                hA, hB, hC = hA + hextra[0], hB + hextra[1], hC + hextra[2]
            tabs = list({id(t): t for t in As + Bs + Cs}.values())
            n = half // 32
            ev = ctx.sc_eval_cubic_batched(As, Bs, Cs) if n >= 2 else b""
            while n >= 2:
                assert ev == b"".join(ol.sc_eval_cubic(a, b, c) for a, b, c in zip(hA, hB, hC)), ("cubic round", lay, n)
                r = chal(pr, ev); rounds += 1
                shared = {id(x): ol.bind_top(x, r) for x in {id(x): x for x in hC}.values()}   # a shared C is bound once, as on the device
                hA, hB = [ol.bind_top(x, r) for x in hA], [ol.bind_top(x, r) for x in hB]
                hC = [shared[id(x)] for x in hC]
                if n >= 4:
                    ev = ctx.sc_bind_eval_cubic_batched(As, Bs, Cs, r)
                else:
                    ctx.bind_top_many(tabs, r)
                n //= 2
            assert [ctx.table_read0(t) for t in As + Bs] == [x[:32] for x in hA + hB], ("final claims", lay)
            assert ctx.table_read0(Cpar) == hC[0][:32]
            for h in halves:
                h[0].free(); h[1].free()
            for t in own:
                t.free()
        return rounds
    hd = [[rand_scalars(nops // 2, 60 + 6 * g + j) for j in range(6)] for g in range(3)]
    dotp = [[ctx.table_upload(x) for x in grp] for grp in hd]
    assert layered(ops_circ, h_ops, dotp, hd, 100) == sum(range(1, LOG_OPS))      # layers of 2^1 .. 2^(LOG_OPS-1) entries per half
    assert layered(mem_circ, h_mem, None, None, 200) == sum(range(1, LOG_MEM))
    for grp in dotp:
        for t in grp:
            t.free()
    for circ in ops_circ + mem_circ:
        for t in circ:
            t.free()

    # ---------------------------------------------------------------- evaluations and the derefs opening (sparse_mlpoly_full.rs:907-1010)
    r_ops = rand_scalars(LOG_OPS + 3, 70)
    hchi = ol.eq_evals(r_ops)
    assert ctx.table_evaluate_many([comb], r_ops) == ol.fr_dot(hcomb, hchi), "evaluate"
    opening(ctx, ol, pr, sbn, comb, hcomb, dl, dr, b"gens_r1cs_eval", 80)
    comb.free()
    for t in mem:
        t.free()
    for p in d_addr + d_rts + d_ats:
        ctx.dev_free(p)

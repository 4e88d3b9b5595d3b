"""GPU parity tests of the sumcheck-round / bind / eq kernels through the C ABI.  Integer work: bit-exact."""
import hashlib

import pytest
from conftest import fr_bytes, golden, rand_scalars

pytestmark = pytest.mark.gpu
H = bytes.fromhex


def test_golden_rounds(ctx):
    g = golden("sumcheck_kat.json")
    for r in g["rounds"]:
        T, A, B, C = (b"".join(H(x) for x in r[k]) for k in "TABC")
        tT, tA, tB, tC = (ctx.table_upload(x) for x in (T, A, B, C))
        assert ctx.sc_eval_cubic(tA, tB, tC) == b"".join(H(x) for x in r["cubic"])
        assert ctx.sc_eval_r1cs(tT, tA, tB, tC) == b"".join(H(x) for x in r["r1cs"])
        assert ctx.sc_eval_quad(tA, tB) == b"".join(H(x) for x in r["quad"])
        for t in (tT, tA, tB, tC):
            t.free()
    for e in g["eq"]:
        t = ctx.eq_evals(b"".join(H(x) for x in e["r"]))
        assert ctx.table_download(t) == b"".join(H(x) for x in e["evals"])
        t.free()
    for b in g["bind"]:
        t = ctx.table_upload(b"".join(H(x) for x in b["Z"]))
        ctx.bind_top(t, H(b["r"]))
        assert ctx.table_download(t) == b"".join(H(x) for x in b["out"])
        t.free()


@pytest.mark.parametrize("n", [2, 4, 256, 1 << 12, 1 << 17])
def test_rounds_vs_oracle(ctx, ol, n):
    T, A, B, C = (rand_scalars(n, s + n) for s in (1, 2, 3, 4))
    tT, tA, tB, tC = (ctx.table_upload(x) for x in (T, A, B, C))
    assert ctx.sc_eval_cubic(tA, tB, tC) == ol.sc_eval_cubic(A, B, C)
    assert ctx.sc_eval_r1cs(tT, tA, tB, tC) == ol.sc_eval_r1cs(T, A, B, C)
    assert ctx.sc_eval_quad(tT, tC) == ol.sc_eval_quad(T, C)
    # prove_cubic_batched: "par" instances share C (sumcheck.rs:201-235), "seq" have their own (238-267)
    got = ctx.sc_eval_cubic_batched([tA, tB, tT], [tB, tT, tA], [tC, tC, tB])
    assert got == ol.sc_eval_cubic(A, B, C) + ol.sc_eval_cubic(B, T, C) + ol.sc_eval_cubic(T, A, B)
    r = rand_scalars(1, 5)
    ctx.bind_top_many([tA, tB, tC], r)
    ctx.bind_top(tT, r)
    for t, h in ((tA, A), (tB, B), (tC, C), (tT, T)):
        assert len(t) == n // 2
        assert ctx.table_download(t) == ol.bind_top(h, r)
    assert ctx.table_read0(tA) == ol.bind_top(A, r)[:32]
    for t in (tT, tA, tB, tC):
        t.free()


def test_full_prove_cubic_loop(ctx, ol, pr):
    """The reference's prove_cubic loop (sumcheck.rs:105-161) driven round by round: device evals -> host UniPoly ->
    challenge -> device binds; compared with the same loop on the oracle, plus the verifier's relations."""
    n = 1 << 10
    A, B, C = (rand_scalars(n, s) for s in (11, 12, 13))
    tA, tB, tC = (ctx.table_upload(x) for x in (A, B, C))
    prod = sum(int.from_bytes(A[32 * i:32 * i + 32], "little") * int.from_bytes(B[32 * i:32 * i + 32], "little") * int.from_bytes(C[32 * i:32 * i + 32], "little") for i in range(n)) % pr.R
    claim = prod
    hA, hB, hC = A, B, C
    for rnd in range(10):
        ev = ctx.sc_eval_cubic(tA, tB, tC)
        assert ev == ol.sc_eval_cubic(hA, hB, hC)
        e0, e2, e3 = (int.from_bytes(ev[32 * i:32 * i + 32], "little") for i in range(3))
        co = ol.unipoly_from_evals(fr_bytes([e0, (claim - e0) % pr.R, e2, e3]))      # host side (unipoly.rs:28-59)
        cs = [int.from_bytes(co[32 * i:32 * i + 32], "little") for i in range(4)]
        assert (cs[0] + sum(cs)) % pr.R == claim
        r = pr.scalar_to_bytes(int.from_bytes(hashlib.sha3_256(co).digest(), "little") % pr.R)
        ctx.bind_top_many([tA, tB, tC], r)
        hA, hB, hC = ol.bind_top(hA, r), ol.bind_top(hB, r), ol.bind_top(hC, r)
        claim = int.from_bytes(ol.unipoly_eval(co, r), "little")
    a0, b0, c0 = (int.from_bytes(ctx.table_read0(t), "little") for t in (tA, tB, tC))
    assert a0 * b0 * c0 % pr.R == claim
    for t in (tA, tB, tC):
        t.free()


@pytest.mark.parametrize("n", [4, 8, 16, 64, 256, 512, 1 << 13, 1 << 16])
def test_fused_bind_eval_vs_oracle(ctx, ol, n):
    """fused round == bind_top on every table, then the next round's sums (one pass instead of two).  n <= 256: the four-lanes-per-
    index kernel of the last rounds (1, 2, 4, 16, 64 indices: part of a wave, one wave, four waves); 512: the single-launch kernel;
    2^13: several blocks per instance with the ticketed fold; 2^16: the streaming kernel"""
    T, A, B, C = (rand_scalars(n, s + 3 * n) for s in (1, 2, 3, 4))
    r = rand_scalars(1, 9)
    bT, bA, bB, bC = (ol.bind_top(x, r) for x in (T, A, B, C))
    tT, tA, tB, tC = (ctx.table_upload(x) for x in (T, A, B, C))
    # batched cubic with a shared C ("par" instances, sumcheck.rs:201-235) and a table used in two roles
    got = ctx.sc_bind_eval_cubic_batched([tA, tB, tT], [tB, tT, tA], [tC, tC, tC], r)
    assert got == ol.sc_eval_cubic(bA, bB, bC) + ol.sc_eval_cubic(bB, bT, bC) + ol.sc_eval_cubic(bT, bA, bC)
    for t, h in ((tT, bT), (tA, bA), (tB, bB), (tC, bC)):
        assert len(t) == n // 2 and ctx.table_download(t) == h
    for t in (tT, tA, tB, tC):
        t.free()
    tT, tA, tB, tC = (ctx.table_upload(x) for x in (T, A, B, C))
    assert ctx.sc_bind_eval_r1cs(tT, tA, tB, tC, r) == ol.sc_eval_r1cs(bT, bA, bB, bC)
    assert ctx.table_download(tT) == bT and ctx.table_download(tC) == bC
    for t in (tT, tA, tB, tC):
        t.free()
    tT, tA = ctx.table_upload(T), ctx.table_upload(A)
    assert ctx.sc_bind_eval_quad(tT, tA, r) == ol.sc_eval_quad(bT, bA)
    assert ctx.table_download(tA) == bA
    # mixing with the unfused calls afterwards keeps working (buffers were swapped)
    if n >= 8:
        r2 = rand_scalars(1, 10)
        ctx.bind_top_many([tT, tA], r2)
        assert ctx.table_download(tT) == ol.bind_top(bT, r2)
    tT.free(); tA.free()


def test_full_prove_cubic_loop_fused(ctx, ol, pr):
    """prove_cubic_batched driven with ONE device call per round (bind of round j fused with the sums of round j+1)"""
    n = 1 << 11
    A1, B1, A2, B2, C = (rand_scalars(n, s) for s in (21, 22, 23, 24, 25))
    ts = [ctx.table_upload(x) for x in (A1, B1, A2, B2, C)]
    tA1, tB1, tA2, tB2, tC = ts
    hs = [A1, B1, A2, B2, C]
    ev = ctx.sc_eval_cubic_batched([tA1, tA2], [tB1, tB2], [tC, tC])
    rounds = 11
    for rnd in range(rounds):
        assert ev == ol.sc_eval_cubic(hs[0], hs[1], hs[4]) + ol.sc_eval_cubic(hs[2], hs[3], hs[4]), rnd
        r = pr.scalar_to_bytes(int.from_bytes(hashlib.sha3_256(ev).digest(), "little") % pr.R)
        hs = [ol.bind_top(h, r) for h in hs]
        if len(tA1) >= 4:
            ev = ctx.sc_bind_eval_cubic_batched([tA1, tA2], [tB1, tB2], [tC, tC], r)
        else:
            ctx.bind_top_many(ts, r)          # last round: nothing left to evaluate
    assert [ctx.table_read0(t) for t in ts] == [h[:32] for h in hs]
    assert ctx.table_read0_many(ts) == [h[:32] for h in hs]
    for t in ts:
        t.free()


def test_read0_many(ctx, sbn):
    """sbn_table_read0_many: entry 0 of many separate tables in one launch (more than one 64-table chunk; empty list; null table)"""
    tabs = [rand_scalars(1 << (i % 4), 900 + i) for i in range(70)]
    ts = [ctx.table_upload(x) for x in tabs]
    assert ctx.table_read0_many(ts) == [x[:32] for x in tabs]
    assert ctx.table_read0_many(ts[:1]) == [tabs[0][:32]]
    assert ctx.table_read0_many([]) == []
    import ctypes as C
    arr = (C.c_void_p * 2)(ts[0].h, None); out = (C.c_uint8 * 64)()
    assert sbn.binding.lib().sbn_table_read0_many(ctx.h, arr, C.c_size_t(2), out) != 0
    for t in ts:
        t.free()


@pytest.mark.parametrize("ell", [0, 1, 2, 7, 12, 13, 15, 16, 21])      # 12 levels direct, then pairs of levels (+ one single)
def test_eq_evals(ctx, ol, ell):
    r = rand_scalars(max(ell, 1), 3)[:32 * ell]
    t = ctx.eq_evals(r)
    want = ol.eq_evals(r) if ell else (1).to_bytes(32, "little")
    assert ctx.table_download(t) == want
    t.free()


def test_table_errors(ctx, sbn, pr):
    with pytest.raises(sbn.SbnError):
        ctx.table_upload(rand_scalars(3, 1))                 # not a power of two
    t = ctx.table_upload(rand_scalars(1, 1))
    with pytest.raises(sbn.SbnError):
        ctx.bind_top(t, rand_scalars(1, 2))                  # no variable left
    t2 = ctx.table_upload(rand_scalars(4, 1))
    with pytest.raises(sbn.SbnError):
        ctx.bind_top(t2, pr.R.to_bytes(32, "little"))        # non-canonical challenge (scalar.rs:87-95)
    t4 = ctx.table_upload(rand_scalars(8, 1))
    with pytest.raises(sbn.SbnError):
        ctx.sc_eval_cubic(t2, t2, t4)                        # length mismatch
    for x in (t, t2, t4):
        x.free()


def test_full_size_round_properties(ctx, ol, pr):
    """Product-circuit layer-0 size (tables of 2^21) checked through size-independent properties:
    the round sums are additive over index ranges (a checksum of checksums), and the bind is checked on slices."""
    import torch
    n = 1 << 21
    raw = [torch.empty((n, 8), dtype=torch.int32, device="cuda") for _ in range(4)]
    for k, x in enumerate(raw):
        ctx.scalars_synthetic(0x5BA27A2B4E254 + 500 + k, 0, n, x.data_ptr())     # uniform in Fr, full width (bits 252 / 253 set on ~3/4)
    torch.cuda.synchronize()
    assert int((raw[0][:, 7] >> 28).max()) >= 2                                   # values above 2^253 are present
    h, q = n // 2, n // 4

    def tables(parts):
        return [ctx.table_from_dev(x.data_ptr(), x.shape[0]) for x in parts]

    def ints(b, k):
        return [int.from_bytes(b[32 * i:32 * i + 32], "little") for i in range(k)]

    full = tables(raw)
    # index range [0, h) split into [0, q) and [q, h): build the two half-size problems with torch.cat on the device
    first = tables([torch.cat([x[:q], x[h:h + q]]).contiguous() for x in raw])
    second = tables([torch.cat([x[q:h], x[h + q:]]).contiguous() for x in raw])
    torch.cuda.synchronize()
    for kind, k in (("cubic", 3), ("r1cs", 3), ("quad", 2)):
        def ev(ts):
            if kind == "cubic":
                return ints(ctx.sc_eval_cubic(ts[1], ts[2], ts[3]), 3)
            if kind == "r1cs":
                return ints(ctx.sc_eval_r1cs(ts[0], ts[1], ts[2], ts[3]), 3)
            return ints(ctx.sc_eval_quad(ts[0], ts[1]), 2)
        a, b, c = ev(full), ev(first), ev(second)
        assert a == [(x + y) % pr.R for x, y in zip(b, c)], kind
    # bind at full size, verified on two slices against the oracle
    r = rand_scalars(1, 77)
    ctx.bind_top_many(full, r)
    got = ctx.table_download(full[0])
    x = raw[0]
    for lo in (0, h - 512):
        small = torch.cat([x[lo:lo + 512], x[h + lo:h + lo + 512]]).cpu().numpy().tobytes()
        assert got[32 * lo:32 * (lo + 512)] == ol.bind_top(small, r)
    for t in full + first + second:
        t.free()


def test_derefs_gather_and_commit(ctx, ol, pr, sbn):
    """MultiSparseMatPolynomialAsDense::deref + Derefs::new + Derefs::commit entirely on the device:
    eq tables (hyrax.rs:355-369) -> gather by address (sparse_mlpoly_full.rs:245-257) -> merge with zero padding
    (hyrax.rs:237-247) -> Hyrax commit (hyrax.rs:283-308).  Small instance of the keyless structure: 3 + 3 polys."""
    import numpy as np
    ell, nops = 6, 40                                    # 64 memory cells, 40 ops per matrix (padded total 6*40 -> 256)
    rx, ry = rand_scalars(ell, 1), rand_scalars(ell, 2)
    mem_rx, mem_ry = ctx.eq_evals(rx), ctx.eq_evals(ry)
    hrx, hry = ol.eq_evals(rx), ol.eq_evals(ry)
    rng = np.random.default_rng(5)
    addrs = [rng.integers(0, 1 << ell, size=nops, dtype=np.uint32) for _ in range(6)]
    addrs[1][10:] = 0                                    # padded ops read cell 0 (sparse_mlpoly_full.rs:89-101)
    dptrs = []
    for a in addrs:
        p = ctx.dev_alloc(a.nbytes); ctx.dev_upload(p, a.tobytes()); dptrs.append(p)
    comb = ctx.gather_merge([mem_rx] * 3 + [mem_ry] * 3, dptrs, nops)
    want = b"".join(b"".join((hrx if k < 3 else hry)[32 * int(i):32 * int(i) + 32] for i in addrs[k]) for k in range(6))
    want += bytes(32 * (256 - 6 * nops))
    assert len(comb) == 256 and ctx.table_download(comb) == want
    # commit the comb polynomial: ell = 8 -> L = 16 rows, R = 16 columns (hyrax.rs:371-373)
    L, R = 1 << sbn.factored_lens(8)[0], 1 << sbn.factored_lens(8)[1]
    bases, gxy = ctx.gens_new(R, b"gens_r1cs_eval")
    out, infs = ctx.commit_table(bases, comb, None, L, R)
    assert out == ol.commit_rows(want, None, L, R, gxy[:64 * R], gxy[64 * R:], 4)
    bl = rand_scalars(L, 9)
    assert ctx.commit_table(bases, comb, bl, L, R)[0] == ol.commit_rows(want, bl, L, R, gxy[:64 * R], gxy[64 * R:], 4)
    # an address outside its table is rejected (sparse_mlpoly_full.rs:228 assert)
    bad = np.full(nops, 1 << ell, dtype=np.uint32); pb = ctx.dev_alloc(bad.nbytes); ctx.dev_upload(pb, bad.tobytes())
    with pytest.raises(sbn.SbnError):
        ctx.gather_merge([mem_rx], [pb], nops)
    for p in dptrs + [pb]:
        ctx.dev_free(p)
    for t in (comb, mem_rx, mem_ry):
        t.free()
    bases.free()


@pytest.mark.parametrize("ell", [1, 4, 9, 13])
def test_opening_pieces_vs_oracle(ctx, ol, sbn, ell):
    """SURVEY 8f-2, the vector work of PolyEvalProof::prove (hyrax.rs:65-116): L, R = compute_factored_evals (hyrax.rs:375-383),
    LZ = poly.bound(L) (hyrax.rs:311-324), evaluate (hyrax.rs:217-222) and the identity <LZ, R> == Z(r)."""
    n = 1 << ell
    Z = rand_scalars(n, 100 + ell); r = rand_scalars(ell, 200 + ell)
    tZ = ctx.table_upload(Z)
    lv, rv = sbn.factored_lens(ell)
    L_size, R_size = 1 << lv, 1 << rv
    tL, tR = ctx.eq_evals(r[:32 * lv]), ctx.eq_evals(r[32 * lv:])
    hL = ol.eq_evals(r[:32 * lv]) if lv else (1).to_bytes(32, "little")
    hR = ol.eq_evals(r[32 * lv:])
    tLZ = ctx.table_bound(tZ, tL)
    want_LZ = ol.bound(Z, hL, L_size, R_size)
    assert len(tLZ) == R_size and ctx.table_download(tLZ) == want_LZ
    zr = ctx.table_evaluate(tZ, r)
    assert zr == ol.fr_dot(Z, ol.eq_evals(r))
    assert ctx.table_dot(tLZ, tR) == zr == ol.fr_dot(want_LZ, hR)
    for t in (tZ, tL, tR, tLZ):
        t.free()


def test_dense_poly_reference_evaluate(ctx):
    """hyrax.rs:435-453: Z=[1,2,3,4] evaluates to 1 at (0,0) and 4 at (1,1)"""
    t = ctx.table_upload(fr_bytes([1, 2, 3, 4]))
    assert ctx.table_evaluate(t, fr_bytes([0, 0])) == fr_bytes([1])
    assert ctx.table_evaluate(t, fr_bytes([1, 1])) == fr_bytes([4])
    t.free()


def test_network_construction_pieces(ctx, ol, pr):
    """Layers::build_hash_layer (sparse_mlpoly_full.rs:745-796) and ProductCircuit::new (product_tree.rs:39-57) on the
    device, then the layer-0 cubic sumcheck of the product circuit on the halves views (product_tree.rs:267-332)."""
    import numpy as np
    ell, nops = 7, 64
    rx = rand_scalars(ell, 1)
    mem = ctx.eq_evals(rx); hmem = ol.eq_evals(rx)
    rng = np.random.default_rng(3)
    addr = rng.integers(0, 1 << ell, size=nops, dtype=np.uint32)
    ts = rng.integers(0, 50, size=nops, dtype=np.uint32)
    audit = rng.integers(0, 9, size=1 << ell, dtype=np.uint32)
    g, tau = rand_scalars(1, 5), rand_scalars(1, 6)
    d_addr = ctx.dev_alloc(addr.nbytes); ctx.dev_upload(d_addr, addr.tobytes())
    d_ts = ctx.dev_alloc(ts.nbytes); ctx.dev_upload(d_ts, ts.tobytes())
    d_au = ctx.dev_alloc(audit.nbytes); ctx.dev_upload(d_au, audit.tobytes())
    deref = ctx.gather_merge([mem], [d_addr], nops)                      # row_ops_val for this matrix
    hderef = b"".join(hmem[32 * int(a):32 * int(a) + 32] for a in addr)
    t_init = ctx.hash_layer(None, mem, None, 0, g, tau)
    t_audit = ctx.hash_layer(None, mem, d_au, 0, g, tau)
    t_read = ctx.hash_layer(d_addr, deref, d_ts, 0, g, tau)
    t_write = ctx.hash_layer(d_addr, deref, d_ts, 1, g, tau)
    assert ctx.table_download(t_init) == ol.hash_layer(None, hmem, None, 0, g, tau)
    assert ctx.table_download(t_audit) == ol.hash_layer(None, hmem, audit, 0, g, tau)
    assert ctx.table_download(t_read) == ol.hash_layer(addr, hderef, ts, 0, g, tau)
    assert ctx.table_download(t_write) == ol.hash_layer(addr, hderef, ts, 1, g, tau)
    # the pairs in one pass each (sbn_hash_layer_pair): read + write of one polynomial, init + audit of a memory
    p_read, p_write = ctx.hash_layer_pair(d_addr, deref, d_ts, 0, d_ts, 1, g, tau)
    p_init, p_audit = ctx.hash_layer_pair(None, mem, None, 0, d_au, 0, g, tau)
    for got, want in ((p_read, t_read), (p_write, t_write), (p_init, t_init), (p_audit, t_audit)):
        assert ctx.table_download(got) == ctx.table_download(want)
        got.free()
    # product circuit layers of the read set, down to two entries, against the oracle; evaluate() = product of everything
    cur, hcur = t_read, ol.hash_layer(addr, hderef, ts, 0, g, tau)
    layers = [cur]
    while len(cur) > 2:
        nxt = ctx.product_layer(cur); hcur = ol.product_layer(hcur)
        assert ctx.table_download(nxt) == hcur
        layers.append(nxt); cur = nxt
    total = 1
    hr = ol.hash_layer(addr, hderef, ts, 0, g, tau)
    for j in range(nops):
        total = total * int.from_bytes(hr[32 * j:32 * j + 32], "little") % pr.R
    assert int.from_bytes(hcur[:32], "little") * int.from_bytes(hcur[32:], "little") % pr.R == total
    # halves views feed the sumcheck: sum_i left[i]*right[i]*eq[i] over layer 0 (product_tree.rs:271-332)
    left, right = ctx.table_halves(layers[0])
    hl, hrr = hr[:32 * nops // 2], hr[32 * nops // 2:]
    assert ctx.table_download(left) == hl and ctx.table_download(right) == hrr
    # slices: any power-of-two run of entries as a view (the polynomials merge() laid end to end: Derefs' row / col polys inside comb)
    for first, ln in ((0, nops), (nops // 4, nops // 4), (nops - 2, 2), (3, 1)):
        sl = ctx.table_slice(layers[0], first, ln)
        assert len(sl) == ln and ctx.table_download(sl) == hr[32 * first:32 * (first + ln)]
        sl.free()
    for first, ln in ((0, 3), (nops - 1, 2), (nops + 1, 1), (0, 0)):
        with pytest.raises(Exception):
            ctx.table_slice(layers[0], first, ln)
    eqr = rand_scalars(5, 8); tC = ctx.eq_evals(eqr); hC = ol.eq_evals(eqr)
    assert ctx.sc_eval_cubic(left, right, tC) == ol.sc_eval_cubic(hl, hrr, hC)
    r = rand_scalars(1, 9)
    assert ctx.sc_bind_eval_cubic_batched([left], [right], [tC], r) == ol.sc_eval_cubic(ol.bind_top(hl, r), ol.bind_top(hrr, r), ol.bind_top(hC, r))
    assert ctx.table_download(left) == ol.bind_top(hl, r)                 # the view now lives in its own second buffer
    for t in (left, right, tC):
        t.free()
    for t in layers[1:] + [t_init, t_audit, t_read, t_write, deref, mem]:
        t.free()
    for p in (d_addr, d_ts, d_au):
        ctx.dev_free(p)


@pytest.mark.parametrize("count", [24, 25, 40])
def test_many_instances_both_launch_paths(ctx, ol, count):
    """up to 24 instances travel as kernel arguments and are folded by the last block (one launch per round); more use the staged
    argument array + finishing kernel.  Both must give the oracle's values, repeatedly (the ticket counters reset themselves)."""
    n = 256
    tabs = [rand_scalars(n, 900 + i) for i in range(count + 2)]
    dev = [ctx.table_upload(x) for x in tabs]
    idx = [(i, i + 1, i + 2) for i in range(count)]
    As, Bs, Cs = ([dev[t[k]] for t in idx] for k in range(3))
    want = b"".join(ol.sc_eval_cubic(tabs[a], tabs[b], tabs[c]) for a, b, c in idx)
    for _ in range(3):
        assert ctx.sc_eval_cubic_batched(As, Bs, Cs) == want
    r = rand_scalars(1, 77)
    bound = [ol.bind_top(x, r) for x in tabs]
    want2 = b"".join(ol.sc_eval_cubic(bound[a], bound[b], bound[c]) for a, b, c in idx)
    assert ctx.sc_bind_eval_cubic_batched(As, Bs, Cs, r) == want2
    for t, h in zip(dev, bound):
        assert ctx.table_download(t) == h
    r2 = rand_scalars(1, 78)
    bound2 = [ol.bind_top(x, r2) for x in bound]
    assert ctx.sc_bind_eval_cubic_batched(As, Bs, Cs, r2) == b"".join(ol.sc_eval_cubic(bound2[a], bound2[b], bound2[c]) for a, b, c in idx)
    for t in dev:
        t.free()


@pytest.mark.parametrize("ell,count", [(1, 1), (6, 5), (11, 30)])
def test_evaluate_many_vs_oracle(ctx, ol, sbn, ell, count):
    """the evaluations of HashLayerProof::prove at one point (sparse_mlpoly_full.rs:907-976) with one shared eq table; more than
    24 tables go through two launches"""
    n = 1 << ell
    r = rand_scalars(ell, 31 + ell)
    chi = ol.eq_evals(r)
    tabs = [rand_scalars(n, 500 + i) for i in range(count)]
    dev = [ctx.table_upload(x) for x in tabs]
    got = ctx.table_evaluate_many(dev, r)
    assert got == b"".join(ol.fr_dot(x, chi) for x in tabs)
    assert got[:32] == ctx.table_evaluate(dev[0], r)
    short = ctx.table_upload(rand_scalars(max(1, n // 2), 7))
    with pytest.raises(sbn.SbnError):
        ctx.table_evaluate_many([dev[0], short], r)
    short.free()
    for t in dev:
        t.free()


def test_async_layers_then_free_keep_results(ctx, ol, pr):
    """sbn_hash_layer / sbn_product_layer return without waiting and sbn_table_free does not wait either: a chain that frees
    its inputs immediately must still produce the oracle's values (buffers are recycled in stream order)"""
    n = 1 << 12
    val = rand_scalars(n, 1234); g, tau = rand_scalars(1, 1), rand_scalars(1, 2)
    want = ol.hash_layer(None, val, None, 0, g, tau)
    for _ in range(3):
        tv = ctx.table_upload(val)
        h = ctx.hash_layer(None, tv, None, 0, g, tau)
        tv.free()
        layers = [h]
        while len(layers[-1]) > 1:
            layers.append(ctx.product_layer(layers[-1]))
            if len(layers) > 2:
                layers[-3].free(); layers[-3] = None
        w = want
        while len(w) > 32:
            w = ol.product_layer(w)
        assert ctx.table_download(layers[-1]) == w
        for t in layers:
            if t is not None:
                t.free()


def test_residue_sharded_sumcheck_two_contexts(sbn, ol, pr):
    """SURVEY 8e, sumcheck rows: tables sharded by index residue over 2 'ranks' (two contexts on this GPU, the exchange done in
    process: sharding.fr_sum is what follows the all-gather).  Every round's sums and the final claims must equal the unsharded
    device run and the oracle."""
    from spartan_bn254_amd import sharding
    n, world = 1 << 10, 2
    full = [rand_scalars(n, 300 + i) for i in range(5)]                        # A1, B1, A2, B2, C (C shared: "par" instances)
    ctxs = [sbn.Context(0) for _ in range(world)]
    ref = sbn.Context(0)
    try:
        loc = [[cx.table_upload(sharding.shard_table_residue(t, g, world)) for t in full] for g, cx in enumerate(ctxs)]
        whole = [ref.table_upload(t) for t in full]
        inst = lambda T: ([T[0], T[2]], [T[1], T[3]], [T[4], T[4]])
        chal = lambda ev: pr.scalar_to_bytes(int.from_bytes(hashlib.sha3_256(ev).digest(), "little") % pr.R)
        ev = sharding.fr_sum([cx.sc_eval_cubic_batched(*inst(T)) for cx, T in zip(ctxs, loc)])
        want = ref.sc_eval_cubic_batched(*inst(whole))
        host = list(full)
        while True:
            assert ev == want == ol.sc_eval_cubic(host[0], host[1], host[4]) + ol.sc_eval_cubic(host[2], host[3], host[4])
            r = chal(ev)
            host = [ol.bind_top(t, r) for t in host]
            m = len(loc[0][0])
            if m >= 4:
                ev = sharding.fr_sum([cx.sc_bind_eval_cubic_batched(*inst(T), r) for cx, T in zip(ctxs, loc)])
                want = ref.sc_bind_eval_cubic_batched(*inst(whole), r)
            else:                                                            # local tables go from 2 entries to 1
                for cx, T in zip(ctxs, loc):
                    cx.bind_top_many(T, r)
                ref.bind_top_many(whole, r)
                break
        # the residual world-entry tables: entry g comes from rank g
        rest = [b"".join(ctxs[g].table_read0(loc[g][t]) for g in range(world)) for t in range(5)]
        assert rest == host == [ref.table_download(t) for t in whole]
        for T in loc + [whole]:
            for t in T:
                t.free()
    finally:
        for cx in ctxs + [ref]:
            cx.close()


@pytest.mark.parametrize("count", [1, 64, 70])
def test_bind_many_both_paths(ctx, ol, count):
    """up to 64 tables: pointers + challenge as kernel arguments (no host wait); more: staged pointer array"""
    n = 64
    tabs = [rand_scalars(n, 1500 + i) for i in range(count)]
    dev = [ctx.table_upload(x) for x in tabs]
    r = rand_scalars(1, 42)
    ctx.bind_top_many(dev, r)
    r2 = rand_scalars(1, 43)
    ctx.bind_top_many(dev, r2)                     # back to back, nothing in between waits for the device
    for t, h in zip(dev, tabs):
        assert len(t) == n // 4 and ctx.table_download(t) == ol.bind_top(ol.bind_top(h, r), r2)
    for t in dev:
        t.free()


@pytest.mark.parametrize("logn", [1, 2, 9, 12, 15])        # 12, 15: streaming layers first, then the <= 2048-entry tail in one launch
def test_product_circuit_one_call(ctx, ol, sbn, logn):
    """ProductCircuit::new (product_tree.rs:39-57) in one call == repeated compute_layer (oracle), down to the product itself"""
    n = 1 << logn
    v = rand_scalars(n, 4000 + logn)
    t = ctx.table_upload(v)
    layers = ctx.product_circuit(t)
    assert [len(x) for x in layers] == [n >> (k + 1) for k in range(logn)]
    w = v
    for x in layers:
        w = ol.product_layer(w)
        assert ctx.table_download(x) == w
    for x in layers:
        x.free()
    one = ctx.table_upload(rand_scalars(1, 5))
    with pytest.raises(sbn.SbnError):
        ctx.product_circuit(one)
    one.free(); t.free()


@pytest.mark.parametrize("logn,count", [(1, 3), (5, 16), (11, 4), (12, 12), (14, 17)])   # 17 circuits: two batches; 2^12, 2^14: streaming layers + the tail
def test_product_circuit_many(ctx, ol, sbn, logn, count):
    """sbn_product_circuit_many: the circuits of several tables of one length built together (Layers::new, sparse_mlpoly_full.rs:813-823)
    == one sbn_product_circuit per table == the oracle's repeated compute_layer"""
    n = 1 << logn
    vs = [rand_scalars(n, 4100 + 7 * logn + i) for i in range(count)]
    ts = [ctx.table_upload(v) for v in vs]
    circ = ctx.product_circuit_many(ts)
    assert len(circ) == count
    for i, layers in enumerate(circ):
        assert [len(x) for x in layers] == [n >> (k + 1) for k in range(logn)]
        w = vs[i]
        for x in layers:
            w = ol.product_layer(w)
            assert ctx.table_download(x) == w, (i, len(x))
    single = ctx.product_circuit(ts[-1])
    assert [ctx.table_download(x) for x in single] == [ctx.table_download(x) for x in circ[-1]]
    for x in single + [x for layers in circ for x in layers]:
        x.free()
    other = ctx.table_upload(rand_scalars(2 * n, 9))
    with pytest.raises(sbn.SbnError):
        ctx.product_circuit_many([ts[0], other])          # one length per call
    other.free()
    for t in ts:
        t.free()


def test_reference_product_tree_known_answers(ctx, pr):
    """the reference's own unit-test values (product_tree.rs:544-590): ProductCircuit over [2, 3, 5, 7] evaluates to 210;
    DotProductCircuit([1,2,3,4], [5,6,7,8], weights 1) evaluates to 70"""
    t = ctx.table_upload(fr_bytes([2, 3, 5, 7]))
    layers = ctx.product_circuit(t)
    assert [len(x) for x in layers] == [2, 1]
    assert ctx.table_download(layers[0]) == fr_bytes([10, 21])            # compute_layer: left half * right half (product_tree.rs:21-37)
    assert ctx.table_download(layers[-1]) == fr_bytes([210])
    for x in layers:
        x.free()
    t.free()
    left, right, weight = (ctx.table_upload(fr_bytes(v)) for v in ([1, 2, 3, 4], [5, 6, 7, 8], [1, 1, 1, 1]))
    assert ctx.table_dot(left, right) == fr_bytes([70])
    # through the cubic round sums: e0 = the lower half of the index range (sumcheck.rs:111-135), e0 + e1 = the circuit's value
    ev = ctx.sc_eval_cubic(left, right, weight)
    assert int.from_bytes(ev[:32], "little") == 1 * 5 + 2 * 6
    for x in (left, right, weight):
        x.free()


def test_fused_round_unspilled_build(sbn, ol):
    """SBN_SC_WAVES=3 selects the 3-waves-per-SIMD build of the streaming fused round (no spills); same values"""
    import os
    n = 1 << 16
    A, B, C = (rand_scalars(n, s + 77) for s in (1, 2, 3))
    r = rand_scalars(1, 78)
    os.environ["SBN_SC_WAVES"] = "3"
    try:
        cx = sbn.Context(0)
    finally:
        del os.environ["SBN_SC_WAVES"]
    try:
        tA, tB, tC = (cx.table_upload(x) for x in (A, B, C))
        got = cx.sc_bind_eval_cubic_batched([tA, tB], [tB, tA], [tC, tC], r)
        bA, bB, bC = (ol.bind_top(x, r) for x in (A, B, C))
        assert got == ol.sc_eval_cubic(bA, bB, bC) + ol.sc_eval_cubic(bB, bA, bC)
        assert cx.table_download(tC) == bC
        for t in (tA, tB, tC):
            t.free()
    finally:
        cx.close()


def _uniform_tables(ctx, count, n, seed):
    """`count` tables of n uniform full-width Fr values (sbn_scalars_synthetic), as (device handles, host numpy uint8 arrays)"""
    import torch
    dev, host = [], []
    for k in range(count):
        x = torch.empty((n, 8), dtype=torch.int32, device="cuda")
        ctx.scalars_synthetic(0x5BA27A2B4E254 + seed + k, 0, n, x.data_ptr())
        torch.cuda.synchronize()
        host.append(x.cpu().numpy().view("uint8").reshape(-1))
        dev.append(ctx.table_from_dev(x.data_ptr(), n))
        del x
    return dev, host


def _run_batched_sumcheck(ctx, ol, n, n_par, n_seq, seed, threads=16):
    """prove_cubic_batched (sumcheck.rs:165-330) on the fused path with caller-chosen challenges: every round's per-instance triples
    and the final claims against the oracle's loop"""
    dev, host = _uniform_tables(ctx, 2 * n_par + 1 + 3 * n_seq, n, seed)
    Ap, Bp, Cp = dev[:n_par], dev[n_par:2 * n_par], dev[2 * n_par]
    o = 2 * n_par + 1
    As, Bs, Cs = dev[o:o + n_seq], dev[o + n_seq:o + 2 * n_seq], dev[o + 2 * n_seq:]
    rounds = n.bit_length() - 1
    ch = rand_scalars(rounds, seed + 1)
    co = rand_scalars(n_par + n_seq, seed + 2)
    As_all, Bs_all, Cs_all = Ap + As, Bp + Bs, [Cp] * n_par + Cs
    got = [ctx.sc_eval_cubic_batched(As_all, Bs_all, Cs_all)]
    for j in range(rounds):
        r = ch[32 * j:32 * j + 32]
        if len(dev[0]) >= 4:
            got.append(ctx.sc_bind_eval_cubic_batched(As_all, Bs_all, Cs_all, r))      # bind to r_j + the sums of round j + 1
        else:
            ctx.bind_top_many(dev, r)                                                   # last round: nothing left to evaluate
    finals = [ctx.table_read0(t) for t in dev]
    want_ev, _, want_fin = ol.sc_prove_cubic_batched(host[:n_par], host[n_par:2 * n_par], host[2 * n_par], host[o:o + n_seq], host[o + n_seq:o + 2 * n_seq], host[o + 2 * n_seq:], co, ch, threads)
    for t in dev:
        t.free()
    return got, finals, want_ev, want_fin


def test_keyless_layer0_sumcheck_every_round(ctx, ol):
    """The real workload shape of the ops product-circuit proof (12 "par" instances sharing C + 6 "seq" instances, tables of
    2^21 uniform full-width Fr values, 2.69 GiB in round 0) on the fused path: ALL 21 rounds' 18 triples and the 43 final values against
    the oracle's prove_cubic_batched loop — streaming kernel, ticketed fold, single-launch rounds and the four-lane tail in one run."""
    n = 1 << 21
    got, finals, want_ev, want_fin = _run_batched_sumcheck(ctx, ol, n, 12, 6, 9000)
    assert len(got) == 21
    for j, g in enumerate(got):
        assert g == b"".join(want_ev[j]), f"round {j}"
    assert finals == want_fin


def test_handover_under_uneven_load(sbn, ctx, ol):
    """The ticketed cross-block fold and the host mailbox of the round kernels (sumcheck_kernels.cuh: sc_last_block_fold) with the
    chip UNEVENLY loaded: a second context keeps 2^22-point MSMs running on its own stream while fused rounds run on 2^17 .. 2^18-entry
    tables, several blocks per instance; every word of every round is compared with the oracle.  On an idle chip a missing
    s_waitcnt vmcnt(0) ahead of the signal hides (MI355X_MICROARCH.md: 'stale only under uneven load')."""
    import threading
    import torch
    S0, DSTEP = 0x1234567, 0x89abcdef
    other = sbn.Context(0)
    stop = threading.Event()
    done = []

    def load():
        n = 1 << 22
        b = other.bases_synthetic(n, 0, S0.to_bytes(32, "little"), DSTEP.to_bytes(32, "little"))
        x = torch.empty((n, 8), dtype=torch.int32, device="cuda")
        other.scalars_synthetic(77, 0, n, x.data_ptr())
        first = None
        while not stop.is_set():
            out = other.msm_bases_dev(b, x.data_ptr(), n)
            first = first or out
            assert out == first
            done.append(1)
        b.free()

    th = threading.Thread(target=load)
    th.start()
    try:
        import time
        while not done and th.is_alive():
            time.sleep(0.05)                    # the load is running before the rounds start
        for rep, (logn, n_par, n_seq) in enumerate([(17, 5, 2), (18, 3, 1), (17, 12, 6), (15, 4, 0)]):
            got, finals, want_ev, want_fin = _run_batched_sumcheck(ctx, ol, 1 << logn, n_par, n_seq, 9500 + 100 * rep)
            for j, g in enumerate(got):
                assert g == b"".join(want_ev[j]), f"case {rep} round {j}"
            assert finals == want_fin
    finally:
        stop.set(); th.join()
        other.close()
    assert len(done) >= 2, "the competing MSM load did not run"


def _run_stateful_sumcheck(ctx, ol, n, n_par, n_seq, seed, coeffs=None, threads=16):
    """sbn_sumcheck_begin / round / finish == the oracle's prove_cubic_batched: the coeffs-combined (e0, e2, e3) of every round
    (sumcheck.rs:269-271) and the final claims (:302-318).  The caller's tables must come back untouched."""
    dev, host = _uniform_tables(ctx, 2 * n_par + (1 if n_par else 0) + 3 * n_seq, n, seed)
    o = 2 * n_par + (1 if n_par else 0)
    Ap, Bp, Cp = dev[:n_par], dev[n_par:2 * n_par], (dev[2 * n_par] if n_par else None)
    As, Bs, Cs = dev[o:o + n_seq], dev[o + n_seq:o + 2 * n_seq], dev[o + 2 * n_seq:]
    rounds = n.bit_length() - 1
    ch = rand_scalars(rounds, seed + 1)
    co = coeffs if coeffs is not None else rand_scalars(n_par + n_seq, seed + 2)
    st, ev = ctx.sumcheck_begin(Ap, Bp, Cp, As, Bs, Cs, co)
    got = [ev]
    for j in range(rounds):
        assert len(st) == n >> j
        ev = st.round(ch[32 * j:32 * j + 32])
        if j + 1 < rounds:
            got.append(ev)
        else:
            assert ev == bytes(96)
    finals = st.finish()
    st.free()
    _, want_comb, want_fin = ol.sc_prove_cubic_batched(host[:n_par], host[n_par:2 * n_par], host[2 * n_par] if n_par else None, host[o:o + n_seq], host[o + n_seq:o + 2 * n_seq], host[o + 2 * n_seq:], co, ch, threads)
    for j in range(rounds):
        assert got[j] == want_comb[j], f"round {j} of {rounds} (n = {n}, {n_par} par + {n_seq} seq)"
    assert finals == want_fin
    # the state never writes the caller's tables
    for k in (0, len(dev) - 1):
        assert len(dev[k]) == n
    if n <= 1 << 12:
        assert ctx.table_download(dev[0]) == host[0].tobytes()
    for t in dev:
        t.free()


@pytest.mark.parametrize("logn,n_par,n_seq", [(1, 1, 0), (1, 3, 2), (2, 2, 1), (3, 12, 6), (6, 1, 1), (9, 4, 0), (10, 12, 6), (10, 0, 3), (12, 18, 6), (14, 5, 2)])
def test_stateful_sumcheck_small(ctx, ol, logn, n_par, n_seq):
    """small tables (below 2^16 entries the "par" A tables are scaled by coeffs when the sumcheck begins, per-instance kernels after that)"""
    _run_stateful_sumcheck(ctx, ol, 1 << logn, n_par, n_seq, 7000 + 10 * logn + n_par)


@pytest.mark.parametrize("logn,n_par,n_seq", [(15, 2, 0), (15, 12, 6), (16, 4, 0), (16, 13, 3), (17, 16, 0), (17, 20, 4), (18, 7, 1), (19, 12, 0)])
def test_stateful_sumcheck_combined_kernels(ctx, ol, logn, n_par, n_seq):
    """2^15 (still the scaled per-instance path) and from 2^16 entries the combined kernels (coefficients folded into A at the first bind, one reduction per index for
    all instances, several groups when the round is small or the instances many), then the per-instance kernels on the scaled tables"""
    _run_stateful_sumcheck(ctx, ol, 1 << logn, n_par, n_seq, 7500 + 10 * logn + n_par)


def test_stateful_sumcheck_modes_agree(ctx, sbn, ol, pr):
    """a zero coefficient among the "par" instances (no inverse: nothing is scaled, the host weighs every triple), and SBN_SC_NO_COMB /
    SBN_SC_NO_COMB_KERNEL forcing the other two modes onto a size the combined kernels would take, SBN_SC_NO_MIXED the separate launches of
    the "par" groups and the "seq" instances, SBN_SC_NO_MIXED_EVAL the two-launch round 0 (k_sc_comb_eval + k_sc_eval with n_seq > 0 on
    2^17-entry tables): all equal to the oracle.  The overrides are read when a context is created: a fresh context per variant."""
    import os
    co = bytearray(rand_scalars(8, 4242)); co[32:64] = bytes(32)
    _run_stateful_sumcheck(ctx, ol, 1 << 15, 5, 3, 7900, bytes(co))
    _run_stateful_sumcheck(ctx, ol, 1 << 6, 5, 3, 7910, bytes(co))
    one = (1).to_bytes(32, "little"); top = (pr.R - 1).to_bytes(32, "little")
    _run_stateful_sumcheck(ctx, ol, 1 << 15, 3, 1, 7920, one + top + one + top)              # coefficients 1 and r - 1
    for var, logn in (("SBN_SC_NO_COMB", 16), ("SBN_SC_NO_COMB_KERNEL", 16), ("SBN_SC_NO_MIXED", 17), ("SBN_SC_NO_MIXED_EVAL", 17)):
        os.environ[var] = "1"
        try:
            cx = sbn.Context(0)
        finally:
            del os.environ[var]
        try:
            _run_stateful_sumcheck(cx, ol, 1 << logn, 4, 2, 7930 + logn)
        finally:
            cx.close()
    # grids asked for beyond the partial-sum area are clamped: 2^18-entry tables, 4096 blocks per group requested
    os.environ["SBN_SC_COMB_GRID"] = "4096"; os.environ["SBN_SC_COMB_EVAL_BLOCKS"] = "65536"
    try:
        cx = sbn.Context(0)
    finally:
        del os.environ["SBN_SC_COMB_GRID"]; del os.environ["SBN_SC_COMB_EVAL_BLOCKS"]
    try:
        _run_stateful_sumcheck(cx, ol, 1 << 18, 3, 1, 7990)
    finally:
        cx.close()


@pytest.mark.parametrize("logn,n_par,n_seq", [(1, 2, 0), (5, 3, 1), (10, 12, 6), (16, 4, 2), (18, 12, 0)])
def test_stateful_sumcheck_begin_eq(ctx, sbn, ol, logn, n_par, n_seq):
    """sbn_sumcheck_begin_eq: poly_C_par = EqPolynomial::new(rand).evals() (product_tree.rs:267-275) built inside the call == the same
    sumcheck begun on an uploaded eq table == the oracle, round by round, final claims included (the eq table's own claim among them)"""
    n = 1 << logn
    dev, host = _uniform_tables(ctx, 2 * n_par + 3 * n_seq, n, 8800 + logn)
    Ap, Bp = dev[:n_par], dev[n_par:2 * n_par]
    o = 2 * n_par
    As, Bs, Cs = dev[o:o + n_seq], dev[o + n_seq:o + 2 * n_seq], dev[o + 2 * n_seq:]
    rand = rand_scalars(logn, 8900 + logn); ch = rand_scalars(logn, 8901 + logn); co = rand_scalars(n_par + n_seq, 8902 + logn)
    eq_host = ol.eq_evals(rand)
    Cp = ctx.table_upload(eq_host)
    runs = []
    for mode in ("eq", "table"):
        st, ev = ctx.sumcheck_begin_eq(Ap, Bp, rand, As, Bs, Cs, co) if mode == "eq" else ctx.sumcheck_begin(Ap, Bp, Cp, As, Bs, Cs, co)
        got = [ev]
        for j in range(logn):
            got.append(st.round(ch[32 * j:32 * j + 32]))
        runs.append((got, st.finish()))
        st.free()
    assert runs[0] == runs[1]
    import numpy as np
    _, want_comb, want_fin = ol.sc_prove_cubic_batched(host[:n_par], host[n_par:2 * n_par], np.frombuffer(eq_host, dtype=np.uint8), host[o:o + n_seq], host[o + n_seq:o + 2 * n_seq], host[o + 2 * n_seq:], co, ch, 16)
    for j in range(logn):
        assert runs[0][0][j] == want_comb[j], f"round {j}"
    assert runs[0][1] == want_fin
    with pytest.raises(sbn.SbnError):
        ctx.sumcheck_begin_eq(Ap, Bp, rand + rand[:32], As, Bs, Cs, co)                     # rand.len() != num_vars (product_tree.rs:268)
    Cp.free()
    for t in dev:
        t.free()


def test_stateful_sumcheck_errors(ctx, sbn, pr):
    a, b, c2 = (ctx.table_upload(rand_scalars(8, s)) for s in (1, 2, 3))
    short = ctx.table_upload(rand_scalars(4, 4))
    with pytest.raises(sbn.SbnError):
        ctx.sumcheck_begin([a], [short], c2, [], [], [], rand_scalars(1, 5))                 # lengths differ
    with pytest.raises(sbn.SbnError):
        ctx.sumcheck_begin([a], [b], c2, [], [], [], pr.R.to_bytes(32, "little"))          # non-canonical coefficient
    # a NULL entry among the "seq" C tables with no "par" instance: SBN_EINVAL, not a crash
    import ctypes as C
    one = (C.c_void_p * 1)(a.h); nul = (C.c_void_p * 1)(None); sth = C.c_void_p(); o96 = (C.c_uint8 * 96)()
    co1 = (C.c_uint8 * 32).from_buffer_copy(rand_scalars(1, 5))
    assert sbn.lib().sbn_sumcheck_begin(ctx.h, None, None, None, C.c_size_t(0), one, one, nul, C.c_size_t(1), co1, o96, C.byref(sth)) == -1
    st, _ = ctx.sumcheck_begin([a], [b], c2, [], [], [], rand_scalars(1, 5))
    with pytest.raises(sbn.SbnError):
        st.finish()                                                                         # variables left
    with pytest.raises(sbn.SbnError):
        st.round(pr.R.to_bytes(32, "little"))                                               # non-canonical challenge
    for j in range(3):
        st.round(rand_scalars(1, 10 + j))
    with pytest.raises(sbn.SbnError):
        st.round(rand_scalars(1, 20))                                                       # no variable left
    assert len(st.finish()) == 3
    st.free()
    for t in (a, b, c2, short):
        t.free()


def test_stateful_sumcheck_keyless_layer0(ctx, ol):
    """the ops product-circuit layer-0 sumcheck at keyless size (12 par + 6 seq, tables of 2^21 uniform Fr): all 21 combined round
    values and the 43 final claims against the oracle"""
    _run_stateful_sumcheck(ctx, ol, 1 << 21, 12, 6, 9900)

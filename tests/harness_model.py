"""CPU model of spartan-bn254_amd/harness/prove_stages.cpp (the compiled caller of the C ABI), built on the oracle: it REPLAYS the
trace a harness run recorded — every value the stand-in transcript absorbed, every challenge it squeezed — recomputing each absorbed
value with oracle_lib from the same synthetic inputs and the recorded challenges, and checking the SHA3 chain itself.  A harness run
whose trace passes has produced, stage by stage, what the CPU restatement of the reference produces on the same inputs.

Test infrastructure (it imports the oracle): used by tests/test_gpu_harness.py and by bench.py's prove_stages parity gate."""
import hashlib
import os
import struct
import sys

import numpy as np

import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
M64 = (1 << 64) - 1


def _bench():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    return bench


def splitmix(x):
    x = (x + 0x9E3779B97F4A7C15) & M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M64
    return x ^ (x >> 31)


def splitmix_np(base, n):
    with np.errstate(over="ignore"):
        x = np.uint64(base & M64) + np.arange(n, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def reduce_to_fr(h):
    v = int.from_bytes(h, "little") & ((1 << 254) - 1)
    return (v - R if v >= R else v).to_bytes(32, "little")


class Replay:
    def __init__(self, trace):
        self.recs = []
        off = 0
        while off < len(trace):
            tag, n = struct.unpack_from("<II", trace, off)
            self.recs.append((tag, trace[off + 8:off + 8 + n])); off += 8 + n
        self.pos = 0
        self.h = bytes(32)
        self.checked = 0

    def challenge(self):
        tag, d = self.recs[self.pos]; self.pos += 1
        assert tag == 0xC0, f"record {self.pos - 1}: expected a challenge, found tag {tag:#x}"
        self.h = hashlib.sha3_256(self.h + b"\x63").digest()
        assert d == reduce_to_fr(self.h), f"record {self.pos - 1}: challenge does not follow from the absorbed values"
        return d

    def cvec(self, k):
        return b"".join(self.challenge() for _ in range(k))

    def absorb(self, tag, expected, what):
        t, d = self.recs[self.pos]; self.pos += 1
        assert t == tag, f"record {self.pos - 1}: expected tag {tag:#x} ({what}), found {t:#x}"
        assert d == expected, f"{what}: the harness absorbed a value the oracle does not reproduce (record {self.pos - 1}, {len(d)} bytes)"
        self.h = hashlib.sha3_256(self.h + d).digest()
        self.checked += 1

    def peek_absorb(self, tag):
        """the recorded value of the next absorb (for outputs whose later inputs depend on them: bullet L / R)"""
        t, d = self.recs[self.pos]
        assert t == tag
        return d


def synth(n, seed, stream):
    return _bench().splitmix_scalars(n, seed + stream, 0)


def inv(u):
    return pow(int.from_bytes(u, "little"), R - 2, R).to_bytes(32, "little")


def opening(rp, Z, log_l, log_r, G_xy, H_xy, what):
    """PolyEvalProof::prove (hyrax.rs:65-116) + the bullet reduction (nizk/bullet.rs:41-126)"""
    rl, rr = rp.cvec(log_l), rp.cvec(log_r)
    hL = ol.eq_evals(rl) if log_l else (1).to_bytes(32, "little")
    hR = ol.eq_evals(rr)
    n = 1 << log_r
    LZ = ol.bound(Z, hL, 1 << log_l, n)
    rp.absorb(0x40, ol.fr_dot(LZ, hR), what + ": <LZ, R>")
    Qb = (1).to_bytes(32, "little") + (2).to_bytes(32, "little")
    Q = ol.g1_mul(Qb, rp.challenge())                                  # Q = r * gens_1.G[0], r from the transcript (nizk/mod.rs:478-494)
    blind = rp.challenge()
    # the challenges u_i depend on L_i, R_i: walk the rounds with growing prefixes of the challenge list (each call re-derives the
    # earlier rounds; sizes here are small)
    gamma_pos = rp.pos
    # first pass: collect the per-round blinds and challenges from the trace (the chain check happens in the second pass)
    save = (rp.pos, rp.h, rp.checked)
    rp.pos += 1                                                        # Gamma
    bvec, us = b"", b""
    for rnd in range(log_r):
        for _ in range(2):
            assert rp.recs[rp.pos][0] == 0xC0; bvec += rp.recs[rp.pos][1]; rp.pos += 1
        rp.pos += 1                                                    # L, R
        assert rp.recs[rp.pos][0] == 0xC0; us += rp.recs[rp.pos][1]; rp.pos += 1
    rp.pos, rp.h, rp.checked = save
    assert rp.pos == gamma_pos
    want = ol.bullet_prove(G_xy, Q, H_xy, LZ, hR, blind, bvec, us)
    rp.absorb(0x41, want["Gamma"], what + ": Gamma")
    for rnd in range(log_r):
        assert rp.challenge() + rp.challenge() == bvec[64 * rnd:64 * rnd + 64]
        rp.absorb(0x42, want["L"][64 * rnd:64 * rnd + 64] + want["R"][64 * rnd:64 * rnd + 64], f"{what}: bullet round {rnd} L, R")
        assert rp.challenge() == us[32 * rnd:32 * rnd + 32]
    rp.absorb(0x43, want["a_hat"] + want["b_hat"] + want["g_hat"], what + ": a_hat, b_hat, g_hat")


def zk_sumcheck(rp, kind, tabs, what):
    n = len(tabs[0]) // 32
    while n >= 2:
        ev = ol.sc_eval_r1cs(*tabs) if kind == 0 else ol.sc_eval_quad(*tabs)
        rp.absorb(0x10 if kind == 0 else 0x11, ev, f"{what}: round sums at length {n}")
        r = rp.challenge()
        tabs = [ol.bind_top(t, r) for t in tabs]
        n //= 2
    rp.absorb(0x12, b"".join(t[:32] for t in tabs), what + ": final claims")


def layered(rp, circ, dotp, what):
    """ProductCircuitEvalProofBatched::prove (product_tree.rs:251-392), every layer's prove_cubic_batched by the oracle's loop"""
    nc, nl = len(circ), len(circ[0])
    rounds = 0
    for lay in range(nl - 1, -1, -1):
        half = len(circ[0][lay]) // 64
        A = [c[lay][:32 * half] for c in circ]; B = [c[lay][32 * half:] for c in circ]
        k = half.bit_length() - 1
        rand = rp.cvec(k)
        C = ol.eq_evals(rand) if k else (1).to_bytes(32, "little")
        As, Bs, Cs = [], [], []
        if lay == 0 and dotp:
            ns = len(dotp) // 3
            As, Bs, Cs = dotp[:ns], dotp[ns:2 * ns], dotp[2 * ns:]
        coeffs = rp.cvec(nc + len(As))
        if half < 2:
            continue
        # the challenges of this sumcheck, read ahead from the trace (checked against the chain as the rounds are replayed below)
        nr = half.bit_length() - 1
        ch, p = b"", rp.pos
        for _ in range(nr):
            assert rp.recs[p][0] == 0x20 and rp.recs[p + 1][0] == 0xC0
            ch += rp.recs[p + 1][1]; p += 2
        _, comb, fin = ol.sc_prove_cubic_batched(A, B, C, As, Bs, Cs, coeffs, ch, 8)
        for j in range(nr):
            rp.absorb(0x20, comb[j], f"{what}: layer {lay} round {j} combined sums")
            assert rp.challenge() == ch[32 * j:32 * j + 32]
            rounds += 1
        rp.absorb(0x21, b"".join(fin), f"{what}: layer {lay} final claims")
    return rounds


def replay(trace, log_ops, log_mem, log_cons, seed=1):
    """-> dict(records checked, sumcheck rounds); raises AssertionError on the first value the oracle does not reproduce"""
    rp = Replay(trace)
    LO, LM, LC = log_ops, log_mem, log_cons
    nops, nmem, ncons = 1 << LO, 1 << LM, 1 << LC
    wl = LC // 2; wr = LC - wl
    dl = (LO + 3) // 2; dr = LO + 3 - dl
    ol_ = (LO + 4) // 2; or_ = LO + 4 - ol_
    ml = (LM + 1) // 2; mr = LM + 1 - ml
    gw, _ = ol.gens_new(1 << wr, b"gens_r1cs_sat")
    gd, _ = ol.gens_new(1 << dr, b"gens_r1cs_eval")
    go, _ = ol.gens_new(1 << or_, b"gens_r1cs_eval")
    gm, _ = ol.gens_new(1 << mr, b"gens_r1cs_eval")
    real = int(nops * 0.57)
    addr, rts = [], []
    for k in range(6):
        x = splitmix_np(seed * 1000003 + (k << 40), nops)
        a = (x % np.uint64(nmem // 2)).astype(np.uint32); a[real:] = 0
        addr.append(a); rts.append(((x >> np.uint64(40)) % np.uint64(64)).astype(np.uint32))
    ats = [(splitmix_np(seed * 7919 + ((s + 8) << 40), nmem) % np.uint64(64)).astype(np.uint32) for s in range(2)]
    # ---- R1CS sat proof
    z = synth(ncons, seed, 1)
    blinds = rp.cvec(1 << wl)
    C = ol.commit_rows(z, blinds, 1 << wl, 1 << wr, gw[:64 << wr], gw[64 << wr:], 8)
    rp.absorb(0x01, b"".join(ol.g1_compress(C[64 * i:64 * i + 64]) for i in range(1 << wl)), "witness commitment")
    rtau = rp.cvec(LC)
    zk_sumcheck(rp, 0, [ol.eq_evals(rtau), synth(ncons, seed, 2), synth(ncons, seed, 3), synth(ncons, seed, 4)], "phase-1 sumcheck")
    zk_sumcheck(rp, 1, [synth(2 * ncons, seed, 5), synth(2 * ncons, seed, 6)], "phase-2 sumcheck")
    opening(rp, z, wl, wr, gw[:64 << wr], gw[64 << wr:], "witness opening")
    # ---- eq tables, derefs, derefs commitment
    rx, ry = rp.cvec(LM), rp.cvec(LM)
    mem = [ol.eq_evals(rx), ol.eq_evals(ry)]
    val = [b"".join(mem[k // 3][32 * int(i):32 * int(i) + 32] for i in addr[k]) for k in range(6)]
    comb = b"".join(val) + bytes(32 * 2 * nops)
    C = ol.commit_rows(comb, None, 1 << dl, 1 << dr, gd[:64 << dr], gd[64 << dr:], 8)
    rp.absorb(0x02, b"".join(ol.g1_compress(C[64 * i:64 * i + 64]) for i in range(1 << dl)), "derefs commitment")
    # ---- network construction
    g, tau = rp.challenge(), rp.challenge()
    ops_circ, mem_circ = [], []
    for side in range(2):
        mem_circ.append([ol.hash_layer(None, mem[side], None, 0, g, tau)])
        mem_circ.append([ol.hash_layer(None, mem[side], ats[side], 0, g, tau)])
        for i in range(3):
            k = 3 * side + i
            for ts_add in (0, 1):
                ops_circ.append([ol.hash_layer(addr[k], val[k], rts[k], ts_add, g, tau)])
    prods = b""
    for c in ops_circ + mem_circ:
        w = c[0]
        while len(w) > 32:
            w = ol.product_layer(w); c.append(w)
        prods += c.pop()
    rp.absorb(0x03, prods, "product-circuit values")
    # ---- network proof
    dotp = [synth(nops // 2, seed, 20 + j) for j in range(18)]
    r_ops_rounds = layered(rp, ops_circ, dotp, "ops product circuits")
    r_mem_rounds = layered(rp, mem_circ, None, "mem product circuits")
    big = [synth(nops, seed, 50 + j) for j in range(3)]
    r_ops, r_mem = rp.cvec(LO), rp.cvec(LM)
    chi = ol.eq_evals(r_ops)
    e3 = [ol.fr_dot(b, chi) for b in big]
    rp.absorb(0x30, b"".join(e3[i % 3] for i in range(21)), "21 evaluations at rand_ops")
    chim = ol.eq_evals(r_mem)
    rp.absorb(0x31, ol.fr_dot(mem[0], chim) + ol.fr_dot(mem[1], chim), "2 evaluations at rand_mem")
    opening(rp, comb, dl, dr, gd[:64 << dr], gd[64 << dr:], "derefs opening")
    opening(rp, synth(1 << (ol_ + or_), seed, 60), ol_, or_, go[:64 << or_], go[64 << or_:], "comb_ops opening")
    opening(rp, synth(1 << (ml + mr), seed, 61), ml, mr, gm[:64 << mr], gm[64 << mr:], "comb_mem opening")
    assert rp.pos == len(rp.recs), f"{len(rp.recs) - rp.pos} trace records were not consumed"
    return {"records_checked": rp.checked, "records": len(rp.recs), "sumcheck_rounds_ops": r_ops_rounds, "sumcheck_rounds_mem": r_mem_rounds, "digest": rp.h}

#!/usr/bin/env python3
"""Generates tests/golden/*.json from the pure-Python big-integer model (tests/pyref.py).

Nothing here touches oracle/ or the HIP library: the vectors are the mathematical definition evaluated with Python
ints, so the C oracle and the GPU path are both checked against something that shares none of their code.
The reference (Rust, arkworks 0.5 un-vendored) ships no golden vectors for this path; the entries
marked "reference test" restate the known answers of its own unit tests.

Run:  python tests/golden/make_golden.py      (about a minute; output is deterministic)
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import pyref as pr  # noqa: E402


def hx(b):
    return bytes(b).hex()


def sc(v):
    return hx(pr.scalar_to_bytes(v))


def pt(p):
    return hx(pr.point_to_xy(p))


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=1, sort_keys=True)
    print("wrote", name)


def field_kat():
    rnd = random.Random(20260401)
    out = {}
    for name, mod in (("fq", pr.P), ("fr", pr.R)):
        rows = []
        specials = [0, 1, 2, mod - 1, mod - 2, (mod + 1) // 2, (1 << 253) % mod, (1 << 128) - 1]
        vals = specials + [rnd.randrange(mod) for _ in range(24)]
        for i in range(0, len(vals), 2):
            a, b = vals[i], vals[i + 1]
            rows.append({"a": hx(a.to_bytes(32, "little")), "b": hx(b.to_bytes(32, "little")),
                         "add": hx(((a + b) % mod).to_bytes(32, "little")), "sub": hx(((a - b) % mod).to_bytes(32, "little")),
                         "mul": hx((a * b % mod).to_bytes(32, "little")),
                         "inv_a": hx((pow(a, -1, mod) if a else 0).to_bytes(32, "little"))})
        out[name] = rows
    out["fr_from_wide"] = []
    for _ in range(6):
        w = rnd.randbytes(64)
        out["fr_from_wide"].append({"in": hx(w), "out": sc(pr.challenge_from_wide(w))})
    dump("field_kat.json", out)


def g1_kat():
    rnd = random.Random(7)
    ks = [1, 2, 3, 5, 7, pr.R - 1, pr.R - 2, (pr.R + 1) // 2] + [rnd.randrange(pr.R) for _ in range(8)]
    muls = [{"k": sc(k), "kG": pt(pr.mul(pr.G, k)), "compressed": hx(pr.compress(pr.mul(pr.G, k)))} for k in ks]
    P1, P2 = pr.mul(pr.G, 1234567), pr.mul(pr.G, 7654321)
    adds = [
        {"p": pt(P1), "q": pt(P2), "sum": pt(pr.add(P1, P2)), "note": "generic"},
        {"p": pt(P1), "q": pt(P1), "sum": pt(pr.add(P1, P1)), "note": "P+P"},
        {"p": pt(P1), "q": pt(pr.neg(P1)), "sum": pt(pr.INF), "note": "P+(-P)"},
        {"p": pt(pr.INF), "q": pt(P2), "sum": pt(P2), "note": "inf+Q"},
        {"p": pt(pr.G), "q": pt(pr.G), "sum": pt(pr.mul(pr.G, 2)), "note": "reference test group.rs:305-311 (g*2 == g+g)"},
    ]
    enc = {"generator": hx(pr.compress(pr.G)), "neg_generator": hx(pr.compress(pr.neg(pr.G))), "infinity": hx(pr.compress(pr.INF))}
    dump("g1_kat.json", {"mul": muls, "add": adds, "compressed": enc})


def msm_kat():
    rnd = random.Random(99)
    cases = []

    def case(note, scalars, points):
        cases.append({"note": note, "scalars": [sc(k) for k in scalars], "points": [pt(p) for p in points], "expected": pt(pr.msm(scalars, points))})

    case("reference test group.rs:313-321: 2G+3G == 5G", [2, 3], [pr.G, pr.G])
    case("n=1", [rnd.randrange(pr.R)], [pr.mul(pr.G, 77)])
    case("n=3 with a zero scalar", [0, rnd.randrange(pr.R), 5], [pr.mul(pr.G, 9), pr.mul(pr.G, 10), pr.mul(pr.G, 11)])
    case("all-zero scalars -> identity", [0, 0, 0, 0], [pr.mul(pr.G, i + 1) for i in range(4)])
    case("(r-1)G + G -> identity", [pr.R - 1, 1], [pr.G, pr.G])
    case("P and -P with equal scalars -> identity", [12345, 12345], [pr.mul(pr.G, 42), pr.neg(pr.mul(pr.G, 42))])
    case("all bases equal to G (the reference's common case, SURVEY 0.6)", [rnd.randrange(pr.R) for _ in range(12)], [pr.G] * 12)
    case("a base at infinity", [5, 6], [pr.INF, pr.mul(pr.G, 3)])
    n = 33
    case("n=33 random", [rnd.randrange(pr.R) for _ in range(n)], [pr.mul(pr.G, rnd.randrange(pr.R)) for _ in range(n)])
    case("n=40 scalars near r and tiny", [pr.R - 1 - i for i in range(20)] + list(range(20)), [pr.mul(pr.G, 1000 + i) for i in range(40)])
    dump("msm_kat.json", {"cases": cases})


def gens_kat():
    out = {}
    for label, n in ((b"gens_r1cs_eval", 8193), (b"gens_r1cs_sat", 1025)):
        d = pr.gens_dlogs(n, label)
        out[label.decode()] = {
            "n": n,
            "first_dlogs": [sc(x) for x in d[:12]],
            "first_points": [pt(pr.mul(pr.G, x)) for x in d[:12]],
            "h_dlog": sc(d[n]),
            "count_equal_to_G_incl_h": sum(1 for x in d if x == 1),
            "dlog_sum_mod_r": sc(sum(d) % pr.R),
        }
    dump("gens_kat.json", out)


def commit_kat():
    """Expected commitments via the discrete-log identity sum k_i*(s_i G) = (sum k_i s_i) G."""
    out = {"cases": []}
    for label, R, L, seed, with_blinds in ((b"gens_r1cs_eval", 8192, 1, 0x5BA27A2B4E254, False), (b"gens_r1cs_eval", 8192, 3, 11, True),
                                          (b"gens_r1cs_sat", 1024, 4, 12, True), (b"gens_r1cs_eval", 64, 8, 13, False), (b"tiny", 5, 3, 14, True)):
        d = pr.gens_dlogs(R, label)
        Z = pr.prng_scalars(L * R, seed)
        if L >= 3:
            Z[R:2 * R] = [0] * R              # zero row (hyrax.rs:245 padding) -> identity when the blind is 0
            Z[2 * R:3 * R] = [Z[2 * R]] * R   # constant row (sparse_mlpoly_full.rs:89-101 repeats mem[0])
        blinds = pr.prng_scalars(L, seed + 1) if with_blinds else [0] * L
        exp = []
        for i in range(L):
            s = (sum(k * dl for k, dl in zip(Z[i * R:(i + 1) * R], d[:R])) + blinds[i] * d[R]) % pr.R
            exp.append(pt(pr.mul(pr.G, s)))
        out["cases"].append({"label": label.decode(), "R": R, "L": L, "seed": seed, "with_blinds": with_blinds,
                             "zero_row": 1 if L >= 3 else None, "const_row": 2 if L >= 3 else None, "expected_rows": exp})
    dump("commit_kat.json", out)


def sumcheck_kat():
    rnd = random.Random(4242)
    out = {"unipoly": [
        {"note": "reference test unipoly.rs:130-155: 2x^2+3x+1", "evals": [sc(1), sc(6), sc(15)], "coeffs": [sc(1), sc(3), sc(2)], "at": sc(3), "value": sc(28)},
        {"note": "reference test unipoly.rs:157-184: x^3+2x^2+3x+1", "evals": [sc(1), sc(7), sc(23), sc(55)], "coeffs": [sc(1), sc(3), sc(2), sc(1)], "at": sc(4), "value": sc(109)},
    ], "rounds": [], "eq": [], "bind": []}
    ev = [rnd.randrange(pr.R) for _ in range(4)]
    out["unipoly"].append({"note": "random cubic", "evals": [sc(x) for x in ev], "coeffs": [sc(x) for x in pr.unipoly_from_evals(ev)],
                           "at": sc(12345), "value": sc(pr.unipoly_eval(pr.unipoly_from_evals(ev), 12345))})
    for n in (2, 8, 32):
        T, A, B, C = ([rnd.randrange(pr.R) for _ in range(n)] for _ in range(4))
        out["rounds"].append({"len": n, "T": [sc(x) for x in T], "A": [sc(x) for x in A], "B": [sc(x) for x in B], "C": [sc(x) for x in C],
                              "cubic": [sc(x) for x in pr.sc_eval_cubic(A, B, C)], "r1cs": [sc(x) for x in pr.sc_eval_r1cs(T, A, B, C)],
                              "quad": [sc(x) for x in pr.sc_eval_quad(A, B)]})
    # reference test sumcheck.rs:818-861 tables: A=[1,2,3,4], B=C=[1,1,1,1], claim = 10
    A, B, C = [1, 2, 3, 4], [1, 1, 1, 1], [1, 1, 1, 1]
    e0, e2, e3 = pr.sc_eval_cubic(A, B, C)
    out["rounds"].append({"len": 4, "note": "reference test sumcheck.rs:818-861 round 0", "T": [sc(0)] * 4, "A": [sc(x) for x in A], "B": [sc(x) for x in B], "C": [sc(x) for x in C],
                          "cubic": [sc(e0), sc(e2), sc(e3)], "r1cs": [sc(x) for x in pr.sc_eval_r1cs([0] * 4, A, B, C)], "quad": [sc(x) for x in pr.sc_eval_quad(A, B)],
                          "claim": sc(10), "e0_plus_e1_is_claim": True})
    for ell in (1, 3, 5):
        r = [rnd.randrange(pr.R) for _ in range(ell)]
        out["eq"].append({"r": [sc(x) for x in r], "evals": [sc(x) for x in pr.eq_evals(r)]})
    out["eq"].append({"note": "reference test hyrax.rs:427-433 (len 4)", "r": [sc(2), sc(3)], "evals": [sc(x) for x in pr.eq_evals([2, 3])]})
    Z = [1, 2, 3, 4]
    out["bind"].append({"note": "reference test hyrax.rs:456-470: bind to 0 keeps [1,2]", "Z": [sc(x) for x in Z], "r": sc(0), "out": [sc(1), sc(2)]})
    Z = [rnd.randrange(pr.R) for _ in range(16)]
    r = rnd.randrange(pr.R)
    out["bind"].append({"Z": [sc(x) for x in Z], "r": sc(r), "out": [sc(x) for x in pr.bind_top(Z, r)]})
    dump("sumcheck_kat.json", out)


def bullet_kat():
    """BulletReductionProof::prove (nizk/bullet.rs:41-126) on the reference test's setup (n = 8, label "test-gens", Q = G,
    bullet.rs:222-238) and on n = 4 with repeated generators, challenges fixed."""
    rnd = random.Random(8128)
    out = {"cases": []}
    for n, label in ((8, b"test-gens"), (4, b"gens_r1cs_eval")):
        lg = n.bit_length() - 1
        d = pr.gens_dlogs(n, label)
        G = [pr.mul(pr.G, s) for s in d[:n]]; H = pr.mul(pr.G, d[n])
        Q = pr.G if n == 8 else pr.mul(pr.G, 12345)
        a = [rnd.randrange(pr.R) for _ in range(n)]; b = [rnd.randrange(pr.R) for _ in range(n)]
        blind = rnd.randrange(pr.R)
        bv = [(rnd.randrange(pr.R), rnd.randrange(pr.R)) for _ in range(lg)]
        us = [rnd.randrange(1, pr.R) for _ in range(lg)]
        o = pr.bullet_prove(G, Q, H, a, b, blind, bv, us)
        out["cases"].append({"n": n, "label": label.decode(), "G": "".join(pt(x) for x in G), "H": pt(H), "Q": pt(Q),
                             "a": "".join(sc(x) for x in a), "b": "".join(sc(x) for x in b), "blind": sc(blind),
                             "blinds_vec": "".join(sc(x) + sc(y) for x, y in bv), "us": "".join(sc(x) for x in us),
                             "L": "".join(pt(x) for x in o["L"]), "R": "".join(pt(x) for x in o["R"]), "Gamma": pt(o["Gamma"]),
                             "a_hat": sc(o["a_hat"]), "b_hat": sc(o["b_hat"]), "g_hat": pt(o["g_hat"]), "blind_hat": sc(o["blind_hat"])})
    dump("bullet_kat.json", out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "bullet":
        bullet_kat()
    else:
        field_kat(); g1_kat(); msm_kat(); gens_kat(); commit_kat(); sumcheck_kat(); bullet_kat()

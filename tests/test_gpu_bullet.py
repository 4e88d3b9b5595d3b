"""BulletReductionProof::prove (nizk/bullet.rs:41-126) round by round through the C ABI vs the oracle's restatement.
The Fiat-Shamir challenges are test inputs (the transcript stays in Rust)."""
import pytest

from conftest import rand_scalars

pytestmark = pytest.mark.gpu


def _inv(pr, ub):
    return pr.scalar_to_bytes(pow(pr.scalar_from_bytes(ub), pr.R - 2, pr.R))


def _run_device(ctx, pr, G_xy, H_xy, Q_xy, a, b, blind, blinds_vec, us, fused=False):
    """fused = False: sbn_bullet_cross + sbn_bullet_fold per round; True: one sbn_bullet_fold_cross per challenge (the prover's call sequence)"""
    n = len(a) // 32
    G = ctx.bases_upload(G_xy, H_xy)
    ta, tb = ctx.table_upload(a), ctx.table_upload(b)
    st, Gamma = ctx.bullet_begin(G, Q_xy, ta, tb, blind)
    ta.free(); tb.free()                                   # the state holds its own copies (bullet.rs:50-52)
    Ls, Rs, dots = [], [], []
    blind_hat = pr.scalar_from_bytes(blind)
    rnd = 0
    nxt = None
    while n > 1:
        bl, br = blinds_vec[64 * rnd:64 * rnd + 32], blinds_vec[64 * rnd + 32:64 * rnd + 64]
        L, _, Rp, _, cL, cR = nxt if nxt is not None else ctx.bullet_cross(st, bl, br)
        Ls.append(L); Rs.append(Rp); dots.append(cL + cR)
        u = us[32 * rnd:32 * rnd + 32]; ui = _inv(pr, u)
        if fused and n >= 4:
            nxt = ctx.bullet_fold_cross(st, u, ui, blinds_vec[64 * (rnd + 1):64 * (rnd + 1) + 32], blinds_vec[64 * (rnd + 1) + 32:64 * (rnd + 1) + 64])
        else:
            ctx.bullet_fold(st, u, ui); nxt = None
        uv, uiv = pr.scalar_from_bytes(u), pr.scalar_from_bytes(ui)
        blind_hat = (uv * uv * pr.scalar_from_bytes(bl) + blind_hat + uiv * uiv * pr.scalar_from_bytes(br)) % pr.R   # bullet.rs:108 (host side)
        n //= 2; rnd += 1
        assert len(st) == n
    a_hat, b_hat, g_hat = ctx.bullet_finish(st)
    out = dict(L=b"".join(Ls), R=b"".join(Rs), Gamma=Gamma, a_hat=a_hat, b_hat=b_hat, g_hat=g_hat, blind_hat=pr.scalar_to_bytes(blind_hat), dots=b"".join(dots))
    st.free(); G.free()
    return out


@pytest.mark.parametrize("n,label", [(2, b"test-gens"), (8, b"test-gens"), (64, b"gens_r1cs_eval")])
def test_bullet_rounds_vs_oracle(ctx, ol, pr, n, label):
    # n = 8 with label "test-gens" and Q = G is the reference's own test setup (bullet.rs:222-238)
    pts, _ = ol.gens_new(n, label)
    G_xy, H_xy = pts[:64 * n], pts[64 * n:]
    Q_xy = pr.point_to_xy((1, 2))
    lg = n.bit_length() - 1
    a, b = rand_scalars(n, 101 + n), rand_scalars(n, 202 + n)
    blind = rand_scalars(1, 303)
    blinds_vec, us = rand_scalars(2 * lg, 404 + n), rand_scalars(lg, 505 + n)
    want = ol.bullet_prove(G_xy, Q_xy, H_xy, a, b, blind, blinds_vec, us)
    got = _run_device(ctx, pr, G_xy, H_xy, Q_xy, a, b, blind, blinds_vec, us)
    for k in ("L", "R", "Gamma", "a_hat", "b_hat", "g_hat", "blind_hat"):
        assert got[k] == want[k], k
    fused = _run_device(ctx, pr, G_xy, H_xy, Q_xy, a, b, blind, blinds_vec, us, fused=True)     # one call per challenge: the same values, cross terms included
    assert fused == got


def test_bullet_edge_values(ctx, ol, pr):
    """zero / one / r-1 entries, equal generators (the reference's derivation repeats G), zero blinds, u = 1"""
    n = 16
    pts, _ = ol.gens_new(n, b"gens_r1cs_eval")      # ~2/3 of these equal the curve generator
    G_xy, H_xy = pts[:64 * n], pts[64 * n:]
    Q_xy = pr.point_to_xy(pr.mul((1, 2), 7))
    vals = [0, 1, pr.R - 1, 2, 0, 0, pr.R - 2, 5] * 2
    a = b"".join(pr.scalar_to_bytes(v) for v in vals)
    b = b"".join(pr.scalar_to_bytes(v) for v in reversed(vals))
    blind = pr.scalar_to_bytes(0)
    blinds_vec = b"".join(pr.scalar_to_bytes(v) for v in [0, 0, 1, pr.R - 1, 0, 3, 9, 0])
    us = b"".join(pr.scalar_to_bytes(v) for v in [1, pr.R - 1, 2, 12345])
    want = ol.bullet_prove(G_xy, Q_xy, H_xy, a, b, blind, blinds_vec, us)
    got = _run_device(ctx, pr, G_xy, H_xy, Q_xy, a, b, blind, blinds_vec, us)
    for k in ("L", "R", "Gamma", "a_hat", "b_hat", "g_hat", "blind_hat"):
        assert got[k] == want[k], k
    assert _run_device(ctx, pr, G_xy, H_xy, Q_xy, a, b, blind, blinds_vec, us, fused=True) == got


def test_bullet_optional_terms(ctx, ol, pr):
    """Q / H left out: L and R are the bare MSMs (what vartime_multiscalar_mul(a_L, G_R) returns, bullet.rs:77); second round too"""
    n = 32
    pts, _ = ol.gens_new(n, b"opt")
    G_xy = pts[:64 * n]
    a, b = rand_scalars(n, 1), rand_scalars(n, 2)
    G = ctx.bases_upload(G_xy, None)
    ta, tb = ctx.table_upload(a), ctx.table_upload(b)
    st, Gamma = ctx.bullet_begin(G, None, ta, tb)
    assert Gamma == ol.msm_naive(a, G_xy)
    L, _, Rp, _, cL, cR = ctx.bullet_cross(st)
    h = n // 2
    assert L == ol.msm_naive(a[:32 * h], G_xy[64 * h:])
    assert Rp == ol.msm_naive(a[32 * h:], G_xy[:64 * h])
    assert cL == ol.fr_dot(a[:32 * h], b[32 * h:]) and cR == ol.fr_dot(a[32 * h:], b[:32 * h])
    # the caller's tables are untouched
    assert ctx.table_download(ta) == a and ctx.table_download(tb) == b
    st.free(); G.free(); ta.free(); tb.free()


@pytest.mark.parametrize("n", [4, 64, 1024])
def test_bullet_begin_scaled_q(ctx, ol, pr, n):
    """sbn_bullet_begin_scaled: Q = r * Q_base as DotProductProofLog::prove passes it (nizk/mod.rs:478-494: gens_1.scale(r).G[0], r fresh per proof):
    Gamma and every round's L, R, c_L, c_R equal the run that is handed the scaled point itself; the derived set of Q_base is built once"""
    pts, _ = ol.gens_new(n, b"gens_r1cs_eval")
    G_xy, H_xy = pts[:64 * n], pts[64 * n:]
    Q_base = pr.point_to_xy(pr.mul((1, 2), 424242))
    lg = n.bit_length() - 1
    a, b = rand_scalars(n, 9100 + n), rand_scalars(n, 9200 + n)
    blind = rand_scalars(1, 9300); blinds_vec = rand_scalars(2 * lg, 9400 + n); us = rand_scalars(lg, 9500 + n)
    G = ctx.bases_upload(G_xy, H_xy)
    for k, r in enumerate((rand_scalars(1, 9600 + n), rand_scalars(1, 9601 + n), pr.scalar_to_bytes(1))):       # several proofs over one generator set
        Q = ol.g1_mul(Q_base, r)
        want = ol.bullet_prove(G_xy, Q, H_xy, a, b, blind, blinds_vec, us)
        ta, tb = ctx.table_upload(a), ctx.table_upload(b)
        st, Gamma = ctx.bullet_begin_scaled(G, Q_base, r, ta, tb, blind)
        ta.free(); tb.free()
        assert Gamma == want["Gamma"]
        m = n; rnd = 0; nxt = None
        while m > 1:
            bl, br = blinds_vec[64 * rnd:64 * rnd + 32], blinds_vec[64 * rnd + 32:64 * rnd + 64]
            L, _, Rp, _, cL, cR = nxt if nxt is not None else ctx.bullet_cross(st, bl, br)
            assert L == want["L"][64 * rnd:64 * rnd + 64] and Rp == want["R"][64 * rnd:64 * rnd + 64], (k, rnd)
            u = us[32 * rnd:32 * rnd + 32]; ui = _inv(pr, u)
            if m >= 4:
                nxt = ctx.bullet_fold_cross(st, u, ui, blinds_vec[64 * (rnd + 1):64 * (rnd + 1) + 32], blinds_vec[64 * (rnd + 1) + 32:64 * (rnd + 1) + 64])
            else:
                ctx.bullet_fold(st, u, ui); nxt = None
            m //= 2; rnd += 1
        a_hat, b_hat, g_hat = ctx.bullet_finish(st)
        assert (a_hat, b_hat, g_hat) == (want["a_hat"], want["b_hat"], want["g_hat"])
        st.free()
    G.free()


def test_bullet_errors(ctx, sbn, ol, pr):
    pts, _ = ol.gens_new(8, b"e")
    G = ctx.bases_upload(pts[:64 * 8], pts[64 * 8:])
    Gnoh = ctx.bases_upload(pts[:64 * 8], None)
    ta, tb = ctx.table_upload(rand_scalars(8, 1)), ctx.table_upload(rand_scalars(4, 2))
    with pytest.raises(sbn.SbnError):
        ctx.bullet_begin(G, None, ta, tb)                 # bullet.rs:43 assert_eq
    with pytest.raises(sbn.SbnError):
        ctx.table_upload(rand_scalars(6, 3))              # tables are powers of two, so bullet.rs:44 cannot be violated
    tc = ctx.table_upload(rand_scalars(8, 5))
    with pytest.raises(sbn.SbnError):
        ctx.bullet_begin(Gnoh, None, ta, tc, rand_scalars(1, 6))     # blind without h
    st, _ = ctx.bullet_begin(G, None, ta, tc)
    bad = b"\xff" * 32
    with pytest.raises(sbn.SbnError):
        ctx.bullet_fold(st, bad, bad)
    with pytest.raises(sbn.SbnError):
        ctx.bullet_cross(st, bad, None)
    with pytest.raises(sbn.SbnError):
        ctx.bullet_fold_cross(st, bad, bad)                # non-canonical challenge
    with pytest.raises(sbn.SbnError):
        ctx.bullet_finish(st)                             # bullet.rs:114-116: length must be 1
    one = pr.scalar_to_bytes(1)
    ctx.bullet_fold_cross(st, one, one); ctx.bullet_fold(st, one, one)          # 8 -> 4 -> 2
    with pytest.raises(sbn.SbnError):
        ctx.bullet_fold_cross(st, one, one)                # length 2: the last fold leaves no cross terms
    ctx.bullet_fold(st, one, one)
    with pytest.raises(sbn.SbnError):
        ctx.bullet_cross(st)                              # nothing left to fold
    ctx.bullet_finish(st)
    st.free()
    for t in (ta, tb, tc):
        t.free()
    G.free(); Gnoh.free()


def test_bullet_full_size_verifier_relation(ctx, ol, pr):
    """n = 8192 (the right-hand vector of the derefs opening): check what the verifier checks (bullet.rs:155-170 and
    nizk/mod.rs DotProductProofLog::verify): g_hat = MSM(s, G), b_hat = <s, b>,
    sum u_i^2 L_i + Gamma + sum u_i^-2 R_i = a_hat*g_hat + a_hat*b_hat*Q + rhat*H."""
    n = 8192; lg = 13
    pts, _ = ol.gens_new(n, b"gens_r1cs_eval")
    G_xy, H_xy = pts[:64 * n], pts[64 * n:]
    Q_xy = pr.point_to_xy(pr.mul((1, 2), 0xABCDEF))
    a, b = rand_scalars(n, 11), rand_scalars(n, 12)
    blind = rand_scalars(1, 13); blinds_vec = rand_scalars(2 * lg, 14); us = rand_scalars(lg, 15)
    got = _run_device(ctx, pr, G_xy, H_xy, Q_xy, a, b, blind, blinds_vec, us, fused=True)       # the prover's call sequence: one call per challenge
    assert got == _run_device(ctx, pr, G_xy, H_xy, Q_xy, a, b, blind, blinds_vec, us)           # ... and the separate cross / fold calls
    h0 = n // 2
    assert got["dots"][:64] == ol.fr_dot(a[:32 * h0], b[32 * h0:]) + ol.fr_dot(a[32 * h0:], b[:32 * h0])
    R = pr.R
    u = [pr.scalar_from_bytes(us[32 * i:32 * i + 32]) for i in range(lg)]
    ui = [pow(x, R - 2, R) for x in u]
    s = []
    for i in range(n):                                   # compute_s (bullet.rs:181-199)
        v = 1
        for j in range(lg):
            v = v * (u[lg - 1 - j] if (i >> j) & 1 else ui[lg - 1 - j]) % R
        s.append(v)
    sb = b"".join(pr.scalar_to_bytes(v) for v in s)
    assert got["g_hat"] == ol.msm_pippenger(sb, G_xy, threads=8)
    assert got["b_hat"] == ol.fr_dot(sb, b)
    ab = ol.fr_dot(a, b)
    Gamma = ol.g1_add(ol.g1_add(ol.msm_pippenger(a, G_xy, threads=8), ol.g1_mul(Q_xy, ab)), ol.g1_mul(H_xy, blind))
    assert got["Gamma"] == Gamma
    lhs = Gamma
    for i in range(lg):
        lhs = ol.g1_add(lhs, ol.g1_mul(got["L"][64 * i:64 * i + 64], pr.scalar_to_bytes(u[i] * u[i] % R)))
        lhs = ol.g1_add(lhs, ol.g1_mul(got["R"][64 * i:64 * i + 64], pr.scalar_to_bytes(ui[i] * ui[i] % R)))
    ah, bh = pr.scalar_from_bytes(got["a_hat"]), pr.scalar_from_bytes(got["b_hat"])
    rhs = ol.g1_add(ol.g1_add(ol.g1_mul(got["g_hat"], got["a_hat"]), ol.g1_mul(Q_xy, pr.scalar_to_bytes(ah * bh % R))), ol.g1_mul(H_xy, got["blind_hat"]))
    assert lhs == rhs


def test_bullet_fuzz_sizes(ctx, ol, pr):
    """every power of two from 2 to 256, random data, with and without Q / H / blinds"""
    for lg in range(1, 9):
        n = 1 << lg
        pts, _ = ol.gens_new(n, b"fz%d" % lg)
        G_xy, H_xy = pts[:64 * n], pts[64 * n:]
        Q_xy = pr.point_to_xy(pr.mul((1, 2), 1000 + lg))
        a, b = rand_scalars(n, 3000 + lg), rand_scalars(n, 3100 + lg)
        blind = rand_scalars(1, 3200 + lg); blinds_vec = rand_scalars(2 * lg, 3300 + lg); us = rand_scalars(lg, 3400 + lg)
        want = ol.bullet_prove(G_xy, Q_xy, H_xy, a, b, blind, blinds_vec, us)
        got = _run_device(ctx, pr, G_xy, H_xy, Q_xy, a, b, blind, blinds_vec, us, fused=bool(lg & 1))
        for k in ("L", "R", "Gamma", "a_hat", "b_hat", "g_hat", "blind_hat"):
            assert got[k] == want[k], (n, k)
        got2 = _run_device(ctx, pr, G_xy, H_xy, Q_xy, a, b, blind, blinds_vec, us, fused=not (lg & 1))
        assert got2 == got, n


def test_bullet_golden(ctx, pr):
    """the committed fixture (pure-Python model, tests/golden/make_golden.py) through the C ABI"""
    from conftest import golden
    H = bytes.fromhex
    for case in golden("bullet_kat.json")["cases"]:
        got = _run_device(ctx, pr, H(case["G"]), H(case["H"]), H(case["Q"]), H(case["a"]), H(case["b"]), H(case["blind"]), H(case["blinds_vec"]), H(case["us"]))
        for k in ("L", "R", "Gamma", "a_hat", "b_hat", "g_hat", "blind_hat"):
            assert got[k] == H(case[k]), (case["n"], k)

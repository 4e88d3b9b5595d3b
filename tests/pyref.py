"""Independent big-integer model of the hot path's mathematics (pure Python ints).

Test infrastructure only.  This is deliberately NOT a translation of oracle/bn254_oracle.c:
it works on plain integers with affine formulas and `pow(x, -1, p)`, so that the C oracle
(Montgomery limbs, Jacobian formulas) and the HIP kernels (8x32 limbs, XYZZ buckets) are each
checked against an implementation that shares none of their machinery.  It is the generator of
tests/golden/*.json (tests/golden/make_golden.py).

Reference call sites restated: group.rs:110-131,135-140,143-158,171-175; commitments.rs:31-62,
118-154; hyrax.rs:195-203,253-308,355-373; sumcheck.rs:105-161,494-530,684-699; unipoly.rs:28-82;
scalar.rs:75-95; transcript.rs:56-67.  Constants: SURVEY.md App. A.
"""
import hashlib

P = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
B = 3
G = (1, 2)
INF = None


# ---------------------------------------------------------------- curve (affine, None = infinity)
def on_curve(pt):
    if pt is INF:
        return True
    x, y = pt
    return (y * y - x * x * x - B) % P == 0


def neg(pt):
    if pt is INF:
        return INF
    return (pt[0], (-pt[1]) % P)


def add(p1, p2):
    if p1 is INF:
        return p2
    if p2 is INF:
        return p1
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return INF
        lam = 3 * x1 * x1 * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    return (x3, (lam * (x1 - x3) - y1) % P)


def mul(pt, k):
    k %= R
    acc = INF
    while k:
        if k & 1:
            acc = add(acc, pt)
        pt = add(pt, pt)
        k >>= 1
    return acc


def msm(scalars, points):
    """group.rs:171-175 msm_affine: sum k_i * P_i; length mismatch -> identity (unwrap_or_default)."""
    if len(scalars) != len(points):
        return INF
    acc = INF
    for k, pt in zip(scalars, points):
        acc = add(acc, mul(pt, k))
    return acc


# ---------------------------------------------------------------- wire formats
def scalar_to_bytes(k):
    return int(k % R).to_bytes(32, "little")


def scalar_from_bytes(b):
    """scalar.rs:87-95: None when the integer is >= r."""
    v = int.from_bytes(b, "little")
    return v if v < R else None


def point_to_xy(pt):
    if pt is INF:
        return bytes(64)
    return pt[0].to_bytes(32, "little") + pt[1].to_bytes(32, "little")


def point_from_xy(b):
    if b == bytes(64):
        return INF
    return (int.from_bytes(b[:32], "little"), int.from_bytes(b[32:], "little"))


def compress(pt):
    """arkworks serialize_compressed for G1Affine (group.rs:135-140): x LE; bit7 = y > p-y; bit6 = infinity."""
    if pt is INF:
        return bytes(31) + b"\x40"
    x, y = pt
    b = bytearray(x.to_bytes(32, "little"))
    if y > P - y:
        b[31] |= 0x80
    return bytes(b)


def challenge_from_wide(b64):
    """transcript.rs:56-67: 64 bytes LE reduced mod r."""
    return int.from_bytes(b64, "little") % R


# ---------------------------------------------------------------- generators / commit
def from_uniform_bytes_dlog(ub):
    """group.rs:110-131 — returns the scalar s with point = s*G."""
    s = scalar_from_bytes(hashlib.sha3_256(ub).digest())
    if s is None:
        s = scalar_from_bytes(hashlib.sha3_256(b"fallback" + ub).digest())
        if s is None:
            s = 1
    return s


def gens_dlogs(n, label):
    """commitments.rs:31-62 — the n+1 discrete logs (G[0..n) then h)."""
    stream = hashlib.shake_256(label + compress(G)).digest(64 * (n + 1))
    return [from_uniform_bytes_dlog(stream[64 * i:64 * (i + 1)]) for i in range(n + 1)]


def commit(scalars, blind, gens_pts, h_pt):
    """commitments.rs:144-154."""
    assert len(scalars) == len(gens_pts)
    return msm(list(scalars) + [blind], list(gens_pts) + [h_pt])


def factored_lens(ell):
    return ell // 2, ell - ell // 2


# ---------------------------------------------------------------- multilinear / sumcheck over Fr
def eq_evals(r):
    """hyrax.rs:355-369 (variable 0 = most significant index bit)."""
    ev = [1]
    for rj in r:
        nxt = []
        for s in ev:
            hi = s * rj % R
            nxt += [(s - hi) % R, hi]
        ev = nxt
    return ev


def bind_top(Z, r):
    n = len(Z) // 2
    return [(Z[i] + r * (Z[i + n] - Z[i])) % R for i in range(n)]


def _pts(lo, hi):
    p2 = (2 * hi - lo) % R
    return p2, (p2 + hi - lo) % R


def sc_eval_cubic(A, Bv, C):
    h = len(A) // 2
    e0 = e2 = e3 = 0
    for i in range(h):
        a2, a3 = _pts(A[i], A[i + h]); b2, b3 = _pts(Bv[i], Bv[i + h]); c2, c3 = _pts(C[i], C[i + h])
        e0 += A[i] * Bv[i] * C[i]; e2 += a2 * b2 * c2; e3 += a3 * b3 * c3
    return e0 % R, e2 % R, e3 % R


def sc_eval_r1cs(T, A, Bv, C):
    h = len(A) // 2
    f = lambda t, a, b, c: t * (a * b - c)
    e0 = e2 = e3 = 0
    for i in range(h):
        t2, t3 = _pts(T[i], T[i + h]); a2, a3 = _pts(A[i], A[i + h]); b2, b3 = _pts(Bv[i], Bv[i + h]); c2, c3 = _pts(C[i], C[i + h])
        e0 += f(T[i], A[i], Bv[i], C[i]); e2 += f(t2, a2, b2, c2); e3 += f(t3, a3, b3, c3)
    return e0 % R, e2 % R, e3 % R


def sc_eval_quad(Z, ABC):
    h = len(Z) // 2
    e0 = e2 = 0
    for i in range(h):
        e0 += Z[i] * ABC[i]
        e2 += (2 * Z[i + h] - Z[i]) * (2 * ABC[i + h] - ABC[i])
    return e0 % R, e2 % R


def unipoly_from_evals(ev):
    """unipoly.rs:28-59 -> coefficients low..high."""
    i2, i6 = pow(2, -1, R), pow(6, -1, R)
    if len(ev) == 3:
        c = ev[0]; a = i2 * (ev[2] - 2 * ev[1] + c) % R; b = (ev[1] - c - a) % R
        return [c, b, a]
    d = ev[0]
    a = i6 * (ev[3] - 3 * ev[2] + 3 * ev[1] - ev[0]) % R
    b = i2 * (2 * ev[0] - 5 * ev[1] + 4 * ev[2] - ev[3]) % R
    c = (ev[1] - d - a - b) % R
    return [d, c, b, a]


def unipoly_eval(co, r):
    return sum(c * pow(r, i, R) for i, c in enumerate(co)) % R


# ---------------------------------------------------------------- deterministic inputs
def splitmix64(state):
    state = (state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return state, z ^ (z >> 31)


def prng_scalars(n, seed=0x5BA27A2B4E254):
    out = []
    st = seed
    for _ in range(n):
        v = 0
        for j in range(4):
            st, w = splitmix64(st)
            v |= w << (64 * j)
        out.append(v % R)
    return out


def bullet_prove(G, Q, H, a, b, blind, blinds_vec, us):
    """nizk/bullet.rs:41-126 with the Fiat-Shamir challenges given (ints / point tuples in, dict out): the definition, with
    per-round generator folds exactly as the reference writes them."""
    G = list(G); a = list(a); b = list(b)
    n = len(G)
    Gamma = add(add(msm(a, G), mul(Q, sum(x * y for x, y in zip(a, b)) % R)), mul(H, blind))
    Ls, Rs = [], []
    bg = blind
    for (bl, br), u in zip(blinds_vec, us):
        n //= 2
        ui = pow(u, R - 2, R)
        cL = sum(x * y for x, y in zip(a[:n], b[n:])) % R
        cR = sum(x * y for x, y in zip(a[n:], b[:n])) % R
        Ls.append(add(add(msm(a[:n], G[n:]), mul(Q, cL)), mul(H, bl)))
        Rs.append(add(add(msm(a[n:], G[:n]), mul(Q, cR)), mul(H, br)))
        G = [add(mul(G[i], ui), mul(G[i + n], u)) for i in range(n)]
        a = [(u * a[i] + ui * a[i + n]) % R for i in range(n)]
        b = [(ui * b[i] + u * b[i + n]) % R for i in range(n)]
        bg = (u * u * bl + bg + ui * ui * br) % R
    return {"L": Ls, "R": Rs, "Gamma": Gamma, "a_hat": a[0], "b_hat": b[0], "g_hat": G[0], "blind_hat": bg}

"""ctypes binding of oracle/libsbn_oracle.so — the checker; tests / smoke / bench cpu_baseline only."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(ORACLE_DIR, "libsbn_oracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.orc_msm_window_bits.restype = C.c_int
        _LIB.orc_msm_window_bits.argtypes = [C.c_size_t]
    return _LIB


def _buf(n):
    return (C.c_uint8 * n)()


def _b(x):
    return bytes(x)


def fe_op(op, which, a, b=None):
    out = _buf(32)
    f = getattr(lib(), "orc_fe_" + op)
    if b is None:
        f(which, a, out)
    else:
        f(which, a, b, out)
    return _b(out)


def fe_is_canonical(which, a):
    return bool(lib().orc_fe_is_canonical(which, a))


def fr_from_wide(b64):
    out = _buf(32); lib().orc_fr_from_wide(b64, out); return _b(out)


def fr_dot(k, s):
    n = len(k) // 32; out = _buf(32); lib().orc_fr_dot(k, s, C.c_size_t(n), out); return _b(out)


def g1_on_curve(p):
    return bool(lib().orc_g1_on_curve(p))


def g1_add(p, q):
    out = _buf(64); lib().orc_g1_add(p, q, out); return _b(out)


def g1_neg(p):
    out = _buf(64); lib().orc_g1_neg(p, out); return _b(out)


def g1_mul(p, k):
    out = _buf(64); lib().orc_g1_mul(p, k, out); return _b(out)


def g1_compress(p):
    out = _buf(32); lib().orc_g1_compress(p, out); return _b(out)


def g1_decompress(c):
    out = _buf(64); ok = lib().orc_g1_decompress(c, out); return _b(out) if ok else None


def g1_mul_gen_batch(s, threads=8):
    n = len(s) // 32; out = _buf(64 * n); lib().orc_g1_mul_gen_batch(s, C.c_size_t(n), out, threads); return _b(out)


def msm_naive(scalars, bases):
    n = len(scalars) // 32; out = _buf(64); lib().orc_msm_naive(scalars, bases, C.c_size_t(n), out); return _b(out)


def msm_pippenger(scalars, bases, threads=1):
    n = len(scalars) // 32; out = _buf(64); lib().orc_msm_pippenger(scalars, bases, C.c_size_t(n), out, threads); return _b(out)


def window_bits(n):
    return lib().orc_msm_window_bits(n)


def gens_new(n, label, threads=8):
    xy = _buf(64 * (n + 1)); dl = _buf(32 * (n + 1))
    lib().orc_gens_new(C.c_size_t(n), label, C.c_size_t(len(label)), xy, dl, threads)
    return _b(xy), _b(dl)


def commit(scalars, blind, G_xy, h_xy):
    n = len(scalars) // 32; out = _buf(64); lib().orc_commit(scalars, C.c_size_t(n), blind, G_xy, h_xy, out); return _b(out)


def commit_rows(Z, blinds, L, R, G_xy, h_xy, threads=8):
    out = _buf(64 * L)
    lib().orc_commit_rows(Z, blinds, C.c_size_t(L), C.c_size_t(R), G_xy, h_xy, out, threads)
    return _b(out)


def eq_evals(r):
    ell = len(r) // 32; out = _buf(32 << ell); lib().orc_eq_evals(r, C.c_size_t(ell), out); return _b(out)


def hash_layer(addr, val, ts, ts_add, g, tau):
    """addr / ts: numpy uint32 arrays or None"""
    n = len(val) // 32; out = _buf(32 * n)
    ap = addr.ctypes.data_as(C.c_void_p) if addr is not None else None
    tp = ts.ctypes.data_as(C.c_void_p) if ts is not None else None
    lib().orc_hash_layer(ap, val, tp, C.c_uint32(ts_add), g, tau, C.c_size_t(n), out); return _b(out)


def product_layer(v):
    n = len(v) // 32; out = _buf(16 * n); lib().orc_product_layer(v, C.c_size_t(n), out); return _b(out)


def bound(Z, L, L_size, R_size):
    out = _buf(32 * R_size); lib().orc_bound(Z, L, C.c_size_t(L_size), C.c_size_t(R_size), out); return _b(out)


def bind_top(Z, r):
    n = len(Z) // 32; buf = (C.c_uint8 * len(Z)).from_buffer_copy(Z); lib().orc_bind_top(buf, C.c_size_t(n), r); return _b(buf)[: 32 * (n // 2)]


def sc_eval_cubic(A, B, Cc):
    out = _buf(96); lib().orc_sc_eval_cubic(A, B, Cc, C.c_size_t(len(A) // 32), out); return _b(out)


def sc_eval_r1cs(T, A, B, Cc):
    out = _buf(96); lib().orc_sc_eval_r1cs(T, A, B, Cc, C.c_size_t(len(A) // 32), out); return _b(out)


def sc_eval_quad(Z, ABC):
    out = _buf(64); lib().orc_sc_eval_quad(Z, ABC, C.c_size_t(len(Z) // 32), out); return _b(out)


def sc_prove_cubic_batched(A_par, B_par, C_par, A_seq, B_seq, C_seq, coeffs, challenges, threads=8):
    """prove_cubic_batched (sumcheck.rs:165-330) with caller-supplied challenges.  Tables: bytes-like (numpy uint8 arrays are taken
    without a copy).  -> (evals[round][inst] = 96 B, combined[round] = 96 B, finals = list of 32 B in the order A_par.., B_par..,
    C_par, A_seq.., B_seq.., C_seq..)"""
    import numpy as np
    keep = []

    def ptr(x):
        a = np.frombuffer(x, dtype=np.uint8) if not isinstance(x, np.ndarray) else x
        keep.append(a)
        return a.ctypes.data_as(C.c_void_p)

    def arr(lst):
        return (C.c_void_p * max(1, len(lst)))(*[ptr(x) for x in lst])
    n_par, n_seq = len(A_par), len(A_seq)
    first = A_par[0] if n_par else A_seq[0]
    ln = len(first) // 32 if not hasattr(first, "nbytes") else first.nbytes // 32
    rounds = len(challenges) // 32
    ninst = n_par + n_seq
    ntab = 2 * n_par + (1 if n_par else 0) + 3 * n_seq
    ev, cb, fin = _buf(96 * rounds * ninst), _buf(96 * rounds), _buf(32 * ntab)
    lib().orc_sc_prove_cubic_batched(arr(A_par), arr(B_par), ptr(C_par) if n_par else None, C.c_size_t(n_par), arr(A_seq), arr(B_seq), arr(C_seq), C.c_size_t(n_seq),
                                     C.c_size_t(ln), C.c_size_t(rounds), bytes(coeffs), bytes(challenges), ev, cb, fin, threads)
    ev, cb, fin = _b(ev), _b(cb), _b(fin)
    evals = [[ev[96 * (j * ninst + i):96 * (j * ninst + i + 1)] for i in range(ninst)] for j in range(rounds)]
    return evals, [cb[96 * j:96 * j + 96] for j in range(rounds)], [fin[32 * t:32 * t + 32] for t in range(ntab)]


def unipoly_from_evals(ev):
    n = len(ev) // 32; out = _buf(32 * n); lib().orc_unipoly_from_evals(ev, C.c_size_t(n), out); return _b(out)


def unipoly_eval(co, r):
    out = _buf(32); lib().orc_unipoly_eval(co, C.c_size_t(len(co) // 32), r, out); return _b(out)


def bullet_prove(G_xy, Q_xy, H_xy, a, b, blind, blinds_vec, us):
    """nizk/bullet.rs:41-126 with caller-supplied challenges -> dict"""
    n = len(a) // 32
    lg = n.bit_length() - 1
    Lv, Rv = _buf(64 * lg), _buf(64 * lg)
    Gm, ah, bh, gh, blh = _buf(64), _buf(32), _buf(32), _buf(64), _buf(32)
    lib().orc_bullet_prove(_b(G_xy), _b(Q_xy), _b(H_xy), _b(a), _b(b), C.c_size_t(n), _b(blind), _b(blinds_vec), _b(us), Lv, Rv, Gm, ah, bh, gh, blh)
    return dict(L=bytes(Lv), R=bytes(Rv), Gamma=bytes(Gm), a_hat=bytes(ah), b_hat=bytes(bh), g_hat=bytes(gh), blind_hat=bytes(blh))


def sha3_256(m):
    out = _buf(32); lib().orc_sha3_256(m, C.c_size_t(len(m)), out); return _b(out)


def shake256(m, n):
    out = _buf(n); lib().orc_shake256(m, C.c_size_t(len(m)), out, C.c_size_t(n)); return _b(out)

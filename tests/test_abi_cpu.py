"""CPU: the C-ABI library loads, exports every symbol include/sbn254.h declares, its host-only helpers agree with the
oracle, and the device entry points fail loudly (no CPU fallback) when no GPU is present."""
import os
import re

import pytest
from conftest import ROOT, golden

H = bytes.fromhex


def header_functions():
    src = open(os.path.join(ROOT, "include", "sbn254.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sbn_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(sbn):
    import ctypes
    lib = ctypes.CDLL(sbn.lib_path())
    names = header_functions()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/sbn254.h but not exported"
    assert sorted(sbn.EXPORTED_SYMBOLS) == names, "binding.py symbol list and the header disagree"


def test_no_oracle_in_product():
    """the product path must not include, link or call oracle/ (③)"""
    pk = os.path.join(ROOT, "spartan-bn254_amd")
    for dirpath, _, files in os.walk(pk):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".hpp", ".inc", ".h", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "bn254_oracle" not in txt and "libsbn_oracle" not in txt and "orc_" not in txt, f
                assert "oracle_lib" not in txt, f
    import subprocess
    out = subprocess.run(["ldd", os.path.join(pk, "libsbn254_hip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_version_and_lens(sbn):
    assert b"gfx950" in sbn.lib().sbn_version()
    # hyrax.rs:371-373 ; derefs at keyless size: ell = 25 -> 4096 x 8192
    assert sbn.factored_lens(25) == (12, 13)
    assert sbn.factored_lens(20) == (10, 10)
    assert sbn.factored_lens(1) == (0, 1)
    assert sbn.factored_lens(0) == (0, 0)


def test_compress_matches_oracle_and_golden(sbn, ol, pr):
    g = golden("g1_kat.json")
    pts = b"".join(H(r["kG"]) for r in g["mul"]) + bytes(64)
    want = b"".join(H(r["compressed"]) for r in g["mul"]) + bytes(31) + b"\x40"
    assert sbn.g1_compress(pts) == want
    assert sbn.g1_compress(pts) == b"".join(ol.g1_compress(pts[64 * i:64 * i + 64]) for i in range(len(pts) // 64))
    assert sbn.g1_compress(b"") == b""


def test_g1_sum_host(sbn, ol, pr):
    G = pr.point_to_xy(pr.G)
    out, inf = sbn.g1_sum(G + G + G)
    assert out == ol.g1_mul(G, (3).to_bytes(32, "little")) and not inf
    out, inf = sbn.g1_sum(G + ol.g1_neg(G))
    assert inf and out == bytes(64)
    out, inf = sbn.g1_sum(b"")
    assert inf
    out, inf = sbn.g1_sum(bytes(64) + G + bytes(64))
    assert out == G
    for row in golden("g1_kat.json")["add"]:
        out, _ = sbn.g1_sum(H(row["p"]) + H(row["q"]))
        assert out == H(row["sum"]), row["note"]
    with pytest.raises(sbn.SbnError):
        sbn.g1_sum(b"\xff" * 64)          # non-canonical coordinates are rejected


def test_unipoly_host_mirror(sbn, ol, pr):
    """UniPoly::from_evals / evaluate (unipoly.rs:28-82) in the product's host code: the reference's own known answers
    (unipoly.rs:130-184: 2x^2+3x+1 from (1,6,15); x^3+2x^2+3x+1 from (1,7,23,55)) and random values against the oracle"""
    from conftest import fr_bytes, rand_scalars
    for u in golden("sumcheck_kat.json")["unipoly"]:
        ev = b"".join(H(x) for x in u["evals"])
        co = sbn.unipoly_from_evals(ev)
        assert co == b"".join(H(x) for x in u["coeffs"]), u["note"]
        assert sbn.unipoly_eval(co, H(u["at"])) == H(u["value"])
    assert sbn.unipoly_from_evals(fr_bytes([1, 6, 15])) == fr_bytes([1, 3, 2])
    assert sbn.unipoly_from_evals(fr_bytes([1, 7, 23, 55])) == fr_bytes([1, 3, 2, 1])
    for n in (3, 4):
        for seed in range(10):
            ev = rand_scalars(n, 100 * n + seed); r = rand_scalars(1, seed)
            co = sbn.unipoly_from_evals(ev)
            assert co == ol.unipoly_from_evals(ev)
            assert sbn.unipoly_eval(co, r) == ol.unipoly_eval(co, r)
            # p(0) + p(1) == e0 + e1 (the verifier's check, sumcheck.rs:62-66)
            cs = [int.from_bytes(co[32 * i:32 * i + 32], "little") for i in range(n)]
            e = [int.from_bytes(ev[32 * i:32 * i + 32], "little") for i in range(2)]
            assert (cs[0] + sum(cs)) % pr.R == (e[0] + e[1]) % pr.R
    with pytest.raises(sbn.SbnError):
        sbn.unipoly_from_evals(rand_scalars(2, 1))
    with pytest.raises(sbn.SbnError):
        sbn.unipoly_from_evals(pr.R.to_bytes(32, "little") * 3)


def test_device_entry_points_fail_loudly_without_gpu(sbn):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(sbn.SbnError):
        sbn.Context(0)


def test_shard_ranges(sbn):
    from spartan_bn254_amd import sharding
    for n in (0, 1, 7, 8, 1000, (1 << 20) + 3):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    rows = sorted(sum((sharding.shard_rows(4096, r, 8) for r in range(8)), []))
    assert rows == list(range(4096))
    # zero-padding rows 3072.. (hyrax.rs:245) spread evenly over the ranks
    assert all(sum(1 for x in sharding.shard_rows(4096, r, 8) if x >= 3072) == 128 for r in range(8))


def test_cols_headroom_arithmetic():
    """fp.cuh's 64-bit product columns (`Cols`): the callers accumulate up to `pend == 6` products between carry passes and then
    run the reduction on top (sumcheck_comb_kernels.cuh).  With the limb width and count parsed from the shipped sources, the
    worst case (every operand limb at its maximum) must stay below 2^64 — an edit to the limb layout or to the callers' carry
    interval that breaks the bound fails here.  (tools/micro/fptest runs the same worst case on the device.)"""
    cs = os.path.join(ROOT, "spartan-bn254_amd", "csrc")
    fp = open(os.path.join(cs, "fp.cuh")).read()
    nl = int(re.search(r"constexpr int NL = (\d+);", fp).group(1))
    bits = int(re.search(r"constexpr uint32_t LMASK = \(1u << (\d+)\) - 1;", fp).group(1))
    assert nl * bits >= 256 + 5
    lmax = (1 << bits) - 1
    pend = set()
    for f in ("sumcheck_comb_kernels.cuh", "sumcheck_kernels.cuh", "g1.cuh", "msm_kernels.cuh", "comb_kernels.cuh"):
        pend |= {int(x) for x in re.findall(r"pend == (\d+)", open(os.path.join(cs, f)).read())}
    assert pend, "no carry interval found in the kernels"
    for cap in pend:
        after_carry = lmax + (1 << (64 - bits))              # a column after cols_carry: below 2^bits, plus the carry that came in from below
        worst = after_carry + cap * nl * lmax * lmax          # `cap` products of nl limb products each in the middle column
        worst += nl * lmax * lmax                            # the reduction: nl quotient digits (< 2^bits) times modulus limbs (< 2^bits)
        worst += worst >> bits                               # the carry the reduction moves up from the column below
        assert worst < 1 << 64, f"Cols overflow with {cap} products between carry passes ({worst.bit_length()} bits)"
    # the lazy two-product form (g1.cuh: Y3): one operand's limbs below 2^30.6
    lazy = int(2 ** 30.6)
    assert 2 * nl * lmax * lazy + nl * lmax * lmax + (1 << 36) < 1 << 64


def test_sc_grids_clamped_to_partial_area():
    """every grid of the combined sumcheck kernels is clamped to the partial-sum area (SC_PART_*_BLOCKS), whatever SBN_SC_* says"""
    src = open(os.path.join(ROOT, "spartan-bn254_amd", "csrc", "abi_sumcheck.inc")).read()
    assert "static_assert(SC_PARTIAL_BYTES ==" in src
    for name in ("gxc", "gxs", "gx_seq"):
        m = re.search(r"const unsigned %s = ([^;]*);" % name, src)
        assert m and "SC_PART_" in m.group(1), name
    for m in re.finditer(r"const unsigned gx = ([^;]*);", src):
        assert "SC_PART_" in m.group(1), m.group(0)
    # no environment lookup on a round's path any more
    for f in ("abi_sumcheck.inc", "abi_tables.inc", "abi_bullet.inc"):
        assert "getenv(" not in open(os.path.join(ROOT, "spartan-bn254_amd", "csrc", f)).read(), f

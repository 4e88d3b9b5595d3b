"""CPU: sanitizer builds (ASan + UBSan) of the host-side product code and of the oracle.  GPU AddressSanitizer / XNACK are
not available on the pool, so sanitizers run on the CPU builds only."""
import os
import subprocess
import sys

from conftest import ROOT


def _run(cmd, **kw):
    return subprocess.run(cmd, capture_output=True, text=True, timeout=600, **kw)


def test_host_code_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "host_sanitize")
    r = _run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
              os.path.join(ROOT, "tests", "host_sanitize.cpp"), "-o", exe])
    assert r.returncode == 0, r.stderr
    r = _run([exe])
    assert r.returncode == 0 and "HOST SANITIZE OK" in r.stdout, r.stdout + r.stderr


def test_oracle_under_asan_ubsan(tmp_path):
    """the checker itself: a sanitizer build of oracle/bn254_oracle.c driven through a few representative calls"""
    so = str(tmp_path / "libsbn_oracle_asan.so")
    r = _run(["gcc", "-O1", "-g", "-fPIC", "-fopenmp", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-std=c11", "-shared",
              os.path.join(ROOT, "oracle", "bn254_oracle.c"), "-o", so])
    assert r.returncode == 0, r.stderr
    libasan = _run(["gcc", "-print-file-name=libasan.so"]).stdout.strip()
    code = f"""
import ctypes, sys
sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})
import oracle_lib as ol
ol._LIB = ctypes.CDLL({so!r}); ol._LIB.orc_msm_window_bits.restype = ctypes.c_int; ol._LIB.orc_msm_window_bits.argtypes = [ctypes.c_size_t]
from conftest import rand_scalars
import pyref as pr
n = 70
sc = rand_scalars(n, 1); dl = rand_scalars(n, 2); pts = ol.g1_mul_gen_batch(dl, 2)
want = ol.g1_mul(pr.point_to_xy(pr.G), ol.fr_dot(sc, dl))
assert ol.msm_pippenger(sc, pts, 2) == want and ol.msm_naive(sc, pts) == want
xy, d = ol.gens_new(9, b"gens_r1cs_eval", 2)
assert ol.commit_rows(sc[:32*18], sc[:64], 2, 9, xy[:64*9], xy[64*9:], 2)
A, B, C = (rand_scalars(16, s) for s in (3, 4, 5))
ol.sc_eval_cubic(A, B, C); ol.sc_eval_r1cs(A, B, C, A); ol.sc_eval_quad(A, B); ol.bind_top(A, sc[:32]); ol.eq_evals(sc[:96])
ol.unipoly_from_evals(sc[:128]); ol.g1_decompress(ol.g1_compress(pts[:64])); ol.shake256(b"x" * 300, 500)
print("ORACLE SANITIZE OK")
"""
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0")
    r = _run([sys.executable, "-c", code], env=env)
    assert r.returncode == 0 and "ORACLE SANITIZE OK" in r.stdout, r.stdout + r.stderr[-3000:]

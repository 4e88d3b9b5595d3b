"""Device groups (sbn_group_*): one process, one call, several GPUs.  The GPU box has one device, so the groups here list device 0
several times (N contexts, N host threads, the same code path as N devices).  Everything against the CPU oracle: bit-exact."""
import pytest
from conftest import rand_scalars

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def group(sbn):
    g = sbn.Group([0, 0, 0])
    yield g
    g.close()


@pytest.mark.parametrize("L,R", [(1, 16), (2, 64), (10, 64), (33, 128)])
def test_group_commit_rows_one_matrix_interleaved(group, ol, L, R):
    """ONE L x R matrix, row i on context i mod 3 (hyrax.rs:259-261 iterates the rows; they are independent): with and without
    blinds, L not a multiple of the group size, fewer rows than contexts"""
    gb, gxy = group.gens_new(R, b"gens_r1cs_eval")
    assert gxy == ol.gens_new(R, b"gens_r1cs_eval")[0]
    Z = bytearray(rand_scalars(L * R, 100 + L))
    if L > 4:
        Z[32 * R * 3:32 * R * 4] = bytes(32 * R)                       # an all-zero row commits to the identity
    Z = bytes(Z)
    out, inf = group.commit_rows(gb, Z, None, L, R)
    assert out == ol.commit_rows(Z, None, L, R, gxy[:64 * R], gxy[64 * R:], 4)
    if L > 4:
        assert inf[3] == 1 and out[64 * 3:64 * 4] == bytes(64)
    bl = rand_scalars(L, 200 + L)
    assert group.commit_rows(gb, Z, bl, L, R)[0] == ol.commit_rows(Z, bl, L, R, gxy[:64 * R], gxy[64 * R:], 4)
    group.bases_precompute(gb, 8 << 20)                                 # lookup tables on every context: same commitments
    assert group.commit_rows(gb, Z, bl, L, R)[0] == ol.commit_rows(Z, bl, L, R, gxy[:64 * R], gxy[64 * R:], 4)
    gb.free()


@pytest.mark.parametrize("L,R", [(2, 64), (10, 64), (32, 128)])
def test_group_commit_rows_dev_interleaved(group, ol, L, R):
    """sbn_group_commit_rows_dev: the matrix is ALREADY on the devices, context d holding the rows d, d + 3, ... (hyrax.rs:253-267 over
    device-resident rows); canonical and ark-Montgomery scalars, with and without blinds, ragged row counts, fewer rows than contexts"""
    import torch
    import numpy as np
    N = 3
    gb, gxy = group.gens_new(R, b"gens_r1cs_eval")
    Z = rand_scalars(L * R, 500 + L); bl = rand_scalars(L, 600 + L)
    want = ol.commit_rows(Z, None, L, R, gxy[:64 * R], gxy[64 * R:], 4)
    want_b = ol.commit_rows(Z, bl, L, R, gxy[:64 * R], gxy[64 * R:], 4)
    Zn = np.frombuffer(Z, dtype=np.uint8).reshape(L, R * 32); Bn = np.frombuffer(bl, dtype=np.uint8).reshape(L, 32)
    zs = [torch.from_numpy(np.ascontiguousarray(Zn[d::N])).cuda() for d in range(N)]
    bs = [torch.from_numpy(np.ascontiguousarray(Bn[d::N])).cuda() for d in range(N)]
    torch.cuda.synchronize()
    zp = [t.data_ptr() if t.numel() else 0 for t in zs]; bp = [t.data_ptr() if t.numel() else 0 for t in bs]
    out, inf = group.commit_rows_dev(gb, zp, None, L, R)
    assert out == want and not any(inf)
    assert group.commit_rows_dev(gb, zp, bp, L, R)[0] == want_b
    group.bases_precompute(gb, 8 << 20)
    assert group.commit_rows_dev(gb, zp, bp, L, R)[0] == want_b
    assert group.commit_rows_dev(gb, zp, None, L, R)[0] == want == group.commit_rows(gb, Z, None, L, R)[0]
    gb.free()


@pytest.mark.parametrize("log_n,count,log_mem,R", [(6, 6, 5, 32), (8, 6, 7, 64), (7, 3, 6, 16), (5, 5, 4, 8)])
def test_group_gather_commit_derefs(group, ol, pr, log_n, count, log_mem, R):
    """sbn_group_gather_commit: Derefs::commit over the group from device-resident inputs (sparse_mlpoly_full.rs:245-257, 293-297, 301-304):
    the eq tables are built on every context, each context gathers and commits only its interleaved rows of the merged polynomial.  Against
    the oracle's row commitments of the host-side gather, and against the single-context sbn_gather_merge + sbn_commit_table."""
    import torch
    import numpy as np
    N = 3
    n = 1 << log_n; nmem = 1 << log_mem
    padded = 1
    while padded < count * n:
        padded <<= 1
    L = padded // R
    rng = np.random.default_rng(1000 + log_n + count)
    addrs = [rng.integers(0, nmem, size=n, dtype=np.uint32) for _ in range(count)]
    rx = rand_scalars(log_mem, 701); ry = rand_scalars(log_mem, 702)
    eqs_host = [ol.eq_evals(rx), ol.eq_evals(ry)]
    comb = b"".join(b"".join(eqs_host[k % 2][32 * int(a):32 * int(a) + 32] for a in addrs[k]) for k in range(count))
    comb += bytes(32 * (padded - count * n))
    gb, gxy = group.gens_new(R, b"gens_r1cs_eval")
    want = ol.commit_rows(comb, None, L, R, gxy[:64 * R], gxy[64 * R:], 4)
    mem, aptr, keep = [], [], []
    for d in range(N):
        cx = group.ctx(d)
        tx, ty = cx.eq_evals(rx), cx.eq_evals(ry)
        at = [torch.from_numpy(a.view(np.int32)).cuda() for a in addrs]
        keep.append((tx, ty, at))
        mem.append([tx if k % 2 == 0 else ty for k in range(count)])
        aptr.append([t.data_ptr() for t in at])
    torch.cuda.synchronize()
    out, inf = group.gather_commit(gb, mem, aptr, n, L, R)
    assert out == want
    assert all(bool(inf[i]) == (out[64 * i:64 * i + 64] == bytes(64)) for i in range(L))
    # the one-context form of the same commitment
    c0 = group.ctx(0)
    t = c0.gather_merge(mem[0], aptr[0], n)
    assert c0.table_download(t) == comb
    t.free()
    # one context's share on its own: rows 1, 4, 7, ...
    rows1 = len(range(1, L, N))
    if rows1:
        t1 = c0.gather_merge_rows(mem[0], aptr[0], n, R, 1, N, rows1)
        assert len(t1) == rows1 * R
        assert c0.table_download(t1) == b"".join(comb[32 * R * r:32 * R * (r + 1)] for r in range(1, L, N))
        t1.free()
    with pytest.raises(Exception):
        c0.gather_merge_rows(mem[0], aptr[0], n, R, 1, N, rows1 + 1)          # a row outside the matrix
    for tx, ty, _ in keep:
        tx.free(); ty.free()
    gb.free()


@pytest.mark.parametrize("n", [1, 2, 5, 1000, 1 << 14])
def test_group_msm_base_point_ranges(group, ol, n):
    """ONE MSM cut into contiguous base-point ranges, the per-context partial sums folded on the host (group.rs:171-175 + the `+` of
    GroupElement); n below the group size leaves contexts without work"""
    sc = rand_scalars(n, 300 + n); dl = rand_scalars(n, 400 + n)
    pts = ol.g1_mul_gen_batch(dl, 8)
    want = ol.msm_pippenger(sc, pts, 8)
    assert group.msm(sc, pts) == (want, False)
    gb = group.bases_upload_ranges(pts)
    assert [gb.range(d) for d in range(3)] == [(n * d // 3, n * (d + 1) // 3) for d in range(3)]
    assert group.msm_bases(gb, sc) == (want, False)
    gb.free()
    # cancelling pairs: P and -P with equal scalars on different ranges -> the identity
    if n >= 2 and n % 2 == 0:
        half = n // 2
        pts2 = pts[:64 * half] + b"".join(ol.g1_neg(pts[64 * i:64 * i + 64]) for i in range(half))
        sc2 = sc[:32 * half] + sc[:32 * half]
        out, inf = group.msm(sc2, pts2)
        assert inf and out == bytes(64)


def test_group_msm_synthetic_ranges_dlog(group, sbn, ol, pr):
    """the synthetic benchmark bases built range by range on the contexts (P_i = (s0 + i d) G), device-resident scalar slices, and the
    discrete-log identity as the oracle (SURVEY 8d config 4's shape, 2^20 points over 3 contexts)"""
    import torch
    import bench
    n = 1 << 20
    S0, D = bench.S0, bench.DSTEP
    gb = group.bases_synthetic_ranges(n, S0.to_bytes(32, "little"), D.to_bytes(32, "little"))
    scal = torch.empty(32 * n, dtype=torch.uint8, device="cuda")
    group.ctx(0).scalars_synthetic(bench.SEED, 0, n, scal.data_ptr())
    torch.cuda.synchronize()
    ptrs = [scal.data_ptr() + 32 * gb.range(d)[0] for d in range(3)]
    got, inf = group.msm_bases_dev(gb, ptrs)
    want = ol.g1_mul(bench.G_XY, bench.dlog_expect(scal, 0, n))
    assert got == want and not inf
    assert group.msm_bases(gb, scal.cpu().numpy().tobytes()) == (want, False)
    gb.free()


def test_group_errors(group, sbn):
    gb, _ = group.gens_new(16, b"x")
    with pytest.raises(sbn.SbnError):
        group.commit_rows(gb, rand_scalars(8, 1), None, 1, 8)            # R != gens.n (commitments.rs:146)
    with pytest.raises(sbn.SbnError):
        group.msm_bases(gb, rand_scalars(16, 1))                         # a replicated set is not a ranged set
    gb.free()
    with pytest.raises(sbn.SbnError):
        sbn.Group([7])                                                   # no such device

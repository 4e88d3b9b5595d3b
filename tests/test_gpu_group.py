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

"""CPU, world_size 2 over gloo: the N>1 path of the MSM — base-point-range shards, one all-gather of the 64-byte partial
sums, local fold — and the row-sharded Hyrax commit.  On the GPU box the same code runs over RCCL ("nccl"); here the
per-rank partial MSM is produced by the oracle because there is no GPU (test-only injection)."""
import os
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from conftest import load_pkg, rand_scalars
    import oracle_lib as ol
    sbn = load_pkg()
    from spartan_bn254_amd import sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = rand_scalars(n, 5); dl = rand_scalars(n, 6)
        pts = ol.g1_mul_gen_batch(dl, 2)
        lo, hi = sharding.shard_range(n, rank, world)
        partial = ol.msm_pippenger(sc[32 * lo:32 * hi], pts[64 * lo:64 * hi], 1)      # stands in for sbn_msm on this rank's GPU
        xy, inf = sharding.allgather_fold(partial)
        full = ol.msm_pippenger(sc, pts, 1)
        # rows: 6 x 16 Hyrax commit, rows interleaved over ranks, no reduction
        L, R = 6, 16
        gx, _ = ol.gens_new(R, b"gens_r1cs_eval", 2)
        Z = rand_scalars(L * R, 7)
        mine = sharding.shard_rows(L, rank, world)
        my_rows = b"".join(ol.commit(Z[32 * R * i:32 * R * (i + 1)], bytes(32), gx[:64 * R], gx[64 * R:]) for i in mine)
        allrows = sharding.gather_rows(my_rows, L, rank, world)
        want_rows = ol.commit_rows(Z, None, L, R, gx[:64 * R], gx[64 * R:], 2)
        q.put((rank, xy == full, inf, allrows == want_rows))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [37, 200])
def test_msm_sharded_world2(n):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500) + n % 7
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, inf, rows_ok in res:
        assert ok and not inf and rows_ok, f"rank {rank}"

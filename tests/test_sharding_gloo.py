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


def _sc_worker(rank, world, port, logn, q):
    """prove_cubic_batched (2 instances sharing C) with residue-sharded tables: local sums + bind by the oracle (no GPU here),
    one all-gather per round; must reproduce the unsharded oracle run round by round and in the final claims."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hashlib
    import torch.distributed as dist
    from conftest import load_pkg, rand_scalars
    import oracle_lib as ol
    load_pkg()
    from spartan_bn254_amd import sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 1 << logn
        full = [rand_scalars(n, 60 + i) for i in range(5)]                   # A1, B1, A2, B2, C
        loc = [sharding.shard_table_residue(t, rank, world) for t in full]
        ok = True
        chal = lambda ev: (int.from_bytes(hashlib.sha3_256(ev).digest(), "little") % sharding.R_MOD).to_bytes(32, "little")
        while len(loc[0]) // 32 >= 2:                                         # local rounds
            mine = ol.sc_eval_cubic(loc[0], loc[1], loc[4]) + ol.sc_eval_cubic(loc[2], loc[3], loc[4])
            ev = sharding.allreduce_round_sums(mine)
            want = ol.sc_eval_cubic(full[0], full[1], full[4]) + ol.sc_eval_cubic(full[2], full[3], full[4])
            ok &= ev == want
            r = chal(ev)
            loc = [ol.bind_top(t, r) for t in loc]
            full = [ol.bind_top(t, r) for t in full]
        rest = sharding.gather_residual_tables(b"".join(loc))                 # world-entry tables, finished on every rank
        ok &= rest == full
        while len(rest[0]) // 32 >= 2:
            ev = ol.sc_eval_cubic(rest[0], rest[1], rest[4]) + ol.sc_eval_cubic(rest[2], rest[3], rest[4])
            r = chal(ev)
            rest = [ol.bind_top(t, r) for t in rest]
            full = [ol.bind_top(t, r) for t in full]
        ok &= rest == full and len(rest[0]) == 32
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("logn", [3, 7])
def test_sumcheck_residue_sharded_world2(logn):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29100 + (os.getpid() % 400) + logn
    procs = [ctx.Process(target=_sc_worker, args=(r, world, port, logn, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)


def test_residue_shard_helpers():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_pkg
    load_pkg()
    from spartan_bn254_amd import sharding
    t = b"".join(i.to_bytes(32, "little") for i in range(8))
    assert sharding.shard_table_residue(t, 1, 4) == (1).to_bytes(32, "little") + (5).to_bytes(32, "little")
    with pytest.raises(ValueError):
        sharding.shard_table_residue(t, 0, 3)
    a = (sharding.R_MOD - 1).to_bytes(32, "little") + (5).to_bytes(32, "little")
    b = (2).to_bytes(32, "little") + (7).to_bytes(32, "little")
    assert sharding.fr_sum([a, b]) == (1).to_bytes(32, "little") + (12).to_bytes(32, "little")

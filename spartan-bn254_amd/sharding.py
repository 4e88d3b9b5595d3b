"""Multi-GPU layer of the MSM path: one process per GPU, base-point-range sharding, one gather of partial sums.

north_star: "MSM shards by base-point range across the 8 GPUs of one node with a single RCCL reduce of partial G1
sums over xGMI".  RCCL has no elliptic-curve reduction op, so the "reduce" is an all-gather of each rank's 64-byte
canonical affine partial (as uint8) followed by a local fold with the group law (sbn_g1_sum) — 8 x 64 B, latency-bound.
Hyrax row commits shard by rows and need no reduction at all (rows are independent, hyrax.rs:259-261): ranks take
interleaved rows so the zero-padding rows (3072..4095 at keyless size, hyrax.rs:245) spread evenly.

Sumcheck rounds shard by index RESIDUE: rank g keeps the entries i = g (mod N) of every table.  bound_poly_var_top pairs entry
i with i + len/2 (hyrax.rs:195-203), and len/2 is a multiple of N, so both live on the same rank: binding is local in every
round, a round's sums e0, e2, e3 (sumcheck.rs:111-135) are plain sums over i, hence sums of the ranks' partial sums, and the
only exchange is one all-gather of 96 B per instance per round (Fr addition on the host).  After log2(len/N) rounds every rank
holds one entry per table; the N-entry tables that remain are gathered once and finished on every rank identically.  No
re-pairing of halves across GPUs is ever needed (the index-range split of SURVEY 8e would need it for the first log2 N rounds).

The functions take `torch.distributed` process groups; on a GPU box the backend is "nccl" (= RCCL), the CPU tests use gloo.
"""
from . import binding


# collectives issued through this module since import (bench.py reports them: evidence that the RCCL path ran, and how often)
STATS = {"all_gather_calls": 0, "all_gather_bytes_per_rank": 0, "device_tensors": 0}


def _count(t):
    STATS["all_gather_calls"] += 1
    STATS["all_gather_bytes_per_rank"] += int(t.numel())
    if t.is_cuda:
        STATS["device_tensors"] += 1


def shard_range(n, rank, world):
    """contiguous base-point range [lo, hi) of rank `rank` (sizes differ by at most one)"""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_rows(L, rank, world):
    """interleaved Hyrax rows of rank `rank`: rank, rank+world, ..."""
    return list(range(rank, L, world))


def allgather_fold(partial_xy, group=None, device=None):
    """all-gather the ranks' 64-byte partial sums and fold them; every rank returns the same (xy, is_inf)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    t = torch.frombuffer(bytearray(partial_xy), dtype=torch.uint8)
    if device is not None:
        t = t.to(device)
    outs = [torch.empty_like(t) for _ in range(world)]
    _count(t)
    dist.all_gather(outs, t, group=group)
    allxy = b"".join(bytes(o.cpu().numpy().tobytes()) for o in outs)
    return binding.g1_sum(allxy)


def gather_rows(rows_xy, L, rank, world, group=None, device=None):
    """all-gather interleaved row commitments (each rank holds rows rank::world, padded to equal count) -> L x 64 B"""
    import torch
    import torch.distributed as dist
    per = (L + world - 1) // world
    buf = bytearray(per * 64)
    buf[: len(rows_xy)] = rows_xy
    t = torch.frombuffer(buf, dtype=torch.uint8)
    if device is not None:
        t = t.to(device)
    outs = [torch.empty_like(t) for _ in range(world)]
    _count(t)
    dist.all_gather(outs, t, group=group)
    full = bytearray(L * 64)
    for r, o in enumerate(outs):
        ob = o.cpu().numpy().tobytes()
        for j, row in enumerate(range(r, L, world)):
            full[64 * row: 64 * row + 64] = ob[64 * j: 64 * j + 64]
    return bytes(full)


# ---- sumcheck rounds, residue-sharded -------------------------------------------------------------------------------
R_MOD = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001


def shard_table_residue(table_bytes, rank, world):
    """entries i = rank (mod world) of a table of 32-byte scalars, in order: the rank's local table (length len/world)"""
    n = len(table_bytes) // 32
    if n % world:
        raise ValueError("table length must be a multiple of the number of ranks")
    return b"".join(table_bytes[32 * i:32 * i + 32] for i in range(rank, n, world))


def fr_sum(blobs):
    """elementwise sum mod r of equally long strings of canonical 32-byte scalars (the ranks' partial round sums)"""
    k = len(blobs[0]) // 32
    out = bytearray(32 * k)
    for j in range(k):
        v = sum(int.from_bytes(b[32 * j:32 * j + 32], "little") for b in blobs) % R_MOD
        out[32 * j:32 * j + 32] = v.to_bytes(32, "little")
    return bytes(out)


def allgather_bytes(local, group=None, device=None):
    """every rank's byte string (equal lengths), in rank order"""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    t = torch.frombuffer(bytearray(local), dtype=torch.uint8)
    if device is not None:
        t = t.to(device)
    outs = [torch.empty_like(t) for _ in range(world)]
    _count(t)
    dist.all_gather(outs, t, group=group)
    return [bytes(o.cpu().numpy().tobytes()) for o in outs]


def allreduce_round_sums(local_evals, group=None, device=None):
    """the round's (e0, e2, e3) per instance over ALL ranks: all-gather of the partial sums + Fr addition"""
    return fr_sum(allgather_bytes(local_evals, group, device))


def gather_residual_tables(local_entries, group=None, device=None):
    """after the local rounds every rank holds ONE entry per table (local_entries = those, 32 B each, in table order);
    returns the remaining world-entry tables, entry g from rank g, identical on every rank"""
    parts = allgather_bytes(local_entries, group, device)
    k = len(local_entries) // 32
    return [b"".join(p[32 * t:32 * t + 32] for p in parts) for t in range(k)]

"""Multi-GPU layer of the MSM path: one process per GPU, base-point-range sharding, one gather of partial sums.

north_star: "MSM shards by base-point range across the 8 GPUs of one node with a single RCCL reduce of partial G1
sums over xGMI".  RCCL has no elliptic-curve reduction op, so the "reduce" is an all-gather of each rank's 64-byte
canonical affine partial (as uint8) followed by a local fold with the group law (sbn_g1_sum) — 8 x 64 B, latency-bound.
Hyrax row commits shard by rows and need no reduction at all (rows are independent, hyrax.rs:259-261): ranks take
interleaved rows so the zero-padding rows (3072..4095 at keyless size, hyrax.rs:245) spread evenly.

The functions take `torch.distributed` process groups; on a GPU box the backend is "nccl" (= RCCL), the CPU tests use gloo.
"""
from . import binding


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
    dist.all_gather(outs, t, group=group)
    full = bytearray(L * 64)
    for r, o in enumerate(outs):
        ob = o.cpu().numpy().tobytes()
        for j, row in enumerate(range(r, L, world)):
            full[64 * row: 64 * row + 64] = ob[64 * j: 64 * j + 64]
    return bytes(full)

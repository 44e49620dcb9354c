"""Batch-axis sharding across ranks (one process per GPU) and the single collective of the path: an all-gather of
result rows.  Works with the RCCL ("nccl") backend on GPUs and with gloo on CPU (tests)."""


def shard_range(B, rank, world):
    """Contiguous slice [lo, hi) of the batch owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_rows(local, B, world):
    """all_gather of per-rank row blocks (uneven shards padded to the largest) -> (B, ...) tensor on every rank."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return local
    rows = (B + world - 1) // world
    pad = torch.zeros((rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    parts = []
    for r in range(world):
        lo, hi = shard_range(B, r, world)
        parts.append(out[r][: hi - lo])
    return torch.cat(parts, dim=0)

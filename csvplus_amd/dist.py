"""Multi-GPU exchange for the sharded Join: one process per GPU,
probe rows split into contiguous ranges, build side replicated, and an
allgatherv of the per-rank match lists so that every rank ends up with the
whole joined row-id list in the reference's emission order.

RCCL has no native allgatherv.  torch.distributed's NCCL(=RCCL) backend lowers
an all_gather with unequal output sizes to one grouped set of broadcasts, i.e.
every rank's shard travels straight to each peer over its own xGMI link; the
gloo backend (CPU tests) needs equal sizes, so shards are padded to the maximum
there.
"""
from __future__ import annotations


def allgatherv(t, group=None):
    """Concatenation over ranks (rank order) of 1-D tensor `t`, whose length may differ per rank.
    Returns (gathered, counts)."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t, [int(t.numel())]
    world = dist.get_world_size(group)
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    counts_t = [torch.zeros(1, dtype=torch.int64, device=t.device) for _ in range(world)]
    dist.all_gather(counts_t, n, group=group)
    counts = [int(c.item()) for c in counts_t]
    total = sum(counts)
    out = torch.empty(total, dtype=t.dtype, device=t.device)
    backend = dist.get_backend(group)
    if backend == "nccl":
        views, off = [], 0
        for c in counts:
            views.append(out[off:off + c])
            off += c
        dist.all_gather(views, t.contiguous(), group=group)   # unequal sizes: grouped broadcasts
    else:
        mx = max(counts) if counts else 0
        pad = torch.zeros(mx, dtype=t.dtype, device=t.device)
        pad[: t.numel()] = t
        bufs = [torch.empty(mx, dtype=t.dtype, device=t.device) for _ in range(world)]
        dist.all_gather(bufs, pad, group=group)
        off = 0
        for c, b in zip(counts, bufs):
            out[off:off + c] = b[:c]
            off += c
    return out, counts


def sharded_chained_join(total_stream_rows: int, local_join, group=None, exchange: bool = True):
    """Runs `local_join(begin, end)` on this rank's row range and (optionally) allgathers
    the resulting triples.

    local_join(begin, end) -> (stream_row, a_row, b_row) 1-D tensors for stream rows
    [begin, end), with stream_row holding GLOBAL row numbers.  Returned: the three
    gathered tensors (emission order of the whole stream) and the per-rank counts."""
    import torch.distributed as dist

    from .engine import shard_range

    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    begin, end = shard_range(total_stream_rows, rank, world)
    s, a, b = local_join(begin, end)
    if not exchange or world == 1:
        return s, a, b, [int(s.numel())]
    gs, counts = allgatherv(s, group)
    ga, _ = allgatherv(a, group)
    gb, _ = allgatherv(b, group)
    return gs, ga, gb, counts

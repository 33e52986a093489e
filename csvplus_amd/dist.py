"""Multi-GPU exchange for the sharded Join: one process per GPU,
probe rows split into contiguous ranges, build side replicated, and an
allgatherv of the per-rank match lists so that every rank ends up with the
whole joined row-id list in the reference's emission order.

RCCL has no native allgatherv.  The exchange is: one `all_gather` of the per-rank
counts, then ONE grouped batch of point-to-point sends/receives
(`batch_isend_irecv` = ncclGroupStart ... ncclSend/ncclRecv ... ncclGroupEnd):
every rank sends its shard straight to each peer and receives each peer's shard
into its slot of the output buffer.  On the 8-GPU xGMI mesh every pair of GPUs
has its own link, so the N-1 transfers of a rank run concurrently, one per link —
a ring all-gather would push (N-1)/N of the data through every single link instead.
The same code runs on the gloo backend (CPU tests).
"""
from __future__ import annotations


def allgatherv(t, group=None, single_rank_shortcut: bool = True):
    """Concatenation over ranks (rank order) of 1-D tensor `t`, whose length may differ per rank.
    Returns (gathered, counts).  single_rank_shortcut=False sends a 1-rank group through the collectives
    too (used by the 1-GPU RCCL smoke test)."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and single_rank_shortcut):
        return t, [int(t.numel())]
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if dist.get_backend(group) == "gloo" and t.is_cuda:   # debug path (several ranks sharing one GPU)
        out, counts = allgatherv(t.cpu(), group)
        return out.to(t.device), counts
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    counts_t = torch.zeros(world, dtype=torch.int64, device=t.device)
    dist.all_gather_into_tensor(counts_t, n, group=group)
    counts = [int(c) for c in counts_t.tolist()]
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + c)
    out = torch.empty(offs[-1], dtype=t.dtype, device=t.device)
    t = t.contiguous()
    if len(set(counts)) == 1:   # equal shards (e.g. every stream row joined): one plain ncclAllGather
        if counts[0]:
            dist.all_gather_into_tensor(out, t, group=group)
        return out, counts
    if counts[rank]:
        out[offs[rank]:offs[rank + 1]].copy_(t)
    ops = []
    for peer in range(world):
        if peer == rank:
            continue
        gpeer = dist.get_global_rank(group, peer) if group is not None else peer
        if counts[rank]:
            ops.append(dist.P2POp(dist.isend, t, gpeer, group=group))
        if counts[peer]:
            ops.append(dist.P2POp(dist.irecv, out[offs[peer]:offs[peer + 1]], gpeer, group=group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out, counts


def sharded_chained_join(total_stream_rows: int, local_join, group=None, exchange: bool = True):
    """Runs `local_join(begin, end)` on this rank's row range and (optionally) allgathers
    the resulting tuples.

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

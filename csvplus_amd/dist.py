"""Multi-GPU exchange for the sharded Join: one process per GPU, probe rows split into
contiguous ranges, build side replicated (or broadcast), and an allgatherv of the per-rank row-id lists so
that every rank ends up with the whole joined list in the reference's emission order.

The exchange itself lives behind the C ABI (`cph_dist_*`, csrc/dist.hip: RCCL count all-gather + ONE
grouped send/recv batch for all arrays of a result).  This module is the thin host side:

  connect(ctx)            ships RCCL's unique id from rank 0 to the other ranks over whatever
                          torch.distributed group the launcher set up, then cph_dist_create
  chain_allgather(...)    ChainResult of this rank -> ChainResult of the whole stream (torch views of the
                          library's gathered device arrays)
  N.Dist.join_chain(...)  cph_dist_join_chain: the shard joined in sub-chunks, chunk k on the wire (or on its way to the
                          node's shared host buffer) while chunk k+1 is joined — what `bench.py --gpus N` times
  build_side_estimate()   replicated builds vs. one build + cph_dist_index_broadcast, in milliseconds (printed by bench.py)

`pipelined_dense_exchange` is the control flow of cph_dist_join_chain written against torch.distributed (same chunk cut,
same displacements, dense slots with the ABSENT sentinel, ONE exchange of totals at the end, local compaction): the
transport of the CPU tests (gloo).

`allgatherv_many` is the same exchange written against torch.distributed (count all_gather + one
batch_isend_irecv for all tensors).  It is the transport of the CPU tests (gloo) and of the debug mode in
which several ranks share one GPU — RCCL refuses that — never of a real multi-GPU run.
"""
from __future__ import annotations

import numpy as np

from . import _native as N


def connect(ctx: N.Context, group=None) -> N.Dist:
    """One cph_dist per rank.  torch.distributed is used for exactly one thing: moving the 128-byte id."""
    import torch.distributed as dist

    if not dist.is_initialized():
        return N.Dist.create(ctx, N.Dist.unique_id(ctx), 0, 1)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    box = [None]
    if rank == 0:
        try:
            box = [N.Dist.unique_id(ctx)]
        except N.CphError as e:      # the other ranks are waiting in the broadcast: tell them instead of leaving them there
            box = [e]
    src = dist.get_global_rank(group, 0) if group is not None else 0
    dist.broadcast_object_list(box, src=src, group=group)
    if not isinstance(box[0], (bytes, bytearray)):
        raise N.CphError(N.CPH_ERR_HIP, f"rank 0 could not create the RCCL unique id: {box[0]}")
    return N.Dist.create(ctx, bytes(box[0]), rank, world)


def chain_allgather(d: N.Dist, res, device):
    """res: engine.ChainResult of this rank's row range -> ChainResult of all ranks' rows, emission order.
    Returns (ChainResult, counts)."""
    from .engine import ChainResult, device_view

    ch = res.keep[0]
    g = d.chain_allgather(ch)
    ptrs = g.data_ptrs
    if g.identity:
        stream, rows = None, ptrs
    else:
        stream, rows = device_view(ptrs[0], g.total, "<i8", g, device), ptrs[1:]
    out = ChainResult(stream, [device_view(p, g.total, "<i4", g, device) for p in rows], g.total, keep=(g,),
                      stream_base=g.stream_base if g.identity else 0)
    return out, g.counts


# ---- the same exchange over torch.distributed (gloo on CPU; debug) ----------------------------------------------
def allgatherv_many(ts, group=None, single_rank_shortcut: bool = True):
    """Concatenation over ranks (rank order) of 1-D tensors that all have the SAME length on a rank (the
    arrays of one join result).  One count exchange and one grouped send/recv batch for all of them.
    Returns ([gathered...], counts)."""
    import torch
    import torch.distributed as dist

    ts = list(ts)
    n_local = int(ts[0].numel())
    assert all(int(t.numel()) == n_local for t in ts)
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and single_rank_shortcut):
        return ts, [n_local]
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if dist.get_backend(group) == "gloo" and ts[0].is_cuda:   # debug path (several ranks sharing one GPU)
        outs, counts = allgatherv_many([t.cpu() for t in ts], group)
        return [o.to(ts[0].device) for o in outs], counts
    dev = ts[0].device
    counts_t = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts_t, torch.tensor([n_local], dtype=torch.int64, device=dev), group=group)
    counts = [int(c) for c in counts_t.tolist()]   # the one host wait of the exchange
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + c)
    outs = [torch.empty(offs[-1], dtype=t.dtype, device=dev) for t in ts]
    ts = [t.contiguous() for t in ts]
    if len(set(counts)) == 1:   # equal shards (e.g. every stream row joined): plain all-gathers
        if counts[0]:
            for o, t in zip(outs, ts):
                dist.all_gather_into_tensor(o, t, group=group)
        return outs, counts
    ops = []
    for o, t in zip(outs, ts):
        if counts[rank]:
            o[offs[rank]:offs[rank + 1]].copy_(t)
        for peer in range(world):
            if peer == rank:
                continue
            gpeer = dist.get_global_rank(group, peer) if group is not None else peer
            if counts[rank]:
                ops.append(dist.P2POp(dist.isend, t, gpeer, group=group))
            if counts[peer]:
                ops.append(dist.P2POp(dist.irecv, o[offs[peer]:offs[peer + 1]], gpeer, group=group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return outs, counts


def allgatherv(t, group=None, single_rank_shortcut: bool = True):
    """One tensor through allgatherv_many: (gathered, counts)."""
    outs, counts = allgatherv_many([t], group, single_rank_shortcut)
    return outs[0], counts


def sharded_chained_join(total_stream_rows: int, local_join, group=None, exchange: bool = True):
    """Runs `local_join(begin, end)` on this rank's row range and (optionally) allgathers the resulting tuples
    over torch.distributed.

    local_join(begin, end) -> (stream_row, a_row, b_row) 1-D tensors for stream rows [begin, end), with
    stream_row holding GLOBAL row numbers.  Returned: the three gathered tensors (emission order of the whole
    stream) and the per-rank counts."""
    import torch.distributed as dist

    from .engine import shard_range

    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    begin, end = shard_range(total_stream_rows, rank, world)
    s, a, b = local_join(begin, end)
    if not exchange or world == 1:
        return s, a, b, [int(s.numel())]
    (gs, ga, gb), counts = allgatherv_many([s, a, b], group)
    return gs, ga, gb, counts


# ---- the pipelined exchange of cph_dist_join_chain over torch.distributed (gloo on CPU) ---------------------------
ABSENT = -1   # 0xFFFFFFFF as int32: "this stream row did not join" in the dense form


def chunk_range(n: int, c: int, nchunks: int):
    """Rows [begin, end) of chunk c when n rows are cut into nchunks (csrc/dist.hip cuts every shard the same way)."""
    q, m = divmod(n, nchunks)
    b = q * c + min(c, m)
    return b, b + q + (1 if c < m else 0)


def packed_bits(limits):
    """CPH_DIST_PACKED's widths (csrc/dist.hip): step s carries values 0 .. limits[s] - 1; step 0 also the absent code limits[0]."""
    return [max(1, int(limits[0]).bit_length())] + [max(1, int(x - 1).bit_length() if x else 1) for x in limits[1:]]


def pack_rows(parts, limits):
    """The wire format of CPH_DIST_PACKED, restated with numpy: B = sum(packed_bits) bits per row, row i of a group of 64 rows in
    bits [i * B, (i + 1) * B) of the group's B little-endian 64-bit words, step 0 in a row's low bits, ABSENT in step 0 as limits[0];
    the last group is padded with zero rows and the word count rounded up to an even number."""
    bits = packed_bits(limits)
    B = sum(bits)
    assert B <= 64
    n = len(parts[0])
    groups = (n + 63) // 64
    v = np.zeros(groups * 64, dtype=np.uint64)
    shift = 0
    for s, (x, b) in enumerate(zip(parts, bits)):
        x = np.asarray(x).astype(np.int64) & 0xFFFFFFFF
        if s == 0:
            x = np.where(x == (ABSENT & 0xFFFFFFFF), int(limits[0]), x)
        v[:n] |= (x.astype(np.uint64) & np.uint64((1 << b) - 1)) << np.uint64(shift)
        shift += b
    nwords = (groups * B + 1) & ~1
    out = np.zeros(nwords, dtype=np.uint64)
    i = np.arange(groups * 64, dtype=np.int64)
    pos = (i % 64) * B
    word = (i // 64) * B + (pos >> 6)
    off = (pos & 63).astype(np.uint64)
    np.bitwise_or.at(out, word, v << off)
    spill = (pos & 63) + B > 64
    np.bitwise_or.at(out, word[spill] + 1, v[spill] >> (np.uint64(64) - off[spill]))
    return out


def unpack_rows(words, n, limits):
    """pack_rows' inverse: n rows -> one int32 array per step (ABSENT restored in step 0)."""
    bits = packed_bits(limits)
    B = sum(bits)
    words = np.asarray(words, dtype=np.uint64)
    i = np.arange(n, dtype=np.int64)
    pos = (i % 64) * B
    word = (i // 64) * B + (pos >> 6)
    off = (pos & 63).astype(np.uint64)
    v = words[word] >> off
    spill = (pos & 63) + B > 64
    v[spill] |= words[word[spill] + 1] << (np.uint64(64) - off[spill])
    if B < 64:
        v &= np.uint64((1 << B) - 1)
    out, shift = [], 0
    for s, b in enumerate(bits):
        x = ((v >> np.uint64(shift)) & np.uint64((1 << b) - 1)).astype(np.int64)
        if s == 0:
            x = np.where(x == int(limits[0]), ABSENT & 0xFFFFFFFF, x)
        out.append(x.astype(np.uint32).view(np.int32))
        shift += b
    return out


def pipelined_dense_exchange(shard_rows, dense_chunk, nsteps: int, nchunks: int, group=None, packed_limits=None):
    """shard_rows: the row counts of ALL ranks' shards (consecutive ranges in rank order).  dense_chunk(begin, end) -> nsteps
    int32 tensors of end - begin slots for THIS rank's shard rows [begin, end): slot i holds the build row of stream row i,
    ABSENT in array 0 where the row did not join.  Chunk c's sends are posted before chunk c + 1 is computed and awaited only
    at the end.  Returns (stream_row or None when every row of every rank joined, [build rows...], per-rank totals).
    packed_limits (one value range per step): the chunks travel in CPH_DIST_PACKED's format — one array of 64-bit words per chunk,
    unpacked by the receiver when all transfers are done."""
    import torch
    import torch.distributed as dist

    on = dist.is_initialized()
    world = dist.get_world_size(group) if on else 1
    rank = dist.get_rank(group) if on else 0
    rows = [int(x) for x in shard_rows]
    displs = [0]
    for x in rows:
        displs.append(displs[-1] + x)
    total = displs[-1]
    slots = [torch.empty(total, dtype=torch.int32) for _ in range(nsteps)]
    pending, inbox, keep_alive = [], [], []
    joined = 0
    for c in range(nchunks):
        b, e = chunk_range(rows[rank], c, nchunks)
        if e > b:
            part = dense_chunk(b, e)
            joined += int((part[0] != ABSENT).sum())
            for a in range(nsteps):
                slots[a][displs[rank] + b:displs[rank] + e].copy_(part[a])
        if world == 1:
            continue
        if packed_limits is not None:
            B = sum(packed_bits(packed_limits))
            nwords = lambda rows_: ((((rows_ + 63) // 64) * B) + 1) & ~1   # noqa: E731
            ops = []
            mine = torch.from_numpy(pack_rows([slots[a][displs[rank] + b:displs[rank] + e].numpy() for a in range(nsteps)], packed_limits).view(np.int64)) if e > b else None
            keep_alive.append(mine)
            for peer in range(world):
                if peer == rank:
                    continue
                gpeer = dist.get_global_rank(group, peer) if group is not None else peer
                if e > b:
                    ops.append(dist.P2POp(dist.isend, mine, gpeer, group=group))
                pb, pe = chunk_range(rows[peer], c, nchunks)
                if pe > pb:
                    buf = torch.empty(nwords(pe - pb), dtype=torch.int64)
                    inbox.append((peer, pb, pe, buf))
                    ops.append(dist.P2POp(dist.irecv, buf, gpeer, group=group))
            if ops:
                pending.extend(dist.batch_isend_irecv(ops))
            continue
        ops = []
        for a in range(nsteps):      # same order on every rank: array-major, then peer
            for peer in range(world):
                if peer == rank:
                    continue
                gpeer = dist.get_global_rank(group, peer) if group is not None else peer
                if e > b:
                    ops.append(dist.P2POp(dist.isend, slots[a][displs[rank] + b:displs[rank] + e], gpeer, group=group))
                pb, pe = chunk_range(rows[peer], c, nchunks)
                if pe > pb:
                    ops.append(dist.P2POp(dist.irecv, slots[a][displs[peer] + pb:displs[peer] + pe], gpeer, group=group))
        if ops:
            pending.extend(dist.batch_isend_irecv(ops))      # posted; chunk c + 1 is computed while they move
    for req in pending:
        req.wait()
    for peer, pb, pe, buf in inbox:
        for a, x in enumerate(unpack_rows(buf.numpy().view(np.uint64), pe - pb, packed_limits)):
            slots[a][displs[peer] + pb:displs[peer] + pe].copy_(torch.from_numpy(x))
    totals = [joined]
    if world > 1:
        t = torch.zeros(world, dtype=torch.int64)
        dist.all_gather_into_tensor(t, torch.tensor([joined], dtype=torch.int64), group=group)   # the one count exchange
        totals = [int(x) for x in t.tolist()]
    if sum(totals) == total:
        return None, slots, totals
    keep = slots[0] != ABSENT
    return torch.nonzero(keep).flatten(), [x[keep] for x in slots], totals


def build_side_estimate(build_rows: int, code_bytes: int, n_ranks: int, build_rows_per_s: float, link_gb_per_s: float = 50.0):
    """SURVEY.md §8e options for the build side, in milliseconds on the critical path of one rank:
    A `replicated`  every rank sorts the same table:            build_rows / build_rows_per_s
    B `broadcast`   rank 0 sorts, then ncclBroadcast of sorted codes + perm (code_bytes + 4 per row) — a ring/tree broadcast
                    is bound by ONE xGMI link:                   A + (code_bytes + 4) * build_rows / link
    B never wins on time (it adds the transfer to the same sort); it saves (N - 1) sorts' worth of energy and HBM traffic."""
    a = build_rows / build_rows_per_s * 1e3
    b = a + (code_bytes + 4) * build_rows / (link_gb_per_s * 1e9) * 1e3 if n_ranks > 1 else a
    return {"replicated_ms": a, "broadcast_ms": b, "choice": "replicated" if a <= b else "broadcast"}

"""Columnar driver above the C ABI: device-resident chained joins and the
row-range sharding of the probe side.

torch is plumbing here — device memory for the staged columns, the current HIP
stream, torch.distributed (RCCL) for the exchange.  All compute happens inside
libcsvplus_hip.so; there is no CPU fallback.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import _native as N
from .columns import StrCol


def shard_range(total_rows: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous row range [begin, end) of `rank`.  Rank-ordered concatenation of the
    per-rank match lists is then exactly the reference's emission order (stream order,
    csvplus.go:553-567)."""
    return (rank * total_rows) // world, ((rank + 1) * total_rows) // world


class _DevArray:
    """Zero-copy torch view of a library-owned device array (__cuda_array_interface__)."""

    def __init__(self, ptr: int, count: int, typestr: str, owner):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": typestr, "data": (ptr, False), "version": 2}
        self._owner = owner


def device_view(ptr: int, count: int, typestr: str, owner, device):
    import torch

    if count == 0 or not ptr:
        dt = {"<i8": torch.int64, "<i4": torch.int32}[typestr]
        return torch.empty(0, dtype=dt, device=device)
    t = torch.as_tensor(_DevArray(ptr, count, typestr, owner), device=device)
    t._cph_owner = owner   # keep the cph_matches alive as long as the view
    return t


@dataclass
class ChainResult:
    """Joined rows of stream JOIN a JOIN b as row-id triples, emission order.
    stream_row: int64 (global stream row), a_row / b_row: int32 bit patterns of uint32 row ids."""
    stream_row: "object"
    a_row: "object"
    b_row: "object"
    n: int
    keep: tuple = ()


class Engine:
    """One GPU: a cph_ctx bound to the torch device and running on torch's current stream."""

    def __init__(self, device: int = 0, use_torch_stream: bool = True):
        import torch

        if not torch.cuda.is_available():
            raise N.CphError(N.CPH_ERR_NO_DEVICE, "no GPU visible to torch; csvplus_amd has no CPU fallback")
        self.torch = torch
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.ctx = N.Context(device)
        if use_torch_stream:
            self.ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        self.ctx.close()

    # ---- IndexOn / UniqueIndexOn -------------------------------------------------------------
    def index_on(self, keycols, unique: bool = False) -> N.DeviceIndex:
        ix = N.DeviceIndex(self.ctx, keycols, unique=unique)
        if unique and ix.status == N.CPH_ERR_DUPLICATE:
            raise N.CphError(N.CPH_ERR_DUPLICATE, self.ctx.last_error())
        return ix

    # ---- Join ----------------------------------------------------------------------------------
    def join(self, index: N.DeviceIndex, probecols, probe_base: int = 0, want_pairs: bool = True) -> N.Matches:
        return index.probe(probecols, probe_base=probe_base, want_pairs=want_pairs, out_mem=N.CPH_MEM_DEVICE)

    def chained_join(self, index_a, key_a: StrCol, index_b, key_b: StrCol, probe_base: int = 0) -> ChainResult:
        """stream.Join(a, key_a).Join(b, key_b) where both keys are columns of the STREAM table
        (README.md:56: orders.Join(customers,"cust_id").Join(products,"prod_id")).  The second
        probe runs over exactly the rows the first join emitted (row selection on the device);
        mergeRows (csvplus.go:571-583) lets the stream's value win on a column-name collision,
        so key_b of a joined row is the stream row's value."""
        torch = self.torch
        m1 = index_a.probe([key_a], probe_base=probe_base, out_mem=N.CPH_MEM_DEVICE)
        p1 = m1.device_ptrs()
        m2 = index_b.probe([key_b], row_sel=(p1["probe_idx"], 64, m1.nmatches), sel_base=probe_base,
                           probe_base=0, out_mem=N.CPH_MEM_DEVICE)
        p2 = m2.device_ptrs()
        dev = self.device
        s1 = device_view(p1["probe_idx"], m1.nmatches, "<i8", m1, dev)
        a1 = device_view(p1["build_row"], m1.nmatches, "<i4", m1, dev)
        b2 = device_view(p2["build_row"], m2.nmatches, "<i4", m2, dev)
        if index_b.first_dup is None and m2.nmatches == m2.nprobe:
            # b has distinct keys (cnt <= 1) and nmatches == nprobe, so every row of join 1 matched
            # exactly once: composition is the identity
            return ChainResult(s1, a1, b2, m2.nmatches, keep=(m1, m2))
        sel = device_view(p2["probe_idx"], m2.nmatches, "<i8", m2, dev)
        return ChainResult(torch.index_select(s1, 0, sel), torch.index_select(a1, 0, sel), b2, m2.nmatches,
                           keep=(m1, m2))

"""Columnar driver above the C ABI: device-resident chained joins and the
row-range sharding of the probe side.

torch is plumbing here — device memory for the staged columns, the current HIP
stream, torch.distributed (RCCL) for the exchange.  All compute happens inside
libcsvplus_hip.so; there is no CPU fallback.
"""
from __future__ import annotations

from dataclasses import dataclass

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
    t._cph_owner = owner   # keep the library object alive as long as the view
    return t


@dataclass
class ChainResult:
    """Joined rows of stream JOIN a JOIN b ... as row-id tuples, emission order.
    stream_row: int64 (global stream row) or None when every stream row joined exactly once
    (then result row m is stream row stream_base + m); build_rows[k]: int32 bit patterns of
    uint32 row ids."""
    stream_row: "object"
    build_rows: list
    n: int
    keep: tuple = ()
    stream_base: int = 0

    def stream_rows(self):
        """stream_row materialised (torch.arange when implicit)."""
        if self.stream_row is not None:
            return self.stream_row
        import torch

        dev = self.build_rows[0].device if self.build_rows else "cpu"
        return torch.arange(self.stream_base, self.stream_base + self.n, dtype=torch.int64, device=dev)

    def release(self):
        for k in self.keep:
            k.release()
        self.keep = ()


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

    def index_on_many(self, specs, unique=False) -> list:
        """Several IndexOn / UniqueIndexOn calls as one batch (cph_index_build_many): specs = [keycols, ...]; unique: one flag for
        all of them or one per spec."""
        flags = list(unique) if isinstance(unique, (list, tuple)) else [bool(unique)] * len(specs)
        res = N.DeviceIndex.build_many(self.ctx, [(cols, u) for cols, u in zip(specs, flags)])
        if any(u and ix.status == N.CPH_ERR_DUPLICATE for u, ix in zip(flags, res)):
            msg = self.ctx.last_error()
            for ix in res:
                ix.close()
            raise N.CphError(N.CPH_ERR_DUPLICATE, msg)
        return res

    # ---- Join ----------------------------------------------------------------------------------
    def join(self, index: N.DeviceIndex, probecols, probe_base: int = 0, want_pairs: bool = True) -> N.Matches:
        return index.probe(probecols, probe_base=probe_base, want_pairs=want_pairs, out_mem=N.CPH_MEM_DEVICE)

    def chained_join(self, steps, probe_base: int = 0, positions: bool = False) -> ChainResult:
        """stream.Join(i0, k0).Join(i1, k1)...  with steps = [(index, [stream key columns]), ...]
        (README.md:56: orders.Join(customers,"cust_id").Join(products,"prod_id")).  Runs as
        cph_join_chain; results stay on the device as torch views."""
        ch = N.join_chain(self.ctx, [(ix, cols if isinstance(cols, (list, tuple)) else [cols]) for ix, cols in steps],
                          probe_base=probe_base, out_mem=N.CPH_MEM_DEVICE, positions=positions)
        p = ch.device_ptrs()
        dev = self.device
        stream = None if ch.identity else device_view(p["stream_row"], ch.nrows, "<i8", ch, dev)
        return ChainResult(stream, [device_view(q, ch.nrows, "<i4", ch, dev) for q in p["build_row"]], ch.nrows,
                           keep=(ch,), stream_base=probe_base)

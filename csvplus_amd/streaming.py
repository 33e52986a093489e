"""Pipelined Join of a host-resident stream (cph_stream_join_*): chunks are uploaded, probed and
downloaded on several HIP streams so that PCIe transfers overlap the kernels (BASELINE config 5)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N
from .columns import StrCol


class PinnedCol:
    """A host string column whose buffers live in pinned memory (cph_pinned_alloc), so that the
    pipeline's H2D copies are truly asynchronous."""

    def __init__(self, ctx: N.Context, col: StrCol):
        self.ctx = ctx
        self._ptrs = []
        data = np.ascontiguousarray(col.data)
        self.data = self._pinned_copy(data, extra=8)
        self.offsets = None if col.fixed_width else self._pinned_copy(np.ascontiguousarray(col.offsets))
        self.col = StrCol(self.data, self.offsets if self.offsets is not None else col.offsets, col.nrows,
                          col.offset_bits, N.CPH_MEM_HOST, fixed_width=col.fixed_width)

    def _pinned_copy(self, arr: np.ndarray, extra: int = 0) -> np.ndarray:
        p = C.c_void_p()
        self.ctx._check(self.ctx.lib.cph_pinned_alloc(self.ctx.handle, arr.nbytes + extra + 8, C.byref(p)))
        self._ptrs.append(p)
        buf = (C.c_uint8 * (arr.nbytes + extra)).from_address(p.value)
        out = np.frombuffer(buf, dtype=arr.dtype, count=arr.size)
        out[:] = arr
        return out

    def free(self):
        for p in self._ptrs:
            self.ctx.lib.cph_pinned_free(self.ctx.handle, p)
        self._ptrs = []


class PinnedArray:
    """nelem elements of `dtype` in pinned host memory (cph_pinned_alloc): e.g. the code arrays of HostEncoder."""

    def __init__(self, ctx: N.Context, nelem: int, dtype=np.uint32):
        self.ctx = ctx
        self.ptr = C.c_void_p()
        nbytes = int(nelem) * np.dtype(dtype).itemsize
        ctx._check(ctx.lib.cph_pinned_alloc(ctx.handle, nbytes + 64, C.byref(self.ptr)))
        buf = (C.c_uint8 * nbytes).from_address(self.ptr.value)
        self.array = np.frombuffer(buf, dtype=dtype, count=int(nelem))

    def free(self):
        if self.ptr:
            self.ctx.lib.cph_pinned_free(self.ctx.handle, self.ptr)
            self.ptr = None


class HostEncoder:
    """cph_host_encoder_*: the key codes of an index formed on the host by a pool of worker threads, so that a stream in host
    memory ships 4 bytes per row and step (StreamJoin.submit_codes) instead of its key strings.  Raises CphError
    (CPH_ERR_INVALID) for an index whose keys do not code in one word below 2^31."""

    def __init__(self, index, nthreads: int = 0):
        self.ctx = index.ctx
        self.lib = index.ctx.lib
        h = C.c_void_p()
        self.ctx._check(self.lib.cph_host_encoder_create(index.handle, int(nthreads), C.byref(h)))
        self.handle = h
        self.threads = int(self.lib.cph_host_encoder_threads(h))

    def run(self, cols, out: np.ndarray):
        """cols: the index's key columns for the chunk (host StrCols); out: uint32[nrows] (a PinnedArray's .array for overlap)."""
        arr = (N.cph_strcol * len(cols))()
        keep = []
        for i, c in enumerate(cols):
            sc, k = c.as_c()
            arr[i] = sc
            keep.append(k)
        assert out.dtype == np.uint32 and out.flags.c_contiguous and len(out) >= cols[0].nrows
        self.ctx._check(self.lib.cph_host_encoder_run(self.handle, arr, len(cols), out.ctypes.data))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.cph_host_encoder_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class StreamJoin:
    def __init__(self, ctx: N.Context, indexes, nslots: int = 3, ncols=None, positions: bool = False):
        """ncols=None: the fused-kernel pipeline (cph_stream_join_create: distinct keys, one key column per index).
        ncols=[columns of the stream per step]: any chain (cph_stream_join_create_general), results as pair lists
        unless the chain qualifies for the fused kernel."""
        self.ctx = ctx
        self.lib = ctx.lib
        self.indexes = list(indexes)
        arr = (C.c_void_p * len(self.indexes))(*[ix.handle for ix in self.indexes])
        h = C.c_void_p()
        if ncols is None:
            ctx._check(self.lib.cph_stream_join_create(ctx.handle, arr, len(self.indexes), nslots, C.byref(h)))
        else:
            assert len(ncols) == len(self.indexes)
            nc = (C.c_int32 * len(ncols))(*[int(x) for x in ncols])
            ctx._check(self.lib.cph_stream_join_create_general(ctx.handle, arr, nc, len(self.indexes), nslots, C.byref(h)))
        self.handle = h
        if positions:
            ctx._check(self.lib.cph_stream_join_set_positions(h, 1))
        self.positions = positions
        self.nslots = nslots
        self._keep = []
        ctx._children.add(self)

    def submit(self, step_cols, probe_base: int = 0):
        """step_cols: the steps' key columns one after the other (one host StrCol per step in the fused mode).  Raises CphError(CPH_ERR_INVALID) when every slot is in flight."""
        arr = (N.cph_strcol * len(step_cols))()
        keep = []
        for i, c in enumerate(step_cols):
            sc, k = c.as_c()
            arr[i] = sc
            keep.append(k)
        self.ctx._check(self.lib.cph_stream_join_submit(self.handle, arr, probe_base))
        self._keep.append(keep)

    def submit_codes(self, step_codes, nrows: int, probe_base: int = 0):
        """step_codes: one uint32 array of host-formed key codes per step (HostEncoder.run; pinned for real overlap), valid
        until the chunk was returned by next()."""
        ptrs = (C.c_void_p * len(step_codes))(*[int(a.ctypes.data) for a in step_codes])
        self.ctx._check(self.lib.cph_stream_join_submit_codes(self.handle, ptrs, int(nrows), int(probe_base)))
        self._keep.append(list(step_codes))

    @property
    def pending(self) -> int:
        return int(self.lib.cph_stream_join_pending(self.handle))

    def next(self, copy: bool = True) -> dict:
        """Waits for the oldest chunk: dict(probe_base, nrows, nmatches, bitmap(uint64), build_row[list of uint32])."""
        ch = N.cph_stream_chunk()
        self.ctx._check(self.lib.cph_stream_join_next(self.handle, C.byref(ch)))
        if self._keep:
            self._keep.pop(0)
        n = int(ch.nrows)
        if not ch.dense:   # pair lists (general chains): cph_chain's layout
            m = int(ch.nmatches)
            rows = [N._ptr_array(ch.build_row[k], m, np.uint32) if m else np.zeros(0, np.uint32) for k in range(int(ch.nsteps))]
            sr = N._ptr_array(ch.stream_row, m, np.uint64) if (m and ch.stream_row) else None
            if copy:
                rows = [r.copy() for r in rows]
                sr = sr.copy() if sr is not None else None
            if sr is None:
                sr = np.arange(int(ch.probe_base), int(ch.probe_base) + m, dtype=np.uint64)
            return {"probe_base": int(ch.probe_base), "nrows": n, "nmatches": m, "dense": False, "stream_row": sr,
                    "build_row": rows}
        words = (n + 1023) // 1024 * 16
        bm = N._ptr_array(ch.match_bitmap, words, np.uint64)
        rows = [N._ptr_array(ch.build_row[k], n, np.uint32) for k in range(int(ch.nsteps))]
        if copy:
            bm, rows = bm.copy(), [r.copy() for r in rows]
        return {"probe_base": int(ch.probe_base), "nrows": n, "nmatches": int(ch.nmatches), "dense": True, "bitmap": bm,
                "build_row": rows}

    def close(self):
        if getattr(self, "handle", None):
            self.lib.cph_stream_join_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def bitmap_to_rows(bitmap: np.ndarray, nrows: int) -> np.ndarray:
    """Row numbers (ascending) whose bit is set."""
    bits = np.unpackbits(bitmap.view(np.uint8), bitorder="little")[:nrows]
    return np.nonzero(bits)[0]

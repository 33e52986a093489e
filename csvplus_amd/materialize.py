"""The step after the path: gathering columns through the joined row ids and writing the
canonical CSV (cph_gather_rows / cph_csv_write; mergeRows csvplus.go:571-583, ToCsv :379-406)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N
from .columns import StrCol


def gather_rows(ctx: N.Context, col: StrCol, row_ids=None, id_base: int = 0, out_mem: int = N.CPH_MEM_HOST):
    """out[i] = col[row_ids[i] - id_base].  Host columns take numpy uint32/uint64 ids; device columns take
    (device_ptr, bits, count).  Returns a host StrCol (out_mem HOST) or a ColBuf handle (DEVICE)."""
    sc, keep = col.as_c()
    ptr, bits, n = C.c_void_p(0), 32, 0
    if row_ids is not None:
        if isinstance(row_ids, np.ndarray):
            if row_ids.dtype != np.uint64:
                row_ids = row_ids.astype(np.uint32)
            row_ids = np.ascontiguousarray(row_ids)
            ptr, bits, n = C.c_void_p(row_ids.ctypes.data), row_ids.dtype.itemsize * 8, len(row_ids)
        else:
            ptr, bits, n = C.c_void_p(row_ids[0]), int(row_ids[1]), int(row_ids[2])
    out = C.POINTER(N.cph_colbuf)()
    ctx._check(ctx.lib.cph_gather_rows(ctx.handle, C.byref(sc), ptr, bits, id_base, n, out_mem, C.byref(out)))
    del keep
    cb = ColBuf(ctx, out)
    if out_mem == N.CPH_MEM_HOST:
        res = cb.to_strcol()
        cb.release()
        return res
    return cb


def permute_col(ctx: N.Context, index, col: StrCol, out_mem: int = N.CPH_MEM_DEVICE):
    """cph_index_permute: `col` (a column of the table `index` was built over) in index order — out[p] = col[perm[p]] — so
    that the sorted positions a Join reports are row subscripts (csvplus.go:736: the reference keeps its index rows
    sorted).  Returns a ColBuf (DEVICE) or a host StrCol."""
    sc, keep = col.as_c()
    out = C.POINTER(N.cph_colbuf)()
    ctx._check(ctx.lib.cph_index_permute(ctx.handle, index.handle, C.byref(sc), out_mem, C.byref(out)))
    del keep
    cb = ColBuf(ctx, out)
    if out_mem == N.CPH_MEM_HOST:
        res = cb.to_strcol()
        cb.release()
        return res
    return cb


class ColBuf:
    def __init__(self, ctx, ptr):
        self.ctx, self.ptr = ctx, ptr
        c = ptr.contents
        self.nrows, self.nbytes, self.mem = int(c.col.nrows), int(c.nbytes), int(c.col.mem)
        ctx._children.add(self)

    def to_strcol(self) -> StrCol:
        assert self.mem == N.CPH_MEM_HOST
        c = self.ptr.contents.col
        offs = N._ptr_array(c.offsets, self.nrows + 1, np.uint64).copy()
        data = N._ptr_array(c.data, self.nbytes, np.uint8).copy()
        return StrCol(data, offs, self.nrows, 64)

    def as_device_strcol(self) -> StrCol:
        """Zero-copy device StrCol view (valid until release)."""
        assert self.mem == N.CPH_MEM_DEVICE
        c = self.ptr.contents.col

        class _Raw:   # minimal object with data_ptr()
            def __init__(self, p):
                self._p = int(p or 0)

            def data_ptr(self):
                return self._p

        return StrCol(_Raw(c.data), _Raw(c.offsets), self.nrows, 64, N.CPH_MEM_DEVICE, fixed_width=0)

    def release(self):
        if self.ptr:
            self.ctx.lib.cph_colbuf_release(self.ptr)
            self.ptr = None

    close = release

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class DeviceBytes:
    """cph_bytes kept in HBM (valid until release())."""

    def __init__(self, ctx, ptr):
        self.ctx, self.ptr = ctx, ptr
        self.size, self.data_ptr = int(ptr.contents.size), int(ptr.contents.data or 0)
        ctx._children.add(self)

    def __len__(self):
        return self.size

    def release(self):
        if self.ptr:
            self.ctx.lib.cph_bytes_release(self.ptr)
            self.ptr = None

    close = release

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def csv_write(ctx: N.Context, cols, header=None, out_mem: int = N.CPH_MEM_HOST, row_ids=None, nrows=None):
    """ToCsv: header (list of names or None) + rows of `cols`, Go csv.Writer format.
    row_ids (optional, one entry per column): None = the column's own rows, else the rows of that column feeding
    the output (numpy uint32/uint64 for host columns, (device_ptr, bits, count[, base]) for device columns) —
    Join(...).ToCsv(...) fused: mergeRows happens inside the writer (cph_csv_write_rows).
    Returns bytes (out_mem HOST) or a DeviceBytes handle (DEVICE)."""
    arr = (N.cph_strcol * len(cols))()
    keep = []
    for i, c in enumerate(cols):
        sc, k = c.as_c()
        arr[i] = sc
        keep.append(k)
    hv = None
    if header is not None:
        hv = (N.cph_strval * len(cols))()
        for i, h in enumerate(header):
            b = np.frombuffer(h.encode() if isinstance(h, str) else bytes(h), dtype=np.uint8)
            keep.append(b)
            hv[i].data = b.ctypes.data if len(b) else None
            hv[i].len = len(b)
    sel = None
    n = cols[0].nrows if nrows is None else int(nrows)
    if row_ids is not None:
        sel = (N.cph_rowsel * len(cols))()
        for i, ids in enumerate(row_ids):
            if ids is None:
                continue
            if isinstance(ids, np.ndarray):
                if ids.dtype != np.uint64:
                    ids = ids.astype(np.uint32)
                ids = np.ascontiguousarray(ids)
                keep.append(ids)
                sel[i].ids, sel[i].bits = ids.ctypes.data if len(ids) else None, ids.dtype.itemsize * 8
                cnt = len(ids)
            else:
                sel[i].ids, sel[i].bits, cnt = int(ids[0]) or None, int(ids[1]), int(ids[2])
                sel[i].base = int(ids[3]) if len(ids) > 3 else 0
            if nrows is None:
                n = cnt
    out = C.POINTER(N.cph_bytes)()
    ctx._check(ctx.lib.cph_csv_write_rows(ctx.handle, arr, sel, len(cols), n, hv, out_mem, C.byref(out)))
    if out_mem == N.CPH_MEM_DEVICE:
        return DeviceBytes(ctx, out)
    res = N._ptr_array(out.contents.data, int(out.contents.size), np.uint8).tobytes()
    ctx.lib.cph_bytes_release(out)
    return res

"""Arrow-style string columns: the SoA staging layout handed across the C ABI.

A column = `data` (uint8, concatenated raw value bytes) + `offsets` (uint32 or
uint64, nrows+1 entries).  A column whose values all have the same length is
marked `fixed_width`: the library then needs no offsets (one dependent load and
4-8 bytes of HBM traffic per row less).  Host columns are numpy arrays; device
columns are torch uint8 tensors (torch is only the device-memory allocator here).
"""
from __future__ import annotations

import numpy as np

from . import _native as N


def _detect_fixed_width(offsets: np.ndarray, nrows: int) -> int:
    if nrows == 0:
        return 0
    w = int(offsets[1]) - int(offsets[0])
    if w <= 0 or int(offsets[0]) != 0:
        return 0
    if int(offsets[nrows]) != w * nrows:
        return 0
    d = np.diff(offsets[: nrows + 1].astype(np.int64))
    return w if bool((d == w).all()) else 0


class StrCol:
    def __init__(self, data, offsets, nrows: int, offset_bits: int, mem: int = N.CPH_MEM_HOST, fixed_width=None):
        self.data = data
        self.offsets = offsets
        self.nrows = int(nrows)
        self.offset_bits = int(offset_bits)
        self.mem = mem
        if fixed_width is None:
            fixed_width = _detect_fixed_width(offsets, self.nrows) if mem == N.CPH_MEM_HOST else 0
        self.fixed_width = int(fixed_width)

    # ---- construction -------------------------------------------------------------
    @staticmethod
    def from_values(values, offset_bits: int = 32, fixed_width=None) -> "StrCol":
        """values: iterable of bytes/str (str is encoded as UTF-8, like Go strings)."""
        bs = [v.encode("utf-8") if isinstance(v, str) else bytes(v) for v in values]
        lens = np.fromiter((len(b) for b in bs), dtype=np.uint64, count=len(bs))
        offs = np.zeros(len(bs) + 1, dtype=np.uint64)
        np.cumsum(lens, out=offs[1:])
        data = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.empty(0, np.uint8)
        odt = np.uint32 if offset_bits == 32 else np.uint64
        if offset_bits == 32 and len(offs) and offs[-1] > 0xFFFFFFFF:
            raise ValueError("column too large for 32-bit offsets")
        return StrCol(data, offs.astype(odt), len(bs), offset_bits, fixed_width=fixed_width)

    @staticmethod
    def from_arrays(data: np.ndarray, offsets: np.ndarray, fixed_width=None) -> "StrCol":
        assert data.dtype == np.uint8 and offsets.dtype in (np.uint32, np.uint64)
        return StrCol(np.ascontiguousarray(data), np.ascontiguousarray(offsets), len(offsets) - 1,
                      offsets.dtype.itemsize * 8, fixed_width=fixed_width)

    def as_variable(self) -> "StrCol":
        """The same column handed over with offsets (fixed_width ignored)."""
        return StrCol(self.data, self.offsets, self.nrows, self.offset_bits, self.mem, fixed_width=0)

    # ---- access ---------------------------------------------------------------------
    def value(self, i: int) -> bytes:
        assert self.mem == N.CPH_MEM_HOST
        return self.data[int(self.offsets[i]):int(self.offsets[i + 1])].tobytes()

    def values(self):
        return [self.value(i) for i in range(self.nrows)]

    def slice(self, begin: int, end: int) -> "StrCol":
        """Row range [begin,end) sharing the data buffer (offsets keep absolute values)."""
        assert self.mem == N.CPH_MEM_HOST
        if self.fixed_width:
            w = self.fixed_width
            offs = (np.arange(end - begin + 1, dtype=np.uint64) * w).astype(self.offsets.dtype)
            return StrCol(self.data[begin * w:end * w], offs, end - begin, self.offset_bits, fixed_width=w)
        return StrCol(self.data, self.offsets[begin:end + 1], end - begin, self.offset_bits, fixed_width=0)

    def head(self, n: int) -> "StrCol":
        """The first n rows, sharing the buffers (host or device)."""
        assert 0 <= n <= self.nrows
        return StrCol(self.data, self.offsets, n, self.offset_bits, self.mem, fixed_width=self.fixed_width)

    def nbytes_values(self) -> int:
        if self.nrows == 0:
            return 0
        if self.fixed_width:
            return self.fixed_width * self.nrows
        if self.mem == N.CPH_MEM_HOST:
            return int(self.offsets[self.nrows]) - int(self.offsets[0])
        raise NotImplementedError

    def nbytes_offsets(self) -> int:
        """Offset bytes the library reads per pass over the column (0 for fixed-width columns)."""
        return 0 if self.fixed_width else (self.offset_bits // 8) * self.nrows

    # ---- device ---------------------------------------------------------------------
    def to_device(self, device="cuda:0") -> "StrCol":
        import torch

        assert self.mem == N.CPH_MEM_HOST
        # +8 slack: kernels read whole aligned 8-byte words around a value
        d = torch.empty(self.data.nbytes + 8, dtype=torch.uint8, device=device)
        if self.data.nbytes:
            d[: self.data.nbytes].copy_(torch.from_numpy(self.data))
        o = None
        if not self.fixed_width:
            ob = np.ascontiguousarray(self.offsets).view(np.uint8)
            o = torch.from_numpy(ob.copy()).to(device)
        return StrCol(d, o, self.nrows, self.offset_bits, N.CPH_MEM_DEVICE, fixed_width=self.fixed_width)

    def as_c(self):
        """(cph_strcol, keepalive)."""
        sc = N.cph_strcol()
        if self.mem == N.CPH_MEM_HOST:
            data = np.ascontiguousarray(self.data)
            offs = np.ascontiguousarray(self.offsets)
            sc.data = data.ctypes.data if data.size else None
            sc.offsets = offs.ctypes.data
            keep = (data, offs)
        else:
            sc.data = self.data.data_ptr()
            sc.offsets = self.offsets.data_ptr() if self.offsets is not None else None
            keep = (self.data, self.offsets)
        sc.nrows = self.nrows
        sc.offset_bits = self.offset_bits
        sc.mem = self.mem
        sc.fixed_width = self.fixed_width
        return sc, keep

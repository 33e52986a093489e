"""ctypes wrapper of oracle/csvplus_oracle.c (the CPU restatement of the reference).

TEST INFRASTRUCTURE ONLY — the checker, never the product path.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
LIB_PATH = _DIR / "_build" / "liboracle.so"
SORT_STABLE, SORT_GO_PDQSORT = 0, 1
UINT64_MAX = 0xFFFFFFFFFFFFFFFF


class orc_strcol(C.Structure):
    _fields_ = [("data", C.c_void_p), ("offsets", C.c_void_p), ("nrows", C.c_uint64), ("offset_bits", C.c_int32),
                ("mem", C.c_int32)]


class orc_strval(C.Structure):
    _fields_ = [("data", C.c_void_p), ("len", C.c_uint64)]


_lib = None


def build():
    """gcc-compiles the C restatement (no GPU needed)."""
    src = _DIR / "csvplus_oracle.c"
    if LIB_PATH.exists() and LIB_PATH.stat().st_mtime >= src.stat().st_mtime:
        return
    LIB_PATH.parent.mkdir(exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall", str(src), "-o",
                           str(LIB_PATH)])


def _load():
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            build()
        lib = C.CDLL(str(LIB_PATH))
        P = C.c_void_p
        lib.orc_index_build.restype = C.c_uint64
        lib.orc_index_build.argtypes = [C.POINTER(orc_strcol), C.c_int32, C.c_int32, P]
        lib.orc_first_dup.restype = C.c_uint64
        lib.orc_first_dup.argtypes = [C.POINTER(orc_strcol), C.c_int32, P]
        lib.orc_find.restype = None
        lib.orc_find.argtypes = [C.POINTER(orc_strcol), P, C.POINTER(orc_strval), C.c_int32, C.POINTER(C.c_uint64),
                                 C.POINTER(C.c_uint64)]
        lib.orc_has.restype = C.c_int32
        lib.orc_has.argtypes = [C.POINTER(orc_strcol), P, C.POINTER(orc_strval), C.c_int32]
        lib.orc_join.restype = C.c_uint64
        lib.orc_join.argtypes = [C.POINTER(orc_strcol), C.c_int32, P, C.POINTER(orc_strcol), C.c_int32, P, C.c_uint64,
                                 C.c_uint64, P, P, P, P, C.c_uint64]
        lib.orc_fnv1a64.restype = C.c_uint64
        lib.orc_fnv1a64.argtypes = [P, C.c_uint64, C.c_uint64]
        _lib = lib
    return _lib


def _cols(cols):
    arr = (orc_strcol * len(cols))()
    keep = []
    for i, c in enumerate(cols):
        data = np.ascontiguousarray(c.data)
        offs = np.ascontiguousarray(c.offsets)
        arr[i].data = data.ctypes.data if data.size else None
        arr[i].offsets = offs.ctypes.data
        arr[i].nrows = c.nrows
        arr[i].offset_bits = c.offset_bits
        arr[i].mem = 0
        keep.append((data, offs))
    return arr, keep


def _vals(values):
    arr = (orc_strval * max(1, len(values)))()
    keep = []
    for i, v in enumerate(values):
        b = np.frombuffer(v.encode() if isinstance(v, str) else bytes(v), dtype=np.uint8)
        keep.append(b)
        arr[i].data = b.ctypes.data if len(b) else None
        arr[i].len = len(b)
    return arr, keep


class OracleIndex:
    """createIndex / createUniqueIndex (csvplus.go:707-756) over SoA key columns."""

    def __init__(self, keycols, mode: int = SORT_STABLE):
        self.cols = list(keycols)
        self.n = self.cols[0].nrows
        self._arr, self._keep = _cols(self.cols)
        self.perm = np.empty(self.n, dtype=np.uint32)
        self.less_calls = int(_load().orc_index_build(self._arr, len(self.cols), mode, self.perm.ctypes.data))
        if self.less_calls == UINT64_MAX:
            raise RuntimeError("orc_index_build failed")

    def first_dup(self):
        r = int(_load().orc_first_dup(self._arr, len(self.cols), self.perm.ctypes.data))
        return None if r == UINT64_MAX else r

    def find(self, *values):
        v, keep = _vals(values)
        lo, hi = C.c_uint64(), C.c_uint64()
        _load().orc_find(self._arr, self.perm.ctypes.data, v, len(values), C.byref(lo), C.byref(hi))
        return int(lo.value), int(hi.value)

    def has(self, *values) -> bool:
        v, keep = _vals(values)
        return bool(_load().orc_has(self._arr, self.perm.ctypes.data, v, len(values)))

    def join(self, probecols, row_sel=None, probe_base: int = 0, want_pairs: bool = True):
        """Join (csvplus.go:545-569): returns dict(lo, cnt, probe_idx, build_row, nmatches)."""
        parr, pkeep = _cols(probecols)
        nprobe = len(row_sel) if row_sel is not None else probecols[0].nrows
        sel = None
        if row_sel is not None:
            sel = np.ascontiguousarray(row_sel, dtype=np.uint32)
        lo = np.empty(nprobe, dtype=np.uint32)
        cnt = np.empty(nprobe, dtype=np.uint32)
        lib = _load()
        total = int(lib.orc_join(self._arr, len(self.cols), self.perm.ctypes.data, parr, len(probecols),
                                 sel.ctypes.data if sel is not None else None, nprobe, probe_base, lo.ctypes.data,
                                 cnt.ctypes.data, None, None, 0))
        out = {"lo": lo, "cnt": cnt, "nmatches": total, "probe_idx": None, "build_row": None}
        if want_pairs:
            pidx = np.empty(total, dtype=np.uint64)
            brow = np.empty(total, dtype=np.uint32)
            lib.orc_join(self._arr, len(self.cols), self.perm.ctypes.data, parr, len(probecols),
                         sel.ctypes.data if sel is not None else None, nprobe, probe_base, None, None,
                         pidx.ctypes.data, brow.ctypes.data, total)
            out["probe_idx"], out["build_row"] = pidx, brow
        return out


def fnv1a64(arr: np.ndarray, h: int = 0) -> int:
    a = np.ascontiguousarray(arr)
    return int(_load().orc_fnv1a64(a.ctypes.data, a.nbytes, h))


def csv_write(cols, header=None) -> bytes:
    """ToCsv (csvplus.go:379-406): header + rows through Go's csv.Writer rules (C restatement)."""
    lib = _load()
    if not hasattr(lib, "_csv_ready"):
        lib.orc_csv_write.restype = C.c_uint64
        lib.orc_csv_write.argtypes = [C.POINTER(orc_strcol), C.c_int32, C.POINTER(orc_strval), C.c_void_p, C.c_uint64]
        lib._csv_ready = True
    arr, keep = _cols(cols)
    hv, hk = (_vals(header) if header is not None else (None, None))
    size = int(lib.orc_csv_write(arr, len(cols), hv, None, 0))
    buf = np.empty(size, dtype=np.uint8)
    lib.orc_csv_write(arr, len(cols), hv, buf.ctypes.data, size)
    return buf.tobytes()


class orc_csv_opts(C.Structure):
    _fields_ = [("comma", C.c_uint8), ("comment", C.c_uint8), ("trim_leading_space", C.c_int32),
                ("fields_per_record", C.c_int32), ("skip_records", C.c_uint64)]


def csv_parse(data: bytes, col_index, comma=b",", comment=None, trim_leading_space=False, fields_per_record=0,
              skip_records=0):
    """Go csv.Reader + csvplus column selection (C restatement).  Returns (columns as lists of bytes,
    err_kind, err_record)."""
    from csvplus_amd.columns import StrCol

    lib = _load()
    if not hasattr(lib, "_csvp_ready"):
        lib.orc_csv_parse.restype = C.c_uint64
        lib.orc_csv_parse.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(orc_csv_opts), C.POINTER(C.c_int32), C.c_int32,
                                      C.POINTER(C.c_uint64), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                      C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]
        lib._csvp_ready = True
    buf = np.frombuffer(data, dtype=np.uint8)
    o = orc_csv_opts(comma[0], comment[0] if comment else 0, 1 if trim_leading_space else 0, fields_per_record, skip_records)
    nc = len(col_index)
    idx = (C.c_int32 * nc)(*col_index)
    nbytes = (C.c_uint64 * nc)()
    ek, er = C.c_int32(), C.c_uint64()
    p = buf.ctypes.data if len(buf) else None
    n = int(lib.orc_csv_parse(p, len(buf), C.byref(o), idx, nc, nbytes, None, None, C.byref(ek), C.byref(er)))
    # the record that fails is written before it is rolled back: leave room for one whole input per column
    datas = [np.empty(int(nbytes[c]) + len(buf) + 1, dtype=np.uint8) for c in range(nc)]
    offs = [np.empty(n + 1, dtype=np.uint64) for _ in range(nc)]
    dp = (C.c_void_p * nc)(*[d.ctypes.data for d in datas])
    op = (C.c_void_p * nc)(*[x.ctypes.data for x in offs])
    lib.orc_csv_parse(p, len(buf), C.byref(o), idx, nc, nbytes, dp, op, C.byref(ek), C.byref(er))
    cols = [StrCol(datas[c][: int(nbytes[c])], offs[c], n, 64) for c in range(nc)]
    return cols, int(ek.value), int(er.value)


# ---------------------------------------------------------------------------------------------------------------
# dedup (csvplus.go:810-867), restated line by line over a list of row dicts (small cases only)
# ---------------------------------------------------------------------------------------------------------------
def dedup_rows(rows, columns, resolve):
    """rows: list of dicts sorted on `columns` (the index's rows).  resolve(list_of_rows) -> row dict | {} | raises.
    Returns the new row list (index.rows[:dest], :863).  Follows the reference literally, tail rule included."""
    rows = list(rows)

    def equal_rows(a, b):                       # equalRows :759-767
        return all(a.get(c, "") == b.get(c, "") for c in columns)

    def cmp_gt(i, values):                      # cmp(i, values, false) :907-920
        for c, v in zip(columns, values):
            x = rows[i].get(c, "")
            xb, vb = (x.encode() if isinstance(x, str) else x), (v.encode() if isinstance(v, str) else v)
            if xb != vb:
                return xb > vb
        return False

    n = len(rows)
    lower = 1
    while lower < n:                            # :815-819
        if equal_rows(rows[lower - 1], rows[lower]):
            break
        lower += 1
    if lower >= n:                              # :821-823
        return rows
    dest = lower - 1
    while lower < n:                            # :828
        values = [rows[lower].get(c, "") for c in columns]
        lo_i, hi_i = 0, n - lower               # sort.Search :831-833
        while lo_i < hi_i:
            h = (lo_i + hi_i) >> 1
            if not cmp_gt(lower + h, values):
                lo_i = h + 1
            else:
                hi_i = h
        upper = lower + lo_i
        row = resolve(rows[lower - 1:upper])    # :838
        lower = upper + 1                       # :842
        if len(row) >= len(columns):            # :845-848
            rows[dest] = row
            dest += 1
        while lower < n:                        # :851-859
            if equal_rows(rows[lower - 1], rows[lower]):
                break
            rows[dest] = rows[lower - 1]
            lower += 1
            dest += 1
    return rows[:dest]                          # :862-864


# ---------------------------------------------------------------------------------------------------------------
# CPU baselines with the reference's cost model (oracle/faithful.cpp): bench.py's cpu_baseline leg only
# ---------------------------------------------------------------------------------------------------------------
_fth = None


def _load_faithful():
    global _fth
    if _fth is None:
        path = Path(__file__).resolve().parent / "_build" / "libfaithful.so"
        if not path.exists():
            raise ImportError(f"{path} not found: run `make oracle`")
        lib = C.CDLL(str(path))
        lib.fth_chain_join.restype = C.c_uint64
        lib.fth_lean_mt_join.restype = C.c_uint64
        lib.fth_max_threads.restype = C.c_int
        _fth = lib
    return _fth


def faithful_chain_join(cust: dict, cust_key: str, prod: dict, prod_key: str, ords: dict, ord_cust: str, ord_prod: str):
    """Map-per-row restatement of orders.Join(customers, cust).Join(products, prod) (single thread).
    Tables are {name: StrCol}.  Returns dict(joined, checksum, rows_s, index_s, join_s)."""
    lib = _load_faithful()

    def pack(tab):
        names = list(tab.keys())
        arr, keep = _cols([tab[n] for n in names])
        cn = (C.c_char_p * len(names))(*[n.encode() for n in names])
        return arr, cn, len(names), keep

    ca, cn, cc, k1 = pack(cust)
    pa, pn, pc, k2 = pack(prod)
    oa, on, oc, k3 = pack(ords)
    times = (C.c_double * 3)()
    chk = C.c_uint64()
    joined = int(lib.fth_chain_join(ca, cn, cc, cust_key.encode(), pa, pn, pc, prod_key.encode(), oa, on, oc, ord_cust.encode(),
                                    ord_prod.encode(), times, C.byref(chk)))
    del k1, k2, k3
    return {"joined": joined, "checksum": int(chk.value), "rows_s": times[0], "index_s": times[1], "join_s": times[2]}


def lean_mt_join(build, probe, threads: int = 0):
    """Lean SoA unique join on `threads` cores (0 = all).  Returns (build_row uint32[m], joined, sort_s, probe_s, threads)."""
    lib = _load_faithful()
    ba, k1 = _cols([build])
    pa, k2 = _cols([probe])
    out = np.empty(probe.nrows, dtype=np.uint32)
    times = (C.c_double * 2)()
    joined = int(lib.fth_lean_mt_join(ba, pa, int(threads), out.ctypes.data_as(C.c_void_p), times))
    del k1, k2
    return out, joined, times[0], times[1], (threads or int(lib.fth_max_threads()))

"""ctypes binding of the C ABI declared in include/csvplus_hip.h.

This is the same boundary a cgo shim would bind.  There is no
CPU implementation behind it: when libcsvplus_hip.so is missing or no GPU is
present every call fails loudly (NativeLibraryMissing / CphError).
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "lib" / "libcsvplus_hip.so"

# status codes (include/csvplus_hip.h)
CPH_OK = 0
CPH_ERR_INVALID = -1
CPH_ERR_HIP = -2
CPH_ERR_NO_DEVICE = -3
CPH_ERR_DUPLICATE = -4
CPH_ERR_TOO_MANY_ROWS = -5
CPH_ERR_TOO_MANY_COLS = -7
CPH_ERR_NOMEM = -8
CPH_MEM_HOST = 0
CPH_MEM_DEVICE = 1
UINT64_MAX = 0xFFFFFFFFFFFFFFFF

_STATUS_NAMES = {
    CPH_ERR_INVALID: "CPH_ERR_INVALID",
    CPH_ERR_HIP: "CPH_ERR_HIP",
    CPH_ERR_NO_DEVICE: "CPH_ERR_NO_DEVICE",
    CPH_ERR_DUPLICATE: "CPH_ERR_DUPLICATE",
    CPH_ERR_TOO_MANY_ROWS: "CPH_ERR_TOO_MANY_ROWS",
    CPH_ERR_TOO_MANY_COLS: "CPH_ERR_TOO_MANY_COLS",
    CPH_ERR_NOMEM: "CPH_ERR_NOMEM",
}


class NativeLibraryMissing(ImportError):
    pass


class CphError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{_STATUS_NAMES.get(code, code)}: {msg}")
        self.code = code
        self.msg = msg


class cph_strcol(C.Structure):
    _fields_ = [
        ("data", C.c_void_p),
        ("offsets", C.c_void_p),
        ("nrows", C.c_uint64),
        ("offset_bits", C.c_int32),
        ("mem", C.c_int32),
        ("fixed_width", C.c_uint32),
        ("reserved_", C.c_uint32),
    ]


class cph_strval(C.Structure):
    _fields_ = [("data", C.c_void_p), ("len", C.c_uint64)]


class cph_matches(C.Structure):
    _fields_ = [
        ("nprobe", C.c_uint64),
        ("nmatches", C.c_uint64),
        ("lo", C.c_void_p),
        ("cnt", C.c_void_p),
        ("probe_idx", C.c_void_p),
        ("build_row", C.c_void_p),
        ("mem", C.c_int32),
        ("reserved_", C.c_int32),
    ]


class cph_index_info(C.Structure):
    _fields_ = [
        ("nrows", C.c_uint64),
        ("nkeycols", C.c_int32),
        ("key_positions", C.c_int32),
        ("code_words", C.c_int32),
        ("code_bits", C.c_int32),
        ("key_bytes", C.c_int32),
        ("sort_passes", C.c_int32),
        ("direct_table", C.c_int32),
        ("dict_entries", C.c_int32),
        ("table_entries", C.c_uint64),
        ("lookup_built", C.c_int32),
        ("hash_mode", C.c_int32),
        ("hash_bytes", C.c_uint64),
        ("build_path", C.c_int32),
        ("split", C.c_int32),
    ]


CPH_MAX_CHAIN = 8


class cph_index_spec(C.Structure):
    _fields_ = [("keycols", C.POINTER(cph_strcol)), ("nkeycols", C.c_int32), ("unique", C.c_int32)]


class cph_chain_step(C.Structure):
    _fields_ = [("index", C.c_void_p), ("cols", C.POINTER(cph_strcol)), ("ncols", C.c_int32), ("source", C.c_int32)]


class cph_chain(C.Structure):
    _fields_ = [("nrows", C.c_uint64), ("stream_row", C.c_void_p), ("build_row", C.c_void_p * CPH_MAX_CHAIN),
                ("nsteps", C.c_int32), ("mem", C.c_int32), ("positions", C.c_int32), ("reserved_", C.c_int32)]


CPH_CHAIN_POSITIONS = 1
CPH_DIST_HOST_GATHER = 0x100
CPH_DIST_PACKED = 0x200


class cph_colbuf(C.Structure):
    _fields_ = [("col", cph_strcol), ("nbytes", C.c_uint64)]


class cph_bytes(C.Structure):
    _fields_ = [("data", C.c_void_p), ("size", C.c_uint64), ("mem", C.c_int32), ("reserved_", C.c_int32)]


CPH_CSV_ERR_BARE_QUOTE, CPH_CSV_ERR_QUOTE, CPH_CSV_ERR_FIELD_COUNT = 1, 2, 3
CPH_MAX_KEY_COLS = 16


class cph_csv_options(C.Structure):
    _fields_ = [("comma", C.c_uint8), ("comment", C.c_uint8), ("trim_leading_space", C.c_uint8),
                ("lazy_quotes", C.c_uint8), ("fields_per_record", C.c_int32), ("skip_records", C.c_uint64)]


class cph_csv_table(C.Structure):
    _fields_ = [("nrecords", C.c_uint64), ("error_record", C.c_uint64), ("error_kind", C.c_int32),
                ("ncols", C.c_int32), ("cols", cph_strcol * CPH_MAX_KEY_COLS)]


class cph_rowsel(C.Structure):
    _fields_ = [("ids", C.c_void_p), ("bits", C.c_int32), ("reserved_", C.c_int32), ("base", C.c_uint64)]


class cph_groups(C.Structure):
    _fields_ = [("ngroups", C.c_uint64), ("lower", C.c_void_p), ("upper", C.c_void_p)]


CPH_DIST_ID_BYTES = 128
CPH_MAX_GATHER = 8


class cph_gathered(C.Structure):
    _fields_ = [("total", C.c_uint64), ("narrays", C.c_int32), ("nranks", C.c_int32),
                ("counts", C.POINTER(C.c_uint64)), ("displs", C.POINTER(C.c_uint64)),
                ("data", C.c_void_p * CPH_MAX_GATHER), ("mem", C.c_int32), ("reserved_", C.c_int32)]


class cph_dist_join_stats(C.Structure):
    _fields_ = [("chunks", C.c_int32), ("pipelined", C.c_int32), ("compute_ms", C.c_double), ("exchange_ms", C.c_double),
                ("exposed_exchange_ms", C.c_double), ("total_ms", C.c_double), ("bytes_sent", C.c_uint64),
                ("bytes_received", C.c_uint64), ("packed_bits", C.c_int32), ("reserved_", C.c_int32)]


class cph_stream_chunk(C.Structure):
    _fields_ = [("probe_base", C.c_uint64), ("nrows", C.c_uint64), ("nmatches", C.c_uint64),
                ("match_bitmap", C.c_void_p), ("build_row", C.c_void_p * CPH_MAX_CHAIN), ("nsteps", C.c_int32),
                ("dense", C.c_int32), ("stream_row", C.c_void_p), ("positions", C.c_int32), ("reserved_", C.c_int32)]


class cph_kernel_stat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint64), ("total_ms", C.c_double),
                ("algo_bytes", C.c_double)]


# every symbol include/csvplus_hip.h declares: (name, restype, argtypes)
_P = C.c_void_p
PROTOTYPES = [
    ("cph_version", C.c_char_p, []),
    ("cph_ctx_create", C.c_int32, [C.c_int32, C.POINTER(_P)]),
    ("cph_ctx_destroy", None, [_P]),
    ("cph_last_error", C.c_char_p, [_P]),
    ("cph_ctx_set_stream", C.c_int32, [_P, _P]),
    ("cph_ctx_set_option", C.c_int32, [_P, C.c_char_p, C.c_int64]),
    ("cph_ctx_synchronize", C.c_int32, [_P]),
    ("cph_ctx_profile", C.c_int32, [_P, C.c_int32]),
    ("cph_ctx_profile_only", C.c_int32, [_P, C.c_char_p]),
    ("cph_ctx_profile_read", C.c_int32,
     [_P, C.POINTER(cph_kernel_stat), C.c_int32, C.POINTER(C.c_int32), C.c_int32]),
    ("cph_pinned_alloc", C.c_int32, [_P, C.c_size_t, C.POINTER(_P)]),
    ("cph_pinned_free", C.c_int32, [_P, _P]),
    ("cph_index_build", C.c_int32,
     [_P, C.POINTER(cph_strcol), C.c_int32, C.c_int32, C.POINTER(_P), C.POINTER(C.c_uint64)]),
    ("cph_index_build_many", C.c_int32,
     [_P, C.POINTER(cph_index_spec), C.c_int32, C.POINTER(_P), C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]),
    ("cph_index_destroy", None, [_P]),
    ("cph_dist_unique_id", C.c_int32, [_P, C.POINTER(C.c_uint8)]),
    ("cph_dist_create", C.c_int32, [_P, C.POINTER(C.c_uint8), C.c_int32, C.c_int32, C.POINTER(_P)]),
    ("cph_dist_create_loopback", C.c_int32, [_P, C.c_char_p, C.c_int32, C.c_int32, C.POINTER(_P)]),
    ("cph_dist_destroy", None, [_P]),
    ("cph_dist_rank", C.c_int32, [_P]),
    ("cph_dist_size", C.c_int32, [_P]),
    ("cph_dist_transport", C.c_char_p, [_P]),
    ("cph_dist_allgatherv", C.c_int32,
     [_P, C.POINTER(_P), C.POINTER(C.c_int32), C.c_int32, C.c_uint64, C.POINTER(C.POINTER(cph_gathered))]),
    ("cph_gathered_release", None, [C.POINTER(cph_gathered)]),
    ("cph_dist_chain_allgather", C.c_int32,
     [_P, C.POINTER(cph_chain), C.c_uint64, C.POINTER(C.POINTER(cph_gathered)), C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]),
    ("cph_dist_join_chain", C.c_int32,
     [_P, C.POINTER(cph_chain_step), C.c_int32, C.c_uint64, C.POINTER(C.c_uint64), C.c_int32, C.c_uint32,
      C.POINTER(C.POINTER(cph_gathered)), C.POINTER(C.c_int32), C.POINTER(C.c_uint64), C.POINTER(cph_dist_join_stats)]),
    ("cph_dist_index_broadcast", C.c_int32, [_P, _P, C.c_int32, C.POINTER(_P)]),
    ("cph_index_nrows", C.c_uint64, [_P]),
    ("cph_index_nkeycols", C.c_int32, [_P]),
    ("cph_index_perm", C.c_int32, [_P, C.c_int32, C.POINTER(_P), C.POINTER(C.c_uint64)]),
    ("cph_index_permute", C.c_int32, [_P, _P, C.POINTER(cph_strcol), C.c_int32, C.POINTER(C.POINTER(cph_colbuf))]),
    ("cph_join_probe", C.c_int32,
     [_P, _P, C.POINTER(cph_strcol), C.c_int32, _P, C.c_int32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int32,
      C.c_int32, C.POINTER(C.POINTER(cph_matches))]),
    ("cph_matches_release", None, [C.POINTER(cph_matches)]),
    ("cph_join_chain", C.c_int32,
     [_P, C.POINTER(cph_chain_step), C.c_int32, C.c_uint64, C.c_int32, C.POINTER(C.POINTER(cph_chain))]),
    ("cph_join_chain_ex", C.c_int32,
     [_P, C.POINTER(cph_chain_step), C.c_int32, C.c_uint64, C.c_int32, C.c_uint32, C.POINTER(C.POINTER(cph_chain))]),
    ("cph_chain_release", None, [C.POINTER(cph_chain)]),
    ("cph_stream_join_create", C.c_int32, [_P, C.POINTER(_P), C.c_int32, C.c_int32, C.POINTER(_P)]),
    ("cph_stream_join_create_general", C.c_int32, [_P, C.POINTER(_P), C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.POINTER(_P)]),
    ("cph_stream_join_set_positions", C.c_int32, [_P, C.c_int32]),
    ("cph_stream_join_destroy", None, [_P]),
    ("cph_stream_join_submit", C.c_int32, [_P, C.POINTER(cph_strcol), C.c_uint64]),
    ("cph_stream_join_submit_codes", C.c_int32, [_P, C.POINTER(C.c_void_p), C.c_uint64, C.c_uint64]),
    ("cph_host_encoder_create", C.c_int32, [_P, C.c_int32, C.POINTER(_P)]),
    ("cph_host_encoder_threads", C.c_int32, [_P]),
    ("cph_host_encoder_run", C.c_int32, [_P, C.POINTER(cph_strcol), C.c_int32, C.c_void_p]),
    ("cph_host_encoder_destroy", None, [_P]),
    ("cph_stream_join_pending", C.c_int32, [_P]),
    ("cph_stream_join_next", C.c_int32, [_P, C.POINTER(cph_stream_chunk)]),
    ("cph_gather_rows", C.c_int32,
     [_P, C.POINTER(cph_strcol), _P, C.c_int32, C.c_uint64, C.c_uint64, C.c_int32, C.POINTER(C.POINTER(cph_colbuf))]),
    ("cph_colbuf_release", None, [C.POINTER(cph_colbuf)]),
    ("cph_csv_write", C.c_int32,
     [_P, C.POINTER(cph_strcol), C.c_int32, C.POINTER(cph_strval), C.c_int32, C.POINTER(C.POINTER(cph_bytes))]),
    ("cph_csv_write_rows", C.c_int32,
     [_P, C.POINTER(cph_strcol), C.POINTER(cph_rowsel), C.c_int32, C.c_uint64, C.POINTER(cph_strval), C.c_int32,
      C.POINTER(C.POINTER(cph_bytes))]),
    ("cph_bytes_release", None, [C.POINTER(cph_bytes)]),
    ("cph_csv_parse", C.c_int32,
     [_P, _P, C.c_uint64, C.c_int32, C.POINTER(cph_csv_options), C.POINTER(C.c_int32), C.c_int32, C.c_int32,
      C.POINTER(C.POINTER(cph_csv_table))]),
    ("cph_csv_table_release", None, [C.POINTER(cph_csv_table)]),
    ("cph_index_dup_groups", C.c_int32, [_P, _P, C.POINTER(C.POINTER(cph_groups))]),
    ("cph_groups_release", None, [C.POINTER(cph_groups)]),
    ("cph_index_select", C.c_int32, [_P, _P, _P, C.c_uint64, C.POINTER(_P)]),
    ("cph_index_save", C.c_int32, [_P, _P, C.c_char_p]),
    ("cph_index_load", C.c_int32, [_P, C.c_char_p, C.POINTER(_P)]),
    ("cph_index_find", C.c_int32,
     [_P, _P, C.POINTER(cph_strval), C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("cph_index_find_many", C.c_int32,
     [_P, _P, C.POINTER(cph_strval), C.c_int32, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("cph_index_get_info", C.c_int32, [_P, C.POINTER(cph_index_info)]),
    ("cph_index_prepare_join", C.c_int32, [_P, C.c_int32]),
    ("cph_calibrate", C.c_int32, [_P, C.c_int32, C.c_uint64, C.c_uint64, C.c_int32, C.POINTER(C.c_double)]),
]

_lib = None


def load_library() -> C.CDLL:
    """Loads libcsvplus_hip.so (built by `make hip` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("CSVPLUS_HIP_LIB", LIB_PATH))
    if not path.exists():
        raise NativeLibraryMissing(
            f"{path} not found: build it with `make hip` (or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "csvplus_amd has no CPU fallback.")
    # One HIP runtime per process: torch ships its own libamdhip64.so (same SONAME as
    # /opt/rocm's).  If ours were loaded first, torch would later bind to the ROCm-tree copy
    # and fail with "No HIP GPUs are available".  Importing torch first makes our DT_NEEDED
    # entry resolve to the copy torch already loaded, so device pointers can be shared.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(str(path))
    for name, restype, argtypes in PROTOTYPES:
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def _ptr_array(ptr: int, n: int, dtype) -> np.ndarray:
    """numpy view (no copy) over `n` items at host address `ptr`."""
    if n == 0 or not ptr:
        return np.empty(0, dtype=dtype)
    nbytes = n * np.dtype(dtype).itemsize
    buf = (C.c_uint8 * nbytes).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n)


class Context:
    """One cph_ctx: single-threaded, bound to one GPU."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        h = _P()
        rc = self.lib.cph_ctx_create(device, C.byref(h))
        if rc != CPH_OK:
            raise CphError(rc, f"cph_ctx_create(device={device}) failed: no usable GPU "
                               "(csvplus_amd has no CPU fallback)")
        self.handle = h
        self.device = device
        self._children = weakref.WeakSet()   # indexes / matches: they borrow the ctx's device pool

    def close(self):
        if getattr(self, "handle", None):
            for child in list(self._children):   # library rule: release indexes and matches before the ctx
                child.close()
            self.lib.cph_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != CPH_OK:
            raise CphError(rc, (self.lib.cph_last_error(self.handle) or b"").decode("utf-8", "replace"))

    def set_stream(self, stream_handle: int | None):
        self._check(self.lib.cph_ctx_set_stream(self.handle, _P(stream_handle or 0)))

    def set_option(self, name: str, value: int):
        """Measurement / tuning knobs of this ctx (include/csvplus_hip.h: cph_ctx_set_option)."""
        self._check(self.lib.cph_ctx_set_option(self.handle, name.encode(), int(value)))

    def set_debug(self, chain_flags: int):
        """Attribution switches of the chained-join kernel (measurement only: results are wrong when set)."""
        self.set_option("chain_debug", chain_flags)

    def synchronize(self):
        self._check(self.lib.cph_ctx_synchronize(self.handle))

    def last_error(self) -> str:
        return (self.lib.cph_last_error(self.handle) or b"").decode("utf-8", "replace")

    def calibrate(self, kind: str, nbytes: int, n: int = 0, reps: int = 5) -> float:
        """ms per launch of the box calibration kernels (cph_calibrate): kind "copy" or "gather"."""
        ms = C.c_double()
        self._check(self.lib.cph_calibrate(self.handle, {"copy": 0, "gather": 1}[kind], nbytes, n, reps, C.byref(ms)))
        return float(ms.value)

    def profile(self, enable: bool = True):
        self._check(self.lib.cph_ctx_profile(self.handle, 1 if enable else 0))

    def profile_only(self, kernel_name):
        """Times only the launches of one kernel (None: off): cph_ctx_profile_only."""
        self._check(self.lib.cph_ctx_profile_only(self.handle, kernel_name.encode() if kernel_name else None))

    def profile_read(self, reset: bool = True) -> dict:
        """{kernel name: {launches, total_ms, algo_bytes}} measured with HIP events on the ctx stream."""
        cap = 64
        arr = (cph_kernel_stat * cap)()
        n = C.c_int32()
        self._check(self.lib.cph_ctx_profile_read(self.handle, arr, cap, C.byref(n), 1 if reset else 0))
        return {arr[i].name.decode(): {"launches": int(arr[i].launches), "total_ms": float(arr[i].total_ms),
                                       "algo_bytes": float(arr[i].algo_bytes)} for i in range(min(cap, n.value))}


def _cols_array(cols):
    arr = (cph_strcol * len(cols))()
    keep = []
    for i, c in enumerate(cols):
        sc, k = c.as_c()
        arr[i] = sc
        keep.append(k)
    return arr, keep


class DeviceIndex:
    """Handle on a GPU-resident index (cph_index)."""

    def __init__(self, ctx: Context, keycols, unique: bool = False):
        self.ctx = ctx
        self.lib = ctx.lib
        arr, keep = _cols_array(keycols)
        h = _P()
        dup = C.c_uint64(UINT64_MAX)
        rc = self.lib.cph_index_build(ctx.handle, arr, len(keycols), 1 if unique else 0, C.byref(h), C.byref(dup))
        del keep
        self.handle = h if h.value else None
        ctx._children.add(self)
        self.first_dup = None if dup.value == UINT64_MAX else int(dup.value)
        self.status = rc
        if rc not in (CPH_OK, CPH_ERR_DUPLICATE):
            ctx._check(rc)

    @staticmethod
    def build_many(ctx: "Context", specs) -> list:
        """specs: [(keycols, unique), ...] -> [DeviceIndex, ...] through cph_index_build_many (one batch: the
        builds share their two host round trips).  Raises for any status other than OK / DUPLICATE; a unique
        spec with equal keys comes back with .status == CPH_ERR_DUPLICATE like the single build."""
        k = len(specs)
        arr = (cph_index_spec * k)()
        keep = []
        for i, (cols, unique) in enumerate(specs):
            ca, kp = _cols_array(cols)
            keep.append((ca, kp))
            arr[i].keycols = ca
            arr[i].nkeycols = len(cols)
            arr[i].unique = 1 if unique else 0
        outs = (_P * k)()
        dups = (C.c_uint64 * k)()
        sts = (C.c_int32 * k)()
        rc = ctx.lib.cph_index_build_many(ctx.handle, arr, k, outs, dups, sts)
        del keep
        res = []
        for i in range(k):
            ix = DeviceIndex.__new__(DeviceIndex)
            ix.ctx, ix.lib = ctx, ctx.lib
            ix.handle = _P(outs[i]) if outs[i] else None
            ix.first_dup = None if dups[i] == UINT64_MAX else int(dups[i])
            ix.status = int(sts[i])
            ctx._children.add(ix)
            res.append(ix)
        bad = [r for r in res if r.status not in (CPH_OK, CPH_ERR_DUPLICATE)]
        if bad or (rc not in (CPH_OK, CPH_ERR_DUPLICATE)):
            msg = ctx.last_error()
            for r in res:
                r.close()
            raise CphError(bad[0].status if bad else rc, msg)
        return res

    @property
    def nrows(self) -> int:
        return int(self.lib.cph_index_nrows(self.handle))

    def perm(self) -> np.ndarray:
        """perm[i] = original row id at sorted position i (host copy)."""
        p = _P()
        n = C.c_uint64()
        self.ctx._check(self.lib.cph_index_perm(self.handle, CPH_MEM_HOST, C.byref(p), C.byref(n)))
        return _ptr_array(p.value, int(n.value), np.uint32).copy()

    def perm_host_view(self) -> np.ndarray:
        """The library's pinned host copy of perm WITHOUT another copy (valid until the index is closed)."""
        p = _P()
        n = C.c_uint64()
        self.ctx._check(self.lib.cph_index_perm(self.handle, CPH_MEM_HOST, C.byref(p), C.byref(n)))
        return _ptr_array(p.value, int(n.value), np.uint32)

    def perm_device_ptr(self) -> int:
        p = _P()
        n = C.c_uint64()
        self.ctx._check(self.lib.cph_index_perm(self.handle, CPH_MEM_DEVICE, C.byref(p), C.byref(n)))
        return int(p.value or 0)

    def info(self) -> dict:
        inf = cph_index_info()
        rc = self.lib.cph_index_get_info(self.handle, C.byref(inf))
        if rc != CPH_OK:
            raise CphError(rc, "cph_index_get_info")
        return {k: int(getattr(inf, k)) for k, _ in cph_index_info._fields_ if k != "reserved_"}

    def prepare_join(self, chained: bool = False) -> None:
        """Builds the Join lookup structure (direct table / hash table) now instead of inside the first Join."""
        self.ctx._check(self.lib.cph_index_prepare_join(self.handle, 1 if chained else 0))

    def probe(self, probecols, row_sel=None, probe_base: int = 0, want_pairs: bool = True,
              out_mem: int = CPH_MEM_HOST, sel_base: int = 0) -> "Matches":
        """cph_join_probe.  row_sel: numpy uint32/uint64 array (host columns) or a tuple
        (device_ptr, bits, count) (device columns); sel_base is subtracted from every entry."""
        arr, keep = _cols_array(probecols)
        sel_ptr, sel_bits, nsel = _P(0), 32, 0
        if row_sel is not None:
            if isinstance(row_sel, np.ndarray):
                if row_sel.dtype != np.uint64:
                    row_sel = np.ascontiguousarray(row_sel, dtype=np.uint32)
                row_sel = np.ascontiguousarray(row_sel)
                sel_ptr, sel_bits, nsel = _P(row_sel.ctypes.data), row_sel.dtype.itemsize * 8, len(row_sel)
            else:  # (device pointer, bits, count)
                sel_ptr, sel_bits, nsel = _P(row_sel[0]), int(row_sel[1]), int(row_sel[2])
        out = C.POINTER(cph_matches)()
        rc = self.lib.cph_join_probe(self.ctx.handle, self.handle, arr, len(probecols), sel_ptr, sel_bits, sel_base,
                                     nsel, probe_base, 1 if want_pairs else 0, out_mem, C.byref(out))
        del keep
        self.ctx._check(rc)
        return Matches(self.lib, out, owner=self)

    def find(self, *values: bytes):
        vals = (cph_strval * max(1, len(values)))()
        keep = []
        for i, v in enumerate(values):
            b = np.frombuffer(bytes(v), dtype=np.uint8)
            keep.append(b)
            vals[i].data = b.ctypes.data if len(b) else None
            vals[i].len = len(b)
        lo, hi = C.c_uint64(), C.c_uint64()
        self.ctx._check(self.lib.cph_index_find(self.ctx.handle, self.handle, vals, len(values), C.byref(lo),
                                                C.byref(hi)))
        return int(lo.value), int(hi.value)

    def find_many(self, keys) -> tuple:
        """keys: list of tuples of bytes (all of the same arity <= key columns) -> (lower, upper) uint64 arrays:
        cph_index_find_many, one launch for the whole batch."""
        keys = [k if isinstance(k, (tuple, list)) else (k,) for k in keys]
        nkeys = len(keys)
        nvalues = len(keys[0]) if nkeys else 0
        assert all(len(k) == nvalues for k in keys)
        vals = (cph_strval * max(1, nkeys * nvalues))()
        blob = b"".join(bytes(v) for k in keys for v in k)
        buf = np.frombuffer(blob, dtype=np.uint8) if blob else np.zeros(1, np.uint8)
        off = 0
        for i, k in enumerate(keys):
            for j, v in enumerate(k):
                ln = len(v)
                vals[i * nvalues + j].data = buf.ctypes.data + off if ln else None
                vals[i * nvalues + j].len = ln
                off += ln
        lo = np.zeros(max(1, nkeys), dtype=np.uint64)
        hi = np.zeros(max(1, nkeys), dtype=np.uint64)
        self.ctx._check(self.lib.cph_index_find_many(self.ctx.handle, self.handle, vals, nvalues, nkeys,
                                                     lo.ctypes.data_as(C.POINTER(C.c_uint64)), hi.ctypes.data_as(C.POINTER(C.c_uint64))))
        return lo[:nkeys], hi[:nkeys]

    @classmethod
    def _from_handle(cls, ctx: "Context", handle) -> "DeviceIndex":
        self = cls.__new__(cls)
        self.ctx, self.lib, self.handle = ctx, ctx.lib, handle
        self.first_dup, self.status = None, CPH_OK
        ctx._children.add(self)
        return self

    def dup_groups(self):
        """(lower, upper) uint64 arrays: every maximal run [lower, upper) of >= 2 equal keys, ascending
        (the groups dedup, csvplus.go:810-867, hands to the resolve callback)."""
        out = C.POINTER(cph_groups)()
        self.ctx._check(self.lib.cph_index_dup_groups(self.ctx.handle, self.handle, C.byref(out)))
        g = out.contents
        n = int(g.ngroups)
        lo = _ptr_array(g.lower, n, np.uint64).copy() if n else np.empty(0, np.uint64)
        hi = _ptr_array(g.upper, n, np.uint64).copy() if n else np.empty(0, np.uint64)
        self.lib.cph_groups_release(out)
        return lo, hi

    def select(self, positions) -> "DeviceIndex":
        """New index over the given strictly ascending sorted positions (dedup's compaction)."""
        pos = np.ascontiguousarray(positions, dtype=np.uint64)
        h = _P()
        self.ctx._check(self.lib.cph_index_select(self.ctx.handle, self.handle, _P(pos.ctypes.data if len(pos) else 0),
                                                  len(pos), C.byref(h)))
        return DeviceIndex._from_handle(self.ctx, h)

    def save(self, path: str) -> None:
        self.ctx._check(self.lib.cph_index_save(self.ctx.handle, self.handle, str(path).encode()))

    @staticmethod
    def load(ctx: "Context", path: str) -> "DeviceIndex":
        h = _P()
        ctx._check(ctx.lib.cph_index_load(ctx.handle, str(path).encode(), C.byref(h)))
        return DeviceIndex._from_handle(ctx, h)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.cph_index_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def join_chain(ctx: Context, steps, probe_base: int = 0, out_mem: int = CPH_MEM_HOST, positions: bool = False) -> "Chain":
    """cph_join_chain[_ex]: steps = [(DeviceIndex, [key columns]) or (DeviceIndex, [key columns], source), ...].  source
    (cph_chain_step.source): 0 = the columns belong to the stream table; k > 0 = to the build table of step k-1 in its original
    row order; -k = the same in step k-1's sorted order.  positions=True (CPH_CHAIN_POSITIONS): build_row[k] holds sorted
    positions in index k (original row = index.perm()[position])."""
    arr = (cph_chain_step * len(steps))()
    keep = []
    for i, step in enumerate(steps):
        index, cols = step[0], step[1]
        carr, k = _cols_array(cols)
        keep.append((carr, k, index))
        arr[i].index = index.handle
        arr[i].cols = carr
        arr[i].ncols = len(cols)
        arr[i].source = int(step[2]) if len(step) > 2 else 0
    out = C.POINTER(cph_chain)()
    if positions:
        rc = ctx.lib.cph_join_chain_ex(ctx.handle, arr, len(steps), probe_base, out_mem, CPH_CHAIN_POSITIONS, C.byref(out))
    else:
        rc = ctx.lib.cph_join_chain(ctx.handle, arr, len(steps), probe_base, out_mem, C.byref(out))
    del keep
    ctx._check(rc)
    return Chain(ctx, out, [s[0] for s in steps], probe_base)


class Chain:
    """Result of a chained join (cph_chain): row-id tuples in emission order."""

    def __init__(self, ctx: Context, ptr, owners, probe_base: int = 0):
        self.probe_base = probe_base
        self.ctx = ctx
        self.lib = ctx.lib
        self.ptr = ptr
        self.owners = owners
        c = ptr.contents
        self.nrows = int(c.nrows)
        self.nsteps = int(c.nsteps)
        self.mem = int(c.mem)
        self.positions = bool(c.positions)
        ctx._children.add(self)

    @property
    def identity(self) -> bool:
        """True when cph_chain.stream_row is NULL: result row m IS stream row probe_base + m."""
        return self.nrows > 0 and not self.ptr.contents.stream_row

    @property
    def stream_row(self) -> np.ndarray:
        assert self.mem == CPH_MEM_HOST
        if self.identity:
            return np.arange(self.probe_base, self.probe_base + self.nrows, dtype=np.uint64)
        return _ptr_array(self.ptr.contents.stream_row, self.nrows, np.uint64).copy()

    def build_row(self, k: int) -> np.ndarray:
        assert self.mem == CPH_MEM_HOST
        return _ptr_array(self.ptr.contents.build_row[k], self.nrows, np.uint32).copy()

    def device_ptrs(self) -> dict:
        c = self.ptr.contents
        return {"stream_row": int(c.stream_row or 0), "build_row": [int(c.build_row[k] or 0) for k in range(self.nsteps)]}

    def release(self):
        if self.ptr:
            self.lib.cph_chain_release(self.ptr)
            self.ptr = None

    close = release

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class Gathered:
    """Result of an exchange (cph_gathered): device arrays of this rank's ctx + host counts / displacements."""

    def __init__(self, ctx: Context, ptr, identity=None, stream_base: int = 0):
        self.ctx, self.lib, self.ptr = ctx, ctx.lib, ptr
        g = ptr.contents
        self.total = int(g.total)
        self.narrays = int(g.narrays)
        self.counts = [int(g.counts[r]) for r in range(int(g.nranks))]
        self.displs = [int(g.displs[r]) for r in range(int(g.nranks))]
        self.data_ptrs = [int(g.data[a] or 0) for a in range(self.narrays)]
        self.mem = int(g.mem)
        self.stats = None
        self.identity = identity
        self.stream_base = stream_base
        ctx._children.add(self)

    def release(self):
        if self.ptr:
            self.lib.cph_gathered_release(self.ptr)
            self.ptr = None

    close = release

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class Dist:
    """The communicator of the sharded Join (cph_dist): RCCL, or the in-process loopback used by the tests."""

    def __init__(self, ctx: Context, handle):
        self.ctx, self.lib, self.handle = ctx, ctx.lib, handle
        self.rank = int(self.lib.cph_dist_rank(handle))
        self.size = int(self.lib.cph_dist_size(handle))
        ctx._children.add(self)

    @staticmethod
    def unique_id(ctx: Context) -> bytes:
        buf = (C.c_uint8 * CPH_DIST_ID_BYTES)()
        ctx._check(ctx.lib.cph_dist_unique_id(ctx.handle, buf))
        return bytes(buf)

    @staticmethod
    def create(ctx: Context, uid: bytes, rank: int, nranks: int) -> "Dist":
        assert len(uid) == CPH_DIST_ID_BYTES
        buf = (C.c_uint8 * CPH_DIST_ID_BYTES).from_buffer_copy(uid)
        h = _P()
        ctx._check(ctx.lib.cph_dist_create(ctx.handle, buf, rank, nranks, C.byref(h)))
        return Dist(ctx, h)

    @staticmethod
    def loopback(ctx: Context, group: str, rank: int, nranks: int) -> "Dist":
        h = _P()
        ctx._check(ctx.lib.cph_dist_create_loopback(ctx.handle, group.encode(), rank, nranks, C.byref(h)))
        return Dist(ctx, h)

    def transport(self) -> str:
        """What moves the bytes ("rccl nranks=8 lib=... " / "loopback ..."): cph_dist_transport."""
        return (self.lib.cph_dist_transport(self.handle) or b"").decode("utf-8", "replace")

    def allgatherv(self, ptrs, elem_bytes, count: int) -> Gathered:
        k = len(ptrs)
        pa = (_P * k)(*[_P(p or 0) for p in ptrs])
        ea = (C.c_int32 * k)(*elem_bytes)
        out = C.POINTER(cph_gathered)()
        self.ctx._check(self.lib.cph_dist_allgatherv(self.handle, pa, ea, k, count, C.byref(out)))
        return Gathered(self.ctx, out)

    def chain_allgather(self, chain: "Chain") -> Gathered:
        out = C.POINTER(cph_gathered)()
        ident = C.c_int32(0)
        base = C.c_uint64(0)
        self.ctx._check(self.lib.cph_dist_chain_allgather(self.handle, chain.ptr, chain.probe_base, C.byref(out),
                                                          C.byref(ident), C.byref(base)))
        return Gathered(self.ctx, out, identity=bool(ident.value), stream_base=int(base.value))

    def join_chain(self, steps, probe_base: int = 0, shard_rows=None, nchunks: int = 0, positions: bool = False,
                   host: bool = False, packed: bool = False) -> Gathered:
        """cph_dist_join_chain: this rank's shard joined in sub-chunks, chunk k exchanged (xGMI; host=True: copied into the
        node's shared host buffer) while chunk k+1 is joined.  steps = [(DeviceIndex, [shard key columns]), ...].
        packed=True: CPH_DIST_PACKED, the chunks cross the links bit-packed (g.stats["packed_bits"] per row)."""
        arr = (cph_chain_step * len(steps))()
        keep = []
        for i, (index, cols) in enumerate(steps):
            carr, k = _cols_array(cols)
            keep.append((carr, k, index))
            arr[i].index = index.handle
            arr[i].cols = carr
            arr[i].ncols = len(cols)
        sr = (C.c_uint64 * self.size)(*shard_rows) if shard_rows is not None else None
        flags = (CPH_CHAIN_POSITIONS if positions else 0) | (CPH_DIST_HOST_GATHER if host else 0) | (CPH_DIST_PACKED if packed else 0)
        out = C.POINTER(cph_gathered)()
        ident, base, st = C.c_int32(0), C.c_uint64(0), cph_dist_join_stats()
        rc = self.lib.cph_dist_join_chain(self.handle, arr, len(steps), probe_base, sr, nchunks, flags, C.byref(out), C.byref(ident),
                                          C.byref(base), C.byref(st))
        del keep
        self.ctx._check(rc)
        g = Gathered(self.ctx, out, identity=bool(ident.value), stream_base=int(base.value))
        g.stats = {f: getattr(st, f) for f, _ in cph_dist_join_stats._fields_}
        return g

    def index_broadcast(self, index, root: int = 0):
        """Root passes its DeviceIndex and gets it back; the other ranks pass None and receive an equal index."""
        h = _P()
        self.ctx._check(self.lib.cph_dist_index_broadcast(self.handle, index.handle if index is not None else None, root,
                                                          C.byref(h)))
        if self.rank == root:
            return index
        return DeviceIndex._from_handle(self.ctx, h)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.cph_dist_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Matches:
    """Result of one probe (cph_matches).  Host results are exposed as numpy copies."""

    def __init__(self, lib, ptr, owner=None):
        self.lib = lib
        self.ptr = ptr
        self.owner = owner  # keeps the index (and its ctx) alive: the arrays come from the ctx's pool
        if owner is not None:
            owner.ctx._children.add(self)
        m = ptr.contents
        self.nprobe = int(m.nprobe)
        self.nmatches = int(m.nmatches)
        self.mem = int(m.mem)

    def _host(self, field, n, dtype):
        assert self.mem == CPH_MEM_HOST
        return _ptr_array(getattr(self.ptr.contents, field), n, dtype).copy()

    @property
    def lo(self):
        return self._host("lo", self.nprobe, np.uint32)

    @property
    def cnt(self):
        return self._host("cnt", self.nprobe, np.uint32)

    @property
    def probe_idx(self):
        return self._host("probe_idx", self.nmatches, np.uint64)

    @property
    def build_row(self):
        return self._host("build_row", self.nmatches, np.uint32)

    def device_ptrs(self) -> dict:
        m = self.ptr.contents
        return {k: int(getattr(m, k) or 0) for k in ("lo", "cnt", "probe_idx", "build_row")}

    def release(self):
        if self.ptr:
            self.lib.cph_matches_release(self.ptr)
            self.ptr = None

    close = release

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

"""Binding of the deterministic synthetic-table generator (csrc/datagen.c).

Tables follow the reference's fixtures (csvplus_test.go:1207-1333) scaled to the
BASELINE.json configs; see SURVEY.md §8d.  Counter-based: any row range can be
generated independently (per-rank shards, streaming chunks).
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

from .columns import StrCol

LIB_PATH = Path(__file__).resolve().parent / "lib" / "libcph_datagen.so"

SEQ_PERM, UNIFORM, NAME, SURNAME, PRODUCT, PRICE, VARKEY, SEQ, UNIFORM_PERM, FK_SUBSET, RANDKEY = range(11)
ITOA, FIXED8 = 0, 1
SEED = 0xC5F1D5


class dg_spec(C.Structure):
    _fields_ = [("kind", C.c_int32), ("encoding", C.c_int32), ("domain", C.c_uint64), ("base", C.c_uint64),
                ("seed", C.c_uint64)]


_lib = None


def _load():
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError(f"{LIB_PATH} not found: run `make datagen`")
        lib = C.CDLL(str(LIB_PATH))
        lib.dg_column_bytes.restype = C.c_uint64
        lib.dg_column_bytes.argtypes = [C.POINTER(dg_spec), C.c_uint64, C.c_uint64]
        lib.dg_column_fill.restype = C.c_uint64
        lib.dg_column_fill.argtypes = [C.POINTER(dg_spec), C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int32]
        lib.dg_value_u64.restype = C.c_uint64
        lib.dg_value_u64.argtypes = [C.POINTER(dg_spec), C.c_uint64]
        _lib = lib
    return _lib


def column(kind: int, nrows: int, domain: int, *, encoding: int = ITOA, base: int = 0, seed: int = SEED,
           row0: int = 0, offset_bits: int = 32) -> StrCol:
    lib = _load()
    spec = dg_spec(kind, encoding, domain, base, seed)
    total = lib.dg_column_bytes(C.byref(spec), row0, nrows)
    data = np.empty(int(total) + 8, dtype=np.uint8)
    offs = np.empty(nrows + 1, dtype=np.uint32 if offset_bits == 32 else np.uint64)
    got = lib.dg_column_fill(C.byref(spec), row0, nrows, data.ctypes.data, offs.ctypes.data, offset_bits)
    if got != total:
        raise RuntimeError(f"datagen produced {got} bytes, expected {total} (32-bit offsets overflow?)")
    return StrCol(data[: int(total)], offs, nrows, offset_bits)


def value_u64(kind: int, row: int, domain: int, *, base: int = 0, seed: int = SEED) -> int:
    spec = dg_spec(kind, ITOA, domain, base, seed)
    return int(_load().dg_value_u64(C.byref(spec), row))


# ---- the tables of SURVEY.md §8d ------------------------------------------------------
def customers(n: int, *, encoding: int = FIXED8, seed: int = SEED, row0: int = 0, nrows: int | None = None) -> dict:
    """customers/people(id, name, surname): unique ids in pseudo-random (unsorted) order."""
    m = n if nrows is None else nrows
    return {
        "id": column(SEQ_PERM, m, n, encoding=encoding, seed=seed + 1, row0=row0),
        "name": column(NAME, m, n, seed=seed + 1, row0=row0),
        "surname": column(SURNAME, m, n, seed=seed + 1, row0=row0),
    }


def products(n: int, *, encoding: int = ITOA, seed: int = SEED) -> dict:
    return {
        "prod_id": column(SEQ_PERM, n, n, encoding=encoding, seed=seed + 2),
        "product": column(PRODUCT, n, n, seed=seed + 2),
        "price": column(PRICE, n, n, seed=seed + 2),
    }


def orders(m: int, n_customers: int, n_products: int, *, cust_encoding: int = FIXED8, prod_encoding: int = ITOA,
           seed: int = SEED, row0: int = 0, nrows: int | None = None) -> dict:
    """orders(cust_id, prod_id, qty): the 3 columns the reference benchmarks select
    (csvplus_test.go:1079, :1135)."""
    k = m if nrows is None else nrows
    return {
        "cust_id": column(UNIFORM, k, n_customers, encoding=cust_encoding, seed=seed + 3, row0=row0),
        "prod_id": column(UNIFORM, k, n_products, encoding=prod_encoding, seed=seed + 4, row0=row0),
        "qty": column(UNIFORM, k, 100, base=1, seed=seed + 5, row0=row0),
    }


def varkeys(n: int, distinct_suffix: int = 100_000, *, seed: int = SEED, row0: int = 0,
            nrows: int | None = None) -> StrCol:
    """config 3: surname "/" name "#" decimal(U[0,distinct_suffix)) — 10-22 bytes, duplicates."""
    k = n if nrows is None else nrows
    return column(VARKEY, k, distinct_suffix, seed=seed + 6, row0=row0)

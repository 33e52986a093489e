#!/usr/bin/env python3
"""Generates tests/golden/hotpath_vectors.json: seeded inputs (csvplus_amd.datagen, deterministic splitmix64) and
the oracle's outputs for them, condensed to FNV-1a 64-bit digests of the little-endian result arrays
(SURVEY.md §8c "parity artefacts": perm, (lo,cnt), the (probe_idx, build_row) pair list, the chained-join tuples)
plus a few literal head values.  The Go reference cannot run in this image (no Go toolchain), so the vectors come
from the C restatement in oracle/ — which test_oracle.py pins to the reference's own literal vector
(TestIndexImpl, csvplus_test.go:198-246) and structural tests.

Run from the repo root:  python tests/golden/make_hotpath_vectors.py
"""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from csvplus_amd import datagen as dg   # noqa: E402
from oracle import orc   # noqa: E402

CASES = [
    # name, customers, products, orders, customer id encoding
    {"name": "config1_1e5", "nc": 100_000, "np": 1_000, "m": 100_000, "enc": "ITOA"},
    {"name": "fixed8_3e5", "nc": 300_000, "np": 5_000, "m": 400_000, "enc": "FIXED8"},
    {"name": "missing_customers", "nc": 50_000, "np": 100, "m": 120_000, "enc": "ITOA", "extra_ids": 5_000},
]


def digest(a):
    return "%016x" % orc.fnv1a64(np.ascontiguousarray(a))


def case_vectors(c):
    enc = getattr(dg, c["enc"])
    cust, prod = dg.customers(c["nc"], encoding=enc), dg.products(c["np"])
    ords = dg.orders(c["m"], c["nc"] + c.get("extra_ids", 0), c["np"], cust_encoding=enc)
    ia, ib = orc.OracleIndex([cust["id"]]), orc.OracleIndex([prod["prod_id"]])
    j1 = ia.join([ords["cust_id"]])
    j2 = ib.join([ords["prod_id"]], row_sel=j1["probe_idx"].astype(np.uint32))
    pick = j2["probe_idx"].astype(np.int64)
    stream, arow, brow = j1["probe_idx"][pick], j1["build_row"][pick], j2["build_row"]
    by_name = orc.OracleIndex([cust["surname"], cust["name"]])          # duplicates: stable order
    jn = by_name.join([cust["surname"].slice(0, 100)])                 # prefix join, many matches per row
    return {
        "inputs": {k: c[k] for k in ("nc", "np", "m", "enc")} | {"extra_ids": c.get("extra_ids", 0), "seed": dg.SEED},
        "customers_perm": digest(ia.perm), "customers_perm_head": [int(x) for x in ia.perm[:8]],
        "products_perm": digest(ib.perm),
        "join1_lo_cnt": [digest(j1["lo"][j1["cnt"] > 0]), digest(j1["cnt"])], "join1_nmatches": int(j1["nmatches"]),
        "join1_pairs": [digest(j1["probe_idx"]), digest(j1["build_row"])],
        "chain_rows": int(len(stream)),
        "chain": [digest(stream.astype(np.uint64)), digest(arow.astype(np.uint32)), digest(brow.astype(np.uint32))],
        "dup_index_perm": digest(by_name.perm), "dup_first_dup": by_name.first_dup(),
        "dup_join_nmatches": int(jn["nmatches"]), "dup_join_pairs": [digest(jn["probe_idx"]), digest(jn["build_row"])],
    }


if __name__ == "__main__":
    out = {c["name"]: case_vectors(c) for c in CASES}
    path = Path(__file__).with_name("hotpath_vectors.json")
    path.write_text(json.dumps(out, indent=1, sort_keys=True) + "\n")
    print("wrote", path)

"""Shared helpers for the parity tests: fixtures shaped like the reference's
(csvplus_test.go:1207-1333) with a seeded PRNG, and oracle-vs-GPU comparison."""
from __future__ import annotations

import numpy as np

from csvplus_amd import StrCol

PEOPLE_NAMES = ["Amelia", "Olivia", "Emily", "Ava", "Isla", "Oliver", "Jack", "Harry", "Jacob", "Charlie"]
PEOPLE_SURNAMES = ["Smith", "Jones", "Taylor", "Williams", "Brown", "Davies", "Evans", "Wilson", "Thomas",
                   "Roberts", "Johnson", "Lewis"]
STOCK = [("banana", 0.01), ("apple", 0.02), ("orange", 0.03), ("pea", 0.04), ("tomato", 0.05), ("potato", 0.06),
         ("cucumber", 0.07), ("iPhone", 0.08)]
NUM_ORDERS = 10000


def people_table():
    """makePersonsCsvFile (csvplus_test.go:1220-1253): 120 rows id,name,surname (born omitted: not a key)."""
    ids, names, surnames = [], [], []
    for i, name in enumerate(PEOPLE_NAMES):
        for j, surname in enumerate(PEOPLE_SURNAMES):
            ids.append(str(i * len(PEOPLE_SURNAMES) + j))
            names.append(name)
            surnames.append(surname)
    return {"id": ids, "name": names, "surname": surnames}


def stock_table():
    """makeStockCsvFile (csvplus_test.go:1277-1295)."""
    return {"prod_id": [str(i) for i in range(len(STOCK))], "product": [s[0] for s in STOCK],
            "price": ["%.2f" % s[1] for s in STOCK]}


def orders_table(seed=12345, n=NUM_ORDERS):
    """makeOrderCsvFile (csvplus_test.go:1300-1333), seeded (the reference's rand is unseeded)."""
    rng = np.random.default_rng(seed)
    cust = rng.integers(0, len(PEOPLE_NAMES) * len(PEOPLE_SURNAMES), n)
    prod = rng.integers(0, len(STOCK), n)
    qty = rng.integers(1, 101, n)
    return {"order_id": [str(i) for i in range(n)], "cust_id": [str(x) for x in cust],
            "prod_id": [str(x) for x in prod], "qty": [str(x) for x in qty]}


def cols_of(table, *names, offset_bits=32):
    return [StrCol.from_values(table[n], offset_bits=offset_bits) for n in names]


def random_keys(rng, n, min_len=0, max_len=12, alphabet=None, distinct=None):
    """n random byte strings; `distinct` limits the number of different values (heavy duplicates)."""
    if alphabet is None:
        alphabet = np.arange(256, dtype=np.uint8)
    alphabet = np.asarray(alphabet, dtype=np.uint8)
    pool_n = distinct if distinct else n
    pool = []
    for _ in range(pool_n):
        ln = int(rng.integers(min_len, max_len + 1))
        pool.append(alphabet[rng.integers(0, len(alphabet), ln)].tobytes())
    if distinct:
        return [pool[i] for i in rng.integers(0, pool_n, n)]
    return pool


def assert_join_equal(gpu_matches, oracle_join, check_lo=True):
    """Bit-exact comparison of a cph_matches (host) with the oracle's join output."""
    cnt = gpu_matches.cnt
    np.testing.assert_array_equal(cnt, oracle_join["cnt"])
    if check_lo:
        lo = gpu_matches.lo
        nz = cnt > 0   # lo is unspecified for rows without a match
        np.testing.assert_array_equal(lo[nz], oracle_join["lo"][nz])
    assert gpu_matches.nmatches == oracle_join["nmatches"]
    if oracle_join["probe_idx"] is not None and gpu_matches.probe_idx is not None:
        np.testing.assert_array_equal(gpu_matches.probe_idx, oracle_join["probe_idx"])
        np.testing.assert_array_equal(gpu_matches.build_row, oracle_join["build_row"])


def assert_bounds_equal(gpu_index, probecols, oracle_join):
    """The same Join asked for bounds only (want_pairs=0: Except / has / counts — a duplicate-free index over a dense
    code space answers through its rank table): (lo, cnt) bit-exact against the oracle's."""
    m = gpu_index.probe(probecols, want_pairs=False)
    np.testing.assert_array_equal(m.cnt, oracle_join["cnt"])
    nz = m.cnt > 0
    np.testing.assert_array_equal(m.lo[nz], oracle_join["lo"][nz])
    assert m.nmatches == oracle_join["nmatches"]
    m.release()

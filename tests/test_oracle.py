"""Pins the CPU oracle (oracle/csvplus_oracle.c) against the reference's own tests.

The reference's only literal known-answer vector is TestIndexImpl
(csvplus_test.go:198-246); the other tests are structural and are restated here on
fixtures of the same shape (seeded, the reference's math/rand is unseeded)."""
import numpy as np
import pytest

from csvplus_amd import StrCol
from oracle import orc
from tests.helpers import (PEOPLE_NAMES, PEOPLE_SURNAMES, cols_of, orders_table, people_table, random_keys,
                           stock_table)

INDEX_IMPL_ROWS = [  # csvplus_test.go:201-209
    ("1", "2", "3", "zzz"), ("5", "6", "8", "nnn"), ("0", "5", "3", "xxx"), ("8", "9", "1", "aaa"),
    ("7", "4", "0", "bbb"), ("5", "6", "9", "iii"), ("2", "6", "7", "mmm"),
]


@pytest.mark.parametrize("mode", [orc.SORT_STABLE, orc.SORT_GO_PDQSORT])
def test_index_impl_known_answer(mode):
    """TestIndexImpl (csvplus_test.go:198-246)."""
    cols = [StrCol.from_values([r[i] for r in INDEX_IMPL_ROWS]) for i in range(3)]
    ix = orc.OracleIndex(cols, mode)
    junk = [INDEX_IMPL_ROWS[i][3] for i in ix.perm]
    assert junk == ["xxx", "zzz", "mmm", "nnn", "iii", "bbb", "aaa"]
    lo, hi = ix.find("1", "2", "3")
    assert hi - lo == 1 and INDEX_IMPL_ROWS[ix.perm[lo]][3] == "zzz"   # :214-223
    lo, hi = ix.find("5", "6", "8")
    assert hi - lo == 1 and INDEX_IMPL_ROWS[ix.perm[lo]][3] == "nnn"   # :225-234
    lo, hi = ix.find("5", "6")
    assert hi - lo == 2                                                  # :236-245
    assert all(INDEX_IMPL_ROWS[ix.perm[i]][:2] == ("5", "6") for i in range(lo, hi))


def test_sorted():
    """TestSorted (csvplus_test.go:454-514): bytewise lexicographic tuple order."""
    p = people_table()
    ix = orc.OracleIndex(cols_of(p, "name", "surname"))
    assert ix.first_dup() is None
    names = [p["name"][i] for i in ix.perm]
    assert names[:12] == ["Amelia"] * 12 and names[12:24] == ["Ava"] * 12
    ix = orc.OracleIndex(cols_of(p, "surname", "name"))
    surnames = [p["surname"][i] for i in ix.perm]
    assert surnames[10:20] == ["Davies"] * 10


def test_simple_unique_join():
    """TestSimpleUniqueJoin (csvplus_test.go:368-452): every order joins exactly once; totals match."""
    p, o = people_table(), orders_table()
    ix = orc.OracleIndex(cols_of(p, "id"))
    assert ix.first_dup() is None
    j = ix.join(cols_of(o, "cust_id"))
    assert j["nmatches"] == len(o["cust_id"])
    np.testing.assert_array_equal(j["probe_idx"], np.arange(len(o["cust_id"]), dtype=np.uint64))
    qty = np.zeros(len(p["id"]), dtype=np.int64)
    for pi, br in zip(j["probe_idx"], j["build_row"]):
        assert p["id"][br] == o["cust_id"][pi]
        qty[int(p["id"][br])] += int(o["qty"][pi])
    orig = np.zeros(len(p["id"]), dtype=np.int64)
    for c, q in zip(o["cust_id"], o["qty"]):
        orig[int(c)] += int(q)
    np.testing.assert_array_equal(qty, orig)


def test_decimal_ids_sort_as_strings():
    """ids are compared as strings: "10" < "9" (csvplus_test.go:1241 writes unpadded strconv.Itoa)."""
    ids = [str(i) for i in range(120)]
    ix = orc.OracleIndex([StrCol.from_values(ids)])
    assert [ids[i] for i in ix.perm] == sorted(ids)


def test_multi_index_find():
    """TestMultiIndex (csvplus_test.go:573-649)."""
    p = people_table()
    ix = orc.OracleIndex(cols_of(p, "name", "surname"))
    assert ix.find("xxx")[0] == ix.find("xxx")[1]
    lo, hi = ix.find("Amelia")
    assert hi - lo == len(PEOPLE_SURNAMES)
    for n, s in zip(p["name"], p["surname"]):
        lo, hi = ix.find(n, s)
        assert hi - lo == 1
    lo, hi = ix.find("Jack", "xxx")
    assert lo == hi
    assert ix.find() == (0, 120)   # csvplus.go:872-874


def test_except_counts():
    """TestExcept (csvplus_test.go:651-693): anti-join = probe rows with cnt == 0."""
    p, o = people_table(), orders_table()
    emily = [i for i, n in enumerate(p["name"]) if n == "Emily"]
    sub = {"id": [p["id"][i] for i in emily]}
    ix = orc.OracleIndex(cols_of(sub, "id"))
    j = ix.join(cols_of(o, "cust_id"), want_pairs=False)
    n = int((j["cnt"] == 0).sum())
    m = sum(1 for c in o["cust_id"] if p["name"][int(c)] != "Emily")
    assert n == m
    for c, k in zip(o["cust_id"], j["cnt"]):
        assert ix.has(c) == (k > 0)


def test_duplicate_detection():
    """TestErrors (csvplus_test.go:836-841): UniqueIndexOn("name") reports the smallest duplicated key."""
    p = people_table()
    ix = orc.OracleIndex(cols_of(p, "name"))
    d = ix.first_dup()
    assert d == 1   # rows[0]==rows[1]=="Amelia"
    assert p["name"][ix.perm[d]] == "Amelia"


def test_non_unique_build_side_long_chain_shape():
    """TestLongChain's join shape (csvplus_test.go:252-285): IndexOn(orders.cust_id) is the build side."""
    p, o = people_table(), orders_table()
    ix = orc.OracleIndex(cols_of(o, "cust_id"))
    j = ix.join(cols_of(p, "id"))
    assert j["nmatches"] == len(o["cust_id"])   # every order has a customer
    for pi, br in zip(j["probe_idx"], j["build_row"]):
        assert o["cust_id"][br] == p["id"][pi]
    # stable canonical order: within one key, build rows ascend
    for k in range(len(p["id"])):
        rows = j["build_row"][j["probe_idx"] == k]
        assert np.all(np.diff(rows.astype(np.int64)) > 0)


def test_prefix_join_and_natural_subindex():
    """prefix join on the leading index column (csvplus.go:546-550, :910; BenchmarkJoinOnBiggerMultiIndex)."""
    p, o = people_table(), orders_table()
    ix = orc.OracleIndex(cols_of(o, "cust_id", "prod_id"))
    j = ix.join(cols_of(p, "id"))
    assert j["nmatches"] == len(o["cust_id"])
    keys = [(o["cust_id"][i], o["prod_id"][i]) for i in ix.perm]
    assert keys == sorted(keys)


def test_pdqsort_emulation_agrees_modulo_equal_keys():
    """P1/P2: Go's unstable sort may permute rows inside an equal-key group only."""
    rng = np.random.default_rng(7)
    for n in (5, 13, 50, 51, 200, 1000, 5000):
        keys = random_keys(rng, n, 0, 3, alphabet=list(b"ab"), distinct=max(2, n // 7))
        col = StrCol.from_values(keys)
        st = orc.OracleIndex([col], orc.SORT_STABLE)
        go = orc.OracleIndex([col], orc.SORT_GO_PDQSORT)
        assert [keys[i] for i in st.perm] == [keys[i] for i in go.perm] == sorted(keys)
        assert sorted(go.perm.tolist()) == list(range(n))
        assert st.first_dup() == go.first_dup()
    # presorted and reversed inputs exercise partialInsertionSort / reverseRange
    for keys in ([b"%05d" % i for i in range(300)], [b"%05d" % i for i in range(300)][::-1]):
        go = orc.OracleIndex([StrCol.from_values(keys)], orc.SORT_GO_PDQSORT)
        assert [keys[i] for i in go.perm] == sorted(keys)


def test_strings_compare_edge_cases():
    """strings.Compare semantics: prefix first, NUL and high bytes are ordinary bytes, empty sorts first."""
    keys = [b"a", b"", b"a\x00", b"ab", b"\xff", b"a\x00\x00", b"\x00", b"b", b"a"]
    ix = orc.OracleIndex([StrCol.from_values(keys)])
    assert [keys[i] for i in ix.perm] == sorted(keys)
    assert ix.first_dup() == sorted(keys).index(b"a") + 1


def test_empty_and_single():
    ix = orc.OracleIndex([StrCol.from_values([])])
    assert ix.first_dup() is None and ix.find("x") == (0, 0)
    j = ix.join([StrCol.from_values(["x", ""])])
    assert j["nmatches"] == 0 and j["cnt"].tolist() == [0, 0]
    ix = orc.OracleIndex([StrCol.from_values([""])])
    assert ix.join([StrCol.from_values(["x", ""])])["cnt"].tolist() == [0, 1]


def test_cost_model_baselines_agree_with_the_oracle():
    """oracle/faithful.cpp (bench.py's cpu_baseline variants): the map-per-row restatement and the all-cores lean
    join must produce the oracle's results — they are only allowed to differ in what they cost."""
    from csvplus_amd import datagen as dg

    nc, npd, m = 3000, 40, 20_000
    cust, prod = dg.customers(nc), dg.products(npd)
    ords = dg.orders(m, 2 * nc, npd)                       # half of the customer ids miss
    oa, ob = orc.OracleIndex([cust["id"]]), orc.OracleIndex([prod["prod_id"]])
    j1 = oa.join([ords["cust_id"]])
    j2 = ob.join([ords["prod_id"]], row_sel=j1["probe_idx"].astype(np.uint32))
    r = orc.faithful_chain_join(cust, "id", prod, "prod_id", ords, "cust_id", "prod_id")
    assert r["joined"] == j2["nmatches"] and r["checksum"] > 0
    rows, joined, _, _, threads = orc.lean_mt_join(cust["id"], ords["cust_id"], threads=2)
    assert joined == j1["nmatches"] and threads == 2
    hit = rows != 0xFFFFFFFF
    np.testing.assert_array_equal(np.nonzero(hit)[0].astype(np.uint64), j1["probe_idx"])
    np.testing.assert_array_equal(rows[hit], j1["build_row"])
    # a duplicate key makes the faithful UniqueIndexOn fail like the reference (:751)
    dup = {"id": cust["id"].slice(0, 10), "name": cust["name"].slice(0, 10)}
    from csvplus_amd import StrCol
    dup["id"] = StrCol.from_values([b"00000001"] * 10)
    assert orc.faithful_chain_join(dup, "id", prod, "prod_id", ords, "cust_id", "prod_id")["joined"] == 2**64 - 1

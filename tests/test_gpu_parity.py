"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle.

Bar: bit-exact.  perm, first_dup, (lo,cnt) and the (probe_idx, build_row)
pair list must equal the oracle's; for duplicate keys both sides use the canonical stable
order (input order inside an equal-key group)."""
import numpy as np
import pytest

from csvplus_amd import DeviceIndex, StrCol, _native as N, datagen as dg
from oracle import orc
from tests.helpers import (PEOPLE_NAMES, PEOPLE_SURNAMES, assert_bounds_equal, assert_join_equal, cols_of, orders_table, people_table,
                           random_keys, stock_table)
from tests.test_oracle import INDEX_IMPL_ROWS

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("both_build_paths")]


def check_index(ctx, keycols, unique=False):
    """Builds on GPU + oracle and compares perm / first_dup bit-exactly; returns both."""
    g = DeviceIndex(ctx, keycols, unique=unique)
    o = orc.OracleIndex(keycols)
    np.testing.assert_array_equal(g.perm(), o.perm)
    assert g.first_dup == o.first_dup()
    if unique:
        assert g.status == (N.CPH_ERR_DUPLICATE if o.first_dup() is not None else N.CPH_OK)
    return g, o


# ---- the reference's own tests, through the GPU path ------------------------------------------
def test_index_impl_known_answer(ctx):
    """TestIndexImpl (csvplus_test.go:198-246)."""
    cols = [StrCol.from_values([r[i] for r in INDEX_IMPL_ROWS]) for i in range(3)]
    g, o = check_index(ctx, cols)
    assert [INDEX_IMPL_ROWS[i][3] for i in g.perm()] == ["xxx", "zzz", "mmm", "nnn", "iii", "bbb", "aaa"]
    assert g.find(b"1", b"2", b"3") == o.find("1", "2", "3") == (1, 2)
    assert g.find(b"5", b"6", b"8") == (3, 4)
    assert g.find(b"5", b"6") == (3, 5)
    assert g.find() == (0, 7)
    lo, hi = g.find(b"9")
    assert lo == hi


def test_sorted(ctx):
    """TestSorted (csvplus_test.go:454-514)."""
    p = people_table()
    g, _ = check_index(ctx, cols_of(p, "name", "surname"), unique=True)
    names = [p["name"][i] for i in g.perm()]
    assert names[:12] == ["Amelia"] * 12 and names[12:24] == ["Ava"] * 12
    g, _ = check_index(ctx, cols_of(p, "surname", "name"), unique=True)
    assert [p["surname"][i] for i in g.perm()][10:20] == ["Davies"] * 10


def test_simple_unique_join(ctx):
    """TestSimpleUniqueJoin (csvplus_test.go:368-452): differently named columns, 6-column rows."""
    p, o = people_table(), orders_table()
    g, orc_ix = check_index(ctx, cols_of(p, "id"), unique=True)
    m = g.probe(cols_of(o, "cust_id"))
    assert_join_equal(m, orc_ix.join(cols_of(o, "cust_id")))
    assert m.nmatches == len(o["cust_id"])
    qty = np.zeros(120, dtype=np.int64)
    for pi, br in zip(m.probe_idx, m.build_row):
        assert p["id"][br] == o["cust_id"][pi]
        qty[int(p["id"][br])] += int(o["qty"][pi])
    orig = np.zeros(120, dtype=np.int64)
    for c, q in zip(o["cust_id"], o["qty"]):
        orig[int(c)] += int(q)
    np.testing.assert_array_equal(qty, orig)


def test_simple_totals_natural_join(ctx):
    """TestSimpleTotals (csvplus_test.go:516-571): natural join on prod_id."""
    s, o = stock_table(), orders_table()
    g, orc_ix = check_index(ctx, cols_of(s, "prod_id"), unique=True)
    m = g.probe(cols_of(o, "prod_id"))
    assert_join_equal(m, orc_ix.join(cols_of(o, "prod_id")))
    assert m.nmatches == len(o["prod_id"])


def test_long_chain_shape(ctx):
    """TestLongChain (csvplus_test.go:252-285): non-unique IndexOn(cust_id) as build side, then a second
    natural join on prod_id of the joined rows (chained probe via row selection)."""
    p, o, s = people_table(), orders_table(), stock_table()
    g_orders, o_orders = check_index(ctx, cols_of(o, "cust_id"))
    assert g_orders.first_dup is not None
    m1 = g_orders.probe(cols_of(p, "id"))
    j1 = o_orders.join(cols_of(p, "id"))
    assert_join_equal(m1, j1)
    assert m1.nmatches == len(o["cust_id"])
    # second join: key prod_id comes from the INDEX side rows of join 1 (orders), so select by build_row
    g_prod, o_prod = check_index(ctx, cols_of(s, "prod_id"), unique=True)
    sel = m1.build_row
    m2 = g_prod.probe(cols_of(o, "prod_id"), row_sel=sel)
    j2 = o_prod.join(cols_of(o, "prod_id"), row_sel=sel)
    assert_join_equal(m2, j2)
    assert m2.nmatches == m1.nmatches
    for k in range(0, m2.nmatches, 97):
        order_row = sel[int(m2.probe_idx[k])]
        assert s["prod_id"][m2.build_row[k]] == o["prod_id"][order_row]


def test_multi_index_find_and_prefix_join(ctx):
    """TestMultiIndex (csvplus_test.go:573-649) + prefix join (csvplus.go:546-550)."""
    p, o = people_table(), orders_table()
    g, oi = check_index(ctx, cols_of(p, "name", "surname"), unique=True)
    lo, hi = g.find(b"xxx")
    assert lo == hi
    assert g.find(b"Amelia") == oi.find("Amelia")
    for n, s in zip(p["name"], p["surname"]):
        assert g.find(n.encode(), s.encode()) == oi.find(n, s)
    lo, hi = g.find(b"Jack", b"xxx")
    assert lo == hi
    # self-join on the leading column only: every row matches its 12 namesakes
    m = g.probe(cols_of(p, "name"))
    assert_join_equal(m, oi.join(cols_of(p, "name")))
    assert m.nmatches == 120 * 12
    # bigger: IndexOn(cust_id, prod_id) probed with people.id (BenchmarkJoinOnBiggerMultiIndex :1161-1186)
    g2, o2 = check_index(ctx, cols_of(o, "cust_id", "prod_id"))
    m = g2.probe(cols_of(p, "id"))
    assert_join_equal(m, o2.join(cols_of(p, "id")))
    assert m.nmatches == len(o["cust_id"])
    for c in ("0", "7", "119"):
        assert g2.find(c.encode()) == o2.find(c)
        assert g2.find(c.encode(), b"3") == o2.find(c, "3")


def test_except_via_cnt(ctx):
    """TestExcept (csvplus_test.go:651-693)."""
    p, o = people_table(), orders_table()
    sub = {"id": [p["id"][i] for i, n in enumerate(p["name"]) if n == "Emily"]}
    g, oi = check_index(ctx, cols_of(sub, "id"))
    m = g.probe(cols_of(o, "cust_id"), want_pairs=False)
    j = oi.join(cols_of(o, "cust_id"), want_pairs=False)
    np.testing.assert_array_equal(m.cnt, j["cnt"])
    assert int((m.cnt == 0).sum()) == sum(1 for c in o["cust_id"] if p["name"][int(c)] != "Emily")


def test_errors(ctx):
    """TestErrors (csvplus_test.go:836-841) + boundary errors."""
    p = people_table()
    g = DeviceIndex(ctx, cols_of(p, "name"), unique=True)
    assert g.status == N.CPH_ERR_DUPLICATE
    assert g.first_dup == 1 and p["name"][g.perm()[g.first_dup]] == "Amelia"
    assert "duplicate value while creating unique index" in ctx.last_error()
    g = DeviceIndex(ctx, cols_of(p, "id"))
    with pytest.raises(N.CphError) as e:   # csvplus.go:548-550 "too many source columns in Join()"
        g.probe(cols_of(p, "id", "name"))
    assert e.value.code == N.CPH_ERR_TOO_MANY_COLS
    long_ix = DeviceIndex(ctx, [StrCol.from_values([b"x" * 200, b"y"])])   # no key-length limit (csvplus.go:794-807)
    assert long_ix.perm().tolist() == [0, 1] and long_ix.info()["key_positions"] == 200
    with pytest.raises(N.CphError) as e:
        DeviceIndex(ctx, [StrCol.from_values(["a", "b"]), StrCol.from_values(["a"])])
    assert e.value.code == N.CPH_ERR_INVALID


# ---- edge cases ---------------------------------------------------------
def test_strings_compare_edge_cases(ctx):
    keys = [b"a", b"", b"a\x00", b"ab", b"\xff", b"a\x00\x00", b"\x00", b"b", b"a", b"\xff\xff", b"a\x01"]
    g, oi = check_index(ctx, [StrCol.from_values(keys)])
    assert [keys[i] for i in g.perm()] == sorted(keys)
    probe = [b"a", b"", b"zzz", b"a\x00", b"\x00\x00", b"a\x00\x00\x00", b"c", b"\xfe"]
    assert_join_equal(g.probe([StrCol.from_values(probe)]), oi.join([StrCol.from_values(probe)]))


def test_empty_tables(ctx):
    g, oi = check_index(ctx, [StrCol.from_values([])])
    m = g.probe([StrCol.from_values(["x", ""])])
    assert m.nmatches == 0 and m.cnt.tolist() == [0, 0]
    assert g.find(b"x") == (0, 0) and g.find() == (0, 0)
    g, oi = check_index(ctx, [StrCol.from_values(["", "", "q"])])
    m = g.probe([StrCol.from_values([])])
    assert m.nprobe == 0 and m.nmatches == 0
    assert_join_equal(g.probe([StrCol.from_values(["", "q", "x"])]), oi.join([StrCol.from_values(["", "q", "x"])]))
    # all keys empty
    g, oi = check_index(ctx, [StrCol.from_values(["", "", ""])])
    assert_join_equal(g.probe([StrCol.from_values(["", "a"])]), oi.join([StrCol.from_values(["", "a"])]))


def test_all_equal_keys(ctx):
    keys = ["same"] * 5000
    g, oi = check_index(ctx, [StrCol.from_values(keys)])
    pr = [StrCol.from_values(["same", "other", "same"])]
    assert_join_equal(g.probe(pr), oi.join(pr))


@pytest.mark.parametrize("offset_bits", [32, 64])
@pytest.mark.parametrize("case", ["bytes_any", "ascii_dups", "short_binary", "long_keys", "two_cols", "three_cols_dups"])
def test_random_property(ctx, case, offset_bits):
    rng = np.random.default_rng(hash(case) % 2**32)
    n, m = 20000, 30000
    if case == "bytes_any":
        b = [random_keys(rng, n, 0, 12)]
        p = [random_keys(rng, m // 2, 0, 12) + [b[0][i] for i in rng.integers(0, n, m - m // 2)]]
    elif case == "ascii_dups":
        b = [random_keys(rng, n, 1, 6, alphabet=list(b"0123456789"), distinct=3000)]
        p = [random_keys(rng, m, 1, 6, alphabet=list(b"0123456789"))]
    elif case == "short_binary":
        b = [random_keys(rng, n, 0, 3, alphabet=[0, 1, 255])]
        p = [random_keys(rng, m, 0, 4, alphabet=[0, 1, 2, 255])]
    elif case == "long_keys":
        b = [random_keys(rng, n, 20, 40, alphabet=list(b"abcdefghijklmnopqrstuvwxyz/#"), distinct=7000)]
        p = [[b[0][i] for i in rng.integers(0, n, m)]]
        p[0][::7] = random_keys(rng, len(p[0][::7]), 20, 40, alphabet=list(b"abcxyz/#"))
    elif case == "two_cols":
        b = [random_keys(rng, n, 0, 5, alphabet=list(b"abc"), distinct=200),
             random_keys(rng, n, 0, 9, alphabet=list(b"0123456789"), distinct=500)]
        idx = rng.integers(0, n, m)
        p = [[b[0][i] for i in idx], [b[1][i] for i in rng.integers(0, n, m)]]
    else:
        b = [random_keys(rng, n, 1, 3, alphabet=list(b"xy"), distinct=4),
             random_keys(rng, n, 0, 2, alphabet=list(b"01"), distinct=5),
             random_keys(rng, n, 1, 2, alphabet=list(b"pq"), distinct=3)]
        p = [[b[c][i] for i in rng.integers(0, n, 2000)] for c in range(3)]
    bcols = [StrCol.from_values(x, offset_bits=offset_bits) for x in b]
    pcols = [StrCol.from_values(x, offset_bits=offset_bits) for x in p]
    g, oi = check_index(ctx, bcols)
    keys = list(zip(*[[x[i] for i in g.perm()] for x in b]))
    assert keys == sorted(keys)
    assert_join_equal(g.probe(pcols), oi.join(pcols))
    assert_bounds_equal(g, pcols, oi.join(pcols))
    for k in range(1, len(bcols)):   # prefix joins on the leading k columns
        assert_join_equal(g.probe(pcols[:k]), oi.join(pcols[:k]))
    for r in rng.integers(0, n, 20):   # Find bounds on full and prefix tuples
        for k in range(1, len(bcols) + 1):
            vals = [x[r] for x in b[:k]]
            assert g.find(*vals) == oi.find(*vals)
    # the same through cph_index_find_many: 500 keys of every arity in one launch (present, absent, unencodable)
    for k in range(1, len(bcols) + 1):
        rows = rng.integers(0, n, 400)
        keys = [tuple(x[r] for x in b[:k]) for r in rows]
        keys += [tuple(x[i] for x in p[:k]) for i in rng.integers(0, len(p[0]), 100)] if len(p) >= k else []
        keys += [tuple([b"\x07\xfe?"] * k), tuple([b""] * k)]
        lo, hi = g.find_many(keys)
        for j, key in enumerate(keys):
            olo, ohi = oi.find(*key)
            assert hi[j] - lo[j] == ohi - olo and (ohi == olo or lo[j] == olo), (k, j, key)
    assert g.find_many([()])[1][0] == n and len(g.find_many([])[0]) == 0


def test_pdqsort_emulation_parity_class(ctx):
    """P1: vs the emulated Go sort.Sort the key sequence and the per-key row multisets agree."""
    o = orders_table()
    cols = cols_of(o, "cust_id")
    g = DeviceIndex(ctx, cols)
    go = orc.OracleIndex(cols, orc.SORT_GO_PDQSORT)
    gp, op = g.perm(), go.perm
    keys_g = [o["cust_id"][i] for i in gp]
    assert keys_g == [o["cust_id"][i] for i in op]
    bounds = [0] + [i for i in range(1, len(keys_g)) if keys_g[i] != keys_g[i - 1]] + [len(keys_g)]
    for a, b in zip(bounds[:-1], bounds[1:]):
        assert sorted(gp[a:b].tolist()) == sorted(op[a:b].tolist())


# ---- generated tables (BASELINE.json configs, scaled to oracle-in-seconds sizes) ---------------
@pytest.mark.parametrize("enc", [dg.FIXED8, dg.ITOA])
def test_config1_unique_index_and_join(ctx, enc):
    """config 1: 1e5 people + orders, UniqueIndexOn(id) then Join."""
    n = 100_000
    cust = dg.customers(n, encoding=enc)
    ords = dg.orders(n, n, 1000, cust_encoding=enc)
    g, oi = check_index(ctx, [cust["id"]], unique=True)
    assert g.status == N.CPH_OK
    m = g.probe([ords["cust_id"]])
    assert_join_equal(m, oi.join([ords["cust_id"]]))
    assert m.nmatches == n
    assert_bounds_equal(g, [ords["cust_id"]], oi.join([ords["cust_id"]]))   # bounds only: through the rank table
    assert g.info()["lookup_built"] & 8
    info = g.info()
    assert info["direct_table"] == 1 and info["key_bytes"] == 4


def test_config3_varlen_duplicates(ctx):
    """config 3 shape: variable-length 10-22 byte keys with duplicates (multi-word codes)."""
    n = 300_000
    keys = dg.varkeys(n, 2000)
    g, oi = check_index(ctx, [keys])
    assert g.info()["code_words"] >= 1
    probe = dg.varkeys(50_000, 2500, seed=99)
    assert_join_equal(g.probe([probe]), oi.join([probe]))


def test_chained_join_config4_shape(ctx):
    """config 4 shape: orders JOIN customers JOIN products, chained on the device via row selection."""
    nc, npd, m = 50_000, 1000, 200_000
    cust, prod = dg.customers(nc), dg.products(npd)
    ords = dg.orders(m, nc, npd)
    gc, oc = check_index(ctx, [cust["id"]], unique=True)
    gp, op = check_index(ctx, [prod["prod_id"]], unique=True)
    m1 = gc.probe([ords["cust_id"]], probe_base=1000)
    j1 = oc.join([ords["cust_id"]], probe_base=1000)
    assert_join_equal(m1, j1)
    sel64 = m1.probe_idx   # uint64, base 1000: exactly what a second chained probe consumes
    m2 = gp.probe([ords["prod_id"]], row_sel=sel64, sel_base=1000)
    j2 = op.join([ords["prod_id"]], row_sel=(sel64 - 1000).astype(np.uint32))
    assert_join_equal(m2, j2)
    assert m2.nmatches == m


def test_device_resident_inputs_and_outputs(ctx):
    """mem = CPH_MEM_DEVICE on both sides gives the same bits as the host path."""
    import torch

    n = 60_000
    cust = dg.customers(n)
    ords = dg.orders(80_000, n, 100)
    gh = DeviceIndex(ctx, [cust["id"]], unique=True)
    gd = DeviceIndex(ctx, [cust["id"].to_device()], unique=True)
    np.testing.assert_array_equal(gh.perm(), gd.perm())
    mh = gh.probe([ords["cust_id"]])
    md = gd.probe([ords["cust_id"].to_device()], out_mem=N.CPH_MEM_DEVICE)
    assert md.nmatches == mh.nmatches
    ptrs = md.device_ptrs()

    def dev_array(ptr, count, dtype):
        class _W:   # zero-copy view through __cuda_array_interface__
            __cuda_array_interface__ = {"shape": (count,), "typestr": dtype, "data": (ptr, False), "version": 2}
        return torch.as_tensor(_W(), device="cuda:0").cpu().numpy()

    np.testing.assert_array_equal(dev_array(ptrs["cnt"], md.nprobe, "<i4").view(np.uint32), mh.cnt)
    np.testing.assert_array_equal(dev_array(ptrs["probe_idx"], md.nmatches, "<i8").view(np.uint64), mh.probe_idx)
    np.testing.assert_array_equal(dev_array(ptrs["build_row"], md.nmatches, "<i4").view(np.uint32), mh.build_row)


def test_config2_properties_1e7(ctx):
    """config 2 at full size (1e7 unique 8-byte keys + 1e7 probes): size-independent properties, plus the
    oracle on a 2e5-row probe sample."""
    n = 10_000_000
    cust_id = dg.column(dg.SEQ_PERM, n, n, encoding=dg.FIXED8, seed=dg.SEED + 1)
    ords = dg.column(dg.UNIFORM, n, n, encoding=dg.FIXED8, seed=dg.SEED + 3)
    g = DeviceIndex(ctx, [cust_id], unique=True)
    assert g.status == N.CPH_OK and g.first_dup is None
    perm = g.perm()
    # perm is a permutation and keys ascend: ids are fixed-width decimals, so sorted order == numeric order,
    # and id(row) = feistel_perm(row): the sorted position of row r must be its id
    ids = np.frombuffer(cust_id.data, dtype=np.uint8).reshape(n, 8) - ord("0")
    val = (ids.astype(np.int64) * (10 ** np.arange(7, -1, -1, dtype=np.int64))).sum(axis=1)
    np.testing.assert_array_equal(val[perm], np.arange(n, dtype=np.int64))
    m = g.probe([ords])
    assert m.nmatches == n and int(m.cnt.min()) == 1 == int(m.cnt.max())
    np.testing.assert_array_equal(m.probe_idx, np.arange(n, dtype=np.uint64))
    ov = (np.frombuffer(ords.data, dtype=np.uint8).reshape(n, 8) - ord("0")).astype(np.int64)
    ov = (ov * (10 ** np.arange(7, -1, -1, dtype=np.int64))).sum(axis=1)
    np.testing.assert_array_equal(val[m.build_row], ov)          # every pair joins equal keys
    np.testing.assert_array_equal(m.lo.astype(np.int64), ov)      # lo == sorted position == id
    info = g.info()
    # distinct keys over a full code space (1e7 states for 1e7 rows): one scatter, no radix pass (radix_sort.hip: direct_sort_distinct)
    assert info["sort_passes"] == 0 and info["key_bytes"] == 4 and info["code_bits"] == 24


def test_fixed_width_columns_match_variable_path(ctx):
    """cph_strcol.fixed_width (no offsets) must give the same bits as the same column with offsets."""
    n, m = 30_000, 70_000
    cust = dg.customers(n)["id"]
    ords = dg.orders(m, n, 50)["cust_id"]
    assert cust.fixed_width == 8 and ords.fixed_width == 8
    gf = DeviceIndex(ctx, [cust], unique=True)
    gv = DeviceIndex(ctx, [cust.as_variable()], unique=True)
    np.testing.assert_array_equal(gf.perm(), gv.perm())
    np.testing.assert_array_equal(gf.perm(), orc.OracleIndex([cust]).perm)
    oj = orc.OracleIndex([cust]).join([ords])
    for ix in (gf, gv):
        for probe in (ords, ords.as_variable(), ords.to_device(), ords.as_variable().to_device()):
            mt = ix.probe([probe])
            assert_join_equal(mt, oj)
    # fixed-width build side, variable-width probes of other lengths
    probe = StrCol.from_values([b"00000007", b"7", b"000000070", b"", b"00000003"])
    assert_join_equal(gf.probe([probe]), orc.OracleIndex([cust]).join([probe]))
    # 3-byte fixed-width keys with NULs and high bytes
    rng = np.random.default_rng(1)
    keys = [bytes(rng.integers(0, 256, 3).astype(np.uint8)) for _ in range(5000)]
    col = StrCol.from_values(keys)
    assert col.fixed_width == 3
    g, o = check_index(ctx, [col])
    assert_join_equal(g.probe([col]), o.join([col]))


def test_index_build_many_matches_single_builds(ctx):
    """cph_index_build_many = the same indexes as separate cph_index_build calls (one batch, two host round trips):
    perm bit-exact vs the oracle for mixed shapes, a duplicate in a `unique` spec reported per index."""
    rng = np.random.default_rng(5)
    tables = [
        [dg.customers(30_000)["id"]],                                   # fixed 8-byte unique ids
        [dg.products(700)["prod_id"]],                                  # variable-length decimal ids
        [StrCol.from_values(random_keys(rng, 5_000, 0, 30, alphabet=np.frombuffer(b"ab\x00\xff", dtype=np.uint8), distinct=300))],   # heavy duplicates, NUL / high bytes
        [StrCol.from_values([b"k%d" % (i % 50) for i in range(2000)]), StrCol.from_values([b"%d" % (i % 7) for i in range(2000)])],
        [StrCol.from_values([])],
    ]
    many = DeviceIndex.build_many(ctx, [(t, False) for t in tables])
    for t, ix in zip(tables, many):
        o = orc.OracleIndex(t)
        np.testing.assert_array_equal(ix.perm(), o.perm)
        single = DeviceIndex(ctx, t)
        np.testing.assert_array_equal(ix.perm(), single.perm())
        assert ix.first_dup == single.first_dup == o.first_dup()
        single.close()
    for ix in many:
        ix.close()
    # unique: index 1 has duplicates -> its status is DUPLICATE, the others are fine; all are returned
    specs = [(tables[0], True), (tables[2], True), (tables[1], True)]
    res = DeviceIndex.build_many(ctx, specs)
    assert [r.status for r in res] == [0, N.CPH_ERR_DUPLICATE, 0]
    assert res[1].first_dup == orc.OracleIndex(tables[2]).first_dup()
    for r in res:
        r.close()


def long_keys(rng, n, max_len, alphabet, distinct, shared_prefix):
    """Keys up to max_len bytes; many share a long prefix so that the order is decided deep inside the key."""
    pool = []
    base = alphabet[rng.integers(0, len(alphabet), max_len)].tobytes()
    for _ in range(distinct):
        ln = int(rng.integers(0, max_len + 1))
        k = bytearray(base[:ln]) if rng.random() < shared_prefix else bytearray(alphabet[rng.integers(0, len(alphabet), ln)].tobytes())
        for _ in range(int(rng.integers(0, 3))):   # a few point mutations, often far behind byte 128
            if ln:
                k[int(rng.integers(0, ln))] = int(alphabet[rng.integers(0, len(alphabet))])
        pool.append(bytes(k))
    return [pool[i] for i in rng.integers(0, distinct, n)]


@pytest.mark.parametrize("seed,max_len,ncols", [(1, 300, 1), (2, 1000, 1), (3, 260, 2), (4, 129, 1), (5, 400, 3)])
def test_keys_longer_than_one_codec_window(ctx, seed, max_len, ncols, tmp_path):
    """Keys of 0..1000 bytes: beyond 128 byte positions the key columns are cut into codec windows (SURVEY.md §7
    "hard parts").  Index order, duplicates, Join, prefix Join, Except and Find against the oracle."""
    rng = np.random.default_rng(seed)
    alphabet = np.frombuffer(b"ab\x00\xffz", dtype=np.uint8)
    n, m = 3000, 4000
    build_vals = [long_keys(rng, n, max_len if c == 0 else max_len // 3, alphabet, n // 3, 0.8) for c in range(ncols)]
    build = [StrCol.from_values(v) for v in build_vals]
    g, o = DeviceIndex(ctx, build), orc.OracleIndex(build)
    assert g.info()["key_positions"] > 128 or max(len(v) for v in build_vals[0]) <= 128
    np.testing.assert_array_equal(g.perm(), o.perm)
    assert g.first_dup == o.first_dup()
    # probes: half taken from the build side (mutated sometimes), half fresh
    for k in range(1, ncols + 1):
        probe_vals = []
        for c in range(k):
            src = build_vals[c]
            pv = [src[int(i)] if rng.random() < 0.6 else long_keys(rng, 1, max_len, alphabet, 1, 0.5)[0] for i in rng.integers(0, n, m)]
            pv[0] = src[0] + b"a"            # longer than anything at that prefix
            pv[1] = src[1][: len(src[1]) // 2]
            probe_vals.append(pv)
        probe = [StrCol.from_values(v) for v in probe_vals]
        mt, ej = g.probe(probe), o.join(probe)
        assert_join_equal(mt, ej)
        mt.release()
        for i in range(0, 40):
            vals = [probe_vals[c][i] for c in range(k)]
            glo, ghi = g.find(*vals)
            olo, ohi = o.find(*vals)
            assert ghi - glo == ohi - olo and (glo == olo or ghi == glo), (k, i)   # an empty range has no position
        mlo, mhi = g.find_many([tuple(probe_vals[c][i] for c in range(k)) for i in range(0, 200)])
        for i in range(0, 200):
            olo, ohi = o.find(*[probe_vals[c][i] for c in range(k)])
            assert mhi[i] - mlo[i] == ohi - olo and (ohi == olo or mlo[i] == olo), (k, i)
    # dup groups and select keep working on multi-window codes
    lo, hi = g.dup_groups()
    sel = g.select(sorted(set(range(0, n, 3))))
    assert sel.nrows == len(range(0, n, 3))
    np.testing.assert_array_equal(sel.perm(), g.perm()[::3])
    sel.close()
    # persistence keeps the windows: the loaded index sorts and probes like the built one
    path = tmp_path / "long.cph"
    g.save(str(path))
    ld = DeviceIndex.load(ctx, str(path))
    np.testing.assert_array_equal(ld.perm(), o.perm)
    probe = [StrCol.from_values(v[:500]) for v in build_vals]
    assert_join_equal(ld.probe(probe), o.join(probe))
    assert ld.info() == g.info()
    ld.close()
    g.close()


def test_small_table_build_path(ctx, both_build_paths):
    """The one-launch build (small_build.hip) takes tables of <= small_build_rows rows whose per-position code fits one
    word; everything else reports "not small" from the device and goes through the general path — same results."""
    small_on = both_build_paths == "small_path"
    rng = np.random.default_rng(77)
    cases = {
        "people_ids": ([StrCol.from_values(people_table()["id"])], True),
        "name_surname": (cols_of(people_table(), "name", "surname"), True),
        "orders_two_cols_10000": (cols_of(orders_table(), "cust_id", "prod_id"), True),
        "one_row": ([StrCol.from_values([b"x"])], True),
        "all_empty_values": ([StrCol.from_values([b""] * 100)], True),
        "fixed_width": ([StrCol.from_values([b"%08d" % int(x) for x in rng.permutation(5000)])], True),
        "variable_offsets64": ([StrCol.from_values([b"%d" % int(x) for x in rng.integers(0, 5000, 7000)], offset_bits=64)], True),
        "full_capacity_16384_dups": ([StrCol.from_values([b"k%d" % int(x) for x in rng.integers(0, 3000, 16384)])], True),
        "one_below_capacity": ([StrCol.from_values([b"%05d" % int(x) for x in rng.permutation(16383)])], True),
        "one_above_capacity": ([StrCol.from_values([b"%05d" % int(x) for x in rng.permutation(16385)])], False),
        "63_bit_codes": ([StrCol.from_values([bytes(rng.integers(97, 123, 13, dtype=np.uint8)) for _ in range(4000)])], True),
        "65_positions": ([StrCol.from_values([bytes([65 + (i % 3)]) * (1 + i % 65) for i in range(300)])], False),
        "two_code_words": ([StrCol.from_values(random_keys(rng, 3000, 14, 14, alphabet=list(range(48, 112))))], False),
        "above_the_row_limit": ([StrCol.from_values([b"%d" % int(x) for x in rng.integers(0, 1000, 20000)])], False),
    }
    for name, (cols, expect_small) in cases.items():
        g, o = check_index(ctx, cols)
        assert g.info()["build_path"] == (1 if small_on and expect_small else 0), name
        pr = [StrCol.from_values([c.value(i) for i in rng.integers(0, c.nrows, 500)] + [b"zz", b""]) for c in cols]
        assert_join_equal(g.probe(pr), o.join(pr))
        assert_bounds_equal(g, pr, o.join(pr))
        g.close()


def test_build_many_mixes_one_launch_and_general_jobs(ctx):
    """One cph_index_build_many batch with a small table (one-launch build), a table above the limit (general path), a
    small table whose key needs two code words (reports back, then the general path) and a duplicate in a unique spec."""
    ctx.set_option("small_build_rows", 8192)
    rng = np.random.default_rng(91)
    cols = [
        [StrCol.from_values([b"%d" % int(x) for x in rng.permutation(5000)])],                       # small
        [StrCol.from_values([b"%07d" % int(x) for x in rng.permutation(30000)])],                    # general
        [StrCol.from_values(random_keys(rng, 2000, 14, 14, alphabet=list(range(48, 112))))],         # small candidate, 2 words
        [StrCol.from_values([b"a", b"b", b"a", b"c"])],                                             # small, duplicate
    ]
    res = DeviceIndex.build_many(ctx, [(c, i == 3) for i, c in enumerate(cols)])
    assert [r.info()["build_path"] for r in res] == [1, 0, 0, 1]
    assert res[3].status == N.CPH_ERR_DUPLICATE and res[3].first_dup == 1
    for r, c in zip(res, cols):
        o = orc.OracleIndex(c)
        np.testing.assert_array_equal(r.perm(), o.perm)
        assert r.first_dup == o.first_dup()
        pr = [StrCol.from_values([c[0].value(i) for i in rng.integers(0, c[0].nrows, 300)] + [b"nope"])]
        assert_join_equal(r.probe(pr), o.join(pr))
        r.close()


@pytest.mark.parametrize("n", [1, 4095, 4096, 4097, 70_001, 1_300_003])
def test_scan_lookback_matches_three_kernel_scan(ctx, n):
    """radix_sort.hip k_scan_lookback (one launch, epoch-tagged state words, relative tickets) against the three-kernel scan
    it replaces: the same index, built with either, has the same order and first duplicate as the oracle's
    (csvplus.go:794-807).  Several builds on ONE ctx exercise the epoch / ticket bookkeeping across calls and across a
    growing state buffer."""
    rng = np.random.default_rng(n)
    keys = StrCol.from_values(random_keys(rng, n, 2, 6, alphabet=np.frombuffer(b"abcdefgh0123", np.uint8), distinct=min(n, 50_000)))
    o = orc.OracleIndex([keys])
    for mode in (1, 0, 1, 1):
        ctx.set_option("scan_lookback", mode)
        try:
            g = DeviceIndex(ctx, [keys])
            np.testing.assert_array_equal(g.perm(), o.perm)
            assert g.first_dup == o.first_dup()
        finally:
            ctx.set_option("scan_lookback", 1)


@pytest.mark.parametrize("side", [1, 0])
def test_build_many_on_two_streams(ctx, side):
    """cph_index_build_many enqueues every second build on the ctx's side stream (ctx option build_side_stream): concurrent
    statistics / encode / radix passes / one-launch scans of neighbouring jobs, pool blocks parked until both streams are idle.
    Same indexes as the oracle's (csvplus.go:707-756), batch after batch on one ctx, device-resident and host-resident columns,
    and the join that follows on the ctx's own stream sees finished indexes."""
    ctx.set_option("build_side_stream", side)
    try:
        rng = np.random.default_rng(17)
        tables = [
            [dg.customers(400_000)["id"]],
            [dg.products(100_000)["prod_id"]],
            [dg.varkeys(80_000)],                                                                       # delimiter-split candidate
            [StrCol.from_values([b"%d" % int(x) for x in rng.permutation(6000)])],                    # one-launch build
            [StrCol.from_values(random_keys(rng, 50_000, 0, 20, alphabet=np.frombuffer(b"abc\x00", np.uint8), distinct=900))],
        ]
        oracles = [orc.OracleIndex(t) for t in tables]
        dev = [[c.to_device("cuda:0") for c in t] for t in tables[:2]] + tables[2:]
        o = dg.orders(200_000, 400_000, 100_000)
        probe = [o["cust_id"], o["prod_id"]]
        for rep in range(3):
            res = DeviceIndex.build_many(ctx, [(t, False) for t in (dev if rep != 1 else tables)])
            for ix, oix in zip(res, oracles):
                np.testing.assert_array_equal(ix.perm(), oix.perm)
                assert ix.first_dup == oix.first_dup()
            for k in range(2):
                assert_join_equal(res[k].probe([probe[k]]), oracles[k].join([probe[k]]))
            for ix in res:
                ix.close()
    finally:
        ctx.set_option("build_side_stream", 1)


@pytest.mark.parametrize("rare_row", [None, 777_777])
def test_alphabets_from_a_sample_fixed_width_ids(ctx, rare_row):
    """IndexOn over one fixed-width key column of >= 2^20 rows takes its alphabets from a sample of the rows (ctx option
    stats_sample) instead of a pass over all of them; the encode kernel checks every row against them.  Same index as the oracle's
    (csvplus.go:794-807) when the sample showed every byte (no statistics pass ran) and when ONE row, which the sample does not
    visit, holds a byte seen nowhere else (the build notices and starts over with the exact pass)."""
    n = (1 << 20) + 12_345
    rng = np.random.default_rng(3)
    ids = rng.permutation(5 * n)[:n]
    raw = np.char.zfill(ids.astype("U8"), 8).astype("S8")
    data = np.frombuffer(raw.tobytes(), np.uint8).copy()
    if rare_row is not None:
        assert rare_row % (n >> 16) != 0                       # not a sampled row
        data[8 * rare_row + 3] = ord("A")
    col = StrCol.from_arrays(data, np.arange(n + 1, dtype=np.uint32) * 8, fixed_width=8)
    o = orc.OracleIndex([col])
    ctx.profile(True)
    ctx.profile_read(reset=True)
    g = DeviceIndex(ctx, [col.to_device("cuda:0")], unique=True)
    prof = ctx.profile_read(reset=True)
    ctx.profile(False)
    assert prof.get("k_col_stats", {"launches": 0})["launches"] == (0 if rare_row is None else 1), prof.keys()
    assert prof["k_split_count"]["launches"] == 1
    np.testing.assert_array_equal(g.perm(), o.perm)
    assert g.first_dup == o.first_dup()
    ctx.set_option("stats_sample", 0)
    try:
        g0 = DeviceIndex(ctx, [col.to_device("cuda:0")], unique=True)
        assert g0.info() == g.info()
        np.testing.assert_array_equal(g0.perm(), o.perm)
    finally:
        ctx.set_option("stats_sample", 1)


@pytest.mark.parametrize("shape", ["full_space", "dense_space", "duplicate", "not_unique_call", "tail_window", "scatter_full_space", "scatter_dense_space",
                                   "scatter_duplicate", "two_level", "two_level_duplicate", "fused_full_space"])
def test_direct_sort_of_distinct_keys_over_a_dense_code_space(ctx, shape):
    """UniqueIndexOn (csvplus.go:740-756) over ids that fill their code space densely sorts without radix passes: rows split by
    the top bits of their codes into LDS-sized windows, slot = code inside the window, windows streamed out (window_sort.hip;
    the round-4 variants — one random store per row, radix_sort.hip: direct_sort_distinct — stay as A/B switches and are
    checked here too): same perm as the oracle when the space is full (the slots ARE the permutation) or up to twice the rows
    (slots compacted); a duplicate is noticed on the device, the build starts over the general way and reports the oracle's
    first duplicate; a build that does not ask for distinct keys never takes the path."""
    rng = np.random.default_rng(11)
    if shape in ("full_space", "fused_full_space", "scatter_full_space"):
        ids = rng.permutation(100_000)                       # "00000".."99999": 10^5 states for 10^5 rows
        width = 5
    elif shape == "tail_window":
        ids = rng.permutation(70_001)                        # states 8 * 10^4 (first digit 0..7), the last window partly outside the ids
        width = 5
    elif shape.startswith("two_level"):
        ids = rng.permutation(1_200_000)                     # 2 x 10^6 states, 1.2e6 rows: slots beyond an L2 -> partition pass first
        width = 7
        if shape == "two_level_duplicate":
            ids[1_000_003] = ids[999]
    else:
        ids = rng.permutation(1_000_000)[:620_000]           # 10^6 states, 6.2e5 rows
        width = 6
    if shape.endswith("duplicate") and not shape.startswith("two_level"):
        ids[123_457] = ids[17]
    raw = np.char.zfill(ids.astype(f"U{width}"), width).astype(f"S{width}")
    col = StrCol.from_arrays(np.frombuffer(raw.tobytes(), np.uint8).copy(), np.arange(len(ids) + 1, dtype=np.uint32) * width, fixed_width=width)
    o = orc.OracleIndex([col])
    unique = shape != "not_unique_call"
    # 1 (default): LDS windows; A/B switches: 4 one random store per row, 2 that behind a partition pass, 3 the encode kernel fills the slots
    opt = 2 if shape.startswith("two_level") else 3 if shape == "fused_full_space" else 4 if shape.startswith("scatter") else 1
    ctx.set_option("direct_sort", opt)
    ctx.set_option("host_build", 0)   # (the device's own encode + sort kernels are what this test looks at: tests/test_gpu_host_build.py has the other path)
    ctx.profile(True)
    ctx.profile_read(reset=True)
    try:
        g = DeviceIndex(ctx, [col], unique=unique)
    finally:
        ctx.set_option("direct_sort", 1)
        ctx.set_option("host_build", 1)
    prof = ctx.profile_read(reset=True)
    ctx.profile(False)
    if opt == 1:
        took_direct = "k_win_partition" in prof and "k_win_place" in prof and "k_direct_scatter" not in prof
        assert prof.get("k_win_partition", {"launches": 1})["launches"] == 1   # these code spaces take one partition level
    else:   # (k_direct_finish closes every round-4 variant of the direct sort)
        took_direct = "k_direct_finish" in prof and ("k_direct_scatter" in prof) == (shape != "fused_full_space")
    took_radix = "k_radix_scatter_u32" in prof
    # (the partition pass of the two-level variant is one radix scatter)
    assert took_direct == unique and took_radix == (shape.endswith("duplicate") or shape in ("not_unique_call", "two_level")), sorted(prof)
    if shape == "two_level":
        assert prof["k_radix_scatter_u32"]["launches"] == 1
    np.testing.assert_array_equal(g.perm(), o.perm)
    assert g.first_dup == o.first_dup()
    assert g.status == (N.CPH_ERR_DUPLICATE if shape.endswith("duplicate") else N.CPH_OK)
    probe = [StrCol.from_values([col.value(int(i)) for i in rng.integers(0, col.nrows, 2000)] + [b"9" * width, b"12", b""])]
    assert_join_equal(g.probe(probe), o.join(probe))
    for v in (col.value(5), b"0" * width, b"5" * (width - 1)):
        assert g.find(v) == o.find(v) or (g.find(v)[0] == g.find(v)[1] and o.find(v)[0] == o.find(v)[1])
    if unique and not shape.endswith("duplicate"):   # the fused chain over this index, both output modes
        from csvplus_amd import join_chain

        want = o.join(probe)
        for positions in (False, True):
            ch = join_chain(ctx, [(g, probe)], positions=positions)
            np.testing.assert_array_equal(ch.stream_row, want["probe_idx"])
            rows = ch.build_row(0)
            np.testing.assert_array_equal(g.perm()[rows] if positions else rows, want["build_row"])
            ch.release()


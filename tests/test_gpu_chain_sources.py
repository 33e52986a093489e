"""Chained Joins whose later key is a column of an EARLIER BUILD TABLE (cph_chain_step.source != 0): the reference's flagship
chain people.Join(orders, "id").Join(products) (csvplus_test.go:280-285) reads prod_id from the ORDERS index row that
mergeRows (csvplus.go:571-583) copied into the joined row.  One device call per chain, bit-exact against the oracle's nested
joins (csvplus.go:545-569) in emission order."""
import numpy as np
import pytest

from csvplus_amd import DeviceIndex, StrCol, _native as N, join_chain
from oracle import orc
from tests.helpers import cols_of, orders_table, people_table, stock_table

pytestmark = pytest.mark.gpu


def oracle_chain_sources(oix, steps, probe_base=0):
    """steps[k] = (key columns, source): source 0 = stream columns, t + 1 = columns of step t's build table (original row
    order).  Returns (stream_row u64, [build_row_k u32]) in the reference's emission order."""
    j = oix[0].join(steps[0][0], probe_base=probe_base)
    stream = j["probe_idx"]
    rows = [j["build_row"]]
    for k in range(1, len(oix)):
        cols, src = steps[k]
        sel = (stream - probe_base).astype(np.uint32) if src == 0 else rows[src - 1]
        jk = oix[k].join(cols, row_sel=sel)
        pick = jk["probe_idx"].astype(np.int64)
        stream = stream[pick]
        rows = [r[pick] for r in rows] + [jk["build_row"]]
    return stream, rows


def take_rows(col: StrCol, rows) -> StrCol:
    vals = col.values()
    return StrCol.from_values([vals[int(i)] for i in rows], offset_bits=col.offset_bits)


def check(ctx, builds, steps, probe_base=0, expect_fused=None):
    """Both ways of answering a build-side key on the fused path: from tables pre-joined with each other (run_prejoined, round 5:
    one gather per stream row) and by the kernel that gathers and encodes the key per stream row (ctx option chain_prejoin = 0)."""
    try:
        for pj in (1, 0):
            ctx.set_option("chain_prejoin", pj)
            n = _check(ctx, builds, steps, probe_base, expect_fused, expect_prejoin=bool(pj) and bool(expect_fused))
    finally:
        ctx.set_option("chain_prejoin", 1)
    return n


def _check(ctx, builds, steps, probe_base=0, expect_fused=None, expect_prejoin=False):
    """builds[k] = key columns of table k; steps[k] = (columns, source >= 0).  Runs row ids, positions, and positions with the
    build-side columns laid out in sorted order (source -k)."""
    gix = [DeviceIndex(ctx, b) for b in builds]
    oix = [orc.OracleIndex(b) for b in builds]
    es, erows = oracle_chain_sources(oix, steps, probe_base)
    gsteps = [(g, c, s) for g, (c, s) in zip(gix, steps)]
    ctx.profile(True)
    ctx.profile_read(reset=True)
    ch = join_chain(ctx, gsteps, probe_base=probe_base)
    prof = ctx.profile_read(reset=True)
    ctx.profile(False)
    if expect_fused is not None:
        assert ("k_chain_dense" in prof) == expect_fused and any(k.startswith("k_probe") for k in prof) != expect_fused, sorted(prof)
        assert ("k_chain_prejoined" in prof) == expect_prejoin, sorted(prof)
    assert ch.nrows == len(es)
    np.testing.assert_array_equal(ch.stream_row, es)
    for k in range(len(gix)):
        np.testing.assert_array_equal(ch.build_row(k), erows[k])
    ch.release()
    # sorted positions; then the same with every build-side column permuted into its source index's order
    sorted_steps = []
    for g, c, s in gsteps:
        if s > 0:
            perm = gix[s - 1].perm()
            sorted_steps.append((g, [take_rows(x, perm) for x in c], -s))
        else:
            sorted_steps.append((g, c, s))
    for variant in (gsteps, sorted_steps):
        chp = join_chain(ctx, variant, probe_base=probe_base, positions=True)
        assert chp.positions and chp.nrows == len(es)
        np.testing.assert_array_equal(chp.stream_row, es)
        for k in range(len(gix)):
            pos = chp.build_row(k)
            np.testing.assert_array_equal(gix[k].perm()[pos], erows[k])
        chp.release()
    return len(es)


def test_long_chain_shape(ctx):
    """TestLongChain (csvplus_test.go:248-285): people.Join(orders on cust_id, "id").Join(products): duplicates on the first
    build side, the second key from the orders row.  The general chain, on the device."""
    people, orders, stock = people_table(), orders_table(), stock_table()
    n = check(ctx, [cols_of(orders, "cust_id"), cols_of(stock, "prod_id")],
              [(cols_of(people, "id"), 0), (cols_of(orders, "prod_id"), 1)], probe_base=5, expect_fused=False)
    assert n == len(orders["order_id"])   # every order has a customer and a product


@pytest.mark.parametrize("layout", ["varlen32", "varlen64", "fixed"])
def test_unique_chain_key_from_build_row_is_fused(ctx, layout):
    """orders.Join(customers, cust_id).Join(regions): region is a column of the CUSTOMERS table.  All indexes duplicate-free:
    one pass of the fused kernel, which gathers the region key from the customer row it matched."""
    rng = np.random.default_rng(17)
    nc, nreg, m = 30_000, 200, 200_000
    ob = {"varlen32": 32, "varlen64": 64, "fixed": 32}[layout]
    fmt = (lambda p, v: b"%s%06d" % (p, v)) if layout == "fixed" else (lambda p, v: b"%s%d" % (p, v))
    cust_ids = [fmt(b"c", i) for i in rng.permutation(40_000)[:nc]]
    cust_region = [fmt(b"r", i) for i in rng.integers(0, 260, nc)]          # some regions do not exist
    regions = [fmt(b"r", i) for i in rng.permutation(260)[:nreg]]
    okeys = [fmt(b"c", i) for i in rng.integers(0, 40_000, m)]               # some customers do not exist
    mk = lambda v: StrCol.from_values(v, offset_bits=ob)
    n = check(ctx, [[mk(cust_ids)], [mk(regions)]], [([mk(okeys)], 0), ([mk(cust_region)], 1)], probe_base=1000, expect_fused=True)
    assert 0 < n < m


def test_three_steps_mixed_sources(ctx):
    """stream -> a (stream key) -> b (key from a's row) -> c (key from a's row again) and -> c (key from b's row)."""
    rng = np.random.default_rng(23)
    na, nb, nc, m = 5000, 300, 40, 60_000
    a_id = [b"%d" % i for i in rng.permutation(6000)[:na]]
    a_b = [b"b%d" % i for i in rng.integers(0, 330, na)]
    a_c = [b"c%d" % i for i in rng.integers(0, 44, na)]
    b_id = [b"b%d" % i for i in rng.permutation(330)[:nb]]
    b_c = [b"c%d" % i for i in rng.integers(0, 44, nb)]
    c_id = [b"c%d" % i for i in rng.permutation(44)[:nc]]
    s_a = [b"%d" % i for i in rng.integers(0, 6000, m)]
    s_c = [b"c%d" % i for i in rng.integers(0, 44, m)]
    mk = StrCol.from_values
    builds = [[mk(a_id)], [mk(b_id)], [mk(c_id)]]
    for third in (([mk(a_c)], 1), ([mk(b_c)], 2), ([mk(s_c)], 0)):
        n = check(ctx, builds, [([mk(s_a)], 0), ([mk(a_b)], 1), third], expect_fused=True)
        assert 0 < n < m


def test_duplicates_then_build_side_key_multicolumn(ctx):
    """Duplicate keys on the first index and a two-column second key taken from its rows: the general path."""
    rng = np.random.default_rng(29)
    na, m = 4000, 3000
    a_k = [b"%d" % i for i in rng.integers(0, 500, na)]
    a_x = [b"%d" % i for i in rng.integers(0, 12, na)]
    a_y = [b"%c" % c for c in rng.integers(97, 101, na)]
    bx = [b"%d" % (i // 4) for i in range(40)]
    by = [b"%c" % (97 + i % 4) for i in range(40)]
    s_k = [b"%d" % i for i in rng.integers(0, 550, m)]
    mk = StrCol.from_values
    n = check(ctx, [[mk(a_k)], [mk(bx), mk(by)]], [([mk(s_k)], 0), ([mk(a_x), mk(a_y)], 1)], expect_fused=False)
    assert n > m


def test_source_errors(ctx):
    mk = StrCol.from_values
    a = DeviceIndex(ctx, [mk([b"1", b"2", b"3"])])
    b = DeviceIndex(ctx, [mk([b"x", b"y"])])
    s = [mk([b"1", b"3"])]
    good = [mk([b"x", b"y", b"x"])]
    for steps, positions in (
        ([(a, s, 1)], False),                          # step 0 reads the stream
        ([(a, s, 0), (b, good, 2)], False),            # a later step as source
        ([(a, s, 0), (b, good, -1)], False),           # sorted-order columns without positions
        ([(a, s, 0), (b, [mk([b"x", b"y"])], 1)], False),   # not the source table's row count
    ):
        with pytest.raises(N.CphError):
            join_chain(ctx, steps, positions=positions)
    ch = join_chain(ctx, [(a, s, 0), (b, good, 1)])
    assert ch.nrows == 2 and list(ch.build_row(1)) == [0, 0]
    ch.release()


@pytest.mark.parametrize("missing", [False, True])
@pytest.mark.parametrize("layout", ["itoa", "fixed"])
def test_duplicate_first_index_second_key_from_its_rows_is_prejoined(ctx, missing, layout):
    """(round 6) people.Join(IndexOn(orders.cust_id), "id").Join(UniqueIndexOn(products), prod_id of the ORDERS row) at a size where the
    general path answers the second Join from a table joined with the products index once (chain.hip: prejoin_general_step) instead
    of probing per tuple — bit-exact either way; with `missing` some orders name products that do not exist (tuples are dropped:
    the compaction pass) and some people have no orders."""
    rng = np.random.default_rng(61 + missing)
    npeople, nord, nprod = 4000, 50_000, 300
    fmt = (lambda p, v: b"%s%06d" % (p, v)) if layout == "fixed" else (lambda p, v: b"%s%d" % (p, v))
    people = [fmt(b"", i) for i in rng.permutation(npeople + 500)[:npeople]]
    o_cust = [fmt(b"", i) for i in rng.integers(0, npeople + 500, nord)]
    o_prod = [fmt(b"p", i) for i in rng.integers(0, nprod + (60 if missing else 0), nord)]
    prods = [fmt(b"p", i) for i in rng.permutation(nprod)]
    mk = StrCol.from_values
    builds = [[mk(o_cust)], [mk(prods)]]
    steps = [([mk(people)], 0), ([mk(o_prod)], 1)]
    gix = [DeviceIndex(ctx, b) for b in builds]
    ctx.profile(True)
    ctx.profile_read(reset=True)
    ch = join_chain(ctx, [(gix[0], steps[0][0], 0), (gix[1], steps[1][0], 1)], positions=bool(missing))
    prof = ctx.profile_read(reset=True)
    ctx.profile(False)
    assert "k_prejoin_tuples" in prof and "k_chain_prejoin_table" in prof, sorted(prof)
    assert ("k_compose" in prof) == missing, sorted(prof)      # nothing is compacted when every tuple survives
    ch.release()
    n = check(ctx, builds, steps, probe_base=3, expect_fused=False)
    assert (n < nord) if missing else (0 < n <= nord)

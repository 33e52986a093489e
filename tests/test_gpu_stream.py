"""cph_stream_join_*: chunked, multi-stream Join of a host-resident stream (BASELINE config 5 shape)
against the oracle, chunk by chunk."""
import numpy as np
import pytest

from csvplus_amd import DeviceIndex, StrCol, _native as N, datagen as dg
from csvplus_amd.streaming import PinnedCol, StreamJoin, bitmap_to_rows
from oracle import orc

pytestmark = pytest.mark.gpu


def oracle_chunk(oix, cols, probe_base):
    j = oix[0].join([cols[0]], probe_base=probe_base)
    stream, rows = j["probe_idx"], [j["build_row"]]
    for k in range(1, len(oix)):
        jk = oix[k].join([cols[k]], row_sel=(stream - probe_base).astype(np.uint32))
        pick = jk["probe_idx"].astype(np.int64)
        stream, rows = stream[pick], [r[pick] for r in rows] + [jk["build_row"]]
    return stream, rows


@pytest.mark.parametrize("pinned", [False, True])
def test_stream_join_chunks_match_oracle(ctx, pinned):
    nc, npd = 20_000, 300
    cust, prod = dg.customers(nc)["id"], dg.products(npd)["prod_id"]
    gix = [DeviceIndex(ctx, [cust], unique=True), DeviceIndex(ctx, [prod], unique=True)]
    oix = [orc.OracleIndex([cust]), orc.OracleIndex([prod])]
    sj = StreamJoin(ctx, gix, nslots=3)
    chunk_rows = [50_000, 1, 1024, 77_777, 1023, 200_000, 4096]
    chunks, base = [], 0
    for i, n in enumerate(chunk_rows):
        o = dg.orders(10**7, 2 * nc if i % 2 else nc, npd, row0=base, nrows=n)   # odd chunks: half the keys miss
        cols = [o["cust_id"], o["prod_id"]]
        pins = [PinnedCol(ctx, c) for c in cols] if pinned else None
        chunks.append((base, cols, pins))
        base += n
    results, submitted = [], 0
    while len(results) < len(chunks):
        while submitted < len(chunks) and sj.pending < sj.nslots:
            b, cols, pins = chunks[submitted]
            sj.submit([p.col for p in pins] if pins else cols, probe_base=b)
            submitted += 1
        results.append(sj.next())
    with pytest.raises(N.CphError):
        sj.next()
    for (b, cols, pins), r in zip(chunks, results):
        es, erows = oracle_chunk(oix, cols, b)
        assert r["probe_base"] == b and r["nrows"] == cols[0].nrows and r["nmatches"] == len(es)
        hit = bitmap_to_rows(r["bitmap"], r["nrows"])
        np.testing.assert_array_equal(hit + b, es.astype(np.int64))
        for k in range(2):
            np.testing.assert_array_equal(r["build_row"][k][hit], erows[k])
        if pins:
            for p in pins:
                p.free()
    sj.close()


def test_stream_join_slot_exhaustion_and_rejects_dup_index(ctx):
    cust = dg.customers(1000)["id"]
    ix = DeviceIndex(ctx, [cust], unique=True)
    sj = StreamJoin(ctx, [ix], nslots=2)
    col = dg.orders(5000, 1000, 10)["cust_id"]
    sj.submit([col]); sj.submit([col])
    with pytest.raises(N.CphError):
        sj.submit([col])
    a, b = sj.next(), sj.next()
    assert a["nmatches"] == b["nmatches"] == 5000
    sj.close()
    dup = DeviceIndex(ctx, [StrCol.from_values(["a", "a", "b"])])
    with pytest.raises(N.CphError) as e:
        StreamJoin(ctx, [dup])
    assert e.value.code == N.CPH_ERR_INVALID


def test_stream_join_returned_chunk_survives_later_submits(ctx):
    """The lifetime the header promises for a chunk handed out WITHOUT a copy: chunk k's pinned arrays are reused
    by chunk k + nslots only (slots are taken round robin).  A caller with nslots = 3 that keeps two chunks in
    flight reads chunk k while k+1 and k+2 run — the next -> submit -> process order a cgo caller would use."""
    nc = 5_000
    cust = dg.customers(nc)["id"]
    ix = DeviceIndex(ctx, [cust], unique=True)
    oix = orc.OracleIndex([cust])
    sj = StreamJoin(ctx, [ix], nslots=3)
    n = 60_000
    cols = [dg.orders(10**6, 2 * nc, 10, row0=i * n, nrows=n)["cust_id"] for i in range(5)]
    expect = [oix.join([c], probe_base=i * n) for i, c in enumerate(cols)]
    sj.submit([cols[0]], probe_base=0)
    sj.submit([cols[1]], probe_base=n)
    for k in range(5):
        r = sj.next(copy=False)                       # views of the slot's pinned arrays
        if k + 2 < 5:
            sj.submit([cols[k + 2]], probe_base=(k + 2) * n)   # submitted BEFORE chunk k is read
        import time
        time.sleep(0.05)                               # let the new chunk run: it must not touch chunk k's arrays
        hit = bitmap_to_rows(r["bitmap"].copy(), r["nrows"])
        np.testing.assert_array_equal(hit + k * n, expect[k]["probe_idx"].astype(np.int64))
        np.testing.assert_array_equal(r["build_row"][0][hit], expect[k]["build_row"])
        assert r["nmatches"] == expect[k]["nmatches"]
    sj.close()


def oracle_chain_general(oix, step_cols, probe_base):
    """The chain as nested Joins (csvplus.go:545-569): (stream_row, [build_row per step]) in emission order."""
    j = oix[0].join(step_cols[0], probe_base=probe_base)
    stream, rows = j["probe_idx"], [j["build_row"]]
    for k in range(1, len(oix)):
        jk = oix[k].join(step_cols[k], row_sel=(stream - probe_base).astype(np.uint32))
        pick = jk["probe_idx"].astype(np.int64)
        stream, rows = stream[pick], [r[pick] for r in rows] + [jk["build_row"]]
    return stream, rows


@pytest.mark.parametrize("shape", ["dup_build_side", "two_column_key_and_prefix", "long_keys_two_words"])
def test_stream_join_general_chains(ctx, shape):
    """cph_stream_join_create_general: chains the fused kernel rejects (TestLongChain's duplicate build side,
    csvplus_test.go:248-366; several key columns; a prefix join; multi-word codes), streamed chunk by chunk from ONE
    host table (chunks are row ranges: offsets do not start at 0) and compared with the oracle per chunk."""
    rng = np.random.default_rng({"dup_build_side": 5, "two_column_key_and_prefix": 6, "long_keys_two_words": 7}[shape])
    n_stream = 120_000
    if shape == "dup_build_side":
        # orders-like build side: many rows per customer; second step against a duplicate-free table
        b0 = [StrCol.from_values([b"%d" % int(x) for x in rng.integers(0, 300, 2000)])]
        b1 = [StrCol.from_values([b"p%03d" % i for i in range(50)])]
        s0 = [StrCol.from_values([b"%d" % int(x) for x in rng.integers(0, 400, n_stream)])]
        s1 = [StrCol.from_values([b"p%03d" % int(x) for x in rng.integers(0, 60, n_stream)])]
        ncols = [1, 1]
    elif shape == "two_column_key_and_prefix":
        names = [b"amelia", b"olivia", b"jack", b"harry", b"isla"]
        b0 = [StrCol.from_values([names[int(i)] for i in rng.integers(0, 5, 3000)]),
              StrCol.from_values([b"%d" % int(x) for x in rng.integers(0, 40, 3000)])]
        b1 = [StrCol.from_values([b"%d" % int(x) for x in rng.integers(0, 30, 500)]),
              StrCol.from_values([b"x%d" % int(x) for x in rng.integers(0, 9, 500)])]
        s0 = [StrCol.from_values([names[int(i)] if i < 5 else b"nobody" for i in rng.integers(0, 6, n_stream)]),
              StrCol.from_values([b"%d" % int(x) for x in rng.integers(0, 45, n_stream)])]
        s1 = [StrCol.from_values([b"%d" % int(x) for x in rng.integers(0, 33, n_stream)])]   # prefix of b1's key
        ncols = [2, 1]
    else:
        pool = [bytes(rng.integers(97, 123, int(rng.integers(18, 30)), dtype=np.uint8)) for _ in range(4000)]
        b0 = [StrCol.from_values([pool[int(i)] for i in rng.integers(0, 3000, 6000)])]
        b1 = [StrCol.from_values([b"%d" % i for i in range(100)])]
        s0 = [StrCol.from_values([pool[int(i)] for i in rng.integers(0, 4000, n_stream)])]
        s1 = [StrCol.from_values([b"%d" % int(x) for x in rng.integers(0, 100, n_stream)])]
        ncols = [1, 1]
    gix = [DeviceIndex(ctx, b0), DeviceIndex(ctx, b1)]
    oix = [orc.OracleIndex(b0), orc.OracleIndex(b1)]
    sj = StreamJoin(ctx, gix, nslots=3, ncols=ncols)
    bounds = [0, 1, 5000, 5001, 40_000, 90_000, 119_999, n_stream]
    chunks = [(bounds[i], [[c.slice(bounds[i], bounds[i + 1]) for c in s0], [c.slice(bounds[i], bounds[i + 1]) for c in s1]])
              for i in range(len(bounds) - 1)]
    results, submitted = [], 0
    while len(results) < len(chunks):
        while submitted < len(chunks) and sj.pending < sj.nslots:
            b, sc = chunks[submitted]
            sj.submit(sc[0] + sc[1], probe_base=b)
            submitted += 1
        results.append(sj.next())
    total = 0
    for (b, sc), r in zip(chunks, results):
        es, erows = oracle_chain_general(oix, sc, b)
        assert not r["dense"] and r["probe_base"] == b and r["nrows"] == sc[0][0].nrows
        assert r["nmatches"] == len(es), (shape, b)
        np.testing.assert_array_equal(r["stream_row"], es)
        for k in range(2):
            np.testing.assert_array_equal(r["build_row"][k], erows[k])
        total += len(es)
    assert total > 0
    sj.close()
    # the plain constructor still refuses such chains, the general one runs fused-kernel chains in the dense mode
    with pytest.raises(N.CphError):
        StreamJoin(ctx, gix)
    u = DeviceIndex(ctx, [StrCol.from_values([b"%d" % i for i in range(100)])], unique=True)
    sj2 = StreamJoin(ctx, [u], nslots=2, ncols=[1])
    sj2.submit([s1[0].slice(0, 1000)] if shape != "two_column_key_and_prefix" else [s0[1].slice(0, 1000)])
    assert sj2.next()["dense"]
    sj2.close()


@pytest.mark.parametrize("general", [False, True])
def test_stream_join_reports_positions(ctx, general):
    """cph_stream_join_set_positions: chunks carry sorted positions; perm[position] is the row the default mode reports."""
    rng = np.random.default_rng(12)
    cust = dg.customers(30_000)["id"]
    prod = StrCol.from_values([b"p%d" % int(x) for x in (rng.integers(0, 200, 900) if general else rng.permutation(900))])
    gix = [DeviceIndex(ctx, [cust], unique=True), DeviceIndex(ctx, [prod], unique=not general)]
    o = dg.orders(200_000, 40_000, 10)
    s1 = StrCol.from_values([b"p%d" % int(x) for x in rng.integers(0, 1000, 200_000)])
    kw = dict(ncols=[1, 1]) if general else {}
    out = {}
    for pos in (False, True):
        sj = StreamJoin(ctx, gix, nslots=2, positions=pos, **kw)
        res = []
        for b in (0, 70_000, 140_000):
            sj.submit([o["cust_id"].slice(b, b + 70_000 if b < 140_000 else 200_000), s1.slice(b, b + 70_000 if b < 140_000 else 200_000)], probe_base=b)
            res.append(sj.next())
        out[pos] = res
        sj.close()
    perms = [g.perm() for g in gix]
    for a, b in zip(out[False], out[True]):
        assert a["nmatches"] == b["nmatches"] and a["dense"] == b["dense"] == (not general)
        if general:
            np.testing.assert_array_equal(a["stream_row"], b["stream_row"])
            for k in range(2):
                np.testing.assert_array_equal(perms[k][b["build_row"][k]], a["build_row"][k])
        else:
            np.testing.assert_array_equal(a["bitmap"], b["bitmap"])
            hit = bitmap_to_rows(a["bitmap"], a["nrows"])
            for k in range(2):
                np.testing.assert_array_equal(perms[k][b["build_row"][k][hit]], a["build_row"][k][hit])
    assert sum(r["nmatches"] for r in out[True]) > 0


def test_stream_join_general_positions_two_chunks_in_flight(ctx):
    """General mode + positions with SEVERAL chunks in flight (round-3 advisor finding): the slot workers used to build the
    rank table lazily and concurrently; since round 4 cph_stream_join_set_positions builds it up front and index_ensure_* is
    locked per index.  Results against the row-id mode through perm."""
    rng = np.random.default_rng(21)
    cust = dg.customers(50_000)["id"]                                      # duplicate-free, dense: positions through the rank table
    dup = StrCol.from_values([b"p%d" % int(x) for x in rng.integers(0, 300, 2000)])   # duplicates: makes the chain general
    gix = [DeviceIndex(ctx, [cust], unique=True), DeviceIndex(ctx, [dup])]
    o = dg.orders(400_000, 60_000, 10)
    s1 = StrCol.from_values([b"p%d" % int(x) for x in rng.integers(0, 330, 400_000)])
    bounds = [(b, min(b + 50_000, 400_000)) for b in range(0, 400_000, 50_000)]
    out = {}
    for pos in (False, True):
        sj = StreamJoin(ctx, gix, nslots=4, positions=pos, ncols=[1, 1])
        res, sub = [], 0
        while len(res) < len(bounds):
            while sub < len(bounds) and sj.pending < 4:                     # four chunks in flight: four workers at once
                b, e = bounds[sub]
                sj.submit([o["cust_id"].slice(b, e), s1.slice(b, e)], probe_base=b)
                sub += 1
            res.append(sj.next())
        out[pos] = res
        sj.close()
    perms = [g.perm() for g in gix]
    for a, b in zip(out[False], out[True]):
        assert a["nmatches"] == b["nmatches"] > 0 and not a["dense"] and not b["dense"]
        np.testing.assert_array_equal(a["stream_row"], b["stream_row"])
        for k in range(2):
            np.testing.assert_array_equal(perms[k][b["build_row"][k]], a["build_row"][k])


@pytest.mark.parametrize("positions", [False, True])
def test_stream_join_host_formed_codes(ctx, positions):
    """cph_host_encoder_* + cph_stream_join_submit_codes: the chunks travel as 4-byte codes formed on the host; results equal
    the string pipeline's, chunk by chunk, including keys with bytes outside the alphabets, too long / too short values."""
    from csvplus_amd.streaming import HostEncoder, PinnedArray

    rng = np.random.default_rng(31)
    nc, npd = 30_000, 700
    cust, prod = dg.customers(nc)["id"], dg.products(npd)["prod_id"]      # fixed-width 8-byte ids / unpadded decimal ids
    gix = [DeviceIndex(ctx, [cust], unique=True), DeviceIndex(ctx, [prod], unique=True)]
    encs = [HostEncoder(g, nthreads=t) for g, t in zip(gix, (3, 0))]
    assert encs[0].threads == 3 and encs[1].threads >= 1
    m = 300_000
    o = dg.orders(m, 2 * nc, npd + 50)
    cv, pv = o["cust_id"].values(), o["prod_id"].values()
    for j in range(0, m, 97):
        cv[j] = [b"0000A000", b"\x0012345\xff", b"00000/00", b"0000:000", b"99999999"][j % 5]
    for j in range(0, m, 89):
        pv[j] = [b"", b"7x", b"1234567", b"-1", b"00"][j % 5]
    cols = [StrCol.from_values(cv), StrCol.from_values(pv)]
    bounds = [(0, 100_000), (100_000, 100_001), (100_001, 223_456), (223_456, m)]
    ref, got = [], []
    sj = StreamJoin(ctx, gix, nslots=2, positions=positions)
    for b, e in bounds:
        sj.submit([c.slice(b, e) for c in cols], probe_base=b)
        ref.append(sj.next())
    pins = [[PinnedArray(ctx, e - b) for _ in range(2)] for b, e in bounds]
    sub = 0
    while len(got) < len(bounds):
        while sub < len(bounds) and sj.pending < 2:
            b, e = bounds[sub]
            for k in range(2):
                encs[k].run([cols[k].slice(b, e)], pins[sub][k].array)
            sj.submit_codes([p.array for p in pins[sub]], e - b, probe_base=b)
            sub += 1
        got.append(sj.next())
    sj.close()
    total = 0
    for a, b in zip(ref, got):
        assert a["nmatches"] == b["nmatches"] and b["dense"]
        np.testing.assert_array_equal(a["bitmap"], b["bitmap"])
        hit = bitmap_to_rows(a["bitmap"], a["nrows"])
        total += len(hit)
        for k in range(2):
            np.testing.assert_array_equal(a["build_row"][k][hit], b["build_row"][k][hit])
    assert 0 < total < m
    # the codes themselves: ABSENT exactly where the oracle finds nothing for that column alone
    codes = np.zeros(m, np.uint32)
    encs[0].run([cols[0]], codes)
    oj = orc.OracleIndex([cust]).join([cols[0]])
    assert ((codes == 0xFFFFFFFF) <= (oj["cnt"] == 0)).all()              # an ABSENT code never hides a match
    for p in sum(pins, []):
        p.free()
    for e_ in encs:
        e_.close()
    # an index whose keys need a dictionary / several words is refused
    long_ix = DeviceIndex(ctx, [StrCol.from_values([bytes(rng.integers(97, 123, 30).astype(np.uint8)) for _ in range(500)])])
    with pytest.raises(N.CphError):
        HostEncoder(long_ix)

"""cph_join_chain (chained Join on the device) against the oracle's nested joins
(csvplus.go:545-569 nested as in README.md:56)."""
import numpy as np
import pytest

from csvplus_amd import DeviceIndex, StrCol, _native as N, datagen as dg, join_chain
from oracle import orc
from tests.helpers import random_keys

pytestmark = pytest.mark.gpu


def oracle_chain(indexes, keycols, probe_base=0):
    """Nested oracle joins: returns (stream_row u64, [build_row_k u32]) in emission order."""
    j = indexes[0].join([keycols[0]] if isinstance(keycols[0], StrCol) else keycols[0], probe_base=probe_base)
    stream = j["probe_idx"]
    rows = [j["build_row"]]
    for k in range(1, len(indexes)):
        sel = (stream - probe_base).astype(np.uint32)
        cols = [keycols[k]] if isinstance(keycols[k], StrCol) else keycols[k]
        jk = indexes[k].join(cols, row_sel=sel)
        pick = jk["probe_idx"].astype(np.int64)
        stream = stream[pick]
        rows = [r[pick] for r in rows] + [jk["build_row"]]
    return stream, rows


def check_chain(ctx, build_tables, stream_keys, probe_base=0, expect_fast=None):
    gix = [DeviceIndex(ctx, cols) for cols in build_tables]
    oix = [orc.OracleIndex(cols) for cols in build_tables]
    steps = [(g, k if isinstance(k, list) else [k]) for g, k in zip(gix, stream_keys)]
    ctx.profile(True)
    ctx.profile_read(reset=True)
    ch = join_chain(ctx, steps, probe_base=probe_base)
    prof = ctx.profile_read(reset=True)
    ctx.profile(False)
    if expect_fast is not None:   # one fused pass, or the general chain (probe / select / compose per step) on the device
        assert ("k_chain_dense" in prof) == expect_fast and any(k.startswith("k_probe") for k in prof) != expect_fast, sorted(prof)
    es, erows = oracle_chain(oix, [k if isinstance(k, list) else k for k in stream_keys], probe_base)
    assert ch.nrows == len(es)
    np.testing.assert_array_equal(ch.stream_row, es)
    for k in range(len(gix)):
        np.testing.assert_array_equal(ch.build_row(k), erows[k])
    # the same chain reporting SORTED POSITIONS (cph_join_chain_ex CPH_CHAIN_POSITIONS): perm[position] is the row
    chp = join_chain(ctx, steps, probe_base=probe_base, positions=True)
    assert chp.positions and not ch.positions and chp.nrows == len(es)
    np.testing.assert_array_equal(chp.stream_row, es)
    for k in range(len(gix)):
        pos = chp.build_row(k)
        assert len(pos) == 0 or int(pos.max()) < gix[k].nrows
        np.testing.assert_array_equal(gix[k].perm()[pos], erows[k])
    chp.release()
    return ch


def test_chain_unique_fast_path_config4_shape(ctx):
    """orders JOIN customers JOIN products, all keys present: one fused pass."""
    nc, npd, m = 50_000, 1000, 300_000
    cust, prod = dg.customers(nc), dg.products(npd)
    ords = dg.orders(m, nc, npd)
    ch = check_chain(ctx, [[cust["id"]], [prod["prod_id"]]], [ords["cust_id"], ords["prod_id"]], probe_base=12345)
    assert ch.nrows == m


def test_chain_unique_with_misses(ctx):
    """Inner-join semantics: rows missing in either index disappear; order stays stream order."""
    rng = np.random.default_rng(5)
    a = [b"%05d" % i for i in rng.permutation(20000)[:12000]]
    b = [b"k%d" % i for i in rng.permutation(500)[:300]]
    m = 100_000
    ka = [b"%05d" % i for i in rng.integers(0, 20000, m)]
    kb = [b"k%d" % i for i in rng.integers(0, 500, m)]
    ka[::13] = [b"zz"] * len(ka[::13])          # symbols outside the alphabet
    kb[::17] = [b"k1234567"] * len(kb[::17])     # longer than any index key
    ch = check_chain(ctx, [[StrCol.from_values(a)], [StrCol.from_values(b)]],
                     [StrCol.from_values(ka), StrCol.from_values(kb)])
    assert 0 < ch.nrows < m


@pytest.mark.parametrize("nsteps", [1, 3, 4, 5, 8])
def test_chain_lengths(ctx, nsteps):
    """1-4 Joins: one fused pass; 5-8 (CPH_MAX_CHAIN, round 5): the general chain on the device, still ONE call."""
    rng = np.random.default_rng(nsteps)
    m = 50_000
    builds, keys = [], []
    for s in range(nsteps):
        dom = 1000 * (s + 1)
        vals = [b"%d" % i for i in rng.permutation(dom)[: dom * 3 // 4]]
        builds.append([StrCol.from_values(vals)])
        keys.append(StrCol.from_values([b"%d" % i for i in rng.integers(0, dom, m)]))
    check_chain(ctx, builds, keys, expect_fast=nsteps <= 4)
    if nsteps == 8:   # one Join more than a call takes
        gix = [DeviceIndex(ctx, b) for b in builds] + [DeviceIndex(ctx, builds[0])]
        with pytest.raises(Exception, match="bad chain"):
            join_chain(ctx, [(g, [k]) for g, k in zip(gix, keys + [keys[0]])])


def test_chain_search_path_sparse_codes(ctx):
    """Distinct keys but a sparse code space (no direct table): hash-table lookups inside the fused pass."""
    rng = np.random.default_rng(11)
    vals = list({bytes(v) for v in random_keys(rng, 30000, 6, 9, alphabet=list(b"abcdefghijklmnopqrstuvwxyz"))})
    ix = DeviceIndex(ctx, [StrCol.from_values(vals)])
    assert ix.info()["direct_table"] == 0 and ix.first_dup is None
    probe = [vals[i] for i in rng.integers(0, len(vals), 80000)]
    probe[::9] = random_keys(rng, len(probe[::9]), 6, 9, alphabet=list(b"abcxyz"))
    check_chain(ctx, [[StrCol.from_values(vals)]], [StrCol.from_values(probe)])
    # the same with the hash probe switched off on a second ctx: the sorted-search fallback inside the fused kernel
    from csvplus_amd import Context
    c2 = Context(0)
    c2.set_option("join_hash", 0)
    check_chain(c2, [[StrCol.from_values(vals)]], [StrCol.from_values(probe)])
    c2.close()


def test_chain_general_path_duplicates(ctx):
    """Duplicate build keys: every combination is emitted, a-matches outer, b-matches inner."""
    rng = np.random.default_rng(3)
    a = [b"%d" % i for i in rng.integers(0, 300, 2000)]      # ~7 dups per key
    b = [b"p%d" % i for i in rng.integers(0, 50, 200)]       # ~4 dups per key
    m = 5000
    ka = [b"%d" % i for i in rng.integers(0, 330, m)]
    kb = [b"p%d" % i for i in rng.integers(0, 55, m)]
    ch = check_chain(ctx, [[StrCol.from_values(a)], [StrCol.from_values(b)]],
                     [StrCol.from_values(ka), StrCol.from_values(kb)], probe_base=77)
    assert ch.nrows > m


def test_chain_general_path_multicolumn(ctx):
    rng = np.random.default_rng(9)
    a0 = [b"%d" % i for i in rng.integers(0, 40, 3000)]
    a1 = [b"%c" % c for c in rng.integers(97, 101, 3000)]
    b = [b"%d" % i for i in range(100)]
    m = 8000
    k0 = [b"%d" % i for i in rng.integers(0, 45, m)]
    k1 = [b"%c" % c for c in rng.integers(97, 102, m)]
    kb = [b"%d" % i for i in rng.integers(0, 120, m)]
    check_chain(ctx, [[StrCol.from_values(a0), StrCol.from_values(a1)], [StrCol.from_values(b)]],
                [[StrCol.from_values(k0), StrCol.from_values(k1)], StrCol.from_values(kb)])


def test_chain_empty_and_no_match(ctx):
    ix = DeviceIndex(ctx, [StrCol.from_values(["a", "b"])])
    ch = join_chain(ctx, [(ix, [StrCol.from_values([])])])
    assert ch.nrows == 0
    ch = join_chain(ctx, [(ix, [StrCol.from_values(["x", "y", "zz"])])])
    assert ch.nrows == 0
    e = DeviceIndex(ctx, [StrCol.from_values([])])
    ch = join_chain(ctx, [(e, [StrCol.from_values(["x", ""])])])
    assert ch.nrows == 0


def test_chain_many_tiles_lookback(ctx):
    """5e6 rows = ~4900 tiles: exercises the decoupled look-back across many workgroups, with a
    match rate that varies along the stream; the result must equal stream order exactly."""
    n, m = 200_000, 5_000_000
    cust = dg.column(dg.SEQ_PERM, n, n, encoding=dg.FIXED8, seed=1)
    probe = dg.column(dg.UNIFORM, m, 2 * n, encoding=dg.FIXED8, seed=2)   # ~half the keys miss
    ix = DeviceIndex(ctx, [cust], unique=True)
    for _ in range(3):   # repeated: scheduling differs run to run, the result must not
        ch = join_chain(ctx, [(ix, [probe])])
        pv = (np.frombuffer(probe.data, dtype=np.uint8).reshape(m, 8) - 48).astype(np.int64)
        pv = (pv * (10 ** np.arange(7, -1, -1, dtype=np.int64))).sum(axis=1)
        hit = np.nonzero(pv < n)[0]
        np.testing.assert_array_equal(ch.stream_row, hit.astype(np.uint64))
        cv = (np.frombuffer(cust.data, dtype=np.uint8).reshape(n, 8) - 48).astype(np.int64)
        cv = (cv * (10 ** np.arange(7, -1, -1, dtype=np.int64))).sum(axis=1)
        np.testing.assert_array_equal(cv[ch.build_row(0)], pv[hit])
        ch.release()


def test_chain_identity_has_no_stream_row_array(ctx):
    """Every stream row joins exactly once: cph_chain.stream_row is NULL (row m == stream row base+m)."""
    nc, m = 5000, 40_000
    cust = dg.customers(nc)["id"]
    ords = dg.orders(m, nc, 10)["cust_id"]
    ix = DeviceIndex(ctx, [cust], unique=True)
    ch = join_chain(ctx, [(ix, [ords])], probe_base=500)
    assert ch.nrows == m and ch.identity
    np.testing.assert_array_equal(ch.stream_row, np.arange(500, 500 + m, dtype=np.uint64))
    # one missing key: the array is materialised
    bad = StrCol.from_values([b"99999999"] + ords.values()[1:])
    ch = join_chain(ctx, [(ix, [bad])], probe_base=500)
    assert ch.nrows == m - 1 and not ch.identity
    np.testing.assert_array_equal(ch.stream_row, np.arange(501, 500 + m, dtype=np.uint64))


def test_config4_properties_3e7(ctx):
    """BASELINE config 4 shape at 3e7 orders x 1e7 customers x 1e5 products (the oracle would need minutes): checked
    through size-independent properties — every order joins exactly once, in stream order, and the customer /
    product row it is paired with carries exactly the order's key bytes (verified with numpy on the host)."""
    m, nc, npd = 30_000_000, 10_000_000, 100_000
    cust, prod = dg.customers(nc), dg.products(npd)            # FIXED8 ids, ITOA product ids
    ords = dg.orders(m, nc, npd)
    ia, ib = DeviceIndex(ctx, [cust["id"]], unique=True), DeviceIndex(ctx, [prod["prod_id"]], unique=True)
    assert ia.status == 0 and ib.status == 0
    ch = join_chain(ctx, [(ia, [ords["cust_id"]]), (ib, [ords["prod_id"]])])
    assert ch.nrows == m and ch.identity                       # row i of the result IS order i
    a, b = ch.build_row(0).astype(np.int64), ch.build_row(1).astype(np.int64)
    # customers: fixed 8-byte ids -> compare the key bytes as one u64 per row
    cid = cust["id"].data[: nc * 8].view(np.uint64)
    oid = ords["cust_id"].data[: m * 8].view(np.uint64)
    assert np.array_equal(cid[a], oid)
    # products: variable-length decimal ids -> same length and same bytes (zero-padded to 8)
    def padded(col):
        offs = col.offsets.astype(np.int64)
        lens = np.diff(offs)
        out = np.zeros((col.nrows, 8), dtype=np.uint8)
        out[np.repeat(np.arange(col.nrows), lens), np.arange(offs[-1]) - np.repeat(offs[:-1], lens)] = col.data[: offs[-1]]
        return out.view(np.uint64).ravel(), lens
    pw, pl = padded(prod["prod_id"])
    ow, ol = padded(ords["prod_id"])
    assert np.array_equal(pw[b], ow) and np.array_equal(pl[b], ol)


def test_config4_full_size_1e8(ctx):
    """The configuration bench.py times, at its full size: 1e8 orders x 1e7 customers x 1e5 products with the
    columns resident in HBM.  Every order joins exactly once; ALL customer pairings are checked on the device
    (fixed 8-byte keys: one u64 compare per row), the variable-length product keys on a 2e5-row sample, both
    indexes through verify.check_index_order (permutation, ascending keys, stable ties)."""
    import torch

    from csvplus_amd import verify as V
    from csvplus_amd.engine import Engine, device_view

    m, nc, npd = 100_000_000, 10_000_000, 100_000
    eng = Engine(0)
    dev = eng.device
    cust_id = dg.column(dg.SEQ_PERM, nc, nc, encoding=dg.FIXED8, seed=dg.SEED + 1)
    prod_id = dg.column(dg.SEQ_PERM, npd, npd, encoding=dg.ITOA, seed=dg.SEED + 2)
    ords = dg.orders(m, nc, npd)
    d_cust, d_prod = cust_id.to_device(dev), prod_id.to_device(dev)
    d_oc, d_op = ords["cust_id"].to_device(dev), ords["prod_id"].to_device(dev)
    ia, ib = eng.index_on([d_cust], unique=True), eng.index_on([d_prod], unique=True)
    for ix, col in ((ia, d_cust), (ib, d_prod)):
        r = V.check_index_order(col, device_view(ix.perm_device_ptr(), ix.nrows, "<i4", ix, dev))
        assert r["ok"], r
    res = eng.chained_join([(ia, d_oc), (ib, d_op)])
    assert res.n == m and res.stream_row is None
    # customers, every row: key of the paired customer == the order's cust_id (csvplus.go:553-567)
    ck = d_cust.data[: nc * 8].view(torch.int64)
    ok_ = d_oc.data[: m * 8].view(torch.int64)
    a = res.build_rows[0].to(torch.int64) & 0xFFFFFFFF
    assert bool((a < nc).all().item())
    assert bool((ck[a] == ok_).all().item())
    del a
    rows = V.sample_rows(m, 200_000, seed=7)
    b = res.build_rows[1][torch.from_numpy(rows).to(dev)].cpu().numpy()
    assert V.check_join_sample(ords["prod_id"], prod_id, b, rows) == 0
    res.release()
    ia.close()
    ib.close()
    eng.close()


def test_device_column_without_a_data_buffer(ctx):
    """A device-resident column whose values are all empty may come with data == NULL (only offsets): the
    branch-free value loads must not touch address 0 — the library substitutes a readable base."""
    import torch

    class NullData:
        def data_ptr(self):
            return 0

    n = 5000
    offs = torch.zeros(n + 1, dtype=torch.int32, device="cuda").view(torch.uint8)
    empty = StrCol(NullData(), offs, n, 32, N.CPH_MEM_DEVICE, fixed_width=0)
    cust = dg.customers(1000)["id"]
    ix = DeviceIndex(ctx, [cust], unique=True)
    ch = join_chain(ctx, [(ix, [empty])])
    assert ch.nrows == 0
    ch.release()
    m = ix.probe([empty])
    assert m.nmatches == 0
    m.release()
    ex = DeviceIndex(ctx, [empty])                 # all keys equal "": one group, input order kept
    assert ex.perm().tolist() == list(range(n)) and ex.first_dup == 1
    mm = ex.probe([StrCol.from_values(["", "x"])])
    assert mm.cnt.tolist() == [n, 0]
    mm.release()
    ex.close()
    ix.close()


def test_profile_only_times_one_kernel(ctx):
    """cph_ctx_profile_only: HIP events around ONE kernel's launches (bench.py's timed region), nothing else."""
    cust, prod = dg.customers(20_000)["id"], dg.products(50)["prod_id"]
    o = dg.orders(100_000, 20_000, 50)

    def one():
        ia, ib = DeviceIndex(ctx, [cust], unique=True), DeviceIndex(ctx, [prod], unique=True)
        ch = join_chain(ctx, [(ia, [o["cust_id"]]), (ib, [o["prod_id"]])], out_mem=N.CPH_MEM_DEVICE)
        n = ch.nrows
        ch.release(); ia.close(); ib.close()
        return n

    one()
    ctx.profile(True)
    ctx.profile_read(reset=True)
    assert one() == 100_000
    full = ctx.profile_read(reset=True)
    assert "k_chain_dense" in full and len(full) > 3
    ctx.profile_only("k_chain_dense")
    one(); one()
    only = ctx.profile_read(reset=True)
    assert list(only) == ["k_chain_dense"] and only["k_chain_dense"]["launches"] == 2 and only["k_chain_dense"]["total_ms"] > 0
    ctx.profile(False)
    one()
    assert ctx.profile_read(reset=True) == {}


# ---- lean steps of the fused kernel (round 4): fixed-width 8-byte stream keys over contiguous alphabets are encoded
# arithmetically (codec_device.hpp: ArithPlan) and, reporting positions, answered from the code itself when the index
# fills its code space (identity) or from a rank table ----------------------------------------------------------------
def _fixed8(vals):
    col = StrCol.from_values(vals)
    assert col.fixed_width == 8
    return col


def _nasty_fixed8_probes(rng, good, m):
    """Stream keys for an index of decimal 8-byte ids: hits, misses inside the alphabets, and bytes chosen to trip a
    bytewise-parallel range check — just below '0', just above '9', NUL, 0x7F, 0x80, 0xFF, and pairs where a borrow /
    carry out of one byte would repair or spoil its neighbour."""
    out = [good[i] for i in rng.integers(0, len(good), m)]
    bad_bytes = [0x2F, 0x3A, 0x00, 0x7F, 0x80, 0xFF, 0x30 + 10, 0x20, 0xB0, 0xC6]
    for j in range(0, m, 3):
        v = bytearray(out[j])
        p = int(rng.integers(0, 8))
        v[p] = bad_bytes[int(rng.integers(0, len(bad_bytes)))]
        if j % 2 and p + 1 < 8:
            v[p + 1] = [0x30, 0x39, 0x2F, 0x3A][int(rng.integers(0, 4))]   # neighbour at the edge of its range
        out[j] = bytes(v)
    for j in range(1, m, 7):   # in-alphabet misses
        out[j] = b"%08d" % int(rng.integers(0, 10 ** 8))
    return out


@pytest.mark.parametrize("dense", [True, False])
def test_chain_lean_step_validity(ctx, dense):
    """One lean step.  dense: the index holds every code of its code space (identity, no lookup); otherwise a rank
    table.  Both output modes against the oracle (check_chain), with adversarial stream bytes."""
    rng = np.random.default_rng(41 if dense else 42)
    n = 40_000
    ids = rng.permutation(n) if dense else rng.permutation(3 * n)[:n] + 20_000_000
    keys = [b"%08d" % int(i) for i in ids]
    probes = _nasty_fixed8_probes(rng, keys, 150_000)
    ch = check_chain(ctx, [[_fixed8(keys)]], [_fixed8(probes)], probe_base=7)
    assert 0 < ch.nrows < len(probes)


def test_chain_lean_masks(ctx):
    """Two- and three-step chains with the lean step first, last, everywhere; letters as well as digits (contiguous
    ranges other than '0'..'9'), and an index whose first positions are constant."""
    rng = np.random.default_rng(43)
    m = 120_000
    a = [b"%08d" % int(i) for i in rng.permutation(90_000)[:60_000]]
    b = [b"k%d" % int(i) for i in rng.permutation(3000)[:2000]]                     # variable length: never lean
    c = [bytes(rng.integers(ord("a"), ord("h") + 1, 8).astype(np.uint8)) for _ in range(30_000)]
    c = sorted(set(c))
    ka = _fixed8(_nasty_fixed8_probes(rng, a, m))
    kb = StrCol.from_values([b"k%d" % int(i) for i in rng.integers(0, 3300, m)])
    kc_vals = [c[i] for i in rng.integers(0, len(c), m)]
    kc_vals[::5] = [bytes(rng.integers(ord("a"), ord("j") + 1, 8).astype(np.uint8)) for _ in kc_vals[::5]]
    kc_vals[::11] = [b"abc\x00defg"] * len(kc_vals[::11])
    kc = _fixed8(kc_vals)
    A, B, C = [_fixed8(a)], [StrCol.from_values(b)], [_fixed8(c)]
    check_chain(ctx, [A, B], [ka, kb])          # mask 01
    check_chain(ctx, [B, A], [kb, ka])          # mask 10
    check_chain(ctx, [A, C], [ka, kc])          # mask 11
    check_chain(ctx, [A, C, A], [ka, kc, ka])   # three steps, all lean
    check_chain(ctx, [A, B, C], [ka, kb, kc])   # three steps, one not lean: the general kernel


def test_chain_lean_needs_aligned_fixed8(ctx):
    """The same keys handed over as a device column that starts at an odd address, and as a variable-length column:
    not lean, same result."""
    import torch

    rng = np.random.default_rng(44)
    keys = [b"%08d" % int(i) for i in rng.permutation(50_000)]
    probes = _nasty_fixed8_probes(rng, keys, 60_000)
    ix = DeviceIndex(ctx, [_fixed8(keys)])
    oix = orc.OracleIndex([_fixed8(keys)])
    col = _fixed8(probes)
    want = oix.join([col])
    raw = torch.zeros(col.data.nbytes + 24, dtype=torch.uint8, device="cuda:0")
    for shift in (0, 1, 4):
        raw[shift: shift + col.data.nbytes].copy_(torch.from_numpy(col.data))
        dcol = StrCol(raw[shift:], None, col.nrows, 32, N.CPH_MEM_DEVICE, fixed_width=8)
        chp = join_chain(ctx, [(ix, [dcol])], positions=True)
        np.testing.assert_array_equal(chp.stream_row, want["probe_idx"])
        np.testing.assert_array_equal(ix.perm()[chp.build_row(0)], want["build_row"])
        chp.release()
    chv = join_chain(ctx, [(ix, [col.as_variable()])], positions=True)
    np.testing.assert_array_equal(chv.stream_row, want["probe_idx"])
    np.testing.assert_array_equal(ix.perm()[chv.build_row(0)], want["build_row"])
    # A/B switches: the LUT walk and the rank-table lookup give the same answer
    from csvplus_amd import Context
    c2 = Context(0)
    c2.set_option("chain_arith", 0)
    c2.set_option("chain_identity", 0)
    ix2 = DeviceIndex(c2, [_fixed8(keys)])
    ch2 = join_chain(c2, [(ix2, [col])], positions=True)
    np.testing.assert_array_equal(ch2.stream_row, want["probe_idx"])
    np.testing.assert_array_equal(ix2.perm()[ch2.build_row(0)], want["build_row"])
    c2.close()

"""IndexOn over 32-bit codes WITH duplicates through counted LDS windows (csrc/counted_sort.hip): sort.Sort(&index.impl)
(csvplus.go:736) under Less (:794-807), rows with equal keys in input order; the first adjacent duplicate of createUniqueIndex
(:749-753) out of the same pass.  Compared bit for bit with the classic radix passes (ctx option counted_sort = 0), with numpy's
stable argsort of the key bytes, and — at sizes it finishes in seconds — with the oracle."""
import numpy as np
import pytest

from csvplus_amd import Context, DeviceIndex, StrCol, _native as N, datagen as dg

pytestmark = pytest.mark.gpu


def fixed8(ids: np.ndarray) -> StrCol:
    raw = np.char.zfill(ids.astype("U8"), 8).astype("S8")
    data = np.frombuffer(raw.tobytes(), np.uint8).copy()
    return StrCol.from_arrays(data, np.arange(ids.size + 1, dtype=np.uint32) * 8, fixed_width=8)


def build(ctx, col, unique=False, device=True):
    ctx.profile(True)
    ctx.profile_read(reset=True)
    g = DeviceIndex(ctx, [col.to_device("cuda:0") if device else col], unique=unique)
    prof = ctx.profile_read(reset=True)
    ctx.profile(False)
    return g, prof


def stable_order(ids: np.ndarray) -> np.ndarray:
    return np.argsort(ids, kind="stable").astype(np.uint32)   # zero-padded decimal ids: numeric order == bytewise order


@pytest.mark.parametrize("shape", ["one_level", "two_levels", "two_levels_sparse_tail"])
def test_counted_windows_equal_stable_argsort(shape):
    rng = np.random.default_rng(6)
    if shape == "one_level":
        n, domain = 3_000_000, 400_000
        ids = rng.integers(0, domain, n)
    elif shape == "two_levels":
        n, domain = 9_000_000, 5_600_000     # code space 6e6 states: 2930 windows of 2048 codes
        ids = rng.integers(0, domain, n)
    else:   # a block of ids + a thin spread over a code space 8x as large: tiles of the second level span thousands of windows
        n = 9_000_000
        ids = np.concatenate([rng.integers(0, 5_000_000, n - 40_000), rng.integers(5_000_000, 40_000_000, 40_000)])
        rng.shuffle(ids)
    ctx = Context(0)
    col = fixed8(ids)
    g, prof = build(ctx, col)
    assert "k_cs_window" in prof and "k_radix_scatter_u32" not in prof, sorted(prof)
    assert prof["k_cs_partition"]["launches"] == (1 if shape == "one_level" else 2), prof["k_cs_partition"]
    want = stable_order(ids)
    np.testing.assert_array_equal(g.perm(), want)
    s = ids[want]
    dup = np.flatnonzero(s[1:] == s[:-1])
    assert g.first_dup == (int(dup[0]) + 1 if dup.size else None)
    ctx.set_option("counted_sort", 0)
    r, prof0 = build(ctx, col)
    ctx.set_option("counted_sort", 1)
    assert "k_cs_window" not in prof0 and "k_radix_scatter_u32" in prof0
    np.testing.assert_array_equal(r.perm(), want)
    assert r.first_dup == g.first_dup
    # UniqueIndexOn over the same rows: the duplicate is reported where createUniqueIndex finds it
    u, _ = build(ctx, col, unique=True)
    assert u.status == N.CPH_ERR_DUPLICATE and u.first_dup == g.first_dup
    g.close(); r.close(); u.close(); ctx.close()


def test_a_dense_cluster_is_sorted_with_narrower_windows():
    """The plan sizes the windows by the AVERAGE rows per code; a dense block of keys in a sparse code space overflows them.  Before
    the classic passes the build tries windows a quarter (then a sixteenth) as wide over the same codes."""
    rng = np.random.default_rng(10)
    n = 9_000_000
    ids = np.concatenate([rng.integers(0, 600_000, n - 40_000), rng.integers(600_000, 9_990_000, 40_000)])   # 15 rows per code in the block
    rng.shuffle(ids)
    ctx = Context(0)
    g, prof = build(ctx, fixed8(ids))
    assert prof["k_cs_scan"]["launches"] == 2 and prof["k_cs_window"]["launches"] == 2 and "k_radix_scatter_u32" not in prof, sorted(prof)
    want = stable_order(ids)
    np.testing.assert_array_equal(g.perm(), want)
    s = ids[want]
    assert g.first_dup == int(np.flatnonzero(s[1:] == s[:-1])[0]) + 1
    g.close(); ctx.close()


@pytest.mark.parametrize("shape", ["groups_beyond_32", "one_key_beyond_a_window", "all_rows_one_key"])
def test_large_groups_and_the_overflow_path(shape):
    """Groups of equal keys beyond 32 members are put in row order by a wave's bitonic network; a key with more duplicates than a
    window holds (> 16384: here 20000, and the whole table) cannot be sorted in LDS — the flag goes up, nothing was sorted, and the classic
    passes run over the same codes."""
    rng = np.random.default_rng(9)
    n = 2_600_000
    ids = rng.integers(0, 900_000, n)
    if shape == "groups_beyond_32":
        for key, cnt in ((5, 33), (77, 64), (123_456, 65), (899_999, 1000), (400_000, 4097), (400_001, 700), (0, 2000)):
            ids[rng.choice(n, cnt, replace=False)] = key
    elif shape == "one_key_beyond_a_window":
        ids[rng.choice(n, 20_000, replace=False)] = 424_242
    else:
        ids[:] = 31_337
    ctx = Context(0)
    col = fixed8(ids)
    g, prof = build(ctx, col)
    if shape == "groups_beyond_32":
        assert "k_cs_window" in prof and "k_radix_scatter_u32" not in prof, sorted(prof)
    elif shape == "one_key_beyond_a_window":
        assert "k_cs_scan" in prof and "k_radix_scatter_u32" in prof, sorted(prof)      # tried, gave up, sorted the classic way
    want = stable_order(ids)
    np.testing.assert_array_equal(g.perm(), want)
    s = ids[want]
    dup = np.flatnonzero(s[1:] == s[:-1])
    assert g.first_dup == int(dup[0]) + 1
    g.close(); ctx.close()


@pytest.mark.parametrize("source", ["device", "host"])
def test_config3_keys_against_the_oracle(source):
    """BASELINE config 3's keys (surname/name#number: variable length, ~8 rows per key) at 2.5e6 rows: the split codec, then the
    counted windows; perm and first duplicate bit-exact against the oracle (csvplus.go:707-756, :794-807)."""
    from oracle import orc

    ctx = Context(0)
    col = dg.varkeys(2_500_000, distinct_suffix=3000)
    g, prof = build(ctx, col, device=source == "device")
    assert "k_cs_window" in prof, sorted(prof)
    o = orc.OracleIndex([col])
    np.testing.assert_array_equal(g.perm(), o.perm)
    assert g.first_dup == o.first_dup()
    g.close(); ctx.close()


def test_join_against_an_index_sorted_through_counted_windows():
    """people.Join(IndexOn(orders.cust_id), "id") (csvplus.go:553-567: every equal index row, ascending position) over an index built this way."""
    from oracle import orc

    ctx = Context(0)
    n_people, m = 150_000, 2_400_000
    ords = dg.orders(m, n_people, 100)
    people = dg.column(dg.SEQ_PERM, n_people, n_people, encoding=dg.FIXED8, seed=3)
    g, prof = build(ctx, ords["cust_id"])
    assert "k_cs_window" in prof, sorted(prof)
    o = orc.OracleIndex([ords["cust_id"]])
    np.testing.assert_array_equal(g.perm(), o.perm)
    mt = g.probe([people])
    oj = o.join([people])
    np.testing.assert_array_equal(mt.probe_idx, oj["probe_idx"])
    np.testing.assert_array_equal(mt.build_row, oj["build_row"])
    g.close(); ctx.close()

"""Dictionary-coded position groups of the key codec (keycodec.hip, codec_try_groups): long keys whose leading
bytes take few distinct values (BASELINE config 3: surname "/" name "#" digits) get codes of one word instead of
two.  Everything observable must stay bit-identical to the oracle: order, duplicates, probes, prefix joins, find,
persistence — and the group path must actually be taken."""
import numpy as np
import pytest

from csvplus_amd import DeviceIndex, StrCol, datagen as dg
from oracle import orc
from tests.helpers import assert_join_equal, random_keys

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _no_split(ctx):
    """These tests are about the dictionary WINDOWS: since round 4 keys like config 3's take the delimiter split first
    (tests/test_gpu_split_codec.py), so it is switched off here."""
    ctx.set_option("codec_split", 0)
    yield
    ctx.set_option("codec_split", 1)


def same_bounds(a, b):
    """find() bounds agree; for an absent value only emptiness is observable (rows[lower:upper], csvplus.go:625-641)."""
    return a == b or (a[0] == a[1] and b[0] == b[1])


def _check(ctx, keycols, probecols, expect_groups=True):
    g = DeviceIndex(ctx, keycols)
    o = orc.OracleIndex(keycols)
    info = g.info()
    if expect_groups:
        assert info["dict_entries"] > 0, info
    np.testing.assert_array_equal(g.perm(), o.perm)
    assert g.first_dup == o.first_dup()
    assert_join_equal(g.probe(probecols), o.join(probecols))
    return g, o, info


def test_config3_shaped_keys_use_groups_and_one_word(ctx):
    n = 200_000
    keys = dg.varkeys(n, 1000)
    probe = dg.varkeys(50_000, 3000, seed=dg.SEED + 77)        # some probe keys do not exist in the index
    g, o, info = _check(ctx, [keys], [probe])
    assert info["code_words"] == 1 and info["key_positions"] >= 15, info
    for v in (keys.value(0), keys.value(12345), b"Smith/Amelia#1", b"", b"Zzz"):
        assert same_bounds(g.find(v), o.find(v))
    lo, hi = g.dup_groups()
    assert len(lo) > 0 and int(hi[0] - lo[0]) >= 2


def test_groups_with_multiple_columns_and_prefix_join(ctx):
    rng = np.random.default_rng(41)
    n = 20_000
    a = random_keys(rng, n, 9, 16, alphabet=np.frombuffer(b"abc", np.uint8), distinct=50)
    b = random_keys(rng, n, 0, 12, distinct=300)               # arbitrary bytes, NULs included
    c = random_keys(rng, n, 8, 8, alphabet=np.frombuffer(b"0123456789", np.uint8))
    cols = [StrCol.from_values(a), StrCol.from_values(b), StrCol.from_values(c)]
    probe = [StrCol.from_values(a[:5000] + random_keys(rng, 500, 9, 16, alphabet=np.frombuffer(b"abcd", np.uint8))),
             StrCol.from_values(b[:5000] + random_keys(rng, 500, 0, 12)),
             StrCol.from_values(c[:5000] + c[:500])]
    g, o, info = _check(ctx, cols, probe)
    assert_join_equal(g.probe(probe[:1]), o.join(probe[:1]))   # prefix joins on 1 and 2 leading columns
    assert_join_equal(g.probe(probe[:2]), o.join(probe[:2]))
    assert g.find(a[7]) == o.find(a[7]) and g.find(a[7], b[7]) == o.find(a[7], b[7])
    assert g.find(a[7], b[7], c[7]) == o.find(a[7], b[7], c[7])
    assert same_bounds(g.find(b"zzzz"), o.find(b"zzzz"))


def test_groups_pad_and_length_edge_cases(ctx):
    """Values ending inside a group, empty values, a value that is a proper prefix of another."""
    base = [b"", b"a", b"ab", b"abcdefg", b"abcdefgh", b"abcdefg\x00", b"abcdefgh\x00\x00", b"abcdefghijklmnopqrstuvwxyz0123456789",
            b"abcdefghijklmnopqrstuvwxyz012345678", b"\xff" * 30, b"\x00" * 30, b"\x00" * 29]
    vals = base * 40
    g, o, _ = _check(ctx, [StrCol.from_values(vals)], [StrCol.from_values(base + [b"abc", b"abcdefghi", b"\x00" * 31])])
    for v in base + [b"abc", b"\xff" * 31]:
        assert same_bounds(g.find(v), o.find(v))


def test_groups_survive_save_load(ctx, tmp_path):
    keys = dg.varkeys(30_000, 500)
    g = DeviceIndex(ctx, [keys])
    assert g.info()["dict_entries"] > 0
    g.save(str(tmp_path / "g.cph"))
    ld = DeviceIndex.load(ctx, str(tmp_path / "g.cph"))
    assert ld.info()["dict_entries"] == g.info()["dict_entries"]
    probe = [dg.varkeys(10_000, 800, seed=dg.SEED + 5)]
    a, b = g.probe(probe), ld.probe(probe)
    assert a.cnt.tolist() == b.cnt.tolist() and a.build_row.tolist() == b.build_row.tolist()
    assert ld.find(keys.value(3)) == g.find(keys.value(3))


def test_high_cardinality_falls_back_to_positions(ctx):
    rng = np.random.default_rng(43)
    vals = random_keys(rng, 60_000, 20, 30)     # every 7-byte group has ~60000 distinct joint symbols
    g, o, info = _check(ctx, [StrCol.from_values(vals)], [StrCol.from_values(vals[:2000])], expect_groups=False)
    assert info["dict_entries"] == 0 and info["code_words"] >= 2


def test_config3_properties_1e7(ctx):
    """BASELINE config 3 at 1e7 rows (the sample-then-prune statistics pass only runs above 2^21 rows): the result
    is checked through size-independent properties computed with numpy on the host — perm is a permutation, keys are
    non-decreasing along it, equal keys keep input order, first_dup is the first adjacent-equal pair, and find() of
    sampled keys returns exactly their run."""
    n = 10_000_000
    keys = dg.varkeys(n)
    g = DeviceIndex(ctx, [keys])
    info = g.info()
    assert info["dict_entries"] > 0 and info["code_words"] == 1, info
    perm = g.perm().astype(np.int64)
    assert np.array_equal(np.sort(perm), np.arange(n))
    # keys as zero-padded 24-byte rows -> three big-endian words per key (config-3 keys hold no NUL byte and are
    # at most 22 bytes long, so zero padding orders exactly like strings.Compare)
    offs = keys.offsets.astype(np.int64)
    lens = np.diff(offs)
    assert lens.max() <= 24 and keys.data[: offs[-1]].min() > 0
    padded = np.zeros((n, 24), dtype=np.uint8)
    row_of = np.repeat(np.arange(n), lens)
    col_of = np.arange(offs[-1]) - np.repeat(offs[:-1], lens)
    padded[row_of, col_of] = keys.data[: offs[-1]]
    words = padded.view(">u8").astype(np.uint64)[perm]          # (n, 3) in sorted order
    a, b = words[:-1], words[1:]
    lt = (a[:, 0] < b[:, 0]) | ((a[:, 0] == b[:, 0]) & ((a[:, 1] < b[:, 1]) | ((a[:, 1] == b[:, 1]) & (a[:, 2] < b[:, 2]))))
    eq = (a == b).all(axis=1)
    assert bool((lt | eq).all())                                  # sorted
    assert bool((perm[1:][eq] > perm[:-1][eq]).all())             # stable inside equal-key runs
    first = int(np.argmax(eq)) + 1 if eq.any() else None
    assert g.first_dup == first
    # find(): the run of a sampled key is exactly where that key sits in the sorted sequence
    starts = np.flatnonzero(np.concatenate(([True], ~eq)))        # first position of every run
    for s in starts[:: max(1, len(starts) // 25)][:25]:
        r = int(perm[s])
        lo, hi = g.find(keys.value(r))
        nxt = starts[np.searchsorted(starts, s) + 1] if s != starts[-1] else n
        assert (lo, hi) == (int(s), int(nxt))


def _launches(ctx):
    return {k: v["launches"] for k, v in ctx.profile_read(reset=True).items()}


def test_speculative_dictionaries_hit_and_miss(ctx):
    """Large inputs: the window sets come from a sample of the rows (every `step`-th).  Exact mode runs one pass over all
    rows for the chosen windows (k_group_stats); speculative mode uses the sample's sets as dictionaries and lets the
    encode kernel complete them when it meets a window the sample did not hold (GroupSpec, keycodec.hip) — then the
    encode runs twice.  Every mode must give the oracle's permutation, on data whose rare windows sit only in rows
    the sample never reads."""
    n = (1 << 21) + 150_001                       # step = n >> 18 = 8: the sample reads rows 0, 8, 16, ...
    base = dg.varkeys(n, 1000)
    vals = base.values()
    odd = list(range(1, n, 104_730))              # odd rows only: never sampled
    miss_vals = list(vals)
    for i, r in enumerate(odd):
        miss_vals[r] = b"Qwertz/Xavier#%d" % (i % 7)
    miss_col = StrCol.from_values(miss_vals)
    # every odd row carries a surname the sample never sees: far more than kSpecGiveUp (2^16) rows miss, the speculative
    # encode stops early and the exact pass takes over
    half_vals = [b"Zz" + v if r & 1 else v for r, v in enumerate(vals)]
    half_col = StrCol.from_values(half_vals)
    probe = dg.varkeys(20_000, 1500, seed=dg.SEED + 5)
    try:
        for col, has_unseen in ((base, False), (miss_col, True), (half_col, True)):
            o = orc.OracleIndex([col])
            bits = {}
            for mode in (2, 0, 1):                # always speculative / exact / decided by the sample's singletons
                ctx.set_option("speculative_groups", mode)
                ctx.profile(True)
                ctx.profile_read(reset=True)
                g = DeviceIndex(ctx, [col])
                runs = _launches(ctx)
                ctx.profile(False)
                info = g.info()
                assert info["dict_entries"] > 0 and info["code_words"] == 1, info
                bits[mode] = info["code_bits"]
                assert runs["k_group_sample"] == 1, runs
                if mode == 2:
                    if col is half_col:
                        assert runs.get("k_group_stats") == 1 and runs["k_encode_build"] == 2, runs   # gave up -> exact pass
                    else:
                        assert "k_group_stats" not in runs, runs
                        if has_unseen:
                            assert runs["k_encode_build"] == 2, runs  # the first encode met unknown windows
                if mode == 0:
                    assert runs["k_group_stats"] == 1 and runs["k_encode_build"] == 1, runs
                np.testing.assert_array_equal(g.perm(), o.perm)
                assert g.first_dup == o.first_dup()
                assert_join_equal(g.probe([probe]), o.join([probe]))
                for v in (col.value(1), col.value(odd[3]), b"Qwertz/Xavier#3", b"Nobody/None#0"):
                    assert same_bounds(g.find(v), o.find(v))
                g.close()
            assert bits[2] <= bits[0] + 1, bits
    finally:
        ctx.set_option("speculative_groups", 1)


@pytest.mark.parametrize("seed", [51, 52, 53])
def test_sampled_window_choice_on_multicolumn_long_keys(ctx, seed):
    """Above 2^19 rows the candidate windows are judged on a SAMPLE of the rows (k_group_stage / k_group_sample) and only the
    chosen ones are completed over all rows (k_group_stats).  Random 2-3 column keys with low-cardinality fields of up to
    40 bytes (values longer than 24 bytes take the long-value code paths), rare values included: order, first duplicate,
    joins with absent keys and prefix joins must match the oracle."""
    rng = np.random.default_rng(seed)
    n = (1 << 19) + 40_000 + seed
    ncols = 2 + seed % 2
    cols, probes = [], []
    for c in range(ncols):
        lo, hi = [(3, 12), (20, 40), (0, 9)][c]
        pool = random_keys(rng, 40 + 30 * c, lo, hi, alphabet=np.frombuffer(b"abcdefgh/#0123456789", np.uint8))
        rare = random_keys(rng, 25, lo, hi, alphabet=np.frombuffer(b"xyz", np.uint8))     # a handful of rows each
        pick = rng.integers(0, len(pool), n)
        vals = [pool[i] for i in pick]
        for j, r in enumerate(rng.integers(0, n, 60)):
            vals[int(r)] = rare[j % len(rare)]
        cols.append(StrCol.from_values(vals))
        pv = [vals[int(i)] for i in rng.integers(0, n, 3000)] + random_keys(rng, 300, lo, hi, alphabet=np.frombuffer(b"abcxyz", np.uint8))
        probes.append(StrCol.from_values(pv))
    g = DeviceIndex(ctx, cols)
    o = orc.OracleIndex(cols)
    info = g.info()
    np.testing.assert_array_equal(g.perm(), o.perm)
    assert g.first_dup == o.first_dup()
    assert_join_equal(g.probe(probes), o.join(probes))
    assert_join_equal(g.probe(probes[:1]), o.join(probes[:1]))
    for r in (0, 12345, n - 1):
        vs = [c.value(r) for c in cols]
        assert g.find(*vs) == o.find(*vs) and g.find(vs[0]) == o.find(vs[0])
    assert info["code_words"] >= 1
    g.close()

"""The delimiter-split key codec (keycodec.hip "split codec", round 4): a variable-length key column whose fields float
("Smith/Amelia#12345") is cut at its first delimiter byte into a dictionary-coded prefix and a per-position suffix, so
BASELINE config 3 codes in 25 bits (32-bit keys, 4 radix passes) instead of 47.  Everything observable must stay
bit-identical to the oracle (csvplus.go:794-807 order, :893-920 probes): perm, first duplicate, Join, prefix Join, Find,
persistence — and the split must actually be taken where expected, and NOT taken where it cannot be."""
import numpy as np
import pytest

from csvplus_amd import Context, DeviceIndex, StrCol, datagen as dg
from oracle import orc
from tests.helpers import assert_join_equal, random_keys

pytestmark = pytest.mark.gpu

N = 70_000   # the split is tried from 2^16 rows on


def same_bounds(a, b):
    return a == b or (a[0] == a[1] and b[0] == b[1])


def _check(ctx, keycols, probecols, expect_split=None):
    g = DeviceIndex(ctx, keycols)
    o = orc.OracleIndex(keycols)
    info = g.info()
    if expect_split is True:
        assert info["split"] != 0, info
    elif expect_split is False:
        assert info["split"] == 0, info
    elif expect_split is not None:   # the delimiter byte
        assert info["split"] & 0xFF == expect_split and info["split"] >> 8 >= 1, info
    np.testing.assert_array_equal(g.perm(), o.perm)
    assert g.first_dup == o.first_dup()
    assert_join_equal(g.probe(probecols), o.join(probecols))
    return g, o, info


def _keys(rng, n, heads, digits=5, delim=b"#", no_delim_every=0):
    out = []
    for i in range(n):
        h = heads[int(rng.integers(0, len(heads)))]
        v = h + delim + (b"%d" % int(rng.integers(0, 10 ** digits)))
        if no_delim_every and i % no_delim_every == 7:
            v = h   # a key without the delimiter: its own prefix, no suffix
        out.append(v)
    return out


def test_config3_keys_split_at_hash_32bit_codes(ctx):
    n = 300_000
    keys = dg.varkeys(n)
    probe = dg.varkeys(60_000, 300_000, seed=dg.SEED + 77)   # wider number range: many probe keys miss
    g, o, info = _check(ctx, [keys], [probe], expect_split=ord("#"))
    assert info["key_bytes"] == 4 and info["code_words"] == 1 and info["code_bits"] <= 32, info
    assert info["dict_entries"] == 120, info                   # 12 surnames x 10 names
    for v in (keys.value(0), keys.value(12345), b"Smith/Amelia#1", b"Smith/Amelia#", b"Smith/Amelia", b"", b"Zzz#1", b"#", b"#5"):
        assert same_bounds(g.find(v), o.find(v)), v
    lo, hi = g.dup_groups()
    olo, ohi = o.dup_groups() if hasattr(o, "dup_groups") else (None, None)
    assert len(lo) > 0 and (olo is None or (np.array_equal(lo, olo) and np.array_equal(hi, ohi)))
    # the same table without the split: same order
    c2 = Context(0)
    c2.set_option("codec_split", 0)
    g2 = DeviceIndex(c2, [keys])
    assert g2.info()["split"] == 0
    np.testing.assert_array_equal(g2.perm(), g.perm())
    c2.close()


def test_split_edge_cases_around_the_delimiter(ctx):
    """Keys without the delimiter, the delimiter first / last / doubled, NUL and 0xFF next to it, empty values, a key that
    is a proper prefix of another key's prefix."""
    rng = np.random.default_rng(71)
    heads = [b"alpha", b"alph", b"alpha/beta", b"", b"b\x00", b"b\xff", b"b", b"\x00", b"\xff\xff", b"gamma-delta-epsilon"]
    keys = _keys(rng, N, heads, digits=4, no_delim_every=150)
    specials = [b"#", b"##", b"#1", b"alpha#", b"alpha##1", b"alpha#1#2", b"alpha", b"alph", b"alpha/beta", b"", b"b\x00#\x00",
                b"b\xff#\xff", b"\x00#", b"b#\x001", b"b#1\x00"]
    keys[100:100 + len(specials)] = specials
    keys[2000:2000 + len(specials)] = specials          # duplicates of the specials
    probe = specials + keys[:3000] + [b"alpha#12345", b"alphb#1", b"alph#1", b"zzz", b"alpha#1\xff", b"b\x00#", b"b\x01#1"]
    g, o, info = _check(ctx, [StrCol.from_values(keys)], [StrCol.from_values(probe)], expect_split=ord("#"))
    for v in specials + [b"alpha#12", b"nope#1", b"alpha#\x00"]:
        assert same_bounds(g.find(v), o.find(v)), v


def test_split_other_delimiter_bytes(ctx):
    rng = np.random.default_rng(72)
    for delim in (b"\x00", b"\xff", b"/", b","):
        heads = [bytes([65 + i]) * (1 + i % 5) for i in range(30)]
        keys = _keys(rng, N, heads, digits=6, delim=delim)
        probe = keys[:2000] + [heads[3] + delim, heads[3], delim + b"1", heads[4] + delim + b"9999999"]
        _check(ctx, [StrCol.from_values(keys)], [StrCol.from_values(probe)], expect_split=delim[0])


def test_split_multi_column_keys_and_prefix_joins(ctx):
    """The split column between two ordinary key columns; prefix Joins / Finds on 1, 2 and 3 leading columns."""
    rng = np.random.default_rng(73)
    heads = [b"north/east", b"north/west", b"south", b"s", b"centre/inner/ring"]
    a = [b"%c" % c for c in rng.integers(97, 101, N)]
    b = _keys(rng, N, heads, digits=5, no_delim_every=200)
    c = [b"%02d" % i for i in rng.integers(0, 50, N)]
    cols = [StrCol.from_values(a), StrCol.from_values(b), StrCol.from_values(c)]
    pa = a[:4000] + [b"e", b"a"]
    pb = b[:4000] + [b"south#1", b"north/east#99999"]
    pc = c[:4000] + [b"07", b"99"]
    probe = [StrCol.from_values(pa), StrCol.from_values(pb), StrCol.from_values(pc)]
    g, o, info = _check(ctx, cols, probe, expect_split=ord("#"))
    assert info["split"] >> 8 == 2, info                      # the second key column is the one that is cut
    assert_join_equal(g.probe(probe[:1]), o.join(probe[:1]))
    assert_join_equal(g.probe(probe[:2]), o.join(probe[:2]))
    for i in (0, 7, 4001):
        assert same_bounds(g.find(pa[i]), o.find(pa[i]))
        assert same_bounds(g.find(pa[i], pb[i]), o.find(pa[i], pb[i]))
        assert same_bounds(g.find(pa[i], pb[i], pc[i]), o.find(pa[i], pb[i], pc[i]))
    # the split column FIRST, joined through a chain step and through the stream join's general path
    cols2 = [StrCol.from_values(b), StrCol.from_values(c)]
    probe2 = [StrCol.from_values(pb), StrCol.from_values(pc)]
    g2, o2, info2 = _check(ctx, cols2, probe2, expect_split=ord("#"))
    assert info2["split"] >> 8 == 1
    assert_join_equal(g2.probe(probe2[:1]), o2.join(probe2[:1]))


def test_split_survives_save_load(ctx, tmp_path):
    keys = dg.varkeys(100_000, 5000)
    g = DeviceIndex(ctx, [keys])
    assert g.info()["split"] != 0
    g.save(str(tmp_path / "s.cph"))
    ld = DeviceIndex.load(ctx, str(tmp_path / "s.cph"))
    assert ld.info()["split"] == g.info()["split"] and ld.info()["dict_entries"] == g.info()["dict_entries"]
    probe = [dg.varkeys(20_000, 8000, seed=dg.SEED + 5)]
    a, b = g.probe(probe), ld.probe(probe)
    assert a.cnt.tolist() == b.cnt.tolist() and a.build_row.tolist() == b.build_row.tolist()
    for v in (keys.value(3), b"Smith/Amelia#", b"nothing"):
        assert ld.find(v) == g.find(v)


def test_split_not_taken_where_it_cannot_be(ctx):
    """Prefixes longer than 32 bytes, suffixes longer than 16, too many distinct prefixes, no common byte: the older paths
    build the index, the result is the oracle's all the same."""
    rng = np.random.default_rng(74)
    long_heads = [b"x" * 33 + bytes([65 + i]) for i in range(5)]
    _check(ctx, [StrCol.from_values(_keys(rng, N, long_heads, digits=4))], [StrCol.from_values([long_heads[0] + b"#12", b"q"])], expect_split=False)
    heads = [b"h%d" % i for i in range(20)]
    long_suffix = [h + b"#" + bytes(rng.integers(48, 58, 17).astype(np.uint8)) for h in heads for _ in range(N // 20)]
    _check(ctx, [StrCol.from_values(long_suffix)], [StrCol.from_values(long_suffix[:100])], expect_split=False)
    many = [b"p%05d#%d" % (int(rng.integers(0, 20000)), int(rng.integers(0, 100))) for _ in range(N)]
    _check(ctx, [StrCol.from_values(many)], [StrCol.from_values(many[:100] + [b"p00001#", b"p#1"])], expect_split=False)
    nodelim = random_keys(rng, N, 10, 20, alphabet=np.frombuffer(b"abcdefghijklmnopqrstuvwxyz", np.uint8))
    _check(ctx, [StrCol.from_values(nodelim)], [StrCol.from_values(nodelim[:100])], expect_split=False)


def test_split_values_of_25_to_40_bytes(ctx):
    """The five-chunk instantiation of the split kernels."""
    rng = np.random.default_rng(75)
    heads = [b"department-of-" + bytes([97 + i]) * (3 + i) for i in range(12)]
    keys = _keys(rng, N, heads, digits=9, delim=b":", no_delim_every=500)
    probe = keys[:3000] + [heads[0] + b":", heads[11] + b":999999999", heads[11] + b":9999999990", b":"]
    g, o, info = _check(ctx, [StrCol.from_values(keys)], [StrCol.from_values(probe)], expect_split=ord(":"))
    for v in (keys[5], heads[2], heads[2] + b":", b""):
        assert same_bounds(g.find(v), o.find(v))


def test_split_chain_and_stream_join_over_a_split_index(ctx):
    """A split index as a step of the chained Join (general path) in both output modes."""
    from csvplus_amd import join_chain

    rng = np.random.default_rng(76)
    heads = [b"eu/de", b"eu/fr", b"us", b"apac/jp/tokyo"]
    keys = sorted(set(_keys(rng, 2 * N, heads, digits=5)))
    cust = [b"%08d" % i for i in range(5000)]
    m = 40_000
    sk = StrCol.from_values([keys[i] for i in rng.integers(0, len(keys), m - 2)] + [b"eu/de#", b"zz"])
    ck = StrCol.from_values([cust[i] for i in rng.integers(0, len(cust), m)])
    gi, gc = DeviceIndex(ctx, [StrCol.from_values(keys)]), DeviceIndex(ctx, [StrCol.from_values(cust)])
    oi, oc = orc.OracleIndex([StrCol.from_values(keys)]), orc.OracleIndex([StrCol.from_values(cust)])
    assert len(keys) >= 1 << 16 and gi.info()["split"] != 0
    j1 = oi.join([sk])
    sel = j1["probe_idx"].astype(np.uint32)
    j2 = oc.join([ck], row_sel=sel)
    pick = j2["probe_idx"].astype(np.int64)
    for positions in (False, True):
        ch = join_chain(ctx, [(gi, [sk]), (gc, [ck])], positions=positions)
        np.testing.assert_array_equal(ch.stream_row, j1["probe_idx"][pick])
        r0, r1 = ch.build_row(0), ch.build_row(1)
        if positions:
            r0, r1 = gi.perm()[r0], gc.perm()[r1]
        np.testing.assert_array_equal(r0, j1["build_row"][pick])
        np.testing.assert_array_equal(r1, j2["build_row"])
        ch.release()


def _replace_rows(col, repl):
    """A copy of a variable-length column with some rows' values replaced ({row: bytes})."""
    data = np.asarray(col.data)
    off = np.asarray(col.offsets).astype(np.int64)
    parts, lens, prev = [], (off[1:] - off[:-1]).copy(), 0
    for r in sorted(repl):
        parts.append(data[off[prev]:off[r]])
        parts.append(np.frombuffer(repl[r], np.uint8))
        lens[r] = len(repl[r])
        prev = r + 1
    parts.append(data[off[prev]:off[-1]])
    noff = np.zeros(len(off), np.uint32)
    np.cumsum(lens, out=noff[1:])
    return StrCol.from_arrays(np.concatenate(parts), noff)


@pytest.mark.parametrize("rare", ["none", "prefix", "suffix_byte", "short_suffix", "long_value"])   # (long_suffix / no_delimiter: tools/fuzz_round5.py)
def test_split_codec_from_the_sample_alone(ctx, rare):
    """Round 5: a large table over ONE variable-length key column (>= 2^22 rows) takes the split codec's dictionary and suffix
    alphabets from the 2^18-row sample; the exact pass over all rows (k_split_stats) is gone, the encode kernel checks every row
    instead — prefix in the dictionary, every suffix byte and END in its position's alphabet, lengths within the sample's.  One
    row the sample does not visit and cannot code makes the build start over with the exact statistics: same index as the
    oracle's (csvplus.go:794-807) either way, and the same index as a build with the speculation switched off."""
    n = (1 << 22) + 12_345
    keys = dg.varkeys(n)
    step = n >> 18
    row = 1_234_567
    assert row % step != 0   # not a sampled row
    repl = {"none": None, "prefix": b"Zeppelin/Qq#12345", "suffix_byte": b"Smith/Amelia#12x45", "long_suffix": b"Smith/Amelia#1234567",
            "short_suffix": b"Smith/Amelia#", "no_delimiter": b"Smith/Amelia", "long_value": b"Smith/Amelia-and-a-very-long-middle-name#123"}[rare]
    if repl is not None:
        keys = _replace_rows(keys, {row: repl})
    o = orc.OracleIndex([keys])
    dk = keys.to_device("cuda:0")
    ctx.profile(True)
    ctx.profile_read(reset=True)
    g = DeviceIndex(ctx, [dk])
    prof = ctx.profile_read(reset=True)
    ctx.profile(False)
    exact_passes = prof.get("k_split_stats", {"launches": 0})["launches"]
    info = g.info()
    if rare == "none":
        assert exact_passes == 0 and info["split"] & 0xFF == ord("#"), (prof.keys(), info)
    else:
        assert exact_passes <= 1, prof
    np.testing.assert_array_equal(g.perm(), o.perm)
    assert g.first_dup == o.first_dup()
    ctx.set_option("split_speculative", 0)
    try:
        ctx.profile(True)
        g0 = DeviceIndex(ctx, [dk])
        prof0 = ctx.profile_read(reset=True)
        ctx.profile(False)
        l0 = prof0.get("k_split_stats", {"launches": 0})["launches"]   # (a value beyond the split kernels' 40 bytes: the exact pass ran and said no)
        assert l0 <= 1 and (l0 == 1 or not g0.info()["split"])
        assert g0.info() == info
        np.testing.assert_array_equal(g0.perm(), o.perm)
    finally:
        ctx.set_option("split_speculative", 1)
    probe = dg.varkeys(50_000, 300_000, seed=dg.SEED + 78)
    if repl is not None:
        probe = _replace_rows(probe, {5: repl, 6: repl + b"0"})
    assert_join_equal(g.probe([probe]), o.join([probe]))

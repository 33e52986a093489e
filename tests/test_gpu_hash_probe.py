"""The hash probe of Join (csrc/hash_device.hpp): every full-key Join against an index whose key codes are too sparse
for a direct-address table — random ids, hashes, several key columns, keys of several codec windows — looks its keys
up in a hash table over the codes instead of binary-searching them.  Bar: bit-exact with the oracle, like every
other path; every test asserts through cph_index_get_info that the hash table is what answered.

Reference semantics: first()/cmp (csvplus.go:893-920) via Join (csvplus.go:545-569); a PREFIX join (fewer columns
than the index has, csvplus.go:546-550, :910) needs the order of the keys and must keep using the sorted path."""
import numpy as np
import pytest

from csvplus_amd import Context, DeviceIndex, StrCol, _native as N, join_chain
from csvplus_amd.streaming import StreamJoin, bitmap_to_rows
from oracle import orc
from tests.helpers import assert_join_equal, random_keys
from tests.test_gpu_chain import check_chain, oracle_chain

pytestmark = pytest.mark.gpu



@pytest.fixture(autouse=True, params=["whole_table_build", "slice_by_slice_build"])
def hash_build_kind(ctx, request):
    """Every test twice: the table of a duplicate-free index built by compare-and-swap all over it, and (round 6: ctx option
    hash_partitioned; 2 = whatever the size) slice by slice in LDS behind a counted partition of the rows by home sector."""
    ctx.set_option("hash_partitioned", 2 if request.param == "slice_by_slice_build" else 0)
    yield request.param
    ctx.set_option("hash_partitioned", 1)


ALNUM36 = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz0123456789", dtype=np.uint8)
ALNUM62 = np.frombuffer(b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789", dtype=np.uint8)
HASH_BUILT = 4   # cph_index_info.lookup_built bit


def fixed_random(rng, n, width, alphabet):
    a = np.asarray(alphabet, dtype=np.uint8)
    return [a[rng.integers(0, len(a), width)].tobytes() for _ in range(n)]


def probe_mix(rng, build, m, absent):
    """m probe keys: about 60 % drawn from the build keys, the rest from `absent` (keys that may or may not occur)."""
    pick = rng.random(m) < 0.6
    bi = rng.integers(0, len(build), m)
    ai = rng.integers(0, len(absent), m)
    return [build[b] if p else absent[a] for p, b, a in zip(pick, bi, ai)]


def check_hash_join(ctx, build_cols, probe_cols, mode, unique_expected=None):
    modes = mode if isinstance(mode, tuple) else (mode,)
    g = DeviceIndex(ctx, build_cols)
    o = orc.OracleIndex(build_cols)
    np.testing.assert_array_equal(g.perm(), o.perm)
    inf = g.info()
    assert inf["direct_table"] == 0, inf           # sparse code space: no direct-address table planned
    assert inf["lookup_built"] == 0                # nothing is built before the first Join
    m = g.probe(probe_cols)
    inf = g.info()
    assert inf["lookup_built"] & HASH_BUILT and inf["hash_mode"] in modes, inf
    oj = o.join(probe_cols)
    assert_join_equal(m, oj)
    if unique_expected is not None:
        assert (o.first_dup() is None) == unique_expected
    # Except / counting: the same lookup without pairs
    m2 = g.probe(probe_cols, want_pairs=False)
    np.testing.assert_array_equal(m2.cnt, oj["cnt"])
    return g, o, m


@pytest.mark.parametrize("dups", [False, True])
def test_one_word_codes_random_alnum(ctx, dups):
    """12 random [a-z0-9] characters: 36^12 < 2^63 -> one code word, entries carry the code itself (kHashK1)."""
    rng = np.random.default_rng(101 + dups)
    n, m = 60_000, 150_000
    build = fixed_random(rng, n, 12, ALNUM36)
    if dups:
        build = [build[i] for i in rng.integers(0, n // 6, n)]   # ~6 rows per key: aux = end of the run
    probe = probe_mix(rng, build, m, fixed_random(rng, 5000, 12, ALNUM36) + [b"", b"abc", b"ABCDEFGHIJKL", b"abcdefghijklm"])
    g, o, mt = check_hash_join(ctx, [StrCol.from_values(build)], [StrCol.from_values(probe)], 1, unique_expected=not dups)
    assert g.info()["code_words"] == 1
    assert 0 < mt.nmatches


@pytest.mark.parametrize("dups", [False, True])
def test_two_word_codes_random_alnum62(ctx, dups):
    """12 random [A-Za-z0-9] characters: 62^12 ~ 2^71 -> two code words, both kept in the entry (kHashK3)."""
    rng = np.random.default_rng(201 + dups)
    n, m = 50_000, 120_000
    build = fixed_random(rng, n, 12, ALNUM62)
    if dups:
        build = [build[i] for i in rng.integers(0, n // 4, n)]
    probe = probe_mix(rng, build, m, fixed_random(rng, 5000, 12, ALNUM62) + [b"", b"0", b"zzzzzzzzzzzzz", b"\xff" * 12])
    g, o, mt = check_hash_join(ctx, [StrCol.from_values(build)], [StrCol.from_values(probe)], 2, unique_expected=not dups)
    assert g.info()["code_words"] == 2


@pytest.mark.parametrize("dups", [False, True])
def test_random_16_byte_keys_three_words(ctx, dups):
    """16 random BYTES (UUID-like, NUL and 0xFF included): three code words, still carried by the entry (kHashK3)."""
    rng = np.random.default_rng(301 + dups)
    n, m = 40_000, 100_000
    build = [rng.integers(0, 256, 16, dtype=np.uint8).tobytes() for _ in range(n)]
    if dups:
        build = [build[i] for i in rng.integers(0, n // 5, n)]
    near = [b[:15] + bytes([b[15] ^ 1]) for b in build[:3000]] + [b[:8] for b in build[:500]] + [b"", b"\x00" * 16]
    probe = probe_mix(rng, build, m, near)
    g, o, mt = check_hash_join(ctx, [StrCol.from_values(build)], [StrCol.from_values(probe)], 2, unique_expected=not dups)
    assert g.info()["code_words"] == 3


@pytest.mark.parametrize("dups", [False, True])
def test_random_30_byte_keys_take_the_tag_path(ctx, dups):
    """20-30 random bytes: four or five code words -> 64-bit tags in the table, verified against the sorted codes."""
    rng = np.random.default_rng(351 + dups)
    n, m = 30_000, 80_000
    build = [rng.integers(0, 256, int(rng.integers(20, 31)), dtype=np.uint8).tobytes() for _ in range(n)]
    if dups:
        build = [build[i] for i in rng.integers(0, n // 5, n)]
    near = [b[:-1] + bytes([b[-1] ^ 1]) for b in build[:3000]] + [b[:8] for b in build[:500]] + [b + b"\x00" for b in build[:500]] + [b""]
    probe = probe_mix(rng, build, m, near)
    g, o, mt = check_hash_join(ctx, [StrCol.from_values(build)], [StrCol.from_values(probe)], 3, unique_expected=not dups)
    assert g.info()["code_words"] >= 4


def test_two_column_keys(ctx):
    """IndexOn(a, b) with two random columns: multi-column codes through the generic probe kernel + hash table;
    a prefix join on the first column alone still answers from the sorted codes (csvplus.go:910)."""
    rng = np.random.default_rng(401)
    n, m = 50_000, 120_000
    a = fixed_random(rng, n // 10, 6, ALNUM62)
    b = fixed_random(rng, n // 10, 7, ALNUM36)
    ba = [a[i] for i in rng.integers(0, len(a), n)]
    bb = [b[i] for i in rng.integers(0, len(b), n)]
    pa = [a[i] for i in rng.integers(0, len(a), m)]
    pb = [b[i] for i in rng.integers(0, len(b), m)]
    for i in range(0, m, 3):      # a third of the probe rows repeat a build row's pair
        k = int(rng.integers(0, n))
        pa[i], pb[i] = ba[k], bb[k]
    pa[5], pb[7] = b"", b"\xfe\xfe"
    bcols = [StrCol.from_values(ba), StrCol.from_values(bb)]
    pcols = [StrCol.from_values(pa), StrCol.from_values(pb)]
    g, o, mt = check_hash_join(ctx, bcols, pcols, (1, 2))
    assert 0 < mt.nmatches
    # prefix join: leading column only -> ordered path, many matches per probe row
    mp = g.probe(pcols[:1])
    assert_join_equal(mp, o.join(pcols[:1]))
    assert mp.nmatches > mt.nmatches


@pytest.mark.parametrize("ncols", [1, 2])
def test_400_byte_keys_several_windows(ctx, ncols):
    """Keys beyond 128 byte positions are cut into codec windows; the full-key Join hashes all windows' words
    (one pass per window), looks the tag up once and verifies window by window — no binary search."""
    rng = np.random.default_rng(500 + ncols)
    n, m = 6000, 15000
    shared = rng.integers(97, 123, 300, dtype=np.uint8).tobytes()    # long common prefix: only the tail tells keys apart

    def key():
        ln = int(rng.integers(330, 401))
        return shared[:300] + rng.integers(0, 256, ln - 300, dtype=np.uint8).tobytes()

    pool = [key() for _ in range(n // 2)]
    build = [pool[i] for i in rng.integers(0, len(pool), n)]          # duplicates: runs of equal 400-byte keys
    other = [key() for _ in range(2000)] + [shared, shared[:299], b"", pool[0] + b"x"]
    probe = probe_mix(rng, build, m, other)
    if ncols == 1:
        bcols, pcols = [StrCol.from_values(build)], [StrCol.from_values(probe)]
    else:
        small = [b"k%d" % i for i in range(7)]
        bcols = [StrCol.from_values([small[i % 7] for i in range(n)]), StrCol.from_values(build)]
        pcols = [StrCol.from_values([small[i % 7] for i in range(m)]), StrCol.from_values(probe)]
    g, o, mt = check_hash_join(ctx, bcols, pcols, 3)
    assert 0 < mt.nmatches
    if ncols == 2:   # prefix join across windows: ordered path
        mp = g.probe(pcols[:1])
        assert_join_equal(mp, o.join(pcols[:1]))


def test_prepare_join_builds_the_table_up_front(ctx):
    rng = np.random.default_rng(601)
    build = fixed_random(rng, 20_000, 12, ALNUM36)
    g = DeviceIndex(ctx, [StrCol.from_values(build)], unique=True)
    assert g.info()["lookup_built"] == 0
    g.prepare_join()
    inf = g.info()
    assert inf["lookup_built"] == HASH_BUILT and inf["hash_mode"] == 1 and inf["hash_bytes"] >= 16 * 2 * 20_000   # one 16-byte slot per distinct key at load 0.5
    probe = StrCol.from_values(build[::3] + [b"nope"])
    assert_join_equal(g.probe([probe]), orc.OracleIndex([StrCol.from_values(build)]).join([probe]))
    # a dense code space gets a direct-address table instead, and prepare_join(chained) its 4-byte form
    d = DeviceIndex(ctx, [StrCol.from_values([b"%06d" % i for i in range(5000)])], unique=True)
    d.prepare_join(chained=True)
    assert d.info()["lookup_built"] == 2
    d.prepare_join()
    assert d.info()["lookup_built"] == 3


def test_small_and_degenerate_tables(ctx):
    """1-row and 2-row indexes, all-equal keys, an empty index: the hash path has no size threshold."""
    for build in ([b"q8Zk3LmN0pQr"], [b"q8Zk3LmN0pQr", b"A8Zk3LmN0pQs"], [b"samekeysamek"] * 50):
        cols = [StrCol.from_values(build)]
        probe = [StrCol.from_values([build[0], b"", b"q8Zk3LmN0pQ", build[-1], b"zzzzzzzzzzzz"] * 3)]
        g, o = DeviceIndex(ctx, cols), orc.OracleIndex(cols)
        assert_join_equal(g.probe(probe), o.join(probe))
    e = DeviceIndex(ctx, [StrCol.from_values([])])
    m = e.probe([StrCol.from_values([b"a", b""])])
    assert m.nmatches == 0 and list(m.cnt) == [0, 0]


def test_chain_over_sparse_unique_keys_uses_the_hash_table(ctx):
    """The fused chained-join kernel with a hash-table step (sparse unique ids) next to a direct-table step."""
    rng = np.random.default_rng(701)
    nc, npd, m = 40_000, 800, 200_000
    cust = fixed_random(rng, nc, 12, ALNUM36)
    prod = [b"%d" % i for i in rng.permutation(npd)]
    kc = [cust[i] for i in rng.integers(0, nc, m)]
    kp = [prod[i] for i in rng.integers(0, npd, m)]
    for i in range(0, m, 11):
        kc[i] = fixed_random(rng, 1, 12, ALNUM36)[0]          # misses
    ch = check_chain(ctx, [[StrCol.from_values(cust)], [StrCol.from_values(prod)]],
                     [StrCol.from_values(kc), StrCol.from_values(kp)], probe_base=777)
    assert 0 < ch.nrows < m


def test_chain_hash_step_info_and_every_row_joins(ctx):
    rng = np.random.default_rng(702)
    nc, m = 30_000, 100_000
    cust = fixed_random(rng, nc, 10, ALNUM36)
    g = DeviceIndex(ctx, [StrCol.from_values(cust)], unique=True)
    o = orc.OracleIndex([StrCol.from_values(cust)])
    keys = StrCol.from_values([cust[i] for i in rng.integers(0, nc, m)])
    ch = join_chain(ctx, [(g, [keys])], probe_base=5)
    inf = g.info()
    assert inf["lookup_built"] == HASH_BUILT and inf["hash_mode"] == 1, inf
    es, erows = oracle_chain([o], [keys], 5)
    assert ch.nrows == m == len(es) and ch.identity
    np.testing.assert_array_equal(ch.build_row(0), erows[0])


def test_stream_join_over_sparse_unique_keys(ctx):
    """cph_stream_join_* accepts an index without a direct table: its chunks go through the hash step."""
    rng = np.random.default_rng(801)
    nc = 25_000
    cust = fixed_random(rng, nc, 12, ALNUM36)
    g = DeviceIndex(ctx, [StrCol.from_values(cust)], unique=True)
    o = orc.OracleIndex([StrCol.from_values(cust)])
    sj = StreamJoin(ctx, [g], nslots=2)
    assert g.info()["lookup_built"] == HASH_BUILT
    base = 0
    for n in (30_000, 1, 4097):
        keys = [cust[i] for i in rng.integers(0, nc, n)]
        keys[::7] = [b"000000000000"] * len(keys[::7])
        col = StrCol.from_values(keys)
        sj.submit([col], probe_base=base)
        r = sj.next()
        j = o.join([col], probe_base=base)
        hit = bitmap_to_rows(r["bitmap"], r["nrows"])
        assert r["nmatches"] == j["nmatches"]
        np.testing.assert_array_equal(hit + base, j["probe_idx"].astype(np.int64))
        np.testing.assert_array_equal(r["build_row"][0][hit], j["build_row"])
        base += n
    sj.close()


def test_index_shared_between_two_ctxs():
    """The lookup structure lives in the INDEX's ctx (pool + stream) whichever ctx runs the first Join: the second ctx
    may go away first, and the index's own ctx keeps using the table (use-after-free otherwise)."""
    rng = np.random.default_rng(901)
    a, b = Context(0), Context(0)
    build = fixed_random(rng, 30_000, 12, ALNUM36)
    dense = [b"%07d" % i for i in rng.permutation(30_000)]
    probe = StrCol.from_values([build[i] for i in rng.integers(0, 30_000, 80_000)])
    dprobe = StrCol.from_values([dense[i] for i in rng.integers(0, 30_000, 80_000)])
    for vals, pr, bit in ((build, probe, HASH_BUILT), (dense, dprobe, 1)):
        col = StrCol.from_values(vals)
        g = DeviceIndex(a, [col], unique=True)
        oj = orc.OracleIndex([col]).join([pr])
        # first Join from the OTHER ctx
        out = N.C.POINTER(N.cph_matches)()
        arr, keep = N._cols_array([pr])
        b._check(b.lib.cph_join_probe(b.handle, g.handle, arr, 1, None, 32, 0, 0, 0, 1, N.CPH_MEM_HOST, N.C.byref(out)))
        m = N.Matches(b.lib, out)
        assert g.info()["lookup_built"] & bit
        assert_join_equal(m, oj)
        m.release()
        assert_join_equal(g.probe([pr]), oj)      # and from its own ctx
        g.close()
    b.close()
    # ctx b is gone; an index of ctx a joined from b earlier must still be usable — rebuild the scenario in that order
    col = StrCol.from_values(build)
    g = DeviceIndex(a, [col], unique=True)
    b = Context(0)
    out = N.C.POINTER(N.cph_matches)()
    arr, keep = N._cols_array([probe])
    b._check(b.lib.cph_join_probe(b.handle, g.handle, arr, 1, None, 32, 0, 0, 0, 1, N.CPH_MEM_HOST, N.C.byref(out)))
    N.Matches(b.lib, out).release()
    b.close()
    assert_join_equal(g.probe([probe]), orc.OracleIndex([col]).join([probe]))
    g.close()
    a.close()


@pytest.mark.parametrize("kind", ["k1", "k3", "tag"])
def test_slice_by_slice_build_at_its_own_size(kind):
    """2.3e6 distinct sparse keys (the size from which the slice-by-slice build is taken by itself): the table answers exactly like the
    one built by compare-and-swap — hits with their build rows, misses — for one-word, three-word and tagged entries; the slices show
    in the table's size (whole 64 KB slices)."""
    rng = np.random.default_rng({"k1": 1, "k3": 2, "tag": 3}[kind])
    n, m = 2_300_000, 400_000
    width = {"k1": 12, "k3": 16, "tag": 28}[kind]
    alphabet = ALNUM36 if kind == "k1" else np.arange(256, dtype=np.uint8)
    mat = alphabet[rng.integers(0, len(alphabet), (n, width))]
    mat = np.unique(mat, axis=0)
    rng.shuffle(mat)
    n = len(mat)
    build = StrCol.from_arrays(np.ascontiguousarray(mat).reshape(-1), (np.arange(n + 1, dtype=np.uint64) * width).astype(np.uint32))
    pm = mat[rng.integers(0, n, m)].copy()
    pm[::3, width - 1] ^= 1                                  # a third of the probe keys: (almost surely) absent neighbours
    probe = StrCol.from_arrays(np.ascontiguousarray(pm).reshape(-1), (np.arange(m + 1, dtype=np.uint64) * width).astype(np.uint32))
    res = {}
    for part in (1, 0):
        c = Context(0)
        c.set_option("hash_partitioned", part)
        g = DeviceIndex(c, [build])
        mt = g.probe([probe])
        inf = g.info()
        assert inf["lookup_built"] & HASH_BUILT and inf["hash_mode"] == {"k1": 1, "k3": 2, "tag": 3}[kind], inf
        assert (inf["hash_bytes"] % 65536 == 0) == (part == 1), inf
        res[part] = (np.asarray(mt.probe_idx).copy(), np.asarray(mt.build_row).copy(), mt.nmatches)
        mt.release(); g.close(); c.close()
    assert res[0][2] == res[1][2] and m // 2 < res[1][2] < m
    np.testing.assert_array_equal(res[0][0], res[1][0])
    np.testing.assert_array_equal(res[0][1], res[1][1])
    # every reported pair is a true match (bytes of the build row == bytes of the probe row)
    pi, br = res[1][0][:: 97], res[1][1][:: 97]
    assert np.array_equal(mat[br], pm[pi])

"""duplicate groups / ResolveDuplicates (csvplus.go:643-653, :810-867) and index
persistence (:655-705) on the device index, against the oracle's literal restatement of dedup."""
from __future__ import annotations

import numpy as np
import pytest

from oracle import orc
from tests.helpers import random_keys


# ---- CPU: the restated dedup, on the cases SURVEY.md §2 derives by hand from csvplus.go:810-867 ----------------
def _rows(keys):
    return [{"k": k, "id": str(i)} for i, k in enumerate(keys)]


def test_oracle_dedup_tail_rule():
    first = lambda rows: rows[0]
    assert [r["k"] for r in orc.dedup_rows(_rows("AAB"), ["k"], first)] == ["A"]
    assert [r["k"] for r in orc.dedup_rows(_rows("AABC"), ["k"], first)] == ["A", "B"]
    assert [r["k"] for r in orc.dedup_rows(_rows("ABB"), ["k"], first)] == ["A", "B"]
    assert [r["k"] for r in orc.dedup_rows(_rows("ABC"), ["k"], first)] == ["A", "B", "C"]      # no duplicates: untouched
    assert [r["k"] for r in orc.dedup_rows(_rows("AABBC"), ["k"], first)] == ["A", "B"]
    assert [r["k"] for r in orc.dedup_rows(_rows("AABCC"), ["k"], lambda rows: {})] == ["B"]     # empty row drops the group
    assert [r["id"] for r in orc.dedup_rows(_rows("AAAB"), ["k"], lambda rows: rows[-1])] == ["2"]


def test_oracle_dedup_like_TestResolver():
    """csvplus_test.go:695-752: one duplicated person, resolver called exactly once with n+1 equal rows."""
    from tests.helpers import people_table
    p = people_table()
    rng = np.random.default_rng(3)
    src = [{"id": i, "name": n, "surname": s} for i, n, s in zip(p["id"], p["name"], p["surname"])]
    for _ in range(20):
        rows = list(src)
        dup = rows[int(rng.integers(0, len(rows)))]
        n = int(rng.integers(1, 101))
        rows += [dup] * n
        rows.sort(key=lambda r: (r["name"].encode(), r["surname"].encode()))
        calls = []

        def resolve(group):
            calls.append(len(group))
            assert all(r == dup for r in group)
            return group[0]
        out = orc.dedup_rows(rows, ["name", "surname"], resolve)
        assert calls == [n + 1]
        assert len(out) in (len(src), len(src) - 1)   # the tail rule costs the final row unless it is in the group


def test_dedup_positions_matches_oracle_without_gpu():
    """The host replay (csvplus_amd.dedup) against the restated loop, groups computed on the CPU."""
    from csvplus_amd.dedup import dedup_positions
    rng = np.random.default_rng(5)
    for it in range(300):
        n = int(rng.integers(0, 40))
        keys = sorted(random_keys(rng, n, 0, 2, alphabet=np.frombuffer(b"ab", np.uint8)))
        rows = [{"k": k, "id": str(i)} for i, k in enumerate(keys)]
        lower, upper, i = [], [], 0
        while i < n:
            j = i
            while j + 1 < n and keys[j + 1] == keys[i]:
                j += 1
            if j > i:
                lower.append(i)
                upper.append(j + 1)
            i = j + 1
        mode = it % 3
        pick = {0: lambda lo, hi: lo, 1: lambda lo, hi: hi - 1, 2: lambda lo, hi: None if (lo & 1) else lo}[mode]
        opick = {0: lambda g: g[0], 1: lambda g: g[-1], 2: lambda g: {} if (int(g[0]["id"]) & 1) else g[0]}[mode]
        want = [int(r["id"]) for r in orc.dedup_rows(rows, ["k"], opick)]
        got = dedup_positions(n, lower, upper, pick).tolist()
        assert got == want, (keys, mode)


# ---- GPU ----------------------------------------------------------------------------------------------------
def _build(ctx, keycols):
    from csvplus_amd import _native as N
    return N.DeviceIndex(ctx, keycols)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["key32", "one_word", "multi_word", "two_cols"])
def test_gpu_groups_and_resolve_match_oracle(ctx, shape):
    from csvplus_amd import StrCol
    from csvplus_amd.dedup import resolve_duplicates
    rng = np.random.default_rng(23)
    for it in range(12):
        n = int(rng.integers(0, 3000)) if it else 0
        if shape == "key32":
            cols_vals = [random_keys(rng, n, 0, 3, alphabet=np.frombuffer(b"abc", np.uint8))]
        elif shape == "one_word":
            cols_vals = [random_keys(rng, n, 3, 9, alphabet=np.frombuffer(b"0123456789", np.uint8), distinct=max(1, n // 3))]
        elif shape == "multi_word":
            cols_vals = [random_keys(rng, n, 10, 40, distinct=max(1, n // 4))]
        else:
            cols_vals = [random_keys(rng, n, 0, 2, alphabet=np.frombuffer(b"xy", np.uint8)),
                         random_keys(rng, n, 0, 2, alphabet=np.frombuffer(b"pq", np.uint8))]
        names = [f"c{i}" for i in range(len(cols_vals))]
        ix = _build(ctx, [StrCol.from_values(v) for v in cols_vals])
        perm = ix.perm()
        o = orc.OracleIndex([StrCol.from_values(v) for v in cols_vals])
        assert perm.tolist() == o.perm.tolist()
        rows = [dict({nm: cols_vals[c][int(r)] for c, nm in enumerate(names)}, id=str(int(r))) for r in perm]
        # groups
        lower, upper = ix.dup_groups()
        want_groups = []
        i = 0
        key = lambda r: tuple(r[nm] for nm in names)
        while i < n:
            j = i
            while j + 1 < n and key(rows[j + 1]) == key(rows[i]):
                j += 1
            if j > i:
                want_groups.append((i, j + 1))
            i = j + 1
        assert list(zip(lower.tolist(), upper.tolist())) == want_groups
        # resolve: keep the row with the largest original id / drop odd groups
        for mode in range(2):
            if mode == 0:
                pick = lambda lo, hi: lo + int(np.argmax(perm[lo:hi]))
                opick = lambda g: max(g, key=lambda r: int(r["id"]))
            else:
                pick = lambda lo, hi: None if (lo % 3 == 0) else hi - 1
                opick = lambda g: {} if (rows.index(g[0]) % 3 == 0) else g[-1]
            want = [int(r["id"]) for r in orc.dedup_rows(rows, names, opick)]
            nx = resolve_duplicates(ix, pick)
            assert nx.perm().tolist() == want
            # the compacted index still answers probes like a fresh index over the surviving rows
            if want:
                sub = [StrCol.from_values([cols_vals[c][r] for r in want]) for c in range(len(names))]
                probe = [StrCol.from_values(v) for v in cols_vals]
                m = nx.probe(probe)
                o2 = orc.OracleIndex(sub).join(probe)
                assert m.cnt.tolist() == o2["cnt"].tolist()
                assert m.probe_idx.tolist() == o2["probe_idx"].tolist()
                assert m.build_row.tolist() == [want[int(b)] for b in o2["build_row"]]
            nx.close()
        ix.close()


@pytest.mark.gpu
def test_gpu_select_rejects_bad_positions(ctx):
    from csvplus_amd import StrCol
    ix = _build(ctx, [StrCol.from_values(["b", "a", "c"])])
    for bad in ([1, 1], [2, 1], [0, 3]):
        with pytest.raises(Exception):
            ix.select(bad)
    assert ix.select([0, 2]).perm().tolist() == [1, 2]
    assert ix.select([]).nrows == 0


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["key32", "multi_word"])
def test_gpu_save_load_roundtrip(ctx, tmp_path, shape):
    from csvplus_amd import StrCol
    from csvplus_amd import _native as N
    rng = np.random.default_rng(29)
    n = 5000
    vals = (random_keys(rng, n, 0, 3, alphabet=np.frombuffer(b"abc", np.uint8)) if shape == "key32"
            else random_keys(rng, n, 10, 40, distinct=n // 2))
    col = StrCol.from_values(vals)
    ix = _build(ctx, [col])
    path = tmp_path / "index.cph"
    ix.save(str(path))
    ld = N.DeviceIndex.load(ctx, str(path))
    assert ld.nrows == n and ld.perm().tolist() == ix.perm().tolist()
    probe = [StrCol.from_values(random_keys(rng, 2000, 0, 40) + vals[:500])]
    a, b = ix.probe(probe), ld.probe(probe)
    assert a.lo.tolist() == b.lo.tolist() and a.cnt.tolist() == b.cnt.tolist()
    assert a.build_row.tolist() == b.build_row.tolist()
    assert ld.find(vals[7]) == ix.find(vals[7])
    # damaged files are rejected, not half-loaded
    raw = path.read_bytes()
    (tmp_path / "short.cph").write_bytes(raw[: len(raw) // 2])
    (tmp_path / "magic.cph").write_bytes(b"GOBGOBGO" + raw[8:])
    for name in ("short.cph", "magic.cph", "missing.cph"):
        with pytest.raises(Exception):
            N.DeviceIndex.load(ctx, str(tmp_path / name))


@pytest.mark.gpu
def test_gpu_load_rejects_corrupt_payload(ctx, tmp_path):
    """A well-formed descriptor followed by a damaged payload (codes outside the codec's state space or out of
    order, perm that is not a set of row ids) must be refused: it would otherwise drive table builds and the
    caller's gathers out of bounds (the reference's gob decode is memory-safe, csvplus.go:693-702)."""
    from csvplus_amd import StrCol
    from csvplus_amd import _native as N
    n = 4000
    col = StrCol.from_values([b"%05d" % ((i * 7919) % n) for i in range(n)])
    ix = _build(ctx, [col])
    assert ix.info()["key_bytes"] == 4
    path = tmp_path / "index.cph"
    ix.save(str(path))
    raw = bytearray(path.read_bytes())
    payload = len(raw) - n * 8            # u32 codes, then u32 perm
    codes = np.frombuffer(bytes(raw[payload:payload + 4 * n]), dtype=np.uint32).copy()
    perm = np.frombuffer(bytes(raw[payload + 4 * n:]), dtype=np.uint32).copy()
    assert sorted(perm.tolist()) == list(range(n)) and (np.diff(codes.astype(np.int64)) >= 0).all()

    def variant(name, c, p):
        out = bytearray(raw)
        out[payload:payload + 4 * n] = c.astype(np.uint32).tobytes()
        out[payload + 4 * n:] = p.astype(np.uint32).tobytes()
        f = tmp_path / name
        f.write_bytes(bytes(out))
        return str(f)

    N.DeviceIndex.load(ctx, variant("same.cph", codes, perm)).close()       # the untouched payload loads
    big = codes.copy(); big[-1] = 0xFFFFFFF0                                 # code beyond the state space
    swapped = codes.copy(); swapped[[10, 2000]] = swapped[[2000, 10]]       # not sorted
    dup = perm.copy(); dup[5] = dup[6]                                      # a row id twice
    far = perm.copy(); far[0] = n + 12345                                   # row id outside the table
    for name, c, p in (("big.cph", big, perm), ("swapped.cph", swapped, perm), ("dup.cph", codes, dup), ("far.cph", codes, far)):
        with pytest.raises(N.CphError) as e:
            N.DeviceIndex.load(ctx, variant(name, c, p))
        assert e.value.code == N.CPH_ERR_INVALID, name

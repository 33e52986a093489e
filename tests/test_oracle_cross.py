"""A SECOND restatement of the path, written row-at-a-time the way csvplus.go reads — rows are dicts (Row = map[string]string, :59),
the index is the sorted list of rows (:612-614, :736), Join walks the stream, calls first() (sort.Search, :893-897) and emits
mergeRows(index row, stream row) while the keys compare equal (:553-567, :571-583), Except emits the rows without a match (:588-608),
UniqueIndexOn reports the first adjacent equal pair (:749-753) — in plain Python, with none of the oracle's code.  The C oracle
(oracle/csvplus_oracle.c) must agree with it on random tables: two independent restatements pin each other where the reference
itself cannot run (no Go toolchain).  Python's sorted() is stable, Go's sort.Sort is not: equal keys are compared as the oracle
defines them (input order), which is one of the orders sort.Sort may produce."""
import bisect

import numpy as np
import pytest

from csvplus_amd import StrCol
from oracle import orc


def py_index(rows, columns):
    """createIndex (:707-738): the rows sorted by the tuple of their key columns, bytewise (strings.Compare), stable."""
    order = sorted(range(len(rows)), key=lambda i: tuple(rows[i][c] for c in columns))
    return order


def py_first_dup(rows, order, columns):
    """createUniqueIndex (:740-756): the first sorted position i >= 1 whose key equals its predecessor's."""
    for i in range(1, len(order)):
        if all(rows[order[i - 1]][c] == rows[order[i]][c] for c in columns):
            return i
    return None


def py_join(index_rows, order, index_cols, stream, stream_cols):
    """Join (:545-569) with a prefix of the index columns: (stream row number, index row number) per emitted row, and the merged rows."""
    keys = [tuple(index_rows[i][c] for c in index_cols[:len(stream_cols)]) for i in order]
    pairs, merged = [], []
    for r, row in enumerate(stream):
        values = tuple(row[c] for c in stream_cols)                   # SelectValues (:138-150)
        i = bisect.bisect_left(keys, values)                          # first(): smallest i with keys[i] >= values
        while i < len(keys) and keys[i] == values:                    # !cmp(i, values, false)
            m = dict(index_rows[order[i]])
            m.update(row)                                             # mergeRows: the stream's value wins (:578-580)
            pairs.append((r, order[i]))
            merged.append(m)
            i += 1
    return pairs, merged


def random_values(rng, n, pool_size, maxlen, alphabet):
    pool = [bytes(rng.choice(alphabet, int(rng.integers(0, maxlen + 1)))) for _ in range(pool_size)]
    return [pool[int(i)] for i in rng.integers(0, pool_size, n)]


@pytest.mark.parametrize("seed", range(12))
def test_two_restatements_agree(seed):
    rng = np.random.default_rng(1000 + seed)
    alphabet = np.frombuffer(b"ab\x00\xffz01", dtype=np.uint8)      # NUL and high bytes are ordinary; prefixes sort first
    n, m = int(rng.integers(0, 400)), int(rng.integers(0, 600))
    ncols = int(rng.integers(1, 4))
    index_cols = [f"k{c}" for c in range(ncols)]
    build = [{**{f"k{c}": v for c, v in enumerate(vals)}, "payload": b"b%d" % i}
             for i, vals in enumerate(zip(*[random_values(rng, n, int(rng.integers(1, 40)), 5, alphabet) for _ in range(ncols)]))] if n else []
    nprobe_cols = int(rng.integers(1, ncols + 1))                     # fewer columns than the index has: a prefix join (:546-550)
    stream_cols = [f"s{c}" for c in range(nprobe_cols)]
    stream = [{**{f"s{c}": v for c, v in enumerate(vals)}, "payload": b"s%d" % i}
              for i, vals in enumerate(zip(*[random_values(rng, m, int(rng.integers(1, 40)), 5, alphabet) for _ in range(nprobe_cols)]))] if m else []
    order = py_index(build, index_cols)
    bcols = [StrCol.from_values([r[c] for r in build]) for c in index_cols]
    scols = [StrCol.from_values([r[c] for r in stream]) for c in stream_cols]
    o = orc.OracleIndex(bcols)
    assert list(o.perm) == order
    assert o.first_dup() == py_first_dup(build, order, index_cols)
    pairs, merged = py_join(build, order, index_cols, stream, stream_cols)
    j = o.join(scols)
    assert j["nmatches"] == len(pairs)
    assert list(zip(j["probe_idx"].tolist(), j["build_row"].tolist())) == pairs
    assert all(mr["payload"] == stream[r]["payload"] for (r, _), mr in zip(pairs, merged))   # the stream's value wins on a shared name
    # Except (:588-608): the stream rows without a match, in stream order
    matched = {r for r, _ in pairs}
    assert [r for r in range(m) if r not in matched] == np.flatnonzero(j["cnt"] == 0).tolist()
    # find (:870-891): [lower, upper) of a key prefix
    if n:
        probe = tuple(build[int(rng.integers(0, n))][c] for c in index_cols[:nprobe_cols])
        keys = [tuple(build[i][c] for c in index_cols[:nprobe_cols]) for i in order]
        lo, hi = o.find(*probe)
        assert (lo, hi) == (bisect.bisect_left(keys, probe), bisect.bisect_right(keys, probe))

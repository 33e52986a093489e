"""Index.ResolveDuplicates (csvplus.go:643-653) over the device index.

The device finds every duplicate group in one pass (cph_index_dup_groups); the callback and the compaction
rule of dedup (csvplus.go:810-867) are replayed here on the host over that list, then cph_index_select builds
the compacted index.  The replay keeps the reference's behaviour to the letter, including its tail rule: once
at least one group was resolved, the rows after the LAST group are copied by the loop :851-859, which moves
rows[lower-1] only while lower < len(rows) — the final row of the index is therefore dropped unless it belongs
to the last duplicate group ([A,A,B] -> [A], [A,A,B,C] -> [A,B]).  `keep_last_row=True` opts out of that.
"""
from __future__ import annotations

import numpy as np

from . import _native as N


def dedup_positions(nrows: int, lower, upper, resolve, keep_last_row: bool = False) -> np.ndarray:
    """Sorted positions that survive dedup.  resolve(lo, hi) -> a position in [lo, hi) (the chosen row), or None
    (the reference's "empty row": the whole group is dropped); an exception propagates (the reference returns
    the callback's error, :835-837)."""
    ng = len(lower)
    if ng == 0:                                   # :821-823 no duplicates: nothing changes
        return np.arange(nrows, dtype=np.uint64)
    parts = [np.arange(0, int(lower[0]), dtype=np.uint64)]            # dest = lower-1 (:825)
    for g in range(ng):
        lo, hi = int(lower[g]), int(upper[g])
        choice = resolve(lo, hi)                                       # :835 resolve(rows[lower-1:upper])
        if choice is not None:                                         # :842-845 store the chosen row
            c = int(choice)
            if not lo <= c < hi:
                raise ValueError(f"resolver returned position {c} outside its group [{lo}, {hi})")
            parts.append(np.array([c], dtype=np.uint64))
        # :848-859 copy the non-duplicates up to the next group; rows[lower-1] moves only while lower < len
        stop = int(lower[g + 1]) if g + 1 < ng else (nrows if keep_last_row else nrows - 1)
        if stop > hi:
            parts.append(np.arange(hi, stop, dtype=np.uint64))
    return np.concatenate(parts)


def resolve_duplicates(index: N.DeviceIndex, resolve, keep_last_row: bool = False) -> N.DeviceIndex:
    """Returns the deduplicated index (the input index is left as it was: a failing callback changes nothing,
    where the reference leaves its rows half-compacted)."""
    lower, upper = index.dup_groups()
    pos = dedup_positions(index.nrows, lower, upper, resolve, keep_last_row)
    return index.select(pos)

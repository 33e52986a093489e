"""Size-independent checks of what the library returned, for outputs too large for the CPU checker.

Nothing here calls the library or the CPU restatement under the repo's checker directory: the checks re-derive the defining properties of the
reference's results from the INPUT columns with plain numpy / torch indexing —

  * an Index is its rows sorted by `Less` (csvplus.go:736, :794-807): `perm` is a permutation of
    0..n-1, the keys read through it ascend bytewise, and rows with equal keys keep input order
    (the canonical order of DESIGN.md §2);
  * a Join against a unique index emits, per stream row in stream order, the ONE build row whose
    key equals the stream row's key (csvplus.go:553-567): key bytes of `build_row[r]` == key bytes
    of stream row r.

bench.py runs them after its timed loop (on the 1e8-row outputs it has just timed) and the full-size
GPU tests use them too.
"""
from __future__ import annotations

import numpy as np


def digest_u64(t) -> int:
    """Order-dependent 64-bit digest of an integer tensor / array: sum((i + 1) * x[i]) mod 2^64."""
    import torch

    if isinstance(t, np.ndarray):
        x = t.astype(np.uint64, copy=False)
        w = np.arange(1, x.size + 1, dtype=np.uint64)
        with np.errstate(over="ignore"):
            return int((x * w).sum(dtype=np.uint64))
    x = t.to(torch.int64)
    if t.dtype == torch.int32:
        x = x & 0xFFFFFFFF
    w = torch.arange(1, x.numel() + 1, dtype=torch.int64, device=x.device)
    return int((x * w).sum().item()) & 0xFFFFFFFFFFFFFFFF


def sample_rows(n: int, k: int, seed: int = 12345) -> np.ndarray:
    rng = np.random.default_rng(seed)
    k = min(k, n)
    rows = rng.integers(0, n, size=k, dtype=np.int64)
    # always include the ends
    if k >= 2:
        rows[0], rows[1] = 0, n - 1
    return rows


def check_join_sample(stream_col, build_col, build_rows_at, rows: np.ndarray, row0: int = 0) -> int:
    """stream_col / build_col: HOST StrCol key columns; build_rows_at: the build row ids the join
    reported for the sampled result rows `rows` (result row m == stream row row0 + m: every stream
    row joined).  Returns the number of sampled rows whose key bytes differ."""
    bad = 0
    br = np.asarray(build_rows_at).astype(np.int64) & 0xFFFFFFFF
    for r, b in zip(rows.tolist(), br.tolist()):
        if b >= build_col.nrows or stream_col.value(r - row0 if row0 else r) != build_col.value(b):
            bad += 1
    return bad


def _be_words(data, begin, length, nwords: int):
    """Per row: the first 8*nwords key bytes as big-endian int64 words biased to compare like
    unsigned (zero padded: equals strings.Compare order when the keys hold no NUL byte)."""
    import torch

    words = []
    for w in range(nwords):
        acc = torch.zeros_like(begin)
        for j in range(8):
            q = 8 * w + j
            inside = length > q
            idx = torch.where(inside, begin + q, torch.zeros_like(begin))
            b = data[idx].to(torch.int64)
            b = torch.where(inside, b, torch.zeros_like(b))
            acc = acc | (b << (8 * (7 - j)))
        words.append(acc ^ (-0x8000000000000000))   # unsigned order on signed int64
    return words


def check_index_order(key_col_dev, perm_dev, nwords: int | None = None) -> dict:
    """key_col_dev: DEVICE StrCol (single key column, no NUL bytes, at most 8*nwords bytes per key);
    perm_dev: torch int32/int64 tensor (sorted position -> input row).  All on the device."""
    import torch

    n = key_col_dev.nrows
    perm = perm_dev.to(torch.int64) & 0xFFFFFFFF
    out = {"n": n, "perm_len_ok": perm.numel() == n}
    if n == 0 or not out["perm_len_ok"]:
        return out
    seen = torch.zeros(n, dtype=torch.uint8, device=perm.device)
    in_range = bool((perm < n).all().item())
    if in_range:
        seen.scatter_(0, perm, torch.ones_like(perm, dtype=torch.uint8))
    out["is_permutation"] = in_range and bool(seen.all().item())
    del seen
    if not out["is_permutation"]:
        return out
    data = key_col_dev.data
    if key_col_dev.fixed_width:
        w = key_col_dev.fixed_width
        begin = perm * w
        length = torch.full_like(perm, w)
        maxlen = w
    else:
        odt = torch.int32 if key_col_dev.offset_bits == 32 else torch.int64
        offs = key_col_dev.offsets.view(odt).to(torch.int64)
        if key_col_dev.offset_bits == 32:
            offs = offs & 0xFFFFFFFF
        begin = offs[:-1][perm]
        length = offs[1:][perm] - begin
        maxlen = int(length.max().item())
        del offs
    nw = nwords or (maxlen + 7) // 8
    assert maxlen <= 8 * nw, f"keys of {maxlen} bytes do not fit {nw} words"
    words = _be_words(data, begin, length, nw)
    # lexicographic comparison of neighbours: less / equal masks
    less = torch.zeros(n - 1, dtype=torch.bool, device=perm.device)
    equal = torch.ones(n - 1, dtype=torch.bool, device=perm.device)
    for wv in words:
        a, b = wv[:-1], wv[1:]
        less |= equal & (a < b)
        equal &= a == b
    la, lb = length[:-1], length[1:]
    less |= equal & (la < lb)       # proper prefix first (only reachable with trailing NULs; kept for completeness)
    equal &= la == lb
    out["keys_ascend"] = bool((less | equal).all().item())
    out["equal_keys_keep_input_order"] = bool((~equal | (perm[:-1] < perm[1:])).all().item())
    out["equal_neighbours"] = int(equal.sum().item())
    out["ok"] = out["is_permutation"] and out["keys_ascend"] and out["equal_keys_keep_input_order"]
    return out

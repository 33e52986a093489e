#!/usr/bin/env python3
"""Differential fuzz of the round-4 build paths against the oracle: batches of 2-4 IndexOn / UniqueIndexOn calls through
cph_index_build_many (every second build on the side stream, pool blocks parked meanwhile), tables shaped for the direct sort
(distinct ids over a dense code space, with and without ONE duplicate somewhere), for the sampled alphabets (fixed-width keys,
>= 2^20 rows now and then, a rare byte in a row the sample does not visit) and ordinary ones; device- and host-resident columns.
usage: tools/fuzz_builds.py [seconds] [seed]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from csvplus_amd import Context, DeviceIndex, StrCol, _native as N
from oracle import orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed)
ctx = Context(0)
print("seed", seed, flush=True)


def fixed_ids(ids, width):
    raw = np.char.zfill(ids.astype(f"U{width}"), width).astype(f"S{width}")
    return StrCol.from_arrays(np.frombuffer(raw.tobytes(), np.uint8).copy(), np.arange(len(ids) + 1, dtype=np.uint32) * width, fixed_width=width)


def table():
    kind = int(rng.integers(0, 6))
    if kind <= 2:      # dense distinct ids: kind 0 full space, 1 up to twice the rows, 2 with one duplicate
        width = int(rng.integers(5, 8))
        space = 10 ** width if width < 7 else int(rng.integers(2, 9)) * 10 ** (width - 1)
        space = min(space, 1_000_000)
        n = space if kind == 0 and space <= 400_000 else int(rng.integers(max(space // 2, 66_000), space + 1))
        if n < 66_000:
            n, space = 100_000, 100_000
        ids = rng.permutation(space)[:n]
        if kind == 2:
            ids[int(rng.integers(1, n))] = ids[int(rng.integers(0, n))] if rng.random() < 0.7 else ids[0]
        # (round 5: every third table zero-padded to 8 bytes — the window sort then codes the keys inside its first partition level)
        w8 = rng.random() < 0.35
        return fixed_ids(ids, 8 if w8 else len(str(space - 1))), True, "dense%d%s" % (kind, "w8" if w8 else "")
    if kind == 3:      # the sampled-alphabet path: >= 2^20 rows, sometimes a byte only an unsampled row holds
        n = (1 << 20) + int(rng.integers(0, 50_000))
        ids = rng.integers(0, 5_000_000, n)
        col = fixed_ids(ids, 8)
        if rng.random() < 0.5:
            r = int(rng.integers(0, n)) | 1
            col.data[8 * r + int(rng.integers(0, 8))] = ord("x")
        return col, False, "sampled"
    if kind == 4:      # unpadded decimal ids (variable length)
        n = int(rng.integers(1000, 200_000))
        return StrCol.from_values([b"%d" % int(x) for x in rng.permutation(n * int(rng.integers(1, 4)))[:n]]), bool(rng.random() < 0.7), "itoa"
    n = int(rng.integers(1, 60_000))
    pool = [bytes(rng.integers(97, 103, int(rng.integers(0, 9))).astype(np.uint8)) for _ in range(int(rng.integers(1, 3000)))]
    return StrCol.from_values([pool[int(i)] for i in rng.integers(0, len(pool), n)]), False, "dups"


t_end = time.time() + budget
batches = builds = 0
kinds = {}
while time.time() < t_end:
    k = int(rng.integers(2, 5))
    tabs = [table() for _ in range(k)]
    ctx.set_option("build_side_stream", int(rng.random() < 0.85))
    specs = [([c.to_device("cuda:0")] if rng.random() < 0.6 else [c], u) for c, u, _ in tabs]
    res = DeviceIndex.build_many(ctx, specs)
    for (col, unique, kind), ix in zip(tabs, res):
        o = orc.OracleIndex([col])
        od = o.first_dup()
        if not np.array_equal(ix.perm(), o.perm) or ix.first_dup != od or ix.status != (N.CPH_ERR_DUPLICATE if unique and od is not None else N.CPH_OK):
            print("MISMATCH seed", seed, "batch", batches, kind, "n", col.nrows, "unique", unique, "first_dup", ix.first_dup, od, "status", ix.status, flush=True)
            sys.exit(1)
        kinds[kind] = kinds.get(kind, 0) + 1
        builds += 1
    pick = int(rng.integers(0, k))
    col = tabs[pick][0]
    probe = [StrCol.from_values([col.value(int(i)) for i in rng.integers(0, col.nrows, 500)] + [b"", b"zz", b"0000000"])]
    g, w = res[pick].probe(probe), orc.OracleIndex([col]).join(probe)
    if not (np.array_equal(g.probe_idx, w["probe_idx"]) and np.array_equal(g.build_row, w["build_row"])):
        print("JOIN MISMATCH seed", seed, "batch", batches, tabs[pick][2], flush=True)
        sys.exit(1)
    for ix in res:
        ix.close()
    batches += 1
print("FUZZ_BUILDS_OK seed", seed, "batches", batches, "builds", builds, kinds, flush=True)

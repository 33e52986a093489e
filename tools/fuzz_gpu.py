#!/usr/bin/env python3
"""Differential fuzz of the HIP path against the oracle on random tables (sizes 1..20000, 1-3 key columns, random alphabets
and lengths, duplicates or not): IndexOn (both build paths), Join (pairs / bounds only), chain (row ids / sorted positions),
Find / find_many.  Round 4: every eighth case is a table of 66 000..90 000 rows whose first key column is built like a
delimiter-split candidate (random heads + delimiter + digits, a few values without the delimiter, random delimiter byte), so
the split codec (keycodec.hip) is fuzzed with and without being taken; every seventh case has fixed-width 8-byte decimal
keys (the lean steps of the fused chain: arithmetic encode, identity / rank-table lookups).
usage: tools/fuzz_gpu.py [seconds] [seed]"""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from csvplus_amd import Context, DeviceIndex, StrCol, join_chain
from oracle import orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed)
ctx = Context(0)
print("seed", seed, flush=True)


def rand_col(n, distinct, alphabet, lo, hi):
    pool = [bytes(alphabet[rng.integers(0, len(alphabet), int(rng.integers(lo, hi + 1)))]) for _ in range(distinct)]
    return [pool[int(i)] for i in rng.integers(0, distinct, n)], pool


ALPHAS = [np.frombuffer(b"0123456789", np.uint8), np.frombuffer(b"abcxyz", np.uint8), np.arange(256, dtype=np.uint8),
          np.frombuffer(b"\x00\xffA", np.uint8), np.frombuffer(b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789", np.uint8)]
t_end = time.time() + budget
cases = splits = leans = 0
ONLY = int(os.environ.get("FUZZ_ONLY", "-1"))   # replay: generate every case of the seed, check only this one (and stop)
FROM = int(os.environ.get("FUZZ_FROM", str(ONLY)))   # ... or the cases FROM..ONLY (state left behind by earlier builds)
if os.environ.get("FUZZ_GUARD") == "1":
    ctx.set_option("pool_guard", 1)
while time.time() < t_end:
    n = int(rng.choice([1, 2, 63, 64, 65, 500, 4096, 8192, 8193, 16384, 16385, 20000])) if rng.random() < 0.5 else int(rng.integers(1, 20000))
    ncols = int(rng.integers(1, 4))
    unique_wanted = rng.random() < 0.4
    split_case = cases % 8 == 5
    lean_case = cases % 7 == 3
    if split_case:
        n = int(rng.integers(66_000, 90_000))
        unique_wanted = False
    build, pools = [], []
    for c in range(ncols):
        if split_case and c == 0:
            delim = bytes([int(rng.choice([35, 47, 58, 0, 255, 44, 124]))])
            nheads = int(rng.choice([1, 3, 40, 300, 1500]))
            hl = int(rng.integers(0, 14))
            heads = list({bytes(ALPHAS[int(rng.integers(0, 2)) + 1][rng.integers(0, 6, int(rng.integers(0, hl + 1)))]) for _ in range(nheads)})
            digits = int(rng.integers(1, 7))
            pool = [heads[int(rng.integers(0, len(heads)))] + delim + (b"%d" % int(rng.integers(0, 10 ** digits))) for _ in range(max(2, n // int(rng.integers(1, 6))))]
            for j in range(0, len(pool), int(rng.integers(150, 5000))):
                pool[j] = heads[int(rng.integers(0, len(heads)))]          # no delimiter at all
            vals = [pool[int(i)] for i in rng.integers(0, len(pool), n)]
            build.append(vals); pools.append(pool)
            continue
        if lean_case and c == 0:
            span = int(rng.choice([n, 3 * n]))
            base = int(rng.choice([0, 20_000_000]))
            pool = [b"%08d" % (base + int(x)) for x in rng.permutation(span)[:n]]
            build.append(pool); pools.append(pool)
            unique_wanted = True
            continue
        a = ALPHAS[int(rng.integers(0, len(ALPHAS)))]
        lo = int(rng.integers(0, 4)); hi = lo + int(rng.integers(0, 12))
        distinct = n * 4 if (unique_wanted and c == 0) else int(rng.integers(1, max(2, n)))
        if unique_wanted and c == 0:
            vals = [b"%d" % int(x) for x in rng.permutation(n * 3)[:n]] if rng.random() < 0.5 else [b"%07d" % int(x) for x in rng.permutation(n)]
            pool = vals
        else:
            vals, pool = rand_col(n, distinct, a, lo, hi)
        build.append(vals); pools.append(pool)
    m = int(rng.integers(1, 30000))
    probe = []
    for c in range(ncols):
        pv = [pools[c][int(i)] for i in rng.integers(0, len(pools[c]), m)]
        for j in rng.integers(0, m, m // 10):
            pv[int(j)] = pv[int(j)] + b"!" if rng.random() < 0.5 else pv[int(j)][:-1]
        probe.append(pv)
    if ONLY >= 0 and not (FROM <= cases <= ONLY):
        cases += 1
        t_end = time.time() + budget
        continue
    bcols = [StrCol.from_values(v) for v in build]
    pcols = [StrCol.from_values(v) for v in probe]
    o = orc.OracleIndex(bcols)
    for limit in (16384, 0):
        ctx.set_option("small_build_rows", limit)
        ctx.set_option("hash_partitioned", 2 if limit == 0 else 0)   # sparse keys: the table slice by slice (round 6) / by CAS over the whole table
        g = DeviceIndex(ctx, bcols)
        tag = (seed, cases, n, ncols, m, limit, g.info()["build_path"], g.info()["split"])
        if limit == 0:
            splits += g.info()["split"] != 0
            leans += lean_case
        if not np.array_equal(g.perm(), o.perm):
            gp, op = g.perm(), o.perm
            i = int(np.argmax(gp != op))
            print("PERM MISMATCH", tag, "first at sorted position", i, "info", g.info(), flush=True)
            print("  rows out of range in the GPU perm:", int((gp >= n).sum()), "| positions that differ:", int((gp != op).sum()), flush=True)
            for j in range(max(0, i - 2), min(n, i + 6)):
                print("  pos", j, "gpu row", int(gp[j]), [build[c][int(gp[j])] for c in range(ncols)] if gp[j] < n else "OUT OF RANGE",
                      "| oracle row", int(op[j]), [build[c][int(op[j])] for c in range(ncols)], flush=True)
            raise AssertionError(tag)
        assert g.first_dup == o.first_dup(), tag
        for k in range(1, ncols + 1):
            oj = o.join(pcols[:k])
            mt = g.probe(pcols[:k])
            assert np.array_equal(mt.cnt, oj["cnt"]) and mt.nmatches == oj["nmatches"], tag
            assert np.array_equal(mt.probe_idx, oj["probe_idx"]) and np.array_equal(mt.build_row, oj["build_row"]), tag
            mt.release()
            mb = g.probe(pcols[:k], want_pairs=False)
            nz = mb.cnt > 0
            assert np.array_equal(mb.cnt, oj["cnt"]) and np.array_equal(mb.lo[nz], oj["lo"][nz]), tag
            mb.release()
        oj = o.join(pcols)
        for pos in (False, True):
            ch = join_chain(ctx, [(g, pcols)], probe_base=7, positions=pos)
            assert ch.nrows == oj["nmatches"] and np.array_equal(ch.stream_row, oj["probe_idx"] + 7), tag
            rows = ch.build_row(0)
            if pos and ch.nrows:
                rows = g.perm()[rows]
            assert np.array_equal(rows, oj["build_row"]), (tag, pos)
            ch.release()
        keys = [tuple(build[c][int(r)] for c in range(ncols)) for r in rng.integers(0, n, 50)] + [tuple(probe[c][int(r)] for c in range(ncols)) for r in rng.integers(0, m, 50)]
        lo_, hi_ = g.find_many(keys)
        for (a_, b_), key in zip(zip(lo_, hi_), keys):
            ol, oh = o.find(*key)
            assert int(b_) - int(a_) == oh - ol and (oh == ol or int(a_) == ol), (tag, key)
        g.close()
    cases += 1
    if ONLY >= 0 and cases > ONLY:
        if os.environ.get("FUZZ_GUARD") == "1":
            ctx.set_option("pool_guard_check", 0)
        break
ctx.set_option("small_build_rows", 8192)
ctx.set_option("hash_partitioned", 1)
print("FUZZ_OK cases", cases, "seed", seed, "| tables coded with the delimiter split:", splits, "| fixed-width 8-byte key tables (lean chain steps):", leans, flush=True)

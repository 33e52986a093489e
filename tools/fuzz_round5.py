#!/usr/bin/env python3
"""Differential fuzz of round 5's paths against the oracle (csvplus.go:545-583, 707-756, 794-807 restated in oracle/):
  chain   chains of 2-3 Joins whose later keys are columns of EARLIER BUILD TABLES (cph_chain_step.source): answered from pre-joined
          tables (run_prejoined) or by the fused gather kernel (ctx option chain_prejoin picked at random), row ids / sorted
          positions / build columns laid out in sorted order; unique and duplicate build sides, missing keys on every side
  hostb   IndexOn over ONE key column in HOST memory (>= 2^20 rows) through host-formed codes: chunks, the window sort's first level
          per chunk, duplicates now and then; same index as the oracle's
  split   IndexOn over one variable-length column of >= 2^22 rows: the split codec from the sample alone, with 0-3 rows the sample
          does not visit and cannot code (unseen prefix / suffix byte / lengths)
usage: tools/fuzz_round5.py [seconds] [seed]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from csvplus_amd import Context, DeviceIndex, StrCol, datagen as dg, join_chain
from oracle import orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed)
ctx = Context(0)
print("seed", seed, flush=True)
mk = StrCol.from_values


def oracle_chain(oix, steps, probe_base):
    j = oix[0].join(steps[0][0], probe_base=probe_base)
    stream, rows = j["probe_idx"], [j["build_row"]]
    for k in range(1, len(oix)):
        cols, src = steps[k]
        sel = (stream - probe_base).astype(np.uint32) if src == 0 else rows[src - 1]
        jk = oix[k].join(cols, row_sel=sel)
        pick = jk["probe_idx"].astype(np.int64)
        stream = stream[pick]
        rows = [r[pick] for r in rows] + [jk["build_row"]]
    return stream, rows


def ids(prefix, space, n, unique):
    v = rng.permutation(space)[:n] if unique else rng.integers(0, space, n)
    fixed = rng.random() < 0.4
    return [(b"%s%07d" if fixed else b"%s%d") % (prefix, int(x)) for x in v]


def chain_case():
    nsteps = int(rng.integers(2, 5)) if rng.random() < 0.9 else int(rng.integers(5, 7))   # (5-6 Joins: the general chain in one call)
    na = int(rng.integers(50, 30_000))
    m = int(rng.integers(2 * na + 1, 12 * na + 2)) if rng.random() < 0.8 else int(rng.integers(1, 2 * na))   # (short streams: no pre-join)
    uniq = [rng.random() < 0.8 for _ in range(nsteps)]
    sizes = [na] + [int(rng.integers(5, 3000)) for _ in range(nsteps - 1)]
    spaces = [int(s * rng.uniform(1.0, 1.4)) + 1 for s in sizes]
    pre = [b"a", b"b", b"c", b"d", b"e", b"f"]
    tables = [ids(pre[k], spaces[k], sizes[k], uniq[k]) for k in range(nsteps)]
    # every table carries one key column per later table
    fk = [[None] * nsteps for _ in range(nsteps)]
    for t in range(nsteps):
        for k in range(t + 1, nsteps):
            fk[t][k] = ids(pre[k], int(spaces[k] * 1.1) + 1, sizes[t], False)
    stream_keys = [ids(pre[k], int(spaces[k] * 1.1) + 1, m, False) for k in range(nsteps)]
    steps = [([mk(stream_keys[0])], 0)]
    for k in range(1, nsteps):
        src = int(rng.integers(0, k + 1))            # 0: the stream; t + 1: table t
        steps.append(([mk(stream_keys[k])], 0) if src == 0 else ([mk(fk[src - 1][k])], src))
    builds = [[mk(t)] for t in tables]
    pb = int(rng.integers(0, 1000))
    ctx.set_option("chain_prejoin", int(rng.random() < 0.6))
    gix = [DeviceIndex(ctx, b) for b in builds]
    oix = [orc.OracleIndex(b) for b in builds]
    es, erows = oracle_chain(oix, steps, pb)
    gsteps = [(g, c, s) for g, (c, s) in zip(gix, steps)]
    ch = join_chain(ctx, gsteps, probe_base=pb)
    assert ch.nrows == len(es) and np.array_equal(ch.stream_row, es), "row ids: stream rows"
    for k in range(nsteps):
        assert np.array_equal(ch.build_row(k), erows[k]), ("row ids", k)
    ch.release()
    variants = [gsteps]
    if rng.random() < 0.5:   # build-side columns laid out in their source index's sorted order
        srt = []
        for g, c, s in gsteps:
            if s > 0:
                perm = gix[s - 1].perm()
                vals = c[0].values()
                srt.append((g, [mk([vals[int(i)] for i in perm])], -s))
            else:
                srt.append((g, c, s))
        variants.append(srt)
    for v in variants:
        chp = join_chain(ctx, v, probe_base=pb, positions=True)
        assert chp.nrows == len(es) and np.array_equal(chp.stream_row, es), "positions: stream rows"
        for k in range(nsteps):
            pos = chp.build_row(k)
            if all(uniq):
                assert np.array_equal(gix[k].perm()[pos], erows[k]), ("positions", k)
            else:   # duplicates: the reference's order inside a group of equal keys is open; ours is the stable one = the oracle's
                assert np.array_equal(gix[k].perm()[pos], erows[k]), ("positions, duplicates", k)
        chp.release()
    for g in gix:
        g.close()
    ctx.set_option("chain_prejoin", 1)
    return "chain%d%s" % (nsteps, "" if all(uniq) else "d")


def fixed8(v):
    raw = np.char.zfill(v.astype("U8"), 8).astype("S8")
    return StrCol.from_arrays(np.frombuffer(raw.tobytes(), np.uint8).copy(), np.arange(len(v) + 1, dtype=np.uint32) * 8, fixed_width=8)


def hostb_case():
    n = (1 << 20) + int(rng.integers(0, 3_500_000))
    kind = int(rng.integers(0, 4))
    if kind == 0:
        col, unique = fixed8(rng.permutation(n)), True                                      # full code space: identity, direct sort
    elif kind == 1:
        col, unique = fixed8(rng.permutation(int(n * rng.uniform(1.0, 1.9)))[:n]), True     # dense, compacted windows
    elif kind == 2:
        v = rng.permutation(int(n * rng.uniform(1.0, 1.9)))[:n]
        v[int(rng.integers(1, n))] = v[int(rng.integers(0, n))]                              # one duplicate somewhere
        col, unique = fixed8(v), True
    else:
        col, unique = dg.column(dg.UNIFORM, n, int(rng.integers(1000, n)), encoding=dg.ITOA, seed=int(rng.integers(1, 1 << 30))), False
    ctx.set_option("host_threads", int(rng.choice([0, 1, 3, 16])))
    g = DeviceIndex(ctx, [col], unique=unique)
    o = orc.OracleIndex([col])
    assert np.array_equal(g.perm(), o.perm), "perm"
    assert g.first_dup == o.first_dup(), ("first_dup", g.first_dup, o.first_dup())
    g.close()
    ctx.set_option("host_threads", 0)
    return "hostb%d" % kind


def split_case():
    n = (1 << 22) + int(rng.integers(0, 300_000))
    keys = dg.varkeys(n, int(rng.choice([1000, 100_000, 3_000_000])), seed=int(rng.integers(1, 1 << 30)))
    step = n >> 18
    rare = [b"Zeppelin/Qq#12345", b"Smith/Amelia#12x45", b"Smith/Amelia#123456789", b"Smith/Amelia#", b"Smith/Amelia", b"#", b"",
            b"Smith/Amelia-and-a-very-long-middle-name#123", b"Smith/Amelia#\x00", b"\xff/x#1"]
    repl = {}
    for _ in range(int(rng.integers(0, 4))):
        r = int(rng.integers(1, n))
        if r % step:
            repl[r] = rare[int(rng.integers(0, len(rare)))]
    if repl:
        data, off = np.asarray(keys.data), np.asarray(keys.offsets).astype(np.int64)
        parts, lens, prev = [], (off[1:] - off[:-1]).copy(), 0
        for r in sorted(repl):
            parts += [data[off[prev]:off[r]], np.frombuffer(repl[r], np.uint8)]
            lens[r] = len(repl[r])
            prev = r + 1
        parts.append(data[off[prev]:off[-1]])
        noff = np.zeros(len(off), np.uint32)
        np.cumsum(lens, out=noff[1:])
        keys = StrCol.from_arrays(np.concatenate(parts), noff)
    g = DeviceIndex(ctx, [keys.to_device("cuda:0")])
    o = orc.OracleIndex([keys])
    assert np.array_equal(g.perm(), o.perm), "perm"
    assert g.first_dup == o.first_dup(), "first_dup"
    g.close()
    return "split%d" % len(repl)


t_end = time.time() + budget
kinds, cases = {}, 0
while time.time() < t_end:
    x = rng.random()
    fn = chain_case if x < 0.80 else hostb_case if x < 0.93 else split_case
    state = rng.bit_generator.state
    try:
        k = fn()
    except AssertionError as ex:
        print("MISMATCH seed", seed, "case", cases, fn.__name__, ex.args, flush=True)
        sys.exit(1)
    kinds[k] = kinds.get(k, 0) + 1
    cases += 1
print("FUZZ_R5_OK seed", seed, "cases", cases, dict(sorted(kinds.items())), flush=True)

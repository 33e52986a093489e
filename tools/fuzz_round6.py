#!/usr/bin/env python3
"""Differential fuzz of round 6's paths:
  counted   IndexOn over >= 2^21 rows with duplicates (counted_sort.hip): fixed-width / unpadded decimal ids and split-codec keys;
            uniform keys, dense clusters in a sparse code space, heavy hitters beyond a window (the overflow path), groups of 33..5000
            equal keys (the bitonic path), one or two partition levels — perm and first duplicate against numpy's STABLE argsort of the
            keys (fixed-width ids) / the classic radix passes (ctx option counted_sort = 0), now and then against the oracle
  csv       cph_csv_parse through the byte-parallel fast path (csv_ingest.hip: k_csv_fast) against the record-parallel kernels
            (ctx option csv_fast = 0) and, for small texts, the oracle: unquoted texts with \n / \r\n / mixed line ends, stray \r,
            empty fields, ragged records, missing final newline, header records skipped, 1-4 columns, duplicates among them, long
            records near the window limit, and texts the fast path must hand over (a quote, a blank line, a comment line)
  chaind    people.Join(IndexOn(orders.key) with duplicates, k).Join(UniqueIndexOn(products), key of the ORDERS row): the general
            chain's pre-joined step (chain.hip: prejoin_general_step) on / off, row ids / positions, against the oracle
  tocsv     cph_csv_write_rows through the one-pass writer (materialize.hip: k_csv_slots + k_csv_onepass — slot tables, LDS-DMA gathers,
            decoupled look-back over 2 .. all workgroups) against the two-pass writer (ctx option csv_onepass = 0) and, for small outputs,
            the oracle's Writer: 1-6 output columns out of stream columns and 1-3 tables (adjacent columns of a table form one slot
            group), 32- / 64-bit row ids with a base, values with every quoting trigger, fragments around the stride limits (15 / 31 /
            63 / 127 bytes), a few records larger than a tile's LDS stage, tables larger than the output, outputs the buffer estimate
            cannot hold (the overflow path)
usage: tools/fuzz_round6.py [seconds] [seed] [only: counted | csv | chaind | tocsv | varkey]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from csvplus_amd import Context, DeviceIndex, StrCol, _native as N, ingest, join_chain
from oracle import orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
only = sys.argv[3] if len(sys.argv) > 3 else None
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed)
ctx = Context(0)
print("seed", seed, flush=True)


def fixed8(ids):
    raw = np.char.zfill(ids.astype("U8"), 8).astype("S8")
    return StrCol.from_arrays(np.frombuffer(raw.tobytes(), np.uint8).copy(), np.arange(ids.size + 1, dtype=np.uint32) * 8, fixed_width=8)


def counted_case():
    n = int(rng.integers(1 << 21, 3 << 21))
    shape = int(rng.integers(0, 5))
    if shape == 0:      # uniform over a space 0.1 .. 4 rows wide
        ids = rng.integers(0, max(1000, int(n * rng.uniform(0.1, 4.0))), n)
    elif shape == 1:    # a dense block + a thin spread far away
        ids = np.concatenate([rng.integers(0, int(n * rng.uniform(0.2, 1.0)), n - 30_000), rng.integers(n, int(n * rng.uniform(4, 12)), 30_000)])
        rng.shuffle(ids)
    elif shape == 2:    # heavy hitters: a few keys with thousands of duplicates (one of them maybe beyond a window)
        ids = rng.integers(0, n // 3, n)
        for _ in range(int(rng.integers(1, 6))):
            ids[rng.choice(n, int(rng.integers(33, 5000 if rng.random() < 0.7 else 40_000)), replace=False)] = int(rng.integers(0, n // 3))
    elif shape == 3:    # few distinct keys: hundreds of rows per code (the plan says no: classic passes)
        ids = rng.integers(0, int(rng.integers(2, 5000)), n)
    else:               # exactly the same key in long runs of the input (stability matters)
        ids = np.repeat(rng.integers(0, n // 4, n // 8 + 1), 8)[:n]
    ids = ids.astype(np.int64)
    col = fixed8(ids)
    dev = rng.random() < 0.7
    g = DeviceIndex(ctx, [col.to_device("cuda:0") if dev else col], unique=bool(rng.random() < 0.2))
    want = np.argsort(ids, kind="stable").astype(np.uint32)
    s = ids[want]
    dup = np.flatnonzero(s[1:] == s[:-1])
    fd = int(dup[0]) + 1 if dup.size else None
    assert np.array_equal(g.perm(), want), ("perm", shape, n)
    assert g.first_dup == fd, ("first_dup", shape, g.first_dup, fd)
    g.close()
    return "counted%d" % shape


def varkey_case():
    """split-codec keys (surname/name#number): counted windows against the classic passes, sometimes the oracle"""
    from csvplus_amd import datagen as dg
    n = int(rng.integers(1 << 22, (1 << 22) + 400_000))
    col = dg.varkeys(n, distinct_suffix=int(rng.integers(50, 20_000)), seed=int(rng.integers(0, 1 << 30)))
    d = col.to_device("cuda:0")
    g = DeviceIndex(ctx, [d])
    ctx.set_option("counted_sort", 0)
    r = DeviceIndex(ctx, [d])
    ctx.set_option("counted_sort", 1)
    assert np.array_equal(g.perm(), r.perm()) and g.first_dup == r.first_dup, "varkeys"
    g.close(); r.close()
    return "varkeys"


ALPH = np.frombuffer(b"abcxyz 0123456789#;\t", dtype=np.uint8)


def gen_text(nrec, nf, crlf_prob, trailing_nl, ragged, long_rec):
    out = bytearray()
    for r in range(nrec):
        k = int(rng.integers(1, nf + 1)) if ragged else nf
        fields = []
        for f in range(k):
            ln = int(rng.integers(0, 9)) if rng.random() < 0.92 else int(rng.integers(20, 90))
            v = ALPH[rng.integers(0, len(ALPH), ln)].tobytes()
            if rng.random() < 0.02:
                v += b"\r"
            fields.append(v)
        if long_rec and r == nrec // 2:
            fields[0] = b"L" * long_rec
        if k == 1 and fields[0] in (b"", b"\r"):
            fields[0] = b"q"
        out += b",".join(fields)
        if r != nrec - 1 or trailing_nl:
            out += b"\r\n" if rng.random() < crlf_prob else b"\n"
    return bytes(out)


def parse(text, cols, **kw):
    t = ingest.csv_parse(ctx, text, cols, out_mem=N.CPH_MEM_HOST, **kw)
    return [t.columns[c].values() for c in range(len(cols))], t.nrecords, t.error_kind, (t.error_record if t.error_kind else 0)


def csv_case():
    nf = int(rng.integers(1, 10))
    big = rng.random() < 0.5
    nrec = int(rng.integers(2000, 40_000)) if big else int(rng.integers(1, 120))
    ragged = rng.random() < 0.3
    long_rec = int(rng.choice([0, 0, 3000, 4200, 9000]))
    text = gen_text(nrec, nf, [0.0, 1.0, 0.4][int(rng.integers(0, 3))], bool(rng.random() < 0.8), ragged, long_rec)
    spoil = int(rng.integers(0, 8))
    if spoil == 0:
        p = text.find(b"\n", len(text) // 2)
        text = text[:p + 1] + b"\n" + text[p + 1:]                    # a blank line
    elif spoil == 1:
        p = text.find(b",", len(text) // 3)
        if p > 0:
            text = text[:p + 1] + b'"q"' + text[p + 1:]               # a quoted field (or a bare quote: the error must agree too)
    elif spoil == 2 and rng.random() < 0.5:
        text += b"\r"
    ncols = int(rng.integers(1, min(nf, 8) + 1))
    cols = sorted(rng.choice(nf, size=ncols, replace=False).tolist())
    if ncols < 8 and rng.random() < 0.2:
        cols = cols + [cols[0]]
    kw = dict(fields_per_record=-1 if ragged else int(rng.choice([0, -1, nf])), skip_records=int(rng.integers(0, 4)))
    if spoil == 3:
        kw["comment"] = b"#"
    got = parse(text, cols, **kw)
    ctx.set_option("csv_fast", 0)
    try:
        ref = parse(text, cols, **kw)
    finally:
        ctx.set_option("csv_fast", 1)
    assert got == ref, ("csv fast != classic", nf, cols, kw, len(text), spoil)
    if not big:
        ocols, oek, oer = orc.csv_parse(text, cols, **kw)
        assert got[1] == ocols[0].nrows and got[2] == oek and got[0] == [ocols[c].values() for c in range(len(cols))], ("csv != oracle", text[:100])
    return "csv%d" % spoil


def chaind_case():
    npeople, nord, nprod = int(rng.integers(200, 6000)), int(rng.integers(2000, 90_000)), int(rng.integers(5, 900))
    fixed = rng.random() < 0.5
    fmt = (lambda p, v: b"%s%06d" % (p, v)) if fixed else (lambda p, v: b"%s%d" % (p, v))
    space = int(npeople * rng.uniform(1.0, 1.3)) + 1
    people = [fmt(b"", int(i)) for i in rng.permutation(space)[:npeople]]
    o_cust = [fmt(b"", int(i)) for i in rng.integers(0, space, nord)]
    o_prod = [fmt(b"p", int(i)) for i in rng.integers(0, int(nprod * rng.uniform(1.0, 1.3)) + 1, nord)]
    prods = [fmt(b"p", int(i)) for i in rng.permutation(nprod)]
    mk = StrCol.from_values
    b0, b1, s0, s1 = mk(o_cust), mk(prods), mk(people), mk(o_prod)
    g0, g1 = DeviceIndex(ctx, [b0]), DeviceIndex(ctx, [b1], unique=True)
    o0, o1 = orc.OracleIndex([b0]), orc.OracleIndex([b1])
    j = o0.join([s0], probe_base=3)
    j2 = o1.join([s1], row_sel=j["build_row"])
    pick = j2["probe_idx"].astype(np.int64)
    es, e0, e1 = j["probe_idx"][pick], j["build_row"][pick], j2["build_row"]
    ctx.set_option("chain_prejoin", int(rng.random() < 0.7))
    try:
        for pos in (False, True):
            ch = join_chain(ctx, [(g0, [s0], 0), (g1, [s1], 1)], probe_base=3, positions=pos)
            assert ch.nrows == len(es) and np.array_equal(ch.stream_row, es), ("chaind rows", pos)
            r0, r1 = ch.build_row(0), ch.build_row(1)
            if pos and ch.nrows:
                r0, r1 = g0.perm()[r0], g1.perm()[r1]
            assert np.array_equal(r0, e0) and np.array_equal(r1, e1), ("chaind tuples", pos)
            ch.release()
    finally:
        ctx.set_option("chain_prejoin", 1)
    g0.close(); g1.close()
    return "chaind"


def tocsv_case():
    from csvplus_amd.materialize import csv_write
    n = int(rng.choice([int(rng.integers(1, 600)), int(rng.integers(600, 20_000)), int(rng.integers(20_000, 400_000))]))
    nasty = rng.random() < 0.5
    alpha = np.frombuffer(b'ab ,"\n\rz#\t0123456789' if nasty else b"abcdefghij0123456789", dtype=np.uint8)

    def values(count, lo, hi):
        lens = rng.integers(lo, hi + 1, count)
        flat = alpha[rng.integers(0, len(alpha), int(lens.sum()))].tobytes()
        offs = np.concatenate([[0], np.cumsum(lens)])
        return [flat[offs[i]:offs[i + 1]] for i in range(count)]

    cols, ids, expect = [], [], []
    ntab = int(rng.integers(0, 4))
    layout = ["s"] * int(rng.integers(0 if ntab else 1, 3))
    for t in range(ntab):
        layout += [t] * int(rng.integers(1, 3))
    if rng.random() < 0.4:
        rng.shuffle(layout)      # a table's columns may end up apart: separate slot groups
    layout = layout[:6]
    tabs = {}
    for what in layout:
        if what == "s":
            hi = int(rng.choice([3, 9, 20]))
            v = values(n, 0, hi)
            if rng.random() < 0.15 and n > 10:
                for r in rng.choice(n, 2, replace=False):
                    v[int(r)] = b"x" * int(rng.integers(15_000, 30_000)) + b'"y'     # beyond the stage
            if rng.random() < 0.1:
                v = [b'""' + x for x in v]                                            # quotes everywhere: the buffer estimate falls short
            c = StrCol.from_values(v)
            cols.append(c); ids.append(None); expect.append(c)
        else:
            if what not in tabs:
                nt = int(rng.choice([int(rng.integers(1, 50)), int(rng.integers(50, 3000)), int(n * rng.uniform(0.5, 3.0)) + 1]))
                wide = rng.random() < 0.5
                tid = rng.integers(0, nt, n)
                tabs[what] = (nt, (tid + 1000).astype(np.uint64) if wide else tid.astype(np.uint32), 1000 if wide else 0, tid)
            nt, tid_arr, base, tid = tabs[what]
            hi = int(rng.choice([4, 7, 14, 15, 16, 30, 31, 32, 62, 64, 126, 130]))
            v = values(nt, 0 if rng.random() < 0.8 else hi, hi)
            cols.append(StrCol.from_values(v)); ids.append((tid_arr, base)); expect.append(StrCol.from_values([v[int(i)] for i in tid]) if n <= 20_000 else None)
    names = ["c%d" % i for i in range(len(cols))] if rng.random() < 0.8 else None
    row_ids = [None if i is None else i[0] - i[1] for i in ids]   # (the wrapper takes ids relative to the column; 64-bit ids stay 64-bit)
    texts = []
    ran = set()
    for mode in (0, int(rng.choice([1, 2, 3, 7, 1 << 20]))):
        ctx.set_option("csv_onepass", mode)
        ctx.profile(True); ctx.profile_read(reset=True)
        try:
            texts.append(csv_write(ctx, cols, names, row_ids=row_ids, nrows=n))
            ran = set(ctx.profile_read(reset=True))
        finally:
            ctx.profile(False)
            ctx.set_option("csv_onepass", 1)
    assert texts[0] == texts[1], ("tocsv one pass != two passes", n, layout)
    if n <= 20_000:
        assert texts[0] == orc.csv_write(expect, names), ("tocsv != oracle", n, layout)
    return "tocsv_" + ("two_pass" if "k_csv_onepass" not in ran else "overflow" if "k_csv_copy" in ran else "slots" if "k_csv_slots" in ran else "plain")


t_end = time.time() + budget
kinds, cases = {}, 0
while time.time() < t_end:
    x = rng.random()
    fn = csv_case if x < 0.40 else tocsv_case if x < 0.62 else chaind_case if x < 0.87 else counted_case if x < 0.97 else varkey_case
    if only:
        fn = {"counted": counted_case, "csv": csv_case, "chaind": chaind_case, "tocsv": tocsv_case, "varkey": varkey_case}[only]
    try:
        k = fn()
    except AssertionError as ex:
        print("MISMATCH seed", seed, "case", cases, fn.__name__, ex.args, flush=True)
        sys.exit(1)
    kinds[k] = kinds.get(k, 0) + 1
    cases += 1
print("FUZZ_R6_OK seed", seed, "cases", cases, dict(sorted(kinds.items())), flush=True)

"""The step after the path: cph_gather_rows (column-wise mergeRows, csvplus.go:571-583) and
cph_csv_write (ToCsv, csvplus.go:379-406) against the oracle's C restatement of Go's csv.Writer."""
import csv
import io

import numpy as np
import pytest

from csvplus_amd import DeviceIndex, StrCol, datagen as dg, join_chain
from oracle import orc

NASTY = [b"", b"plain", b"a,b", b'q"uote', b'""', b"line\nbreak", b"cr\rhere", b" lead", b"trail ", b"\ttab", b"\\.", b"\\..",
         "\u00a0nbsp".encode(), "\u2003em".encode(), "\u3000cjk".encode(), "é".encode(), b"\xc2", b"\xe2\x80", b"\xff\xfe",
         b"x" * 100, b'"', b",", b"\n", b"1234567", b"12345678", b"123456789"]


def table(rng, n, ncols):
    return [[NASTY[i] for i in rng.integers(0, len(NASTY), n)] for _ in range(ncols)]


# ---- CPU: pin the oracle's Writer restatement --------------------------------------------------------
def test_oracle_csv_known_answers():
    """Hand-checked records: what Go's csv.Writer produces for these fields (quoting rules of
    encoding/csv: `\\.`, separator, quote, CR/LF, leading space)."""
    cols = [StrCol.from_values(["a", "1,2", "\\.", "plain", ""]),
            StrCol.from_values(["b c", 'q"uote', "line\nbreak", "\ttab", " "])]
    out = orc.csv_write(cols, ["h1", "h,2"])
    assert out == b'h1,"h,2"\na,b c\n"1,2","q""uote"\n"\\.","line\nbreak"\nplain,"\ttab"\n," "\n'
    assert orc.csv_write(cols) == out.split(b"\n", 1)[1]


def test_oracle_csv_agrees_with_python_csv_on_common_subset():
    """Python's csv (QUOTE_MINIMAL, \\n terminator) and Go's Writer agree when no field starts with a
    space character and none equals `\\.`."""
    rng = np.random.default_rng(0)
    safe = [v for v in NASTY if v[:1] not in (b" ", b"\t", b"\n", b"\r", b"\xc2", b"\xe2", b"\xe3") and v != b"\\."
            and b"\r" not in v and all(c < 0x80 for c in v)]
    rows = [[safe[i].decode() for i in rng.integers(0, len(safe), 3)] for _ in range(300)]
    cols = [StrCol.from_values([r[c] for r in rows]) for c in range(3)]
    s = io.StringIO()
    w = csv.writer(s, lineterminator="\n")
    w.writerow(["x", "y", "z"])
    w.writerows(rows)
    # Python quotes an empty single-field record; with 3 fields the outputs coincide
    assert orc.csv_write(cols, ["x", "y", "z"]).decode() == s.getvalue()


# ---- GPU ---------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("n", [0, 1, 255, 256, 257, 5000])
def test_gather_rows_matches_numpy_take(ctx, n):
    from csvplus_amd.materialize import gather_rows

    rng = np.random.default_rng(n)
    vals = table(rng, 777, 1)[0]
    col = StrCol.from_values(vals)
    ids = rng.integers(0, len(vals), n)
    for dtype, base in ((np.uint32, 0), (np.uint64, 1000)):
        g = gather_rows(ctx, col, (ids + base).astype(dtype), id_base=base)
        assert g.values() == [vals[i] for i in ids]
    assert gather_rows(ctx, col).values() == vals     # identity copy
    big = StrCol.from_values([b"y" * 500] * 300)       # tiles beyond the LDS stage: direct byte path
    assert gather_rows(ctx, big, np.arange(299, -1, -1, dtype=np.uint32)).values() == [b"y" * 500] * 300
    fixed = dg.customers(1000)["id"]
    assert fixed.fixed_width == 8
    assert gather_rows(ctx, fixed, ids[:100].astype(np.uint32) % 1000).values() == [fixed.value(int(i) % 1000) for i in ids[:100]]


@pytest.mark.gpu
@pytest.mark.parametrize("n,ncols", [(0, 2), (1, 1), (300, 3), (4000, 4)])
def test_csv_write_matches_oracle(ctx, n, ncols):
    from csvplus_amd.materialize import csv_write

    rng = np.random.default_rng(n + ncols)
    cols = [StrCol.from_values(c) for c in table(rng, n, ncols)]
    header = ["id", "na,me", " x", 'q"'][:ncols]
    assert csv_write(ctx, cols, header) == orc.csv_write(cols, header)
    assert csv_write(ctx, cols) == orc.csv_write(cols)
    dcols = [c.to_device() for c in cols]
    assert csv_write(ctx, dcols, header) == orc.csv_write(cols, header)


@pytest.mark.gpu
def test_joined_table_to_csv_end_to_end(ctx):
    """orders JOIN customers JOIN products -> gather -> ToCsv, all on the device, against the oracle join
    materialised on the host with mergeRows semantics (stream wins on a collision: none here)."""
    from csvplus_amd.materialize import csv_write, gather_rows

    nc, npd, m = 3000, 40, 20_000
    cust, prod = dg.customers(nc, encoding=dg.ITOA), dg.products(npd)
    ords = dg.orders(m, nc + 500, npd, cust_encoding=dg.ITOA)      # some customers are missing
    gi = [DeviceIndex(ctx, [cust["id"]], unique=True), DeviceIndex(ctx, [prod["prod_id"]], unique=True)]
    ch = join_chain(ctx, [(gi[0], [ords["cust_id"]]), (gi[1], [ords["prod_id"]])])
    s, a, b = ch.stream_row, ch.build_row(0), ch.build_row(1)
    assert 0 < ch.nrows < m
    out_cols = [gather_rows(ctx, ords["cust_id"], s), gather_rows(ctx, ords["qty"], s),
                gather_rows(ctx, cust["name"], a), gather_rows(ctx, cust["surname"], a),
                gather_rows(ctx, prod["product"], b), gather_rows(ctx, prod["price"], b)]
    header = ["cust_id", "qty", "name", "surname", "product", "price"]
    got = csv_write(ctx, out_cols, header)
    # oracle: nested joins + host-side materialisation
    oi = [orc.OracleIndex([cust["id"]]), orc.OracleIndex([prod["prod_id"]])]
    j1 = oi[0].join([ords["cust_id"]])
    j2 = oi[1].join([ords["prod_id"]], row_sel=j1["probe_idx"].astype(np.uint32))
    pick = j2["probe_idx"].astype(np.int64)
    es, ea, eb = j1["probe_idx"][pick], j1["build_row"][pick], j2["build_row"]
    lines = [",".join(header)]
    for r, x, y in zip(es, ea, eb):
        lines.append(",".join([ords["cust_id"].value(int(r)).decode(), ords["qty"].value(int(r)).decode(),
                               cust["name"].value(int(x)).decode(), cust["surname"].value(int(x)).decode(),
                               prod["product"].value(int(y)).decode(), prod["price"].value(int(y)).decode()]))
    assert got.decode() == "\n".join(lines) + "\n"


@pytest.mark.gpu
def test_csv_write_rows_fused_equals_gather_then_write(ctx):
    """cph_csv_write_rows (mergeRows folded into ToCsv) == gather every column, then cph_csv_write == oracle,
    with host columns + host row ids, uint32 and uint64 ids, an id base, and nasty values."""
    from csvplus_amd.materialize import csv_write
    rng = np.random.default_rng(31)
    a = StrCol.from_values([b"x", b'q"q', b"", b" lead", b"a,b", b"line\nbreak", b"\\.", b"plain", b"\xc2\xa0nbsp"])
    b = StrCol.from_values([b"%d" % i for i in range(50)])
    n = 4000
    ia = rng.integers(0, a.nrows, n).astype(np.uint32)
    ib = (rng.integers(0, b.nrows, n) + 1000).astype(np.uint64)
    s = StrCol.from_values([b"row%d" % i for i in range(n)])
    got = csv_write(ctx, [s, a, b], ["s", "a", "b"], row_ids=[None, ia, ib - 1000])
    want = orc.csv_write([s, StrCol.from_values([a.value(int(i)) for i in ia]),
                          StrCol.from_values([b.value(int(i) - 1000) for i in ib])], ["s", "a", "b"])
    assert got == want
    # no rows at all: the header only
    assert csv_write(ctx, [s.head(0), a, b], ["s", "a", "b"], row_ids=[None, ia[:0], ib[:0]], nrows=0) == b"s,a,b\n"


@pytest.mark.gpu
def test_csv_write_many_tiles_and_oversized_records(ctx):
    """Several hundred 256-record tiles, quoted and empty values, gathered columns through 32- and 64-bit row ids, and records
    larger than a tile's LDS stage (they are written to global memory directly)."""
    from csvplus_amd.materialize import csv_write
    rng = np.random.default_rng(97)
    n = 100_000
    alphabet = np.frombuffer(b'ab ,"\n\rz#\t0123456789', dtype=np.uint8)
    stream = [alphabet[rng.integers(0, len(alphabet), int(rng.integers(0, 12)))].tobytes() for _ in range(n)]
    for r in (5, 40_000, 99_999):
        stream[r] = b"x" * 20_000 + b'"' + b"y" * 3000          # larger than the 16 KiB stage
    small = [alphabet[rng.integers(0, len(alphabet), int(rng.integers(0, 20)))].tobytes() for _ in range(700)]
    nums = [b"%d" % i for i in range(5000)]
    s, a, b = StrCol.from_values(stream), StrCol.from_values(small), StrCol.from_values(nums)
    ia = rng.integers(0, a.nrows, n).astype(np.uint32)
    ib = rng.integers(0, b.nrows, n).astype(np.uint64)
    want = orc.csv_write([s, StrCol.from_values([small[int(i)] for i in ia]), StrCol.from_values([nums[int(i)] for i in ib]), s],
                         ["s", "a", "b", "s2"])
    assert csv_write(ctx, [s, a, b, s], ["s", "a", "b", "s2"], row_ids=[None, ia, ib, None]) == want


# ---- (round 6) the one-pass writer: slot tables + decoupled look-back (materialize.hip: k_csv_onepass) -----------------------------
def _write_with(ctx, mode, *args, **kw):
    """csv_write under ctx option csv_onepass = mode; returns (text, names of the kernels that ran)."""
    from csvplus_amd.materialize import csv_write
    ctx.set_option("csv_onepass", mode)
    ctx.profile(True); ctx.profile_read(reset=True)
    try:
        text = csv_write(ctx, *args, **kw)
        return text, set(ctx.profile_read(reset=True))
    finally:
        ctx.profile(False)
        ctx.set_option("csv_onepass", 1)


def _rand_values(rng, count, lo, hi, alphabet=b'ab ,"\n\rz#\t0123456789'):
    al = np.frombuffer(alphabet, dtype=np.uint8)
    return [al[rng.integers(0, len(al), int(rng.integers(lo, hi + 1)))].tobytes() for _ in range(count)]


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 255, 256, 257, 511, 512, 513, 1024, 5000, 70_000])
@pytest.mark.parametrize("grid", [2, 3, 1 << 20])
def test_onepass_writer_equals_oracle_and_two_pass(ctx, n, grid):
    """Stream columns (every quoting rule), a two-column group and a one-column group gathered from small tables (slot strides 32
    and 16), 32- and 64-bit row ids with a base — through 2, 3 and all resident workgroups (several tiles per workgroup, look-back
    across them), against the oracle and the two-pass writer."""
    rng = np.random.default_rng(n * 7 + grid % 5)
    s1 = StrCol.from_values([NASTY[i] if len(NASTY[i]) < 50 else b"x" for i in rng.integers(0, len(NASTY), n)])
    s2 = StrCol.from_values([b"%d" % i for i in rng.integers(0, 1000, n)])
    ta, tb = _rand_values(rng, 300, 0, 9), _rand_values(rng, 300, 0, 4)   # one table, two columns: fragments up to 9 + 1 + 4 (+ quotes)
    tc = [b"%d" % (i * 7) for i in range(40)]
    A, B, Cc = StrCol.from_values(ta), StrCol.from_values(tb), StrCol.from_values(tc)
    ia = rng.integers(0, 300, n).astype(np.uint32)
    ic = (rng.integers(0, 40, n) + 500).astype(np.uint64)
    cols, hdr = [s1, A, B, s2, Cc], ["s1", "a", "b", "s2", "c"]
    want = orc.csv_write([s1, StrCol.from_values([ta[i] for i in ia]), StrCol.from_values([tb[i] for i in ia]), s2,
                          StrCol.from_values([tc[int(i) - 500] for i in ic])], hdr)
    from csvplus_amd.materialize import csv_write
    ids = [None, ia, ia, None, ic - 500]
    got, ran = _write_with(ctx, grid, cols, hdr, row_ids=ids)
    assert got == want
    assert "k_csv_onepass" in ran and ("k_csv_slots" in ran) == (n >= 40) and "k_csv_copy" not in ran, sorted(ran)
    two, ran0 = _write_with(ctx, 0, cols, hdr, row_ids=ids)
    assert two == want and "k_csv_onepass" not in ran0


@pytest.mark.gpu
@pytest.mark.parametrize("maxlen,stride_ok", [(14, True), (15, True), (16, True), (31, True), (32, True), (63, True), (64, True), (127, False), (128, False)])
def test_onepass_slot_strides(ctx, maxlen, stride_ok):
    """Fragments of every length up to maxlen: strides 16 / 32 / 64 / 128 (the text beyond a slot's first 32 bytes is fetched, not
    held); records beyond ~72 bytes by the estimate (a 256-record tile would not fit its LDS stage) and fragments beyond 127 bytes
    leave the call to the two-pass writer."""
    rng = np.random.default_rng(maxlen)
    tab = [bytes(rng.integers(97, 123, L).astype(np.uint8)) for L in list(range(maxlen + 1)) * 3]
    T = StrCol.from_values(tab)
    n = 6000
    it = rng.integers(0, len(tab), n).astype(np.uint32)
    s = StrCol.from_values([b"r%d" % i for i in range(n)])
    want = orc.csv_write([s, StrCol.from_values([tab[i] for i in it])], ["s", "t"])
    got, ran = _write_with(ctx, 1, [s, T], ["s", "t"], row_ids=[None, it])
    assert got == want
    assert ("k_csv_onepass" in ran) == stride_ok and ("k_csv_copy" in ran) != stride_ok, sorted(ran)


@pytest.mark.gpu
def test_onepass_buffer_estimate_too_small_falls_back(ctx):
    """Every value of the stream column needs quotes and doubles its quote characters: the text is far beyond the column's bytes
    + 1/8; a tile notices, nothing of the one pass is used, the two-pass writer renders the same text."""
    n = 20_000
    s = StrCol.from_values([b'""""""""'] * n)
    want = orc.csv_write([s], ["q"])
    got, ran = _write_with(ctx, 1 << 20, [s], ["q"])
    assert got == want and len(got) == 2 + n * 19
    assert "k_csv_onepass" in ran and "k_csv_copy" in ran, sorted(ran)


@pytest.mark.gpu
def test_onepass_tiles_beyond_the_stage_and_a_table_larger_than_the_output(ctx):
    """A few records larger than a tile's LDS stage (written to global memory directly) among short ones; a column gathered from a
    table LARGER than the output (no slot table: its values are fetched and quoted per row); empty values; an empty table row."""
    rng = np.random.default_rng(5)
    n = 30_000
    stream = _rand_values(rng, n, 0, 12)
    for r in (5, 12_345, n - 1):
        stream[r] = b"x" * 30_000 + b'"' + b"y" * 9000
    big = _rand_values(rng, 50_000, 0, 20)
    s, Bg = StrCol.from_values(stream), StrCol.from_values(big)
    ib = np.sort(rng.choice(50_000, n, replace=False)).astype(np.uint64)
    want = orc.csv_write([s, StrCol.from_values([big[int(i)] for i in ib]), s], ["s", "b", "s2"])
    got, ran = _write_with(ctx, 1 << 20, [s, Bg, s], ["s", "b", "s2"], row_ids=[None, ib, None])
    assert got == want
    assert "k_csv_onepass" in ran and "k_csv_slots" not in ran, sorted(ran)
    for grid in (1, 2, 5):   # (option value 1 = "auto": a single workgroup is asked for with 2 tiles' worth of rows below)
        got, ran = _write_with(ctx, max(grid, 2), [s, Bg, s], ["s", "b", "s2"], row_ids=[None, ib, None])
        assert got == want


@pytest.mark.gpu
def test_onepass_device_columns_and_device_row_ids(ctx):
    """The pipeline's shape: device columns, device row ids out of a chained Join (positions), text left on the device."""
    from csvplus_amd import pipeline, ingest
    from csvplus_amd import _native as N
    nc, npd, m = 20_000, 300, 150_000
    cust, prod, ords = dg.customers(nc), dg.products(npd), dg.orders(m, nc, npd)
    def text_of(cols, names):
        return orc.csv_write(cols, names)
    tc = pipeline.read_table(ctx, text_of([cust["id"], cust["name"], cust["surname"]], ["id", "name", "surname"]))
    tp = pipeline.read_table(ctx, text_of([prod["prod_id"], prod["product"], prod["price"]], ["prod_id", "product", "price"]))
    to = pipeline.read_table(ctx, text_of([ords["cust_id"], ords["prod_id"], ords["qty"]], ["cust_id", "prod_id", "qty"]))
    out_cols = [("cust_id", to, "cust_id"), ("qty", to, "qty"), ("name", tc, "name"), ("surname", tc, "surname"),
                ("product", tp, "product"), ("price", tp, "price")]
    steps = [(tc, "id", "cust_id"), (tp, "prod_id", "prod_id")]
    texts = {}
    for mode in (0, 1):
        for positions in (True, False):
            ctx.set_option("csv_onepass", mode)
            try:
                texts[(mode, positions)] = bytes(pipeline.join_to_csv(ctx, to, steps, out_cols, positions=positions))
            finally:
                ctx.set_option("csv_onepass", 1)
    assert len(set(texts.values())) == 1
    # the oracle's joined table
    cid = {cust["id"].value(i): i for i in range(nc)}
    pid = {prod["prod_id"].value(i): i for i in range(npd)}
    lines = [b"cust_id,qty,name,surname,product,price"]
    for r in range(0, m, 997):
        x, y = cid[ords["cust_id"].value(r)], pid[ords["prod_id"].value(r)]
        lines.append(b",".join([ords["cust_id"].value(r), ords["qty"].value(r), cust["name"].value(x), cust["surname"].value(x),
                                prod["product"].value(y), prod["price"].value(y)]))
    got = texts[(1, True)].split(b"\n")
    assert got[0] == lines[0] and len(got) == m + 2
    assert [got[1 + r] for r in range(0, m, 997)] == lines[1:]
    for t in (tc, tp, to):
        t.release()


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_onepass_from_two_contexts_at_once():
    """Two ctxs (two streams) of one process write CSV through the one-pass kernel from two threads at the same time: its persistent
    grid waits for its own lower tiles, so the library lets one such grid at a time onto the device — neither call may hang, both
    texts are right."""
    import threading
    from csvplus_amd import Context
    from csvplus_amd.materialize import csv_write
    rng = np.random.default_rng(404)
    n = 300_000
    tab = [b"%d-%s" % (i, b"x" * int(rng.integers(0, 12))) for i in range(5000)]
    T = StrCol.from_values(tab)
    jobs = []
    for k in range(2):
        it = rng.integers(0, len(tab), n).astype(np.uint32)
        s = StrCol.from_values([b"r%d" % i for i in range(k, n + k)])
        jobs.append((s, it, orc.csv_write([s, StrCol.from_values([tab[i] for i in it])], ["s", "t"])))
    ctxs = [Context(0), Context(0)]
    got, errs = [None, None], []

    def work(k):
        try:
            for _ in range(6):
                got[k] = csv_write(ctxs[k], [jobs[k][0], T], ["s", "t"], row_ids=[None, jobs[k][1]])
                assert got[k] == jobs[k][2]
        except Exception as ex:   # noqa: BLE001
            errs.append(repr(ex))

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=240)
    assert not any(t.is_alive() for t in th), "a writer hangs"
    assert not errs, errs
    for c in ctxs:
        c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["1 plain", "2 plain", "3 plain", "1 plain + slots", "5 apart"])
def test_onepass_with_more_tiles_than_resident_workgroups(ctx, shape):
    """Enough rows that the persistent grid is as large as it gets (tiles > CUs x workgroups per CU): every workgroup of the grid must
    be resident, or tiles wait for tiles nobody runs.  (The runtime's occupancy figure was one too high for the 1-column kernel: a 5e7-row
    call stalled until the watchdog gave it to the two-pass writer; the grid is now sized from the kernel's own registers and LDS.)"""
    import time
    n = 700_000     # 2735 tiles of 256 records; 256 CUs x 4-8 workgroups = 1024-2048
    rng = np.random.default_rng(len(shape))
    num = np.char.mod("%d", rng.integers(0, 100000, n)).astype("S")
    def plain(k):
        vals = np.char.add(num, np.bytes_(b"abcdefgh"[k:k + 1])) if k else num
        return StrCol.from_values(vals.tolist())
    tab = [b"t%d" % i for i in range(3000)]
    T = StrCol.from_values(tab)
    it = rng.integers(0, len(tab), n).astype(np.uint32)
    G = StrCol.from_values([tab[i] for i in it])
    if shape.endswith("plain"):
        k = int(shape[0]); cols, ids, exp = [plain(j) for j in range(k)], None, None
    elif shape == "1 plain + slots":
        p0 = plain(0); cols, ids, exp = [p0, T], [None, it], [p0, G]
    else:
        p0, p1, p2 = plain(0), plain(1), plain(2); cols, ids, exp = [p0, T, p1, T, p2], [None, it, None, it, None], [p0, G, p1, G, p2]
    want = orc.csv_write(exp or cols)
    t0 = time.perf_counter()
    got, ran = _write_with(ctx, 1 << 20, cols, None, row_ids=ids)   # (forced: by default only shapes with a slot table take the one pass)
    dt = time.perf_counter() - t0
    assert got == want
    assert "k_csv_onepass" in ran and "k_csv_copy" not in ran and dt < 2.5, (sorted(ran), dt)
    got1, ran1 = _write_with(ctx, 1, cols, None, row_ids=ids)
    assert got1 == want and ("k_csv_onepass" in ran1) == ("slots" in shape or "apart" in shape), sorted(ran1)

"""CSV ingest: Reader.Iterate's parse loop (csvplus.go:1080-1146) as columns.

CPU: the oracle's restatement of Go's csv.Reader is pinned to known-answer cases (tests/golden) and cross-checked
against Python's csv module on the dialect subset where both agree.  GPU: cph_csv_parse == oracle, bit for bit,
including the kind and record index of the first error, on hand cases, seeded random text and mutated
(malformed) text."""
from __future__ import annotations

import csv
import io

import numpy as np
import pytest

from oracle import orc
from tests.golden.go_csv_reader_cases import CASES

WIDTH = 6   # fields compared per record (missing ones read as "")


def oracle_records(text, opts, width=WIDTH, skip=0):
    cols, ek, er = orc.csv_parse(text, list(range(width)), comma=opts.get("comma", b","), comment=opts.get("comment"),
                                 trim_leading_space=opts.get("trim", False), fields_per_record=opts.get("fpr", 0),
                                 skip_records=skip)
    n = cols[0].nrows
    return [[cols[c].value(r) for c in range(width)] for r in range(n)], ek, er


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_oracle_matches_go_known_answers(case):
    _, text, opts, want, err = case
    got, ek, er = oracle_records(text, opts)
    assert got == [r[:WIDTH] + [b""] * (WIDTH - len(r)) for r in want]
    assert (ek, er) == (err if err else (0, 0))


def _gen_wellformed(rng, nrec, nfields, alphabet, crlf=False, quote_prob=0.3):
    recs, out = [], io.BytesIO()
    for _ in range(nrec):
        rec = []
        for f in range(nfields):
            ln = int(rng.integers(0, 7))
            v = bytes(alphabet[rng.integers(0, len(alphabet), ln)])
            rec.append(v)
        if nfields == 1 and rec[0] == b"":
            rec[0] = b"x"   # a lone empty field is an empty line (skipped)
        recs.append(rec)
        parts = []
        for v in rec:
            needs = any(ch in v for ch in b',"\n\r') or v[:1] in (b" ", b"#") or rng.random() < quote_prob
            parts.append(b'"' + v.replace(b'"', b'""') + b'"' if needs else v)
        out.write(b",".join(parts) + (b"\r\n" if crlf else b"\n"))
    return recs, out.getvalue()


def test_oracle_vs_python_csv_on_common_subset():
    rng = np.random.default_rng(7)
    alphabet = np.frombuffer(b'ab ,"\nz1', dtype=np.uint8)   # no \r: Python keeps "\r\n" inside quotes, Go does not
    for it in range(200):
        nf = int(rng.integers(1, 5))
        recs, text = _gen_wellformed(rng, int(rng.integers(0, 30)), nf, alphabet)
        got, ek, er = oracle_records(text, {"fpr": -1}, width=nf)
        assert (ek, er) == (0, 0)
        py = [[f.encode() for f in row] for row in csv.reader(io.StringIO(text.decode(), newline=""), strict=True) if row]
        assert got == recs == py, (it, text)


# ---------------------------------------------------------------------------------------------------------------
# GPU parity
# ---------------------------------------------------------------------------------------------------------------
def gpu_records(ctx, text, opts, width=WIDTH, skip=0, out_mem=None):
    from csvplus_amd import ingest
    from csvplus_amd import _native as N
    t = ingest.csv_parse(ctx, text, list(range(width)), comma=opts.get("comma", b","), comment=opts.get("comment"),
                         trim_leading_space=opts.get("trim", False), fields_per_record=opts.get("fpr", 0),
                         skip_records=skip, out_mem=N.CPH_MEM_HOST if out_mem is None else out_mem)
    recs = [[t.columns[c].value(r) for c in range(width)] for r in range(t.nrecords)]
    return recs, t.error_kind, t.error_record if t.error_kind else 0


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_gpu_known_answers(ctx, case):
    _, text, opts, want, err = case
    got, ek, er = gpu_records(ctx, text, opts)
    assert got == [r[:WIDTH] + [b""] * (WIDTH - len(r)) for r in want]
    assert (ek, er) == (err if err else (0, 0))


@pytest.mark.gpu
def test_gpu_random_wellformed_matches_oracle(ctx):
    rng = np.random.default_rng(11)
    alphabet = np.frombuffer(b'ab ,"\n\rz#\t', dtype=np.uint8)
    for it in range(60):
        nf = int(rng.integers(1, 6))
        nrec = int(rng.integers(0, 4000 if it % 10 == 0 else 60))
        _, text = _gen_wellformed(rng, nrec, nf, alphabet, crlf=bool(it & 1))
        for opts in ({"fpr": -1}, {"fpr": 0, "trim": True}, {"fpr": nf}):
            skip = int(rng.integers(0, 3))
            assert gpu_records(ctx, text, opts, skip=skip) == oracle_records(text, opts, skip=skip), (it, opts)


@pytest.mark.gpu
def test_gpu_malformed_matches_oracle(ctx):
    """Random byte mutations: the first error (kind, record) and every record before it must agree."""
    rng = np.random.default_rng(13)
    alphabet = np.frombuffer(b'ab ,"\n\rz', dtype=np.uint8)
    junk = np.frombuffer(b'",\n\r a', dtype=np.uint8)
    nerr = 0
    for it in range(300):
        nf = int(rng.integers(1, 5))
        _, text = _gen_wellformed(rng, int(rng.integers(1, 80)), nf, alphabet, crlf=bool(it & 1))
        buf = bytearray(text)
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(0, len(buf)))
            op = int(rng.integers(0, 3))
            if op == 0:
                buf[pos] = int(junk[rng.integers(0, len(junk))])
            elif op == 1:
                del buf[pos]
            else:
                buf.insert(pos, int(junk[rng.integers(0, len(junk))]))
            if not buf:
                buf = bytearray(b"a")
        text = bytes(buf)
        for opts in ({"fpr": -1}, {"fpr": 0}, {"fpr": -1, "trim": True}):
            want = oracle_records(text, opts)
            nerr += want[1] != 0
            assert gpu_records(ctx, text, opts) == want, (it, opts, text)
    assert nerr > 100   # the mutations do produce errors


@pytest.mark.gpu
def test_gpu_comments_and_tiles(ctx):
    """Comment lines, and records/quoted fields straddling the 4 KiB tile boundaries of the parity scan."""
    rng = np.random.default_rng(17)
    lines = []
    for i in range(5000):
        r = rng.random()
        if r < 0.1:
            lines.append(b"# a comment, with commas")
        elif r < 0.15:
            lines.append(b"")
        elif r < 0.3:
            lines.append(b'%d,"multi\nline ""q"" %s",x' % (i, b"y" * int(rng.integers(0, 300))))
        else:
            lines.append(b"%d,%s,z" % (i, b"v" * int(rng.integers(0, 40))))
    text = b"\n".join(lines) + b"\n"
    opts = {"comment": b"#", "fpr": 3}
    assert gpu_records(ctx, text, opts, width=3, skip=1) == oracle_records(text, opts, width=3, skip=1)


@pytest.mark.gpu
def test_gpu_64bit_offsets_path(ctx, monkeypatch):
    """Texts of 4 GiB and more get 64-bit offsets; CPH_CSV_OFFSETS64=1 forces that code path on a small text."""
    monkeypatch.setenv("CPH_CSV_OFFSETS64", "1")
    rng = np.random.default_rng(21)
    _, text = _gen_wellformed(rng, 3000, 3, np.frombuffer(b'ab ,"\n\rz', dtype=np.uint8), crlf=True)
    from csvplus_amd import ingest
    t = ingest.csv_parse(ctx, text, [0, 2], fields_per_record=3, skip_records=2)
    assert t.columns[0].offset_bits == 64
    for opts in ({"fpr": 3}, {"fpr": -1, "trim": True}):
        assert gpu_records(ctx, text, opts, skip=1) == oracle_records(text, opts, skip=1)
    bad = text[:2000] + b'x"y\n' + text[2000:]
    assert gpu_records(ctx, bad, {"fpr": -1}) == oracle_records(bad, {"fpr": -1})


@pytest.mark.gpu
def test_gpu_device_resident_text_and_output(ctx):
    import torch
    from csvplus_amd import ingest, materialize
    from csvplus_amd import _native as N
    rng = np.random.default_rng(19)
    _, text = _gen_wellformed(rng, 20000, 4, np.frombuffer(b"abcdef,\" ", dtype=np.uint8))
    dev = torch.frombuffer(bytearray(text), dtype=torch.uint8).to("cuda:0")
    t = ingest.csv_parse(ctx, None, [3, 0], fields_per_record=4, out_mem=N.CPH_MEM_DEVICE, device_ptr=dev.data_ptr(), size=len(text))
    assert t.error_kind == 0
    want, ek, _ = orc.csv_parse(text, [3, 0], fields_per_record=4)
    assert ek == 0 and t.nrecords == want[0].nrows
    for c in range(2):
        host = materialize.gather_rows(ctx, t.columns[c])   # identity gather = device -> host copy of the column
        assert list(host.values()) == list(want[c].values())
    t.release()


@pytest.mark.gpu
def test_gpu_rejects_what_it_cannot_parse(ctx):
    from csvplus_amd import ingest
    with pytest.raises(Exception):
        ingest.csv_parse(ctx, b"a,b\n", [0], lazy_quotes=True)
    with pytest.raises(Exception):
        ingest.csv_parse(ctx, b'a,b\n# "quoted" comment\nc,d\n', [0], comment=b"#")


@pytest.mark.gpu
def test_read_csv_header_modes(ctx):
    """Reader.Iterate header handling (csvplus.go:1097-1102, makeHeader :1149-1206) on the people fixture."""
    from csvplus_amd import ingest
    from tests.helpers import people_table
    p = people_table()
    text = ("id,name,surname\n" + "".join(f"{i},{n},{s}\n" for i, n, s in zip(p["id"], p["name"], p["surname"]))).encode()
    t = ingest.read_csv(ctx, text)
    assert t.names == [b"id", b"name", b"surname"] and t.nrecords == 120 and t.error_kind == 0
    assert [v.decode() for v in t.columns[2].values()] == p["surname"]
    t = ingest.read_csv(ctx, text, select=["surname", "id"])
    assert t.names == [b"id", b"surname"] or t.names == [b"surname", b"id"]
    assert [v.decode() for v in t.columns[t.names.index(b"id")].values()] == p["id"]
    t = ingest.read_csv(ctx, text, expect_header={"name": 1, "surname": -1})
    assert [v.decode() for v in t.columns[t.names.index(b"name")].values()] == p["name"]
    with pytest.raises(KeyError):
        ingest.read_csv(ctx, text, expect_header={"name": 2})
    with pytest.raises(KeyError):
        ingest.read_csv(ctx, text, select=["xxx"])
    body = text.split(b"\n", 1)[1]
    t = ingest.read_csv(ctx, body, assume_header={"id": 0, "surname": 2})
    assert t.nrecords == 120 and [v.decode() for v in t.columns[t.names.index(b"surname")].values()] == p["surname"]
    with pytest.raises(KeyError):
        ingest.read_csv(ctx, body, assume_header={"id": 0, "zzz": 3})
    # NumFields mismatch is reported at the record where it happens, with the rows before it delivered
    bad = text + b"1,2\n"
    t = ingest.read_csv(ctx, bad)
    assert (t.error_kind, t.error_record, t.nrecords) == (3, 121, 120)
    with pytest.raises(EOFError):
        ingest.read_csv(ctx, b"")


@pytest.mark.gpu
@pytest.mark.parametrize("positions", [True, False])
@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("missing", [0, 500])
def test_pipeline_csv_to_csv(ctx, missing, fused, positions):
    """The README chain end to end (README.md:33-66): three CSV files in, the joined CSV out, every stage on the
    device (csvplus_amd.pipeline), byte-compared with the same pipeline run through the oracle's pieces."""
    from csvplus_amd import datagen as dg
    from csvplus_amd import pipeline
    nc, npd, m = 3000, 40, 20_000
    cust, prod = dg.customers(nc, encoding=dg.ITOA), dg.products(npd)
    ords = dg.orders(m, nc + missing, npd, cust_encoding=dg.ITOA)
    # make a few values need quoting so that the reader and the writer both leave their fast paths
    names = cust["name"].values()
    names[5], names[77] = b'Ann "Annie", Jr', b"multi\nline"
    from csvplus_amd import StrCol
    cust = dict(cust, name=StrCol.from_values(names))
    files = {"customers": orc.csv_write([cust["id"], cust["name"], cust["surname"]], ["id", "name", "surname"]),
             "products": orc.csv_write([prod["prod_id"], prod["product"], prod["price"]], ["prod_id", "product", "price"]),
             "orders": orc.csv_write([ords["cust_id"], ords["prod_id"], ords["qty"]], ["cust_id", "prod_id", "qty"])}
    # device pipeline
    tc = pipeline.read_table(ctx, files["customers"])
    tp = pipeline.read_table(ctx, files["products"])
    to = pipeline.read_table(ctx, files["orders"], select=["cust_id", "prod_id", "qty"])
    out_cols = [("cust_id", to, "cust_id"), ("qty", to, "qty"), ("name", tc, "name"), ("surname", tc, "surname"),
                ("product", tp, "product"), ("price", tp, "price")]
    # positions: the Join reports sorted positions and the build tables' payload columns are kept in index order
    # (cph_index_permute); otherwise original row ids into the tables as parsed
    got = pipeline.join_to_csv(ctx, to, [(tc, "id", "cust_id"), (tp, "prod_id", "prod_id")], out_cols, fused=fused, positions=positions)
    # oracle pipeline
    def parse(text, idx):
        cols, ek, _ = orc.csv_parse(text, idx, skip_records=1)
        assert ek == 0
        return cols
    oc, op, oo = parse(files["customers"], [0, 1, 2]), parse(files["products"], [0, 1, 2]), parse(files["orders"], [0, 1, 2])
    j1 = orc.OracleIndex([oc[0]]).join([oo[0]])
    j2 = orc.OracleIndex([op[0]]).join([oo[1]], row_sel=j1["probe_idx"].astype(np.uint32))
    pick = j2["probe_idx"].astype(np.int64)
    es, ea, eb = j1["probe_idx"][pick], j1["build_row"][pick], j2["build_row"]
    assert (len(es) < m) == bool(missing)
    want_cols = [StrCol.from_values([oo[0].value(int(r)) for r in es]), StrCol.from_values([oo[2].value(int(r)) for r in es]),
                 StrCol.from_values([oc[1].value(int(x)) for x in ea]), StrCol.from_values([oc[2].value(int(x)) for x in ea]),
                 StrCol.from_values([op[1].value(int(y)) for y in eb]), StrCol.from_values([op[2].value(int(y)) for y in eb])]
    want = orc.csv_write(want_cols, [n for n, _, _ in out_cols])
    assert got == want
    for t in (tc, tp, to):
        t.release()


@pytest.mark.gpu
def test_read_csv_more_than_16_columns(ctx):
    from csvplus_amd import ingest
    ncol, nrow = 37, 500
    header = ",".join(f"c{i}" for i in range(ncol))
    body = "".join(",".join(f"v{r}_{c}" for c in range(ncol)) + "\n" for r in range(nrow))
    t = ingest.read_csv(ctx, (header + "\n" + body).encode())
    assert len(t.columns) == ncol and t.nrecords == nrow and t.error_kind == 0
    assert t.names[36] == b"c36" and t.columns[36].value(499) == b"v499_36" and t.columns[16].value(0) == b"v0_16"


# ---- header logic on the host (no GPU): makeHeader, csvplus.go:1149-1206 -----------------------------------------
def test_resolve_header_like_makeHeader():
    from csvplus_amd.ingest import first_record, resolve_header
    first = [b"id", b"name", b"surname", b"name"]
    assert resolve_header(first) == {b"id": 0, b"name": 3, b"surname": 2}            # a repeated name keeps its last position
    assert resolve_header(first, select=["surname", "id"]) == {b"id": 0, b"surname": 2}
    assert resolve_header(first, expect_header={"id": 0, "surname": -1}) == {b"id": 0, b"surname": 2}
    with pytest.raises(KeyError, match="misplaced column"):
        resolve_header(first, expect_header={"surname": 1})
    with pytest.raises(KeyError, match="column not found: zip"):
        resolve_header(first, select=["id", "zip"])
    with pytest.raises(KeyError, match="columns not found: "):
        resolve_header(first, select=["zip", "born"])
    with pytest.raises(ValueError):
        resolve_header(first, select=[])
    with pytest.raises(ValueError):
        resolve_header(first, select=["id", "id"])
    with pytest.raises(ValueError):
        resolve_header([])
    # the header record itself: comments and blank lines before it are skipped, quoted names may span lines
    assert first_record(b"# c\n\n\"a\nb\",c\r\nx,y\n", comment=b"#") == [b"a\nb", b"c"]
    assert first_record(b"\n\n") is None
    assert first_record(b" a, b\n", trim_leading_space=True) == [b"a", b"b"]


def test_first_record_follows_go_reader_rules():
    """ingest.first_record (the header line, parsed on the host) against the oracle's restatement of Go's
    readRecord on the first record of random, mostly malformed texts: same fields or the same error kind.
    Includes the cases Python's csv module gets differently ("\\r\\r\\n" is the record ["\\r"], a bare quote in an
    unquoted field is an error, TrimLeadingSpace trims Unicode spaces, a lone "\\r" does not end a record)."""
    from csvplus_amd.ingest import CsvError, first_record

    rng = np.random.default_rng(41)
    alphabet = np.frombuffer(b'ab ,"\n\r#\t', dtype=np.uint8)
    texts = [b"\r\r\nx\n", b'x"y,z\n', b"a\rb,c\n", b"\xc2\xa0a,\xe3\x80\x80b\n", b'"a\r\nb",c\r\n', b"a,b\r", b"", b"\n\n", b'"abc',
             b"#c\n\n a,\tb\n"]
    texts += [alphabet[rng.integers(0, len(alphabet), int(rng.integers(0, 40)))].tobytes() for _ in range(3000)]
    kinds = {1: "bare", 2: "extraneous"}
    for t in texts:
        for trim in (False, True):
            try:
                got = first_record(t, comment=b"#", trim_leading_space=trim)
            except CsvError as e:
                got = ("ERR", e.kind.split()[0])
            cols, ek, er = orc.csv_parse(t, list(range(12)), comment=b"#", trim_leading_space=trim, fields_per_record=-1)
            if ek and er == 0:
                assert got == ("ERR", kinds[ek]), (t, trim, got)
            elif cols[0].nrows == 0:
                assert got is None, (t, trim, got)
            else:
                want = [c.value(0) for c in cols]
                assert isinstance(got, list) and len(got) <= 12, (t, trim, got)
                assert want[:len(got)] == got and all(v == b"" for v in want[len(got):]), (t, trim, got, want)


def _people_csv():
    """makePersonsCsvFile (csvplus_test.go:1220-1253): header id,name,surname,born + 120 rows, "\\n" line ends (what
    Go's csv.Writer emits; none of the fields needs quoting)."""
    from tests.helpers import PEOPLE_NAMES, PEOPLE_SURNAMES
    rng = np.random.default_rng(1916)
    rows = [["id", "name", "surname", "born"]]
    for i, name in enumerate(PEOPLE_NAMES):
        for j, surname in enumerate(PEOPLE_SURNAMES):
            rows.append([str(i * len(PEOPLE_SURNAMES) + j), name, surname, str(1916 + int(rng.integers(0, 90)))])
    return ("\n".join(",".join(r) for r in rows) + "\n").encode(), rows


@pytest.mark.gpu
def test_reference_reader_and_writer_tests(ctx):
    """The reference's own tests that touch CSV text, through ingest.read_csv / materialize.csv_write:
    TestSimpleDataSource (csvplus_test.go:117-151), TestWriteFile (:174-199), the ExpectHeader of TestSorted (:455-458)
    and the header errors of TestErrors (:810-822, :886-909), error texts included.  These pin what the reference pins;
    Go's encoding/csv itself is still restated from memory (parity for rows 8f2 / 8f3 stays UNPINNED)."""
    from csvplus_amd import ingest
    from csvplus_amd.materialize import csv_write
    from tests.helpers import PEOPLE_SURNAMES
    text, rows = _people_csv()
    header = ["id", "name", "surname", "born"]
    # TestSimpleDataSource: SelectColumns(sorted header), Filter(name is Jack or Amelia) -> 24 rows of 4 columns
    t = ingest.read_csv(ctx, text, select=sorted(header))
    assert sorted(n.decode() for n in t.names) == sorted(header) and len(t.names) == 4 and t.nrecords == 120
    names = [v.decode() for v in t.columns[t.names.index(b"name")].values()]
    assert sum(1 for n in names if n in ("Jack", "Amelia")) == len(PEOPLE_SURNAMES) * 2
    # TestWriteFile: SelectColumns(peopleHeader...).ToCsvFile(peopleHeader...) reproduces the file
    t = ingest.read_csv(ctx, text, select=header)
    cols = [t.columns[t.names.index(h.encode())] for h in header]
    out = csv_write(ctx, cols, header)
    assert out.strip() == text.strip()
    # TestSorted: ExpectHeader{name: 1, surname: 2}
    t = ingest.read_csv(ctx, text, expect_header={"name": 1, "surname": 2})
    assert [v.decode() for v in t.columns[t.names.index(b"surname")].values()] == [r[2] for r in rows[1:]]
    # TestErrors: column not found (the reference reports it at row 1), duplicate names panic, misplaced columns
    with pytest.raises(KeyError) as e:
        ingest.read_csv(ctx, text, select=["id", "name", "xxx"])
    assert "column not found: xxx" in str(e.value)
    with pytest.raises(ValueError):
        ingest.read_csv(ctx, text, select=["id", "name", "id"])
    for pos in (3, 25):
        with pytest.raises(KeyError) as e:
            ingest.read_csv(ctx, text, expect_header={"name": 1, "surname": pos})
        assert f'misplaced column "surname": expected at pos. {pos}, but found at pos. 2' in str(e.value)


def _gen_unquoted(rng, nrec, nf, crlf_prob, blank_prob, trailing_nl, alphabet=b"abc xyz0123456789#\t;"):
    """Text without any quote: `nrec` records of `nf` fields (empty fields included), line ends "\\n" or "\\r\\n",
    blank lines ("\\n", "\\r\\n") sprinkled in, stray '\\r' inside fields."""
    al = np.frombuffer(alphabet, dtype=np.uint8)
    out = bytearray()
    for r in range(nrec):
        while rng.random() < blank_prob:
            out += b"\r\n" if rng.random() < 0.5 else b"\n"
        fields = []
        for f in range(nf):
            ln = int(rng.integers(0, 9)) if rng.random() < 0.9 else int(rng.integers(20, 70))
            v = al[rng.integers(0, len(al), ln)].tobytes()
            if rng.random() < 0.03:
                v += b"\r"                      # a '\r' that is data unless it ends the line before "\n"
            fields.append(v)
        if nf == 1 and fields[0] in (b"", b"\r"):
            fields[0] = b"q"                    # an empty single-field line would be a blank line
        out += b",".join(fields)
        last = r == nrec - 1
        if not last or trailing_nl:
            out += b"\r\n" if rng.random() < crlf_prob else b"\n"
    return bytes(out)


@pytest.mark.gpu
def test_gpu_unquoted_texts_match_oracle(ctx):
    """Texts without quotes, the bulk of real CSV: line-end styles, blank lines, a missing final newline, a final lone
    '\r', stray '\r' inside fields, empty fields, records crossing the scan tiles, skipped header records, column subsets."""
    from csvplus_amd import ingest
    from csvplus_amd import _native as N
    rng = np.random.default_rng(77)
    for it in range(70):
        nf = int(rng.integers(1, 7))
        nrec = int(rng.integers(1, 60)) if it % 7 else int(rng.integers(3000, 9000))     # the large ones span several tiles
        text = _gen_unquoted(rng, nrec, nf, crlf_prob=[0.0, 1.0, 0.4][it % 3], blank_prob=0.05 if it % 2 else 0.0,
                             trailing_nl=bool(it % 5))
        if it % 11 == 0:
            text += b"\r"                                                                  # dropped, or a lone-'\r' last line
        want_cols = sorted(rng.choice(nf, size=int(rng.integers(1, nf + 1)), replace=False).tolist())
        skip = int(rng.integers(0, 3))
        fpr = [0, -1, nf][it % 3]
        ocols, oek, oer = orc.csv_parse(text, want_cols, fields_per_record=fpr, skip_records=skip)
        t = ingest.csv_parse(ctx, text, want_cols, fields_per_record=fpr, skip_records=skip, out_mem=N.CPH_MEM_HOST)
        got = [[t.columns[c].value(r) for c in range(len(want_cols))] for r in range(t.nrecords)]
        want = [[ocols[c].value(r) for c in range(len(want_cols))] for r in range(ocols[0].nrows)]
        assert (got, t.error_kind) == (want, oek), (it, nf, want_cols, skip, fpr, text[:200])


@pytest.mark.gpu
def test_gpu_small_mixed_cases_match_oracle(ctx):
    """A quoted field among plain records, comment lines, ragged records, a line longer than a scan tile, a wanted field
    beyond the records, a field-count error."""
    from csvplus_amd import ingest
    from csvplus_amd import _native as N
    cases = [
        (b"a,b\nc,\"d\"\ne,f\n", {}, [0, 1]),
        (b"a,b\n#x,y\nc,d\n", {"comment": b"#"}, [0, 1]),
        (b"a,b\nc\nd,e,f\n", {"fields_per_record": -1}, [0, 1, 2]),
        (b"a,b\n" + b"x" * 40_000 + b",y\nc,d\n", {}, [0, 1]),
        (b"a,b\nc,d\n", {"fields_per_record": -1}, [0, 3]),
        (b"a,b\nc,d,e\n", {}, [0, 1]),                                  # ErrFieldCount at record 1
    ]
    for text, kw, cols in cases:
        ocols, oek, oer = orc.csv_parse(text, cols, **kw)
        t = ingest.csv_parse(ctx, text, cols, out_mem=N.CPH_MEM_HOST, **kw)
        got = [[t.columns[c].value(r) for c in range(len(cols))] for r in range(t.nrecords)]
        assert got == [[ocols[c].value(r) for c in range(len(cols))] for r in range(ocols[0].nrows)]
        assert (t.error_kind, t.error_record if t.error_kind else 0) == (oek, oer if oek else 0)


# ---------------------------------------------------------------------------------------------------------------
# (round 6) the byte-parallel fast path (csv_ingest.hip: k_csv_fast): texts without quotes, <= 8 columns
# ---------------------------------------------------------------------------------------------------------------
def _parse_profiled(ctx, text, cols, **kw):
    from csvplus_amd import ingest
    from csvplus_amd import _native as N
    ctx.profile(True)
    ctx.profile_read(reset=True)
    t = ingest.csv_parse(ctx, text, cols, out_mem=N.CPH_MEM_HOST, **kw)
    prof = ctx.profile_read(reset=True)
    ctx.profile(False)
    vals = [t.columns[c].values() for c in range(len(cols))]
    return vals, t.nrecords, t.error_kind, (t.error_record if t.error_kind else 0), prof


def _fast_vs_classic(ctx, text, cols, expect_fast=True, oracle=True, **kw):
    got = _parse_profiled(ctx, text, cols, **kw)
    assert ("k_csv_fast_copy" in got[4]) == expect_fast and ("k_csv_copy_fields" in got[4]) != expect_fast, sorted(got[4])
    ctx.set_option("csv_fast", 0)
    try:
        ref = _parse_profiled(ctx, text, cols, **kw)
    finally:
        ctx.set_option("csv_fast", 1)
    assert "k_csv_fast_count" not in ref[4]
    assert got[1:4] == ref[1:4], (got[1:4], ref[1:4])
    for c in range(len(cols)):
        if got[0][c] != ref[0][c]:
            r = next(i for i, (x, y) in enumerate(zip(got[0][c], ref[0][c])) if x != y)
            raise AssertionError(f"column {c} (field {cols[c]}) record {r}: fast {got[0][c][r - 1:r + 2]} classic {ref[0][c][r - 1:r + 2]}; "
                                 f"{len(text)} bytes, kw {kw}")
    if oracle:
        ocols, oek, oer = orc.csv_parse(text, cols, **kw)
        assert got[1] == ocols[0].nrows and (got[2], got[3]) == (oek, oer if oek else 0)
        assert got[0] == [ocols[c].values() for c in range(len(cols))]
    return got


@pytest.mark.gpu
def test_fast_path_random_unquoted_texts(ctx):
    """Line-end styles, a missing final newline, a final lone '\\r', stray '\\r' inside fields, empty fields, ragged records with
    FieldsPerRecord < 0, skipped header records, one field asked for twice, 1-4 columns — through k_csv_fast, equal to the classic
    kernels and to the oracle."""
    rng = np.random.default_rng(606)
    for it in range(40):
        nf = int(rng.integers(1, 7))
        nrec = int(rng.integers(1, 80)) if it % 4 else int(rng.integers(4000, 30000))     # the large ones span many 16 KiB tiles
        text = _gen_unquoted(rng, nrec, nf, crlf_prob=[0.0, 1.0, 0.4][it % 3], blank_prob=0.0, trailing_nl=bool(it % 5))
        if it % 11 == 0:
            text += b"\r"
        ncols = int(rng.integers(1, min(nf, 4) + 1))
        cols = sorted(rng.choice(nf, size=ncols, replace=False).tolist())
        if it % 6 == 0 and ncols < 4:
            cols = cols + [cols[0]]                         # the same field as two columns
        if it % 9 == 0:
            cols[-1] = nf + 2                               # a field no record has: "" everywhere (FieldsPerRecord < 0 only)
        fpr = -1 if it % 9 == 0 else [0, -1, nf][it % 3]
        _fast_vs_classic(ctx, text, cols, fields_per_record=fpr, skip_records=int(rng.integers(0, 4)))


@pytest.mark.gpu
def test_fast_path_ragged_records_and_tile_edges(ctx):
    rng = np.random.default_rng(607)
    # ragged: records with 1..5 fields, FieldsPerRecord < 0: missing values are ""
    lines = [b",".join(b"%d" % int(v) for v in rng.integers(0, 10 ** int(rng.integers(1, 8)), int(rng.integers(1, 6)))) for _ in range(20000)]
    text = b"\n".join(lines) + b"\n"
    _fast_vs_classic(ctx, text, [0, 2, 4], fields_per_record=-1)
    _fast_vs_classic(ctx, text, [1], fields_per_record=-1, skip_records=2)
    # differing field counts with FieldsPerRecord = 0: the classic kernels report the first wrong record
    got = _fast_vs_classic(ctx, text, [0, 1], expect_fast=False, fields_per_record=0)
    assert got[2] != 0
    # newlines exactly at, before and behind the 16 KiB tile boundaries; texts of exactly k tiles; one record per tile
    for pad in (16382, 16383, 16384, 16385, 32767, 32768):
        first = b"h1,h2\n"
        body = b"x" * (pad - len(first) - 3) + b",y\n" + b"a,b\nc,d\n"
        fits = pad < 20000                                                     # (a record may run 4 KiB past the end of its tile)
        _fast_vs_classic(ctx, first + body, [0, 1], expect_fast=fits)
        _fast_vs_classic(ctx, first + body[:-1], [1, 0], expect_fast=fits, skip_records=1)      # no final newline
    text = (b"k" * 8 + b",v\n") * 2048 * 3                                       # 11 bytes x 6144 records: not a multiple of the tile
    _fast_vs_classic(ctx, text, [0, 1])
    text = (b"k" * 13 + b",v\n") * 1024 * 4                                      # 16-byte records: every tile ends with a newline
    assert len(text) % 16384 == 0
    _fast_vs_classic(ctx, text, [0, 1])
    _fast_vs_classic(ctx, text[:-1], [0, 1])
    # a record that runs ~4 KiB past the end of its tile still fits the staged window; a longer one goes to the classic kernels
    fits = b"a,b\n" * 4000 + b"x" * 3500 + b",y\n" + b"c,d\n" * 10
    _fast_vs_classic(ctx, fits, [0, 1])
    too_long = b"a,b\n" * 4000 + b"x" * 30000 + b",y\n" + b"c,d\n" * 10
    _fast_vs_classic(ctx, too_long, [0, 1], expect_fast=False)


@pytest.mark.gpu
def test_fast_path_gives_way_to_the_classic_kernels(ctx):
    """A quote anywhere, TrimLeadingSpace, a blank line in the middle, a comment line, more than 8 columns: the record-parallel kernels."""
    base = b"id,name,qty\n" + b"".join(b"%d,n%d,%d\n" % (i, i % 97, i % 13) for i in range(5000))
    _fast_vs_classic(ctx, base, [0, 2], skip_records=1)
    _fast_vs_classic(ctx, base, [0, 2], comment=b"#")                                        # a comment character, no comment line
    _fast_vs_classic(ctx, base + b'7,"q",1\n', [0, 2], expect_fast=False)
    _fast_vs_classic(ctx, base, [0, 2], expect_fast=False, trim_leading_space=True)
    _fast_vs_classic(ctx, base[:3000] + b"\n" + base[3000:], [0, 2], expect_fast=False, fields_per_record=-1)
    _fast_vs_classic(ctx, base.replace(b"\n2500,", b"\n#2500,"), [0, 2], expect_fast=False, comment=b"#")
    _fast_vs_classic(ctx, b"#c\n" + base, [0, 2], expect_fast=False, comment=b"#")
    _fast_vs_classic(ctx, base + b"\n", [0, 2], expect_fast=False)                             # a blank line at the very end
    wide = b"".join(b",".join(b"%d" % (i * j) for j in range(1, 12)) + b"\n" for i in range(3000))
    _fast_vs_classic(ctx, wide, [0, 1, 2, 3, 4])                                               # 5-8 columns: the 8-column instantiation
    _fast_vs_classic(ctx, wide, [10, 0, 3, 3, 7, 1, 2, 9], skip_records=2)
    _fast_vs_classic(ctx, wide, [0, 1, 2, 3, 4, 5, 6, 7, 8], expect_fast=False)               # 9 columns: the record-parallel kernels


@pytest.mark.gpu
def test_fast_path_at_bench_shape(ctx):
    """The orders file of tools/microbench/csv_ingest.py at 2e6 records, device-resident text and output: both paths, same columns."""
    import torch
    from csvplus_amd import datagen as dg, ingest, _native as N

    ords = dg.orders(2_000_000, 100_000, 1000)
    a, b, c = (ords[k].values() for k in ("cust_id", "prod_id", "qty"))
    text = b"cust_id,prod_id,qty\n" + b"".join(x + b"," + y + b"," + z + b"\n" for x, y, z in zip(a, b, c))
    dtext = torch.frombuffer(bytearray(text), dtype=torch.uint8).to("cuda:0")
    res = []
    for fast in (1, 0):
        ctx.set_option("csv_fast", fast)
        t = ingest.csv_parse(ctx, None, [0, 1], skip_records=1, out_mem=N.CPH_MEM_HOST, device_ptr=dtext.data_ptr(), size=dtext.numel())
        assert t.nrecords == 2_000_000 and t.error_kind == 0
        res.append([(np.asarray(t.columns[k].data).tobytes(), np.asarray(t.columns[k].offsets).tobytes()) for k in range(2)])
    ctx.set_option("csv_fast", 1)
    assert res[0] == res[1]
    assert res[0][0][0] == b"".join(a) and res[0][1][0] == b"".join(b)

"""The step before the path: CSV text -> SoA string columns on the GPU (cph_csv_parse), mirroring the
reference's Reader (csvplus.go:922-1227).  The header logic (makeHeader, csvplus.go:1149-1206) runs here on
the host over the first record only; the parse of the body is the device's."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N
from .columns import StrCol

ERR_NAMES = {0: None, N.CPH_CSV_ERR_BARE_QUOTE: "bare \" in non-quoted field",
             N.CPH_CSV_ERR_QUOTE: "extraneous or missing \" in quoted-field",
             N.CPH_CSV_ERR_FIELD_COUNT: "wrong number of fields"}


class CsvError(Exception):
    """Mirrors csv.ParseError as csvplus reports it: kind + the record it happened in."""

    def __init__(self, kind: int, record: int):
        super().__init__(f"record {record}: {ERR_NAMES.get(kind, kind)}")
        self.kind, self.record = kind, record


class CsvTable:
    """Library-owned result of cph_csv_parse.  `columns` are StrCols (host copies or zero-copy device views
    valid until release())."""

    def __init__(self, ctx, ptr, out_mem):
        self.ctx, self.ptr = ctx, ptr
        t = ptr.contents
        self.nrecords, self.error_kind, self.error_record = int(t.nrecords), int(t.error_kind), int(t.error_record)
        self.columns = []
        for c in range(int(t.ncols)):
            sc = t.cols[c]
            bits = int(sc.offset_bits)
            if out_mem == N.CPH_MEM_HOST:
                offs = N._ptr_array(sc.offsets, self.nrecords + 1, np.uint32 if bits == 32 else np.uint64).copy()
                data = N._ptr_array(sc.data, int(offs[-1]), np.uint8).copy() if int(offs[-1]) else np.empty(0, np.uint8)
                self.columns.append(StrCol(data, offs, self.nrecords, bits))
            else:
                self.columns.append(StrCol(_Raw(sc.data), _Raw(sc.offsets), self.nrecords, bits, N.CPH_MEM_DEVICE, fixed_width=0))
        ctx._children.add(self)
        if out_mem == N.CPH_MEM_HOST:
            self.release()

    def release(self):
        if self.ptr:
            self.ctx.lib.cph_csv_table_release(self.ptr)
            self.ptr = None

    close = release

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class _Raw:   # minimal object with data_ptr()
    def __init__(self, p):
        self._p = int(p or 0)

    def data_ptr(self):
        return self._p


def csv_parse(ctx: N.Context, data, col_index, *, comma=b",", comment=None, trim_leading_space=False,
              lazy_quotes=False, fields_per_record=0, skip_records=0, out_mem=N.CPH_MEM_HOST,
              device_ptr=None, size=None) -> CsvTable:
    """data: bytes / numpy uint8 (host), or pass device_ptr + size for text already in HBM."""
    opt = N.cph_csv_options(ord(comma), ord(comment) if comment else 0, 1 if trim_leading_space else 0,
                            1 if lazy_quotes else 0, int(fields_per_record), int(skip_records))
    idx = (C.c_int32 * len(col_index))(*[int(i) for i in col_index])
    out = C.POINTER(N.cph_csv_table)()
    if device_ptr is not None:
        ptr, n, mem, keep = C.c_void_p(int(device_ptr)), int(size), N.CPH_MEM_DEVICE, None
    else:
        keep = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data)
        ptr, n, mem = C.c_void_p(keep.ctypes.data if len(keep) else 0), len(keep), N.CPH_MEM_HOST
    ctx._check(ctx.lib.cph_csv_parse(ctx.handle, ptr, n, mem, C.byref(opt), idx, len(col_index), out_mem, C.byref(out)))
    del keep
    return CsvTable(ctx, out, out_mem)


class CsvError(ValueError):
    """A parse error in the first record, with Go's error kinds (encoding/csv: ErrBareQuote, ErrQuote)."""

    def __init__(self, kind: str, line: int):
        super().__init__(f"record on line {line}: {kind}")
        self.kind, self.line = kind, line


_GO_SPACES = {0x09, 0x0A, 0x0B, 0x0C, 0x0D, 0x20, 0x85, 0xA0, 0x1680, 0x2028, 0x2029, 0x202F, 0x205F, 0x3000} | set(
    range(0x2000, 0x200B))


def _trim_left_go(line: bytes) -> bytes:
    """strings.TrimLeftFunc(line, unicode.IsSpace) on UTF-8 bytes (an invalid byte is RuneError: not a space)."""
    i, n = 0, len(line)
    while i < n:
        b0 = line[i]
        if b0 < 0x80:
            cp, w = b0, 1
        else:
            w = 2 if b0 >> 5 == 0b110 else 3 if b0 >> 4 == 0b1110 else 4 if b0 >> 3 == 0b11110 else 0
            try:
                cp = ord(line[i:i + w].decode("utf-8")) if w else -1
            except UnicodeDecodeError:
                cp = -1
        if cp not in _GO_SPACES:
            break
        i += w
    return line[i:]


def first_record(text: bytes, comma=b",", comment=None, trim_leading_space=False):
    """The first record of `text` by the rules of Go's encoding/csv Reader.readRecord — the record makeHeader
    (csvplus.go:1149-1206) reads before the body is handed to the device.  Line endings: "\r\n" counts as "\n",
    a final "\r" before EOF is dropped; lines that are empty after that, or start with the comment rune, are
    skipped ("\r\r\n" is NOT empty: it is the record ["\r"]); a quote inside an unquoted field is ErrBareQuote,
    anything but a separator or the line end behind a closing quote is ErrQuote, a quoted field may span lines.
    Returns the list of fields (bytes), or None at EOF.  Raises CsvError."""
    pos, n, lineno = 0, len(text), 0
    sep = comma[0]

    def read_line():
        nonlocal pos, lineno
        if pos >= n:
            return None
        end = text.find(b"\n", pos)
        if end < 0:
            line, pos = text[pos:], n
            if line.endswith(b"\r"):
                line = line[:-1]          # a trailing "\r" before EOF is dropped
        else:
            line, pos = text[pos:end + 1], end + 1
            if line.endswith(b"\r\n"):
                line = line[:-2] + b"\n"
        lineno += 1
        return line

    while True:
        line = read_line()
        if line is None:
            return None
        if comment and line.startswith(comment):
            continue
        if line in (b"\n", b""):
            continue
        break
    rec_line = lineno
    fields = []
    while True:
        if trim_leading_space:
            line = _trim_left_go(line)
        if not line.startswith(b'"'):                       # unquoted field
            i = line.find(comma)
            field = line[:i] if i >= 0 else line[:len(line) - (1 if line.endswith(b"\n") else 0)]
            if b'"' in field:
                raise CsvError("bare \" in non-quoted field", rec_line)
            fields.append(field)
            if i >= 0:
                line = line[i + 1:]
                continue
            return fields
        line = line[1:]                                     # quoted field
        buf = bytearray()
        while True:
            i = line.find(b'"')
            if i >= 0:
                buf += line[:i]
                line = line[i + 1:]
                if line[:1] == b'"':                       # "" -> one quote
                    buf += b'"'
                    line = line[1:]
                elif line[:1] and line[0] == sep:            # closing quote, next field
                    line = line[1:]
                    fields.append(bytes(buf))
                    break
                elif line in (b"", b"\n"):                  # closing quote at the end of the line
                    fields.append(bytes(buf))
                    return fields
                else:
                    raise CsvError("extraneous or missing \" in quoted-field", rec_line)
            elif line:                                       # the field continues on the next line
                buf += line
                nxt = read_line()
                if nxt is None:                              # EOF inside quotes
                    raise CsvError("extraneous or missing \" in quoted-field", rec_line)
                line = nxt
            else:
                raise CsvError("extraneous or missing \" in quoted-field", rec_line)


def _b(x):
    return x.encode() if isinstance(x, str) else bytes(x)


def resolve_header(first, *, select=None, expect_header=None):
    """makeHeader (csvplus.go:1149-1206) on the first record `first` (list of bytes): name -> field index.
    No spec: every name, a repeated name keeps its LAST position (:1159-1167).  select = SelectColumns names
    (:1009-1026, each looked up); expect_header = {name: index, -1 = look the name up} (:985-1003): a name found at
    another position is an error (:1176-1181), names that never occur are reported together (:1186-1202)."""
    if not first:
        raise ValueError("empty header")                       # :1156-1158
    spec = None
    if select is not None:
        names = [_b(n) for n in select]
        if not names:
            raise ValueError("empty header spec")              # panic :1010-1012
        if len(set(names)) != len(names):
            raise ValueError("header spec: duplicate column name")   # panic :1017-1019
        spec = {n: -1 for n in names}
    elif expect_header is not None:
        if not expect_header:
            raise ValueError("empty header spec")              # panic :986-988
        spec = {_b(k): int(v) for k, v in expect_header.items()}
    hdr = {}
    if spec is None:
        for i, nm in enumerate(first):
            hdr[nm] = i
        return hdr
    for i, nm in enumerate(first):
        if nm in spec:
            if spec[nm] == -1 or spec[nm] == i:
                hdr[nm] = i
            else:
                q = '"' + nm.decode("utf-8", "replace").replace("\\", "\\\\").replace('"', '\\"') + '"'   # Go %q for plain names
                raise KeyError(f"misplaced column {q}: expected at pos. {spec[nm]}, but found at pos. {i}")
    missing = [n for n in spec if n not in hdr]
    if missing:
        raise KeyError(("columns not found: " if len(missing) > 1 else "column not found: ")
                       + ", ".join(m.decode("utf-8", "replace") for m in missing))
    return hdr


def read_csv(ctx: N.Context, text: bytes, *, select=None, expect_header=None, assume_header=None, comma=b",",
             comment=None, trim_leading_space=False, lazy_quotes=False, num_fields=0, out_mem=N.CPH_MEM_HOST) -> CsvTable:
    """A csvplus Reader materialised as columns: FromFile(...)[.SelectColumns(select...) |
    .ExpectHeader(expect_header) | .AssumeHeader(assume_header)][.NumFields(num_fields)].

    Header modes as in the reference (resolve_header; assume_header = {name: index}, no header line: AssumeHeader
    :963-980).  num_fields is csv.Reader.FieldsPerRecord (0 = as the first record, the header included; <0 = any,
    short records padded with "" :1121-1122).
    Returns the table with `.names`; `.error_kind/.error_record` report a parse error the way the reference
    returns it after delivering the rows before it.  Header problems raise (the reference fails at line 1).
    """
    if lazy_quotes:   # Reader.LazyQuotes (csvplus.go:1040-1043): the device parser has no lazy mode
        raise NotImplementedError("LazyQuotes is not supported by the device CSV reader")
    skip = 0
    if assume_header is not None:
        if not assume_header:
            raise ValueError("Empty header spec")
        hdr = {_b(k): int(v) for k, v in assume_header.items()}
        if any(v < 0 for v in hdr.values()):
            raise ValueError("header spec: negative index")
        if num_fields >= 0:   # csvplus.go:1123-1127: an index beyond the record is an error unless padding is allowed
            first = first_record(text, comma, comment, trim_leading_space)
            if first is not None:
                for nm, ix in hdr.items():
                    if ix >= (num_fields if num_fields > 0 else len(first)):
                        raise KeyError(f"column not found: {nm!r} ({ix})")
    else:
        first = first_record(text, comma, comment, trim_leading_space)
        if first is None:
            raise EOFError("EOF")   # io.EOF from the header read, csvplus.go:1150-1154
        skip = 1
        hdr = resolve_header(first, select=select, expect_header=expect_header)
    names = list(hdr.keys())
    parts = []
    for i in range(0, len(names), N.CPH_MAX_KEY_COLS):   # the C ABI takes up to 16 columns per call
        batch = names[i:i + N.CPH_MAX_KEY_COLS]
        parts.append(csv_parse(ctx, text, [hdr[n] for n in batch], comma=comma, comment=comment,
                               trim_leading_space=trim_leading_space, fields_per_record=num_fields, skip_records=skip,
                               out_mem=out_mem))
    t = parts[0] if len(parts) == 1 else CsvTableGroup(parts)
    t.names = names
    return t


class CsvTableGroup:
    """More than CPH_MAX_KEY_COLS columns: the text was parsed once per batch of columns (every batch reports the
    same records and the same first error)."""

    def __init__(self, parts):
        self.parts = parts
        self.nrecords, self.error_kind, self.error_record = parts[0].nrecords, parts[0].error_kind, parts[0].error_record
        self.columns = [c for p in parts for c in p.columns]

    def release(self):
        for p in self.parts:
            p.release()

    close = release

"""Known-answer cases for Go's encoding/csv Reader — the third-party (stdlib) dependency csvplus' Reader.Iterate
delegates the parse to (csvplus.go:1091-1097; go.mod: go 1.23).  encoding/csv is not under /root/reference, so
these cases restate its documented behaviour (package doc + the behaviours its reader tests pin: CRLF handling,
bare CR, blank lines, comments, quote errors, field counts, trailing commas).  They pin oracle/orc.csv_parse;
the GPU parser is then compared with that oracle.

Each case: (name, input bytes, options, expected) where expected is a list of records (lists of bytes) plus an
optional (error_kind, error_record) — error kinds 1 = ErrBareQuote, 2 = ErrQuote, 3 = ErrFieldCount.
Options: comma, comment, trim (TrimLeadingSpace), fpr (FieldsPerRecord; csv.Reader's zero value 0 = "as the
first record").
"""
BARE, QUOTE, FIELDS = 1, 2, 3

CASES = [
    ("Simple", b"a,b,c\n", {}, [[b"a", b"b", b"c"]], None),
    ("CRLF", b"a,b\r\nc,d\r\n", {}, [[b"a", b"b"], [b"c", b"d"]], None),
    ("BareCR", b"a,b\rc,d\r\n", {}, [[b"a", b"b\rc", b"d"]], None),
    ("RFC4180", b'#field1,field2,field3\n"aaa","bb\nb","ccc"\n"a,a","b""bb","ccc"\nzzz,yyy,xxx\n', {},
     [[b"#field1", b"field2", b"field3"], [b"aaa", b"bb\nb", b"ccc"], [b"a,a", b'b"bb', b"ccc"], [b"zzz", b"yyy", b"xxx"]], None),
    ("NoEOL", b"a,b,c", {}, [[b"a", b"b", b"c"]], None),
    ("Semicolon", b"a;b;c\n", {"comma": b";"}, [[b"a", b"b", b"c"]], None),
    ("MultiLine", b'"two\nline","one line","three\nline\nfield"', {},
     [[b"two\nline", b"one line", b"three\nline\nfield"]], None),
    ("BlankLine", b"a,b,c\n\nd,e,f\n\n", {}, [[b"a", b"b", b"c"], [b"d", b"e", b"f"]], None),
    ("BlankLineFieldCount", b"a,b,c\n\nd,e,f\n\n", {"fpr": 0}, [[b"a", b"b", b"c"], [b"d", b"e", b"f"]], None),
    ("TrimSpace", b" a,  b,   c\n", {"trim": True}, [[b"a", b"b", b"c"]], None),
    ("LeadingSpace", b" a,  b,   c\n", {}, [[b" a", b"  b", b"   c"]], None),
    ("Comment", b"#1,2,3\na,b,c\n#comment", {"comment": b"#"}, [[b"a", b"b", b"c"]], None),
    ("NoComment", b"#1,2,3\na,b,c", {}, [[b"#1", b"2", b"3"], [b"a", b"b", b"c"]], None),
    ("BadDoubleQuotes", b'a""b,c', {}, [], (BARE, 0)),
    ("BadBareQuote", b'a "word","b"', {}, [], (BARE, 0)),
    ("BadTrailingQuote", b'"a word",b"', {}, [], (BARE, 0)),
    ("ExtraneousQuote", b'"a "word","b"', {}, [], (QUOTE, 0)),
    ("BadFieldCount", b"a,b,c\nd,e", {"fpr": 0}, [[b"a", b"b", b"c"]], (FIELDS, 1)),
    ("BadFieldCountMultiple", b"a,b,c\nd,e\nf", {"fpr": 0}, [[b"a", b"b", b"c"]], (FIELDS, 1)),
    ("BadFieldCount1", b"a,b,c", {"fpr": 2}, [], (FIELDS, 0)),
    ("FieldCount", b"a,b,c\nd,e", {"fpr": -1}, [[b"a", b"b", b"c"], [b"d", b"e"]], None),
    ("TrailingCommaEOF", b"a,b,c,", {}, [[b"a", b"b", b"c", b""]], None),
    ("TrailingCommaEOL", b"a,b,c,\n", {}, [[b"a", b"b", b"c", b""]], None),
    ("TrailingCommaSpaceEOF", b"a,b,c, ", {"trim": True}, [[b"a", b"b", b"c", b""]], None),
    ("TrailingCommaSpaceEOL", b"a,b,c, \n", {"trim": True}, [[b"a", b"b", b"c", b""]], None),
    ("TrailingCommaLine3", b"a,b,c\nd,e,f\ng,hi,", {"trim": True},
     [[b"a", b"b", b"c"], [b"d", b"e", b"f"], [b"g", b"hi", b""]], None),
    ("NotTrailingComma3", b"a,b,c, \n", {}, [[b"a", b"b", b"c", b" "]], None),
    ("CommaFieldTest",
     b'x,y,z,w\nx,y,z,\nx,y,,\nx,,,\n,,,\n"x","y","z","w"\n"x","y","z",""\n"x","y","",""\n"x","","",""\n"","","",""\n', {},
     [[b"x", b"y", b"z", b"w"], [b"x", b"y", b"z", b""], [b"x", b"y", b"", b""], [b"x", b"", b"", b""],
      [b"", b"", b"", b""], [b"x", b"y", b"z", b"w"], [b"x", b"y", b"z", b""], [b"x", b"y", b"", b""],
      [b"x", b"", b"", b""], [b"", b"", b"", b""]], None),
    ("TrailingCommaIneffective1", b"a,b,\nc,d,e", {"trim": True}, [[b"a", b"b", b""], [b"c", b"d", b"e"]], None),
    ("CRLFInQuotedField", b'A,"Hello\r\nHi",B\r\n', {}, [[b"A", b"Hello\nHi", b"B"]], None),
    ("TrailingCR", b"field1,field2\r", {}, [[b"field1", b"field2"]], None),
    ("QuotedTrailingCR", b'"field"\r', {}, [[b"field"]], None),
    ("QuotedTrailingCRCR", b'"field"\r\r', {}, [], (QUOTE, 0)),
    ("FieldCR", b"field\rfield\r", {}, [[b"field\rfield"]], None),
    ("FieldCRCR", b"field\r\rfield\r\r", {}, [[b"field\r\rfield\r"]], None),
    ("FieldCRCRLF", b"field\r\r\nfield\r\r\n", {}, [[b"field\r"], [b"field\r"]], None),
    ("FieldCRCRLFCR", b"field\r\r\n\rfield\r\r\n\r", {}, [[b"field\r"], [b"\rfield\r"]], None),
    ("FieldCRCRLFCRCR", b"field\r\r\n\r\rfield\r\r\n\r\r", {}, [[b"field\r"], [b"\r\rfield\r"], [b"\r"]], None),
    ("MultiFieldCRCRLFCRCR", b"field1,field2\r\r\n\r\rfield1,field2\r\r\n\r\r,", {},
     [[b"field1", b"field2\r"], [b"\r\rfield1", b"field2\r"], [b"\r\r", b""]], None),
    ("QuotedFieldMultipleLF", b'"\n\n\n\n"', {}, [[b"\n\n\n\n"]], None),
    ("MultipleCRLF", b"\r\n\r\n\r\n\r\n", {}, [], None),
    ("DoubleQuoteWithTrailingCRLF", b'"foo""bar"\r\n', {}, [[b'foo"bar']], None),
    ("EvenQuotes", b'""""""""', {}, [[b'"""']], None),
    ("OddQuotes", b'"""""""', {}, [], (QUOTE, 0)),
    ("QuoteWithTrailingCRLF", b'"foo"bar"\r\n', {}, [], (QUOTE, 0)),
    ("StartLine2Error", b'a,b\n"d\n\n,e', {}, [[b"a", b"b"]], (QUOTE, 1)),
    ("ErrorAfterRows", b'a,b\nc,d\ne,f"x\ng,h\n', {}, [[b"a", b"b"], [b"c", b"d"]], (BARE, 2)),
    ("Empty", b"", {}, [], None),
    ("OnlyNewline", b"\n", {}, [], None),
    ("SpaceOnlyLineTrim", b"a\n  \nb\n", {"trim": True, "fpr": -1}, [[b"a"], [b""], [b"b"]], None),
    ("UnicodeSpaceTrim", "a,  b\n".encode(), {"trim": True}, [[b"a", b"b"]], None),
]

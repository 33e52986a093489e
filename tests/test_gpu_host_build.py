"""IndexOn over a key column in HOST memory through host-formed codes (csrc/host_encode.hip: build_from_host_codes; createIndex,
csvplus.go:707-738, as a cgo caller hands it over): the index must be the one the general path (upload of the strings, device
encode) builds, bit for bit — and the oracle's."""
import numpy as np
import pytest

from csvplus_amd import Context, DeviceIndex, StrCol, _native as N, datagen as dg
from oracle import orc

pytestmark = pytest.mark.gpu

NROWS = (1 << 20) + 12345


def both(ctx, col, unique):
    """(host-coded build, general build) of the same host column."""
    ctx.set_option("host_build", 1)
    h = DeviceIndex(ctx, [col], unique=unique)
    ctx.set_option("host_build", 0)
    g = DeviceIndex(ctx, [col], unique=unique)
    ctx.set_option("host_build", 1)
    return h, g


@pytest.mark.parametrize("shape", ["fixed8_full", "fixed8_sparse", "itoa", "fixed6", "duplicates"])
def test_host_coded_build_equals_general_and_oracle(shape):
    ctx = Context(0)
    unique = shape != "duplicates"
    if shape == "fixed8_full":
        col = dg.column(dg.SEQ_PERM, NROWS, NROWS, encoding=dg.FIXED8, seed=5)          # dense ids: the direct sort
    elif shape == "fixed8_sparse":
        col = dg.column(dg.SEQ_PERM, NROWS, 90_000_000, encoding=dg.FIXED8, seed=6)     # sparse ids: radix passes
    elif shape == "itoa":
        col = dg.column(dg.SEQ_PERM, NROWS, 3 * NROWS, encoding=dg.ITOA, seed=7)        # variable length 1-7 digits, 32-bit offsets
    elif shape == "fixed6":
        rng = np.random.default_rng(8)
        vals = rng.permutation(26 ** 5)[:NROWS]
        b = np.empty((NROWS, 6), dtype=np.uint8)
        for q in range(5):
            b[:, 4 - q] = 97 + (vals // 26 ** q) % 26
        b[:, 5] = 35
        col = StrCol.from_arrays(b.reshape(-1), (np.arange(NROWS + 1, dtype=np.uint64) * 6).astype(np.uint32))
    else:
        col = dg.column(dg.UNIFORM, NROWS, 50_000, encoding=dg.ITOA, seed=9)            # ~21 rows per key
    h, g = both(ctx, col, unique)
    assert h.info()["build_path"] == 2 and g.info()["build_path"] == 0
    assert h.status == g.status == N.CPH_OK and h.first_dup == g.first_dup
    np.testing.assert_array_equal(h.perm(), g.perm())
    o = orc.OracleIndex([col])
    np.testing.assert_array_equal(h.perm(), o.perm)
    assert h.first_dup == o.first_dup()
    # the index answers like any other: a Join against it
    probe = col.slice(1000, 1000 + 50_000)
    m = h.probe([probe], want_pairs=True)
    oj = o.join([probe])
    assert m.nmatches == oj["nmatches"]
    np.testing.assert_array_equal(m.build_row, oj["build_row"])
    m.release()
    h.close(); g.close(); ctx.close()


def test_host_coded_build_falls_back():
    """A byte the sample never sees (one row out of a million) and duplicates under UniqueIndexOn: the general path takes over
    and reports what it always reports."""
    ctx = Context(0)
    col = dg.column(dg.SEQ_PERM, NROWS, NROWS, encoding=dg.FIXED8, seed=11)
    col.data[8 * 777_777 + 3] = ord("x")          # row 777 777 is not a multiple of the sample's stride
    h, g = both(ctx, col, True)
    assert h.info()["build_path"] == 0 and h.status == N.CPH_OK
    np.testing.assert_array_equal(h.perm(), g.perm())
    h.close(); g.close()
    col = dg.column(dg.SEQ_PERM, NROWS, NROWS, encoding=dg.FIXED8, seed=12)
    col.data[8 * 500_001: 8 * 500_002] = col.data[8 * 17: 8 * 18]
    h, g = both(ctx, col, True)
    assert h.status == g.status == N.CPH_ERR_DUPLICATE and h.first_dup == g.first_dup is not None
    np.testing.assert_array_equal(h.perm(), g.perm())
    h.close(); g.close(); ctx.close()


def test_host_coded_build_not_taken():
    """Small tables, long keys and several key columns keep the general path."""
    ctx = Context(0)
    small = dg.column(dg.SEQ_PERM, 100_000, 100_000, encoding=dg.FIXED8, seed=1)
    ix = DeviceIndex(ctx, [small], unique=True)
    assert ix.info()["build_path"] == 0
    ix.close()
    keys = dg.varkeys(NROWS)                       # 10-22 bytes
    ix = DeviceIndex(ctx, [keys])
    assert ix.info()["build_path"] == 0
    ix.close()
    a = dg.column(dg.UNIFORM, NROWS, 1000, encoding=dg.ITOA, seed=2)
    b = dg.column(dg.UNIFORM, NROWS, 1000, encoding=dg.ITOA, seed=3)
    ix = DeviceIndex(ctx, [a, b])
    assert ix.info()["build_path"] == 0
    ix.close(); ctx.close()


SPLIT_ROWS = (1 << 22) + 4321


def _varkeys_with(rows: dict):
    """config 3's keys (surname "/" name "#" decimal) with some rows replaced; rows not on the sample's stride (n >> 18 = 16)."""
    col = dg.varkeys(SPLIT_ROWS)
    if not rows:
        return col
    vals = None
    offs = col.offsets.astype(np.int64)
    data = col.data
    pieces, last = [], 0
    for r in sorted(rows):
        pieces.append(data[offs[last]:offs[r]])
        pieces.append(np.frombuffer(rows[r], np.uint8))
        last = r + 1
    pieces.append(data[offs[last]:])
    lens = np.diff(offs)
    for r, v in rows.items():
        lens[r] = len(v)
    new_offs = np.zeros(len(lens) + 1, dtype=np.int64)
    np.cumsum(lens, out=new_offs[1:])
    del vals
    return StrCol.from_arrays(np.concatenate(pieces), new_offs.astype(np.uint32))


def test_host_split_codec_build_equals_general_and_oracle():
    """Round 6: a variable-length key column in host memory that wants the delimiter split is coded by the host (the split codec's
    twin): 4 bytes per row cross PCIe instead of the strings; the index is the general path's and the oracle's."""
    ctx = Context(0)
    ctx.set_option("host_split", 2)   # (1 = by the estimate: a host throttled to a few CPUs uploads the strings)
    col = _varkeys_with({})
    h, g = both(ctx, col, False)
    assert h.info()["build_path"] == 2 and g.info()["build_path"] == 0
    assert h.status == g.status == N.CPH_OK and h.first_dup == g.first_dup
    np.testing.assert_array_equal(h.perm(), g.perm())
    o = orc.OracleIndex([col])
    np.testing.assert_array_equal(h.perm(), o.perm)
    assert h.first_dup == o.first_dup()
    probe = col.slice(5000, 5000 + 20_000)
    m = h.probe([probe], want_pairs=True)
    oj = o.join([probe])
    assert m.nmatches == oj["nmatches"]
    np.testing.assert_array_equal(m.build_row, oj["build_row"])
    m.release()
    # the index built from the device-resident copy of the column: the same sample, the same codec, the same codes
    d = DeviceIndex(ctx, [col.to_device("cuda:0")])
    np.testing.assert_array_equal(h.perm(), d.perm())
    assert h.info()["code_bits"] == d.info()["code_bits"]
    d.close(); h.close(); g.close(); ctx.close()


@pytest.mark.parametrize("case", ["unknown_prefix", "foreign_suffix_byte", "no_delimiter", "long_value", "long_suffix", "switched_off"])
def test_host_split_codec_falls_back(case):
    """One row the sampled split codec cannot code (never on the sample's stride): the strings are uploaded and the general path
    builds what it always builds."""
    ctx = Context(0)
    row = 1_234_567   # odd: not a multiple of 16
    bad = {"unknown_prefix": b"Zzyzx/Qq#123", "foreign_suffix_byte": None, "no_delimiter": b"Smith/Amelia", "long_value": b"S" * 30 + b"/A#1" + b"2" * 9,
           "long_suffix": None, "switched_off": None}[case]
    col = dg.varkeys(SPLIT_ROWS)
    if case in ("foreign_suffix_byte", "long_suffix"):
        v = col.value(row)
        cut = v.index(b"#") + 1
        bad = v[:cut] + (b"12x" if case == "foreign_suffix_byte" else b"1" * 17)
    col = _varkeys_with({row: bad} if bad is not None else {})
    if bad is not None:
        assert col.value(row) == bad and col.value(row + 1) == dg.varkeys(SPLIT_ROWS).value(row + 1)
    ctx.set_option("host_split", 0 if case == "switched_off" else 2)
    h, g = both(ctx, col, False)
    assert h.info()["build_path"] == 0 and h.status == g.status == N.CPH_OK
    np.testing.assert_array_equal(h.perm(), g.perm())
    assert h.first_dup == g.first_dup
    h.close(); g.close(); ctx.close()

#!/usr/bin/env python3
"""Debug aid: replays the malformed-text fuzz of tests/test_csv_ingest.py, logging each input before the call."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np
from csvplus_amd import _native as N
from tests.test_csv_ingest import _gen_wellformed, gpu_records, oracle_records
ctx = N.Context(0)
rng = np.random.default_rng(13)
alphabet = np.frombuffer(b'ab ,"\n\rz', dtype=np.uint8)
junk = np.frombuffer(b'",\n\r a', dtype=np.uint8)
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
        print(it, opts, repr(text), flush=True)
        want = oracle_records(text, opts)
        print("  oracle:", want[1], want[2], len(want[0]), flush=True)
        got = gpu_records(ctx, text, opts)
        if got != want:
            print("  MISMATCH", got[1:], flush=True)

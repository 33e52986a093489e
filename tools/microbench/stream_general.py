#!/usr/bin/env python3
"""Pipelined host-stream Join through the GENERAL chain (cph_stream_join_create_general): the build side has duplicate
keys (two rows per customer id), so every chunk's pair list has to be sized on the host after its probe — one worker
thread per slot.  Orders streamed from pinned host memory in 2^23-row chunks; 1 slot (no overlap) against 2, 3, 4."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np
from csvplus_amd import Context, DeviceIndex, StrCol, datagen as dg
from csvplus_amd.streaming import PinnedCol, StreamJoin

total_rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
chunk = 1 << 23
NC, NP = 1_000_000, 100_000
ctx = Context(0)
ids = dg.column(dg.SEQ_PERM, NC, NC, encoding=dg.FIXED8, seed=dg.SEED + 1)
twice = StrCol.from_arrays(np.concatenate([ids.data, ids.data]), np.arange(2 * NC + 1, dtype=np.uint32) * 8)   # every id twice
ia = DeviceIndex(ctx, [twice])
ib = DeviceIndex(ctx, [dg.column(dg.SEQ_PERM, NP, NP, encoding=dg.ITOA, seed=dg.SEED + 2)], unique=True)
assert ia.first_dup is not None
nbuf = 4
bufs = []
for i in range(nbuf):
    o = dg.orders(10**9, NC, NP, row0=i * chunk, nrows=chunk)
    bufs.append(([PinnedCol(ctx, o["cust_id"]), PinnedCol(ctx, o["prod_id"])],
                 o["cust_id"].nbytes_values() + o["prod_id"].nbytes_values() + o["prod_id"].nbytes_offsets()))
for slots in (1, 2, 3, 4):
    sj = StreamJoin(ctx, [ia, ib], nslots=slots, ncols=[1, 1])
    nchunks = max(slots + 1, total_rows // chunk)
    for warm in (True, False):
        t0 = time.perf_counter()
        sub = done = joined = h2d = 0
        while done < nchunks:
            while sub < nchunks and sj.pending < slots:
                cols, nb = bufs[sub % nbuf]
                sj.submit([c.col for c in cols], probe_base=sub * chunk)
                h2d += nb
                sub += 1
            r = sj.next(copy=False)
            assert not r["dense"]
            joined += r["nmatches"]
            done += 1
        dt = time.perf_counter() - t0
    rows = nchunks * chunk
    d2h = joined * 16
    print(f"slots={slots}: {rows:.3e} stream rows -> {joined:.3e} joined rows in {dt * 1e3:8.1f} ms -> {rows / dt / 1e9:6.2f} G stream rows/s, "
          f"{joined / dt / 1e9:6.2f} G joined rows/s | H2D {h2d / dt / 1e9:5.1f} GB/s  D2H {d2h / dt / 1e9:5.1f} GB/s", flush=True)
    sj.close()

#!/usr/bin/env python3
"""End-to-end rate of the pipelined host-stream Join (BASELINE config 5 on one GPU):
1e7-row customers index + 1e5-row products index, orders streamed from PINNED host memory in
2^24-row chunks through cph_stream_join (3 slots = 3 HIP streams).  PCIe-inclusive."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from csvplus_amd import Context, DeviceIndex, datagen as dg
from csvplus_amd.streaming import PinnedCol, StreamJoin

total_rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
chunk = 1 << 24
NC, NP = 10_000_000, 100_000
ctx = Context(0)
ia = DeviceIndex(ctx, [dg.column(dg.SEQ_PERM, NC, NC, encoding=dg.FIXED8, seed=dg.SEED + 1)], unique=True)
ib = DeviceIndex(ctx, [dg.column(dg.SEQ_PERM, NP, NP, encoding=dg.ITOA, seed=dg.SEED + 2)], unique=True)
nbuf = 4
bufs = []
for i in range(nbuf):
    o = dg.orders(10**9, NC, NP, row0=i * chunk, nrows=chunk)
    bufs.append(([PinnedCol(ctx, o["cust_id"]), PinnedCol(ctx, o["prod_id"])],
                 o["cust_id"].nbytes_values() + o["prod_id"].nbytes_values() + o["prod_id"].nbytes_offsets()))
for zc, slots in [(z, k) for z in (1, 0) for k in (1, 2, 3, 4)]:
    ctx.set_option("stream_role_streams", zc)
    sj = StreamJoin(ctx, [ia, ib], nslots=slots)
    nchunks = max(slots + 1, total_rows // chunk)
    t0 = time.perf_counter()
    sub = done = joined = h2d = 0
    while done < nchunks:
        while sub < nchunks and sj.pending < slots:
            cols, nb = bufs[sub % nbuf]
            sj.submit([c.col for c in cols], probe_base=sub * chunk)
            h2d += nb
            sub += 1
        r = sj.next(copy=False)
        joined += r["nmatches"]
        done += 1
    dt = time.perf_counter() - t0
    rows = nchunks * chunk
    d2h = rows * 8 + rows // 8
    print(f"role_streams={zc} slots={slots}: {rows:.3e} rows in {dt * 1e3:8.1f} ms -> {rows / dt / 1e9:6.2f} G rows/s | "
          f"H2D {h2d / dt / 1e9:5.1f} GB/s  D2H {d2h / dt / 1e9:5.1f} GB/s | joined {joined:.3e}", flush=True)
    sj.close()

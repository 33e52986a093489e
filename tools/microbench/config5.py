#!/usr/bin/env python3
"""BASELINE config 5 on ONE GPU: a 1e9-row orders stream against a 1e7-row customers index (+ the 1e5-row products
index), every chunk DISTINCT (2^24 rows each, generated straight into pinned host memory), streamed through
cph_stream_join_* with H2D / kernel / D2H of consecutive chunks overlapped.  PCIe-inclusive end-to-end rate; every
chunk's match count is checked and a row sample of every 8th chunk is verified key by key (verify.check_join_sample).
The 8-GPU version of this config shards the chunks over ranks; only one GPU is reachable here."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from csvplus_amd import Context, DeviceIndex, datagen as dg, verify as V
from csvplus_amd.streaming import PinnedCol, StreamJoin

total_rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 2
chunk = 1 << 24
NC, NP = 10_000_000, 100_000
ctx = Context(0)
cust = dg.column(dg.SEQ_PERM, NC, NC, encoding=dg.FIXED8, seed=dg.SEED + 1)
prod = dg.column(dg.SEQ_PERM, NP, NP, encoding=dg.ITOA, seed=dg.SEED + 2)
ia, ib = DeviceIndex.build_many(ctx, [([cust], True), ([prod], True)])
bounds = [(b, min(b + chunk, total_rows)) for b in range(0, total_rows, chunk)]
t0 = time.perf_counter()
host, pins = [], []
for b, e in bounds:
    o = dg.orders(total_rows, NC, NP, row0=b, nrows=e - b)
    host.append(o)
    pins.append([PinnedCol(ctx, o["cust_id"]), PinnedCol(ctx, o["prod_id"])])
print(f"generated + pinned {total_rows:.3e} rows in {len(bounds)} chunks: {time.perf_counter() - t0:.1f} s", flush=True)
h2d = sum(o["cust_id"].nbytes_values() + o["prod_id"].nbytes_values() + o["prod_id"].nbytes_offsets() for o in host)
nslots = int(sys.argv[3]) if len(sys.argv) > 3 else 4   # an EVEN number of slot streams: odd counts measured 25% slower
sj = StreamJoin(ctx, [ia, ib], nslots=nslots)             # `slots` chunks in flight, the others being read (round-robin lifetime)
ctx.synchronize()
t0 = time.perf_counter()
sub = done = joined = bad = checked = 0
samples = []
while done < len(bounds):
    while sub < len(bounds) and sj.pending < slots:
        sj.submit([c.col for c in pins[sub]], probe_base=bounds[sub][0])
        sub += 1
    r = sj.next(copy=False)
    n = bounds[done][1] - bounds[done][0]
    assert r["nrows"] == n and r["probe_base"] == bounds[done][0]
    joined += r["nmatches"]
    if done % 8 == 0:   # keep a row sample of the pinned result arrays (valid until chunk done + nslots is submitted)
        rows = V.sample_rows(n, 2000, seed=done)
        samples.append((done, rows, r["build_row"][0][rows], r["build_row"][1][rows]))
    done += 1
dt = time.perf_counter() - t0
for k, rows, b0, b1 in samples:   # key-by-key check of the sampled rows, outside the timed region
    bad += V.check_join_sample(host[k]["cust_id"], cust, b0, rows)
    bad += V.check_join_sample(host[k]["prod_id"], prod, b1, rows)
    checked += 2 * len(rows)
d2h = total_rows * 8 + total_rows // 8
print(f"config 5, 1 GPU, in flight={slots} of {nslots} slots: {total_rows:.3e} rows in {dt * 1e3:9.1f} ms -> {total_rows / dt / 1e9:6.2f} G rows/s "
      f"(PCIe inclusive) | H2D {h2d / dt / 1e9:5.1f} GB/s  D2H {d2h / dt / 1e9:5.1f} GB/s | "
      f"joined {joined} of {total_rows} | sampled key checks: {checked} rows, {bad} mismatches", flush=True)
assert joined == total_rows and bad == 0

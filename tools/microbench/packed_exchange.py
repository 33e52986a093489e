#!/usr/bin/env python3
"""What CPH_DIST_PACKED costs on the device: two thread-ranks on ONE GPU (loopback transport: the "link" is a device copy at HBM
speed, so the packed format cannot win here — the difference to the plain format is the pack + unpack kernels' time, and that is
what a real link has to save).  usage: tools/microbench/packed_exchange.py [rows per rank] [chunks]"""
import sys, threading, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np
from csvplus_amd import Context, DeviceIndex, _native as N, datagen as dg

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 0
world, nc, npd = 2, 10_000_000, 100_000
cust, prod = dg.customers(nc)["id"], dg.products(npd)["prod_id"]
o = dg.orders(rows * world, nc, npd)
res, lock = {}, threading.Lock()


def body(r):
    ctx = Context(0)
    d = N.Dist.loopback(ctx, "packed-mb", r, world)
    ia, ib = DeviceIndex(ctx, [cust.to_device("cuda:0")], unique=True), DeviceIndex(ctx, [prod.to_device("cuda:0")], unique=True)
    b = r * rows
    ca, cb = o["cust_id"].slice(b, b + rows).to_device("cuda:0"), o["prod_id"].slice(b, b + rows).to_device("cuda:0")
    for packed in (False, True, False, True):
        ts, st = [], None
        for rep in range(6):
            t0 = time.perf_counter()
            g = d.join_chain([(ia, [ca]), (ib, [cb])], probe_base=b, shard_rows=[rows] * world, nchunks=chunks, positions=True, packed=packed)
            ctx.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
            st = g.stats
            g.release()
        with lock:
            res.setdefault(packed, []).append((r, round(min(ts[1:]), 3), st))
    d.close()
    ctx.close()


th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
[t.start() for t in th]
[t.join() for t in th]
for packed, v in res.items():
    for r, ms, st in sorted(v, key=lambda x: x[0]):
        print("packed" if packed else "plain ", "rank", r, "ms", ms, "chunks", st["chunks"], "compute_ms", round(st["compute_ms"], 3), "exchange_ms",
              round(st["exchange_ms"], 3), "bytes_sent", st["bytes_sent"], "bits", st["packed_bits"], flush=True)

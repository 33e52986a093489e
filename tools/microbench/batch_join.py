#!/usr/bin/env python3
"""What one Join of a BATCH costs through the C ABI from host memory (the cgo shim's unit of work: joinBatch = 8192 rows, key
columns in host memory, bounds / positions back in host memory), and what page-locking a result block of that size costs by
itself (cph_pinned_alloc + cph_pinned_free) — round 5 takes the result blocks of cph_join_probe / cph_join_chain[_ex] out of
the ctx's cache of pinned blocks instead of allocating one per call."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import ctypes as C
import numpy as np
from csvplus_amd import Context, DeviceIndex, StrCol, _native as N, datagen as dg, join_chain

ctx = Context(0)
cust = dg.customers(100_000)
prod = dg.products(1000)
ia = DeviceIndex(ctx, [cust["id"]], unique=True)
ib = DeviceIndex(ctx, [prod["prod_id"]], unique=True)
for batch in (1, 64, 8192, 65536):
    o = dg.orders(batch, 100_000, 1000)
    def probe():
        m = ia.probe([o["cust_id"]], want_pairs=False)
        m.release()
    def chain():
        ch = join_chain(ctx, [(ia, [o["cust_id"]]), (ib, [o["prod_id"]])], positions=True)
        ch.release()
    def pin():
        p = C.c_void_p()
        ctx._check(ctx.lib.cph_pinned_alloc(ctx.handle, 16 * batch + 64, C.byref(p)))
        ctx.lib.cph_pinned_free(ctx.handle, p)
    for name, fn in (("cph_join_probe (bounds)", probe), ("cph_join_chain_ex (2 steps, positions)", chain), ("cph_pinned_alloc + free of the result's size", pin)):
        for _ in range(5):
            fn()
        reps = 300 if batch <= 8192 else 50
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        dt = (time.perf_counter() - t0) / reps
        print(f"batch {batch:6d}  {name:48s} {dt * 1e6:9.1f} us per call", flush=True)

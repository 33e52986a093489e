#!/usr/bin/env python3
"""Throughput of cph_csv_parse on device-resident text: the orders file (cust_id,prod_id,qty + header)
written by cph_csv_write, parsed back into 2 key columns / all 3 columns; plus a quoted variant."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import ctypes as C
import torch
from csvplus_amd import _native as N, datagen as dg, ingest
from csvplus_amd.engine import Engine

M = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
eng = Engine(0); ctx = eng.ctx; dev = eng.device
ords = dg.orders(M, 10_000_000, 100_000)
names = ["cust_id", "prod_id", "qty"]
cols = [ords[n].to_device(dev) for n in names]
arr = (N.cph_strcol * 3)(); keep = []
for i, c in enumerate(cols):
    sc, k = c.as_c(); arr[i] = sc; keep.append(k)
hv = (N.cph_strval * 3)(); hk = []
for i, h in enumerate(names):
    b = (C.c_uint8 * len(h)).from_buffer_copy(h.encode()); hk.append(b)
    hv[i].data = C.cast(b, C.c_void_p).value; hv[i].len = len(h)
out = C.POINTER(N.cph_bytes)()
ctx._check(ctx.lib.cph_csv_write(ctx.handle, arr, 3, hv, N.CPH_MEM_DEVICE, C.byref(out)))
size = int(out.contents.size); ptr = int(out.contents.data)
print(f"orders.csv: {M} rows, {size / 1e9:.3f} GB on device", flush=True)


def timed(label, fn, reps=3):
    r = fn(); r.release()
    ctx.profile(True); ctx.profile_read(reset=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        r = fn(); assert r.error_kind == 0 and r.nrecords == M, (r.error_kind, r.error_record, r.nrecords); r.release()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    p = ctx.profile_read(reset=True); ctx.profile(False)
    ks = ", ".join(f"{k}={v['total_ms'] / reps:.3f}" for k, v in sorted(p.items(), key=lambda kv: -kv[1]['total_ms'])[:6])
    print(f"{label:40s} wall {dt * 1e3:8.2f} ms  {size / dt / 1e9:7.1f} GB/s  {M / dt / 1e9:6.2f} G rows/s | {ks}", flush=True)


timed("parse -> cust_id, prod_id (device out)", lambda: ingest.csv_parse(ctx, None, [0, 1], fields_per_record=3, skip_records=1,
                                                                        out_mem=N.CPH_MEM_DEVICE, device_ptr=ptr, size=size))
timed("parse -> all 3 columns (device out)", lambda: ingest.csv_parse(ctx, None, [0, 1, 2], fields_per_record=3, skip_records=1,
                                                                     out_mem=N.CPH_MEM_DEVICE, device_ptr=ptr, size=size))

import os
if os.environ.get("CSV_DBG"):   # attribution of the fast path's two kernels (results are wrong with a bit set): csv_ingest.hip `dbg`
    for dbg in (0, 1, 2, 4, 8, 2 | 4 | 8):
        ctx.set_option("chain_debug", dbg << 16)
        ctx.profile(True); ctx.profile_read(reset=True)
        for _ in range(3):
            try:
                ingest.csv_parse(ctx, None, [0, 1], fields_per_record=-1, skip_records=1, out_mem=N.CPH_MEM_DEVICE, device_ptr=ptr, size=size).release()
            except Exception as ex:
                print("dbg", dbg, type(ex).__name__); break
        p = ctx.profile_read(reset=True); ctx.profile(False)
        print("dbg", dbg, " ".join(f"{k}={v['total_ms'] / max(1, v['launches']):.3f}" for k, v in p.items() if k.startswith("k_csv")), flush=True)
    ctx.set_option("chain_debug", 0)

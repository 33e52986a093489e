#!/usr/bin/env python3
"""Throughput of the materialisation kernels on device-resident data: gather of string columns through
joined row ids, and ToCsv of the joined table (1e8 orders JOIN customers JOIN products)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from csvplus_amd import _native as N, datagen as dg
from csvplus_amd.engine import Engine
from csvplus_amd.materialize import gather_rows
import ctypes as C

M = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
NC, NP = 10_000_000, 100_000
eng = Engine(0); ctx = eng.ctx; dev = eng.device
cust = dg.customers(NC); prod = dg.products(NP)
ords = dg.orders(M, NC, NP)
d = {k: v.to_device(dev) for k, v in {"cid": cust["id"], "name": cust["name"], "surname": cust["surname"], "pid": prod["prod_id"],
                                       "product": prod["product"], "price": prod["price"], "o_cid": ords["cust_id"],
                                       "o_pid": ords["prod_id"], "o_qty": ords["qty"]}.items()}
ia = eng.index_on([d["cid"]], unique=True); ib = eng.index_on([d["pid"]], unique=True)
res = eng.chained_join([(ia, d["o_cid"]), (ib, d["o_pid"])])
a_ptr = (res.build_rows[0].data_ptr(), 32, res.n); b_ptr = (res.build_rows[1].data_ptr(), 32, res.n)


def timed(label, fn, reps=3):
    r = fn(); 
    eng.ctx.profile(True); eng.ctx.profile_read(reset=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs = [fn() for _ in range(reps)]
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    p = eng.ctx.profile_read(reset=True); eng.ctx.profile(False)
    ks = ", ".join(f"{k}={v['total_ms'] / reps:.3f}" for k, v in sorted(p.items(), key=lambda kv: -kv[1]['total_ms'])[:5])
    print(f"{label:44s} wall {dt * 1e3:8.3f} ms | {ks}", flush=True)
    for o in outs[:-1]:
        if hasattr(o, "release"): o.release()
    if hasattr(r, "release"): r.release()
    return outs[-1]


g_name = timed("gather customers.name via build_row (1e8)", lambda: gather_rows(ctx, d["name"], a_ptr, out_mem=N.CPH_MEM_DEVICE))
g_sur = timed("gather customers.surname (1e8)", lambda: gather_rows(ctx, d["surname"], a_ptr, out_mem=N.CPH_MEM_DEVICE))
g_prod = timed("gather products.product (1e8)", lambda: gather_rows(ctx, d["product"], b_ptr, out_mem=N.CPH_MEM_DEVICE))
g_price = timed("gather products.price (1e8)", lambda: gather_rows(ctx, d["price"], b_ptr, out_mem=N.CPH_MEM_DEVICE))
cols = [d["o_cid"], d["o_qty"], g_name.as_device_strcol(), g_sur.as_device_strcol(), g_prod.as_device_strcol(), g_price.as_device_strcol()]
arr = (N.cph_strcol * len(cols))(); keep = []
for i, c in enumerate(cols):
    sc, k = c.as_c(); arr[i] = sc; keep.append(k)


def csv():
    out = C.POINTER(N.cph_bytes)()
    ctx._check(ctx.lib.cph_csv_write(ctx.handle, arr, len(cols), None, N.CPH_MEM_DEVICE, C.byref(out)))
    size = int(out.contents.size); ctx.lib.cph_bytes_release(out); return size


size = csv()
timed(f"ToCsv 6 columns x 1e8 rows ({size / 1e9:.2f} GB out)", csv)

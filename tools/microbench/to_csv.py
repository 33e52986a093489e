#!/usr/bin/env python3
"""ToCsv of the README chain's joined rows (orders JOIN customers JOIN products, 6 output columns) on device-resident columns:
the one-pass writer (materialize.hip: k_csv_onepass) against the two-pass writer, and its attribution switches."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from csvplus_amd import _native as N, datagen as dg
from csvplus_amd.engine import Engine
from csvplus_amd.materialize import csv_write, permute_col

M = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
MODES = sys.argv[2].split(",") if len(sys.argv) > 2 else ["0:0", "1:0", "1:1", "1:2", "1:3", "1:4", "1:5"]
NC, NP = 10_000_000, 100_000
eng = Engine(0); ctx = eng.ctx; dev = eng.device
cust = dg.customers(NC); prod = dg.products(NP); ords = dg.orders(M, NC, NP)
d = {k: v.to_device(dev) for k, v in {"cid": cust["id"], "name": cust["name"], "surname": cust["surname"], "pid": prod["prod_id"],
                                       "product": prod["product"], "price": prod["price"], "o_cid": ords["cust_id"],
                                       "o_pid": ords["prod_id"], "o_qty": ords["qty"]}.items()}
ia = N.DeviceIndex(ctx, [d["cid"]], unique=True); ib = N.DeviceIndex(ctx, [d["pid"]], unique=True)
ch = N.join_chain(ctx, [(ia, [d["o_cid"]]), (ib, [d["o_pid"]])], out_mem=N.CPH_MEM_DEVICE, positions=True)
ptrs = ch.device_ptrs(); n = ch.nrows
keep = [permute_col(ctx, ia, d["name"]), permute_col(ctx, ia, d["surname"]), permute_col(ctx, ib, d["product"]), permute_col(ctx, ib, d["price"])]
cols = [d["o_cid"], d["o_qty"]] + [k.as_device_strcol() for k in keep]
ids = [None, None] + [(ptrs["build_row"][0], 32, n)] * 2 + [(ptrs["build_row"][1], 32, n)] * 2
names = ["cust_id", "qty", "name", "surname", "product", "price"]
for mode in MODES:
    op, dbg = (int(x) for x in mode.split(":"))
    ctx.set_option("csv_onepass", op); ctx.set_option("csv_onepass_debug", dbg)
    t = csv_write(ctx, cols, names, out_mem=N.CPH_MEM_DEVICE, row_ids=ids, nrows=n); size = len(t); t.release()
    ctx.profile(True); ctx.profile_read(reset=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        csv_write(ctx, cols, names, out_mem=N.CPH_MEM_DEVICE, row_ids=ids, nrows=n).release()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    p = ctx.profile_read(reset=True); ctx.profile(False)
    ks = ", ".join(f"{k}={v['total_ms'] / 3:.3f}" for k, v in sorted(p.items(), key=lambda kv: -kv[1]['total_ms'])[:6])
    print(f"csv_onepass={op} debug={dbg}: wall {dt * 1e3:7.3f} ms  {size / 1e9:.3f} GB  {size / dt / 1e12:.3f} TB/s | {ks}", flush=True)

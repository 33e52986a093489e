#!/usr/bin/env python3
"""The one-pass ToCsv writer against the two-pass writer on other shapes than the README chain's: stream columns only, 1-8 output
columns after grouping (<= 4: column kinds compiled in; 5-8: read from the arguments), a column gathered from a table larger than the
output (no slot table)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
from csvplus_amd import _native as N, datagen as dg
from csvplus_amd.engine import Engine
from csvplus_amd.materialize import csv_write, permute_col

M = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
NC, NP = 10_000_000, 100_000
eng = Engine(0); ctx = eng.ctx; dev = eng.device
cust = dg.customers(NC); prod = dg.products(NP); ords = dg.orders(M, NC, NP)
d = {k: v.to_device(dev) for k, v in {"cid": cust["id"], "name": cust["name"], "surname": cust["surname"], "pid": prod["prod_id"],
                                       "product": prod["product"], "price": prod["price"], "o_cid": ords["cust_id"],
                                       "o_pid": ords["prod_id"], "o_qty": ords["qty"]}.items()}
ia = N.DeviceIndex(ctx, [d["cid"]], unique=True); ib = N.DeviceIndex(ctx, [d["pid"]], unique=True)
ch = N.join_chain(ctx, [(ia, [d["o_cid"]]), (ib, [d["o_pid"]])], out_mem=N.CPH_MEM_DEVICE, positions=True)
p = ch.device_ptrs(); n = ch.nrows
pay = {k: permute_col(ctx, ix, d[k]) for k, ix in (("name", ia), ("surname", ia), ("product", ib), ("price", ib))}
col = {k: v.as_device_strcol() for k, v in pay.items()}
A, B = (p["build_row"][0], 32, n), (p["build_row"][1], 32, n)
big_ids = torch.arange(0, 2 * n, 2, dtype=torch.int32, device=dev)   # a table twice the output: no slot table
big = dg.orders(2 * M, NC, NP)["qty"].to_device(dev) if M <= 50_000_000 else None
shapes = {
    "3 stream columns (orders.csv)": ([d["o_cid"], d["o_pid"], d["o_qty"]], [None, None, None]),
    "1 stream column": ([d["o_cid"]], [None]),
    "README chain: 2 stream + (name,surname) + (product,price) = 4 groups": ([d["o_cid"], d["o_qty"], col["name"], col["surname"], col["product"], col["price"]], [None, None, A, A, B, B]),
    "5 groups: cust_id, name, qty, surname, (product,price)": ([d["o_cid"], col["name"], d["o_qty"], col["surname"], col["product"], col["price"]], [None, A, None, A, B, B]),
    "6 groups: cust_id, name, qty, surname, product, prod_id, price": ([d["o_cid"], col["name"], d["o_qty"], col["surname"], col["product"], d["o_pid"], col["price"]], [None, A, None, A, B, None, B]),
    "8 groups: every column apart": ([d["o_cid"], col["name"], d["o_qty"], col["surname"], d["o_pid"], col["product"], d["o_cid"], col["price"]], [None, A, None, A, None, B, None, B]),
}
if big is not None:
    shapes["2 stream + one column gathered from a table of 2n rows (no slots)"] = ([d["o_cid"], d["o_qty"], big], [None, None, (big_ids.data_ptr(), 32, n)])
for name, (cols, ids) in shapes.items():
    res = []
    for mode in (0, 1):
        ctx.set_option("csv_onepass", mode)
        t = csv_write(ctx, cols, None, out_mem=N.CPH_MEM_DEVICE, row_ids=ids, nrows=n); size = len(t); t.release()
        ctx.profile(True); ctx.profile_read(reset=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            csv_write(ctx, cols, None, out_mem=N.CPH_MEM_DEVICE, row_ids=ids, nrows=n).release()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        pr = ctx.profile_read(reset=True); ctx.profile(False)
        res.append((dt * 1e3, size, "k_csv_onepass" in pr and "k_csv_copy" not in pr))
    ctx.set_option("csv_onepass", 1)
    print(f"{name:78s} {res[0][1] / 1e9:6.2f} GB  two passes {res[0][0]:7.2f} ms ({res[0][1] / res[0][0] / 1e9:5.2f} TB/s)   one pass {res[1][0]:7.2f} ms ({res[1][1] / res[1][0] / 1e9:5.2f} TB/s){'' if res[1][2] else '  [not taken / fell back]'}", flush=True)

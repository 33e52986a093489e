#!/usr/bin/env python3
"""IndexOn of small tables: wall time per cph_index_build call and the time of the kernels inside it, with the one-launch
path (small_build.hip) and with the general path (ctx option small_build_rows = 0)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from csvplus_amd import DeviceIndex, datagen as dg
from csvplus_amd.engine import Engine
eng = Engine(0); ctx = eng.ctx


def wall(fn, reps=300):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for n in (120, 1000, 4000, 10000, 16384):
    people = dg.customers(n, encoding=dg.ITOA)
    orders = dg.orders(n, 1200, 8, cust_encoding=dg.ITOA)
    cases = {"id (unpadded decimal)": [people["id"].to_device(eng.device)],
             "(cust_id, prod_id)": [orders["cust_id"].to_device(eng.device), orders["prod_id"].to_device(eng.device)]}
    for name, cols in cases.items():
        line = f"{n:>6} rows {name:<24}"
        for label, limit in (("one launch", 16384), ("general", 0)):
            ctx.set_option("small_build_rows", limit)
            w = wall(lambda: DeviceIndex(ctx, cols).close())
            ctx.profile(True); ctx.profile_read(reset=True)
            for _ in range(20):
                DeviceIndex(ctx, cols).close()
            p = ctx.profile_read(reset=True); ctx.profile(False)
            kms = sum(v["total_ms"] for v in p.values()) / 20 * 1e3
            nl = sum(v["launches"] for v in p.values()) / 20
            line += f" | {label}: wall {w:7.1f} us, kernels {kms:7.1f} us in {nl:4.1f} launches"
        print(line, flush=True)
ctx.set_option("small_build_rows", 8192)

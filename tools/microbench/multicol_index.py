#!/usr/bin/env python3
"""IndexOn(cust_id, prod_id) over 1e8 orders (the reference's BenchmarkCreateBiggerMultiIndex shape, scaled): per-kernel times,
then a full two-column Join of the same rows against it."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from csvplus_amd import DeviceIndex, _native as N, datagen as dg
from csvplus_amd.engine import Engine
eng = Engine(0); ctx = eng.ctx
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
for enc, name in ((dg.ITOA, "unpadded ids"), (dg.FIXED8, "fixed8 ids")):
    orders = dg.orders(n, 1_200_000, 8, cust_encoding=enc)
    d_c, d_p = orders["cust_id"].to_device(eng.device), orders["prod_id"].to_device(eng.device)
    DeviceIndex(ctx, [d_c, d_p]).close()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        DeviceIndex(ctx, [d_c, d_p]).close()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 3 * 1e3
    ctx.profile(True); ctx.profile_read(reset=True)
    ix = DeviceIndex(ctx, [d_c, d_p])
    p = ctx.profile_read(reset=True)
    print(f"{name}: IndexOn(cust_id, prod_id) {n} rows wall {wall:.3f} ms | " + " ".join(f"{k.replace('k_', '')}={v['total_ms']:.3f}" for k, v in p.items()), ix.info(), flush=True)
    m = ix.probe([d_c, d_p], out_mem=N.CPH_MEM_DEVICE, want_pairs=False); m.release()
    ctx.profile_read(reset=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m = ix.probe([d_c, d_p], out_mem=N.CPH_MEM_DEVICE, want_pairs=False); m.release()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
    p = ctx.profile_read(reset=True); ctx.profile(False)
    print(f"   probe of the same {n} rows (bounds only) wall {wall:.3f} ms | " + " ".join(f"{k.replace('k_', '')}={v['total_ms']:.3f}" for k, v in p.items()), flush=True)
    ix.close()

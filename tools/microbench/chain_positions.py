#!/usr/bin/env python3
"""The bench's chained Join (1e8 orders x 1e7 customers x 1e5 products) reporting original row ids (default) and sorted
positions (cph_join_chain_ex CPH_CHAIN_POSITIONS), the latter with the small rank table in LDS and in global memory;
kernel times from the library's HIP events, several repetitions inside one process on one box."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from csvplus_amd import _native as N, datagen as dg
from csvplus_amd.engine import Engine
eng = Engine(0); ctx = eng.ctx; dev = eng.device
M, NC, NP = 100_000_000, 10_000_000, 100_000
cust = dg.column(dg.SEQ_PERM, NC, NC, encoding=dg.FIXED8, seed=dg.SEED + 1).to_device(dev)
prod = dg.column(dg.SEQ_PERM, NP, NP, encoding=dg.ITOA, seed=dg.SEED + 2).to_device(dev)
o = dg.orders(M, NC, NP)
oc, op = o["cust_id"].to_device(dev), o["prod_id"].to_device(dev)
ia, ib = eng.index_on_many([[cust], [prod]], unique=True)


def run(label, positions, **opts):
    for k, v in opts.items():
        ctx.set_option(k, v)
    N.join_chain(ctx, [(ia, [oc]), (ib, [op])], out_mem=N.CPH_MEM_DEVICE, positions=positions).release()
    ctx.profile(True); ctx.profile_read(reset=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        N.join_chain(ctx, [(ia, [oc]), (ib, [op])], out_mem=N.CPH_MEM_DEVICE, positions=positions).release()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 5 * 1e3
    p = ctx.profile_read(reset=True); ctx.profile(False)
    print(f"{label:<52} wall {wall:7.3f} ms | " + " ".join(f"{k.replace('k_', '')}={v['total_ms'] / v['launches']:.3f}" for k, v in p.items()), flush=True)


for rep in range(2):
    run("row ids (4-byte row table, 40 MB)", False)
    run("positions, rank tables in global memory / L2", True, chain_rank_lds=0)
    run("positions, products rank table (30 KB) in LDS", True, chain_rank_lds=1)
    run("positions, LDS + non-temporal streams", True, chain_rank_lds=1, chain_nt_streams=1)
    run("row ids, non-temporal streams", False, chain_nt_streams=1)
    ctx.set_option("chain_nt_streams", 0)

#!/usr/bin/env python3
"""Attribution of k_chain_dense in positions mode (ctx option chain_debug: 1 = no lookups, 2 = no encode, 4 = no stores;
results are wrong while a bit is set): per step alone and both together."""
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
for name, steps in (("cust", [(ia, [oc])]), ("prod", [(ib, [op])]), ("cust+prod", [(ia, [oc]), (ib, [op])])):
    for dbg in (0, 1, 2, 3, 4, 5, 7):
        ctx.set_option("chain_debug", dbg)
        N.join_chain(ctx, steps, out_mem=N.CPH_MEM_DEVICE, positions=True).release()
        ctx.profile(True); ctx.profile_read(reset=True)
        for _ in range(3):
            N.join_chain(ctx, steps, out_mem=N.CPH_MEM_DEVICE, positions=True).release()
        p = ctx.profile_read(reset=True); ctx.profile(False)
        print(f"{name:<10} dbg={dbg} k_chain_dense {p['k_chain_dense']['total_ms'] / 3:.3f} ms", flush=True)
ctx.set_option("chain_debug", 0)

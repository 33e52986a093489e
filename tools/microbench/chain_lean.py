#!/usr/bin/env python3
"""Round 4: the bench's chained Join reporting positions with the lean-step paths switched on and off (ctx options
chain_arith / chain_identity: arithmetic encode + one aligned load for the fixed-width 8-byte customer ids, no lookup for an
index that fills its code space), per step alone and both together; kernel times from the library's HIP events, one process."""
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
CHAINS = {"cust": [(ia, [oc])], "prod": [(ib, [op])], "cust+prod": [(ia, [oc]), (ib, [op])]}


def run(label, chain, positions=True, **opts):
    for k, v in opts.items():
        ctx.set_option(k, v)
    steps = CHAINS[chain]
    N.join_chain(ctx, steps, out_mem=N.CPH_MEM_DEVICE, positions=positions).release()
    ctx.profile(True); ctx.profile_read(reset=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        N.join_chain(ctx, steps, out_mem=N.CPH_MEM_DEVICE, positions=positions).release()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 5 * 1e3
    p = ctx.profile_read(reset=True); ctx.profile(False)
    print(f"{chain:<10} {label:<46} wall {wall:7.3f} ms | k_chain_dense {p['k_chain_dense']['total_ms'] / p['k_chain_dense']['launches']:.3f}", flush=True)


for rep in range(2):
    for chain in ("cust+prod", "cust", "prod"):
        run("round-3 kernel (LUT walk, rank tables)", chain, chain_arith=0, chain_identity=0, chain_nt_streams=0)
        run("lean step, rank table lookup", chain, chain_arith=1, chain_identity=0)
        run("lean step, identity (no lookup)", chain, chain_arith=1, chain_identity=1)
        run("lean + identity + non-temporal streams", chain, chain_nt_streams=1)
        ctx.set_option("chain_nt_streams", 0)
    run("row ids (unchanged kernel)", "cust+prod", positions=False)

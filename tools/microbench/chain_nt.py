#!/usr/bin/env python3
"""Does marking the streamed loads / result stores of the fused chain kernel non-temporal keep more of the
direct-address table in L2 / MALL?  CPH_CHAIN_DEBUG=16: the debug instantiation with nothing switched off
(baseline for that instantiation), 8: non-temporal stream accesses."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from csvplus_amd import datagen as dg
from csvplus_amd.engine import Engine

eng = Engine(0); dev = eng.device
M, NC, NP = 100_000_000, 10_000_000, 100_000
cust = dg.column(dg.SEQ_PERM, NC, NC, encoding=dg.FIXED8, seed=dg.SEED + 1).to_device(dev)
prod = dg.column(dg.SEQ_PERM, NP, NP, encoding=dg.ITOA, seed=dg.SEED + 2).to_device(dev)
o = dg.orders(M, NC, NP)
oc, op = o["cust_id"].to_device(dev), o["prod_id"].to_device(dev)
ia = eng.index_on([cust], unique=True); ib = eng.index_on([prod], unique=True)
steps = [(ia, oc), (ib, op)]
for rep in range(2):
    for dbg in (0, 16, 8):
        os.environ["CPH_CHAIN_DEBUG"] = str(dbg)
        eng.chained_join(steps).release()
        eng.ctx.profile(True); eng.ctx.profile_read(reset=True)
        for _ in range(5):
            eng.chained_join(steps).release()
        p = eng.ctx.profile_read(reset=True); eng.ctx.profile(False)
        print(f"dbg={dbg:2d} k_chain_dense {p['k_chain_dense']['total_ms'] / 5:.3f} ms", flush=True)
os.environ["CPH_CHAIN_DEBUG"] = "0"

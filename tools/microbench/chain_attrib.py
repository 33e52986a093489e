#!/usr/bin/env python3
"""Attribution of the fused chain kernel's time: ctx.set_debug switches off the table lookup (1),
the encode (2) and the output stores (4).  The flag is per ctx."""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from csvplus_amd import datagen as dg
from csvplus_amd.engine import Engine

eng = Engine(0); dev = eng.device
M, NC, NP = 100_000_000, 10_000_000, 100_000
cust = dg.column(dg.SEQ_PERM, NC, NC, encoding=dg.FIXED8, seed=dg.SEED + 1).to_device(dev)
prod = dg.column(dg.SEQ_PERM, NP, NP, encoding=dg.ITOA, seed=dg.SEED + 2).to_device(dev)
o = dg.orders(M, NC, NP)
oc, op = o["cust_id"].to_device(dev), o["prod_id"].to_device(dev)
ia = eng.index_on([cust], unique=True); ib = eng.index_on([prod], unique=True)
for name, steps in (("cust", [(ia, oc)]), ("prod", [(ib, op)]), ("cust+prod", [(ia, oc), (ib, op)])):
    for dbg in (0, 1, 2, 3, 4, 5, 7):
        eng.ctx.set_debug(dbg)
        eng.chained_join(steps).release()
        eng.ctx.profile(True); eng.ctx.profile_read(reset=True)
        for _ in range(3):
            eng.chained_join(steps).release()
        p = eng.ctx.profile_read(reset=True); eng.ctx.profile(False)
        print(f"{name:10s} dbg={dbg} (skip{' lookup' if dbg & 1 else ''}{' encode' if dbg & 2 else ''}{' stores' if dbg & 4 else ''}) "
              f"k_chain_dense {p['k_chain_dense']['total_ms'] / 3:.3f} ms", flush=True)
eng.ctx.set_debug(0)

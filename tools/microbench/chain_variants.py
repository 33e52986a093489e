#!/usr/bin/env python3
"""Times single pieces of the hot path on device-resident columns (HIP-event kernel profile + wall)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from csvplus_amd import _native as N, datagen as dg
from csvplus_amd.engine import Engine

eng = Engine(0)
dev = eng.device


def timed(label, fn, reps=3):
    fn()
    eng.ctx.profile(True); eng.ctx.profile_read(reset=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    prof = eng.ctx.profile_read(reset=True); eng.ctx.profile(False)
    ks = ", ".join(f"{k}={v['total_ms'] / reps:.3f}" for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['total_ms'])[:6])
    print(f"{label:50s} wall {dt * 1e3:8.3f} ms | {ks}", flush=True)
    return r


M = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
NC, NP = 10_000_000, 100_000
cust = dg.column(dg.SEQ_PERM, NC, NC, encoding=dg.FIXED8, seed=dg.SEED + 1).to_device(dev)
prod = dg.column(dg.SEQ_PERM, NP, NP, encoding=dg.ITOA, seed=dg.SEED + 2).to_device(dev)
o = dg.orders(M, NC, NP)
oc, op = o["cust_id"].to_device(dev), o["prod_id"].to_device(dev)
ia = eng.index_on([cust], unique=True); ib = eng.index_on([prod], unique=True)
print("cust", ia.info()); print("prod", ib.info())


def chain(steps):
    def f():
        r = eng.chained_join(steps); n = r.n; r.release(); return n
    return f


timed("chain S=1 cust (80MB table)", chain([(ia, oc)]))
timed("chain S=1 prod (1.2MB table)", chain([(ib, op)]))
timed("chain S=2 cust+prod", chain([(ia, oc), (ib, op)]))
timed("generic probe cust lo/cnt only", lambda: eng.join(ia, [oc], want_pairs=False).release())
timed("generic probe cust + expand", lambda: eng.join(ia, [oc]).release())
timed("IndexOn cust 1e7 unique", lambda: eng.index_on([cust], unique=True).close())
del o
big = dg.column(dg.SEQ_PERM, M, M, encoding=dg.FIXED8, seed=7).to_device(dev)
ix = timed(f"IndexOn {M} unique fixed8 ids", lambda: eng.index_on([big], unique=True), reps=2)
print(ix.info()); ix.close(); del big
vk = dg.varkeys(M).to_device(dev)
ix = timed(f"IndexOn {M} varkeys (config 3)", lambda: eng.index_on([vk]), reps=2)
print(ix.info())

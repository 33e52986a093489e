#!/usr/bin/env python3
"""Generic probe path (cph_join_probe: k_probe + scan + k_expand) at scale, incl. a duplicate build side
(TestLongChain shape: IndexOn(orders.cust_id) probed with customers.id)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from csvplus_amd import datagen as dg
from csvplus_amd.engine import Engine

eng = Engine(0); dev = eng.device


def timed(label, fn, reps=3):
    fn()
    eng.ctx.profile(True); eng.ctx.profile_read(reset=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    p = eng.ctx.profile_read(reset=True); eng.ctx.profile(False)
    ks = ", ".join(f"{k}={v['total_ms'] / reps:.3f}" for k, v in sorted(p.items(), key=lambda kv: -kv[1]['total_ms'])[:7])
    print(f"{label:52s} wall {dt * 1e3:8.3f} ms | {ks}", flush=True)


M, NC = 100_000_000, 10_000_000
cust = dg.column(dg.SEQ_PERM, NC, NC, encoding=dg.FIXED8, seed=dg.SEED + 1).to_device(dev)
oc = dg.column(dg.UNIFORM, M, NC, encoding=dg.FIXED8, seed=dg.SEED + 3).to_device(dev)
ia = eng.index_on([cust], unique=True)
timed("unique build, probe 1e8 bounds only", lambda: eng.join(ia, [oc], want_pairs=False).release())
timed("unique build, probe 1e8 + expand", lambda: eng.join(ia, [oc]).release())
timed("IndexOn(orders.cust_id) 1e8 rows, ~10 dups/key", lambda: eng.index_on([oc]).close(), reps=2)
io = eng.index_on([oc])
print(io.info())
timed("dup build (1e8), probe customers 1e7 -> 1e8 pairs", lambda: eng.join(io, [cust]).release())
timed("dup build (1e8), probe orders 1e8 bounds only", lambda: eng.join(io, [oc], want_pairs=False).release())

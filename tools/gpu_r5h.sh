#!/bin/bash
# round 5, run h: why the one-rank N > 1 path reports a 29 ms n1 step at 3e5 rows; then the whole GPU suite (no -x)
mkdir -p gpurun_out/r5h
python - <<'PY' 2>&1 | tee gpurun_out/r5h/n1_debug.txt
import os, time, torch
from csvplus_amd import _native as N, datagen as dg
from csvplus_amd.dist import connect
from csvplus_amd.engine import Engine
dev = torch.device("cuda", 0)
eng = Engine(0)
cdist = connect(eng.ctx)
rows, ncust, nprod = 300000, 20000, 700
cust_id = dg.column(dg.SEQ_PERM, ncust, ncust, encoding=dg.FIXED8, seed=dg.SEED + 1)
prod_id = dg.column(dg.SEQ_PERM, nprod, nprod, encoding=dg.ITOA, seed=dg.SEED + 2)
ords = dg.orders(rows, ncust, nprod)
d_cust, d_prod = cust_id.to_device(dev), prod_id.to_device(dev)
d_ord = {k: v.to_device(dev) for k, v in ords.items()}
torch.cuda.synchronize()
def t(f, name, reps=4):
    for r in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize()
        print(f"{name} rep {r}: {1e3*(time.perf_counter()-t0):.3f} ms", flush=True)
def build():
    a, b = eng.index_on_many([[d_cust], [d_prod]], unique=True); a.close(); b.close()
def dstep():
    a, b = eng.index_on_many([[d_cust], [d_prod]], unique=True)
    g = cdist.join_chain([(a, [d_ord["cust_id"]]), (b, [d_ord["prod_id"]])], probe_base=0, shard_rows=[rows], nchunks=3, positions=True, host=False)
    g.release(); a.close(); b.close()
def step1():
    a, b = eng.index_on_many([[d_cust], [d_prod]], unique=True)
    t0 = time.perf_counter()
    c = N.join_chain(eng.ctx, [(a, [d_ord["cust_id"]]), (b, [d_ord["prod_id"]])], out_mem=N.CPH_MEM_DEVICE, positions=True)
    t1 = time.perf_counter()
    c.release(); a.close(); b.close()
    print(f"    join_chain {1e3*(t1-t0):.3f} ms nrows {c.nrows if hasattr(c,'nrows') else None}")
t(build, "build")
t(step1, "step1 (before any dist step)")
t(dstep, "dist step")
t(step1, "step1 (after dist steps)")
t(dstep, "dist step")
t(build, "build")
PY
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r5h/pytest.txt
tail -8 gpurun_out/r5h/pytest.txt
true

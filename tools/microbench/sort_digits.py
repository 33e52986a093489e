#!/usr/bin/env python3
"""IndexOn kernel times with 8- and 9-bit radix digits and 256- / 512-thread sort workgroups (ctx options sort_rbits,
sort_threads): 27-bit codes (1e8 decimal ids) and 25-bit codes (config 3 through the split codec) take 4 passes of 8 bits or
3 of 9."""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from csvplus_amd import datagen as dg
from csvplus_amd.engine import Engine

eng = Engine(0)
eng.ctx.set_option("direct_sort", 0)   # the radix path is what is measured here (UniqueIndexOn over dense ids would take one scatter)
cols = {"1e8 fixed8": (dg.column(dg.SEQ_PERM, 100_000_000, 100_000_000, encoding=dg.FIXED8, seed=7).to_device(eng.device), True),
        "1e7 fixed8": (dg.column(dg.SEQ_PERM, 10_000_000, 10_000_000, encoding=dg.FIXED8, seed=7).to_device(eng.device), True),
        "1e8 varkeys": (dg.varkeys(100_000_000).to_device(eng.device), False)}
for rep in range(1):
    for rbits, threads in ((0, 0), (8, 256), (9, 256), (9, 512), (8, 512)):
        eng.ctx.set_option("sort_rbits", rbits)
        eng.ctx.set_option("sort_threads", threads)
        for name, (col, unique) in cols.items():
            eng.index_on([col], unique=unique).close()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                ix = eng.index_on([col], unique=unique); inf = ix.info(); ix.close()
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / 3 * 1e3
            eng.ctx.profile(True); eng.ctx.profile_read(reset=True)
            eng.index_on([col], unique=unique).close()
            p = eng.ctx.profile_read(reset=True); eng.ctx.profile(False)
            hist = sum(v["total_ms"] for k, v in p.items() if "hist" in k)
            scat = sum(v["total_ms"] for k, v in p.items() if "scatter" in k)
            scan = sum(v["total_ms"] for k, v in p.items() if "scan" in k)
            print(f"rbits={rbits} threads={threads} {name:<12} wall {wall:7.3f} ms  passes {inf['sort_passes']} hist {hist:6.3f}  scan {scan:6.3f} scatter {scat:6.3f}  all kernels {sum(v['total_ms'] for v in p.values()):7.3f}", flush=True)

#!/usr/bin/env python3
"""IndexOn kernel times with / without the scatter's digit stream (ctx option sort_digit_stream)."""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from csvplus_amd import datagen as dg
from csvplus_amd.engine import Engine

eng = Engine(0)
cols = {"1e8 fixed8": (dg.column(dg.SEQ_PERM, 100_000_000, 100_000_000, encoding=dg.FIXED8, seed=7).to_device(eng.device), True),
        "1e7 fixed8": (dg.column(dg.SEQ_PERM, 10_000_000, 10_000_000, encoding=dg.FIXED8, seed=7).to_device(eng.device), True),
        "1e8 varkeys": (dg.varkeys(100_000_000).to_device(eng.device), False)}
for rep in range(2):
    for stream in (0, 1):
        eng.ctx.set_option("sort_digit_stream", stream)
        for name, (col, unique) in cols.items():
            eng.index_on([col], unique=unique).close()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                eng.index_on([col], unique=unique).close()
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / 3 * 1e3
            eng.ctx.profile(True); eng.ctx.profile_read(reset=True)
            eng.index_on([col], unique=unique).close()
            p = eng.ctx.profile_read(reset=True); eng.ctx.profile(False)
            hist = sum(v["total_ms"] for k, v in p.items() if "hist" in k)
            scat = sum(v["total_ms"] for k, v in p.items() if "scatter" in k)
            print(f"sort_digit_stream={stream} {name:<12} wall {wall:7.3f} ms  hist {hist:6.3f}  scatter {scat:6.3f}  all kernels {sum(v['total_ms'] for v in p.values()):7.3f}", flush=True)

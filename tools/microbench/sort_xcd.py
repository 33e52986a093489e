#!/usr/bin/env python3
"""Radix scatter with / without the XCD-contiguous tile mapping (ctx option sort_xcd_tiles), same process, same box."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from csvplus_amd import datagen as dg
from csvplus_amd.engine import Engine

eng = Engine(0)
cols = {"1e7 fixed8": dg.column(dg.SEQ_PERM, 10**7, 10**7, encoding=dg.FIXED8, seed=7).to_device(eng.device),
        "1e8 fixed8": dg.column(dg.SEQ_PERM, 10**8, 10**8, encoding=dg.FIXED8, seed=7).to_device(eng.device),
        "1e8 varkeys": dg.varkeys(10**8).to_device(eng.device)}
for rep in range(2):
    for mode in (0, 1):
        eng.ctx.set_option("sort_xcd_tiles", mode)
        for name, col in cols.items():
            eng.index_on([col]).close()
            eng.ctx.profile(True); eng.ctx.profile_read(reset=True)
            for _ in range(3):
                eng.index_on([col]).close()
            p = eng.ctx.profile_read(reset=True); eng.ctx.profile(False)
            sc = sum(v["total_ms"] for k, v in p.items() if "scatter" in k) / 3
            tot = sum(v["total_ms"] for v in p.values()) / 3
            print(f"sort_xcd_tiles={mode} {name:12s} scatter {sc:7.3f} ms  all kernels {tot:7.3f} ms", flush=True)

#!/usr/bin/env python3
"""IndexOn kernel times for sort configurations (ctx options sort_threads / sort_rbits)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from csvplus_amd import datagen as dg
from csvplus_amd.engine import Engine

eng = Engine(0)
for threads, rbits in ((256, 8), (512, 8), (256, 9), (512, 9)):
    eng.ctx.set_option("sort_threads", threads)
    eng.ctx.set_option("sort_rbits", rbits)
    print(f"sort_threads={threads} sort_rbits={rbits}", flush=True)
    for n in (100_000, 10_000_000, 100_000_000):
        col = dg.column(dg.SEQ_PERM, n, n, encoding=dg.FIXED8, seed=7).to_device(eng.device)
        eng.index_on([col], unique=True).close()
        eng.ctx.profile(True); eng.ctx.profile_read(reset=True)
        for _ in range(3):
            ix = eng.index_on([col], unique=True); inf = ix.info(); ix.close()
        p = eng.ctx.profile_read(reset=True); eng.ctx.profile(False)
        tot = sum(v["total_ms"] for v in p.values()) / 3
        print(f"  n={n:>11} passes={inf['sort_passes']} total {tot:7.3f} ms | " +
              " ".join(f"{k.replace('k_radix_','').replace('exclusive_','')}={v['total_ms'] / 3:.3f}" for k, v in p.items()), flush=True)
        del col

#!/usr/bin/env python3
"""BASELINE config 3 (IndexOn over 1e8 variable-length duplicate keys): kernel times for the statistics / encode
configurations (ctx options codec_split [round 4: the delimiter split], speculative_groups, plan_threads, gstats_threads).  Usage: config3.py [rows]"""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from csvplus_amd import datagen as dg
from csvplus_amd.engine import Engine

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
eng = Engine(0)
col = dg.varkeys(n).to_device(eng.device)
cases = [dict(codec_split=1, codec_debug=1), dict(codec_split=1), dict(codec_split=0, speculative_groups=0, codec_debug=1),
         dict(codec_split=0, speculative_groups=1)]
if len(sys.argv) > 2:
    cases = cases[:int(sys.argv[2])]
for case in cases:
    for k in ("plan_threads", "gstats_threads", "codec_debug"):
        eng.ctx.set_option(k, 0)
    eng.ctx.set_option("codec_split", 1)
    eng.ctx.set_option("speculative_groups", 1)
    for k, v in case.items():
        eng.ctx.set_option(k, v)
    eng.index_on([col]).close()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        ix = eng.index_on([col]); inf = ix.info(); ix.close()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 3 * 1e3
    eng.ctx.profile(True); eng.ctx.profile_read(reset=True)
    eng.index_on([col]).close()
    p = eng.ctx.profile_read(reset=True); eng.ctx.profile(False)
    print(f"{case}: wall {wall:.3f} ms, bits {inf['code_bits']}, passes {inf['sort_passes']}, dict {inf['dict_entries']} | " +
          " ".join(f"{k.replace('k_', '')}={v['total_ms']:.3f}" for k, v in p.items()), flush=True)

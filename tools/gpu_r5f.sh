#!/bin/bash
# round 5, run f: where the host-coded IndexOn spends its time (threads / chunking A/B)
mkdir -p gpurun_out/r5f
timeout 600 python -m pytest tests/test_gpu_host_build.py tests/test_gpu_window_sort.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -5
python - <<'PY' 2>&1 | tee gpurun_out/r5f/host_build.txt
import time, numpy as np, torch
from csvplus_amd import _native as N, datagen as dg
from csvplus_amd.engine import Engine
from csvplus_amd.streaming import PinnedCol
eng = Engine(0)
n = 100_000_000
col = dg.column(dg.SEQ_PERM, n, n, encoding=dg.FIXED8, seed=7)
pc = PinnedCol(eng.ctx, col)
eng.ctx.set_option("codec_debug", 1)
for threads in (0, 24):
    eng.ctx.set_option("host_threads", threads)
    for rep in range(3):
        t0 = time.perf_counter()
        ix = N.DeviceIndex(eng.ctx, [pc.col], unique=True)
        t1 = time.perf_counter()
        pv = ix.perm_host_view()
        t2 = time.perf_counter()
        ix.close()
        print(f"threads {threads} rep {rep}: build {1e3*(t1-t0):.2f} ms, perm to host {1e3*(t2-t1):.2f} ms, path {ix.info()['build_path'] if False else ''}", flush=True)
PY
true

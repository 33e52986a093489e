#!/usr/bin/env python3
"""IndexOn(config 3's 1e8 variable-length keys) from PINNED HOST memory -> host perm: the split codec's host twin (4 B/row uploaded)
against the upload of the strings, by thread count of the encode pool.  usage: tools/microbench/host_split.py [rows] [threads,...]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np
from csvplus_amd import Context, _native as N, datagen as dg, verify as V
from csvplus_amd.streaming import PinnedCol

rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
threads = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 32, 64, 128]
ctx = Context(0)
col = dg.varkeys(rows)
pc = PinnedCol(ctx, col)
print(f"{rows} rows, {col.nbytes_values() / 1e9:.2f} GB of strings + {col.nbytes_offsets() / 1e9:.2f} GB of offsets in pinned host memory", flush=True)


def timed(reps=6):
    ix = N.DeviceIndex(ctx, [pc.col])
    ix.perm_host_view()
    ix.close()
    best, path, dig, allb = 1e9, None, None, []
    for _ in range(reps):
        t0 = time.perf_counter()
        ix = N.DeviceIndex(ctx, [pc.col])
        t1 = time.perf_counter()
        pv = ix.perm_host_view()
        t2 = time.perf_counter()
        path = ix.info()["build_path"]
        if dig is None:
            dig = V.digest_u64(np.asarray(pv))
        ix.close()
        best = min(best, t2 - t0)
        last = (t1 - t0, t2 - t1)
        allb.append(round((t1 - t0) * 1e3, 1))
    print("      builds:", allb, flush=True)
    return best * 1e3, last[0] * 1e3, last[1] * 1e3, path, dig


ctx.set_option("host_split", 0)
ms, b, p, path, dig0 = timed()
print(f"strings uploaded      : {ms:7.2f} ms (build {b:.2f} + perm to host {p:.2f}) path {path}", flush=True)
try:
    print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip(), flush=True)
except OSError:
    pass
ctx.set_option("host_split", 1)
ms, b, p, path, dig = timed()
print(f"host_split = 1 (auto) : {ms:7.2f} ms (build {b:.2f} + perm to host {p:.2f}) path {path}", flush=True)
ctx.set_option("host_split", 2)
import os
for t in threads:
    numa = 1
    if t < 0:
        t, numa = -t, 0
    ctx.set_option("host_numa", numa)
    print("host_numa", numa, flush=True)
    ctx.set_option("host_split_threads", t)
    ix = N.DeviceIndex(ctx, [pc.col]); ix.close()
    ctx.set_option("codec_debug", 1)
    ms, b, p, path, dig = timed()
    ctx.set_option("codec_debug", 0)
    print(f"host split, threads {t:3d}: {ms:7.2f} ms (build {b:.2f} + perm to host {p:.2f}) path {path} same perm: {dig == dig0}", flush=True)

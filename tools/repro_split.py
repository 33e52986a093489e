#!/usr/bin/env python3
"""Stress of the split codec against the same build without it (ctx option codec_split): many random delimiter-structured
tables in one process, 32- and 64-bit codes, NUL / 0xFF delimiters, heads that contain the delimiter themselves.
(Round 4: this is what isolated the lost LDS histogram increments behind device_utils.hpp: lds_atomics_barrier — ~1 % of the
64-bit builds came out wrong before it, none in 836 iterations after.)
usage: tools/repro_split.py [seed] [seconds] [nul: mostly NUL delimiters] [codec_debug value]; REPRO_OPTS=opt=val,... sets ctx options"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from csvplus_amd import Context, DeviceIndex, StrCol

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120
rng = np.random.default_rng(seed)
ctx = Context(0)
import os
for kv in os.environ.get("REPRO_OPTS", "").split(","):
    if "=" in kv:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
A = [np.frombuffer(b"abcxyz", np.uint8), np.arange(256, dtype=np.uint8)]
t_end = time.time() + budget
it = bad = nsplit = 0
while time.time() < t_end:
    n = int(rng.integers(66_000, 90_000))
    delim = bytes([int(rng.choice([0, 0, 0, 35]))]) if len(sys.argv) > 3 else bytes([int(rng.choice([35, 47, 58, 0, 255, 44, 124]))])
    nheads = int(rng.choice([1, 3, 40, 300]))
    hl = int(rng.integers(0, 14))
    heads = list({bytes(A[int(rng.integers(0, 2))][rng.integers(0, 6, int(rng.integers(0, hl + 1)))]) for _ in range(nheads)})
    digits = int(rng.integers(1, 9))
    pool = [heads[int(rng.integers(0, len(heads)))] + delim + (b"%d" % int(rng.integers(0, 10 ** digits))) for _ in range(max(2, n // int(rng.integers(1, 6))))]
    for j in range(0, len(pool), int(rng.integers(150, 5000))):
        pool[j] = heads[int(rng.integers(0, len(heads)))]
    vals = [pool[int(i)] for i in rng.integers(0, len(pool), n)]
    col = StrCol.from_values(vals)
    # a few small unrelated builds in between (pool / ring / scratch state)
    for _ in range(int(rng.integers(0, 4))):
        DeviceIndex(ctx, [StrCol.from_values([b"%d" % int(x) for x in rng.integers(0, 1000, int(rng.integers(1, 30000)))])]).close()
    ctx.set_option("codec_split", 1)
    if len(sys.argv) > 4:
        ctx.set_option("codec_debug", int(sys.argv[4]))
    g = DeviceIndex(ctx, [col])
    ctx.set_option("codec_debug", 0)
    inf = g.info()
    gp = g.perm().copy()
    ctx.set_option("codec_split", 0)
    r = DeviceIndex(ctx, [col])
    rp = r.perm().copy()
    ctx.set_option("codec_split", 1)
    nsplit += inf["split"] != 0
    if not np.array_equal(gp, rp):
        bad += 1
        i = int(np.argmax(gp != rp))
        print("MISMATCH it", it, "n", n, "delim", delim, "info", inf, "first at", i, "out of range", int((gp >= n).sum()), "differ", int((gp != rp).sum()), flush=True)
        if bad <= 3:
            out = Path(__file__).resolve().parents[1] / "gpurun_out" / "repro"
            out.mkdir(parents=True, exist_ok=True)
            np.save(out / f"data_{bad}.npy", col.data[: int(col.offsets[-1])])
            np.save(out / f"offs_{bad}.npy", col.offsets)
            np.save(out / f"gperm_{bad}.npy", gp)
            np.save(out / f"rperm_{bad}.npy", rp)
            g.save(str(out / f"index_{bad}.cph"))
        g2 = DeviceIndex(ctx, [col])
        print("   rebuilt: equal to reference now?", bool(np.array_equal(g2.perm(), rp)), "info", g2.info()["split"], g2.info()["code_bits"], flush=True)
        g2.close()
    g.close(); r.close()
    it += 1
print("REPRO done: iterations", it, "split taken", nsplit, "mismatches", bad, flush=True)

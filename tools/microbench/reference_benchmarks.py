#!/usr/bin/env python3
"""The reference's six Benchmark* functions (csvplus_test.go:1052-1186) in shape, through the C ABI, at the fixture size
(120 people, 10 000 orders) and scaled up; beside each the lean C restatement (oracle/, one thread) on the host.  The GPU
side is timed device-resident in -> device-resident out, one operation per call like b.N iterations of the reference.
Neither column is the Go binary (no Go toolchain here).  Usage: reference_benchmarks.py [scales...]"""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from csvplus_amd import DeviceIndex, _native as N, datagen as dg
from csvplus_amd.engine import Engine
from oracle import orc

scales = [int(a) for a in sys.argv[1:]] or [1, 100]   # 10 000 (1e8 orders) spends ~7 minutes in the single-thread C port
eng = Engine(0)
ctx = eng.ctx


def timed(fn, min_s=0.15, max_reps=2000):
    fn()                                  # warm-up
    torch.cuda.synchronize()
    reps, t0 = 0, time.perf_counter()
    while True:
        fn()
        reps += 1
        if reps >= max_reps or time.perf_counter() - t0 > min_s:
            break
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def fmt(s):
    return f"{s * 1e6:10.1f} us" if s < 1e-3 else f"{s * 1e3:10.3f} ms"


print(f"{'benchmark':<44}{'rows':>12}  {'GPU (C ABI)':>14}  {'lean C port, 1 thread':>22}")
for s in scales:
    npeople, norders = 120 * s, 10_000 * s
    people = dg.customers(npeople, encoding=dg.ITOA)
    orders = dg.orders(norders, npeople, 8, cust_encoding=dg.ITOA)
    d_id = people["id"].to_device(eng.device)
    d_cust, d_prod = orders["cust_id"].to_device(eng.device), orders["prod_id"].to_device(eng.device)

    def close(ix):
        ix.close()

    rows = []
    rows.append(("CreateSmallSingleIndex", npeople, timed(lambda: close(DeviceIndex(ctx, [d_id], unique=True))),
                 timed(lambda: orc.OracleIndex([people["id"]]), max_reps=20)))
    rows.append(("CreateBiggerMultiIndex", norders, timed(lambda: close(DeviceIndex(ctx, [d_cust, d_prod]))),
                 timed(lambda: orc.OracleIndex([orders["cust_id"], orders["prod_id"]]), max_reps=5)))
    gp, gm = DeviceIndex(ctx, [d_id], unique=True), DeviceIndex(ctx, [d_cust, d_prod])
    op, om = orc.OracleIndex([people["id"]]), orc.OracleIndex([orders["cust_id"], orders["prod_id"]])
    rows.append(("SearchSmallSingleIndex (Find)", npeople, timed(lambda: gp.find(b"0")), timed(lambda: op.find(b"0"))))
    rows.append(("SearchBiggerMultiIndex (Find)", norders, timed(lambda: gm.find(b"0", b"0")), timed(lambda: om.find(b"0", b"0"))))

    # the same searches 1000 at a time through cph_index_find_many (one launch per batch): time per key
    ids = [(people["id"].value(int(i)),) for i in range(0, npeople, max(1, npeople // 1000))][:1000]
    pairs = [(orders["cust_id"].value(int(i)), orders["prod_id"].value(int(i))) for i in range(0, norders, max(1, norders // 1000))][:1000]
    rows.append((f"  ... {len(ids)} keys per find_many, per key", npeople, timed(lambda: gp.find_many(ids)) / len(ids),
                 timed(lambda: [op.find(*k) for k in ids], max_reps=20) / len(ids)))
    rows.append((f"  ... {len(pairs)} keys per find_many, per key", norders, timed(lambda: gm.find_many(pairs)) / len(pairs),
                 timed(lambda: [om.find(*k) for k in pairs], max_reps=20) / len(pairs)))

    def gpu_join(ix, col):
        m = ix.probe([col], out_mem=N.CPH_MEM_DEVICE)
        m.release()

    rows.append(("JoinOnSmallSingleIndex", norders, timed(lambda: gpu_join(gp, d_cust)),
                 timed(lambda: op.join([orders["cust_id"]]), max_reps=5)))
    rows.append(("JoinOnBiggerMultiIndex (prefix)", npeople, timed(lambda: gpu_join(gm, d_id)),
                 timed(lambda: om.join([people["id"]]), max_reps=5)))
    for name, n, g, c in rows:
        print(f"{name:<44}{n:>12}  {fmt(g):>14}  {fmt(c):>22}", flush=True)
    gp.close(); gm.close()
    print()

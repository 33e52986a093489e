"""A Join against a NON-unique index at bench size — the reference's TestLongChain / BenchmarkJoinOnBiggerMultiIndex shape
(csvplus_test.go:252-285, 1161-1186; csvplus.go:559 loops over every equal index row):

    people(1e7).Join(IndexOn(orders.cust_id) over 1e8 rows, "id")                       -> 1e8 pairs  (probe -> scan -> k_expand)
    people(1e7).Join(IndexOn(orders.cust_id), "id").Join(UniqueIndexOn(products), prod_id of the ORDERS row)   (cph_chain_step.source)

usage: dup_build_side.py [orders rows] [people rows]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np
import torch
from csvplus_amd import _native as N, datagen as dg
from csvplus_amd.engine import Engine

M = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
NP = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10_000_000
NPROD = 100_000
eng = Engine(0); ctx = eng.ctx; dev = eng.device
people_id = dg.column(dg.SEQ_PERM, NP, NP, encoding=dg.FIXED8, seed=dg.SEED + 1)
prod_id = dg.column(dg.SEQ_PERM, NPROD, NPROD, encoding=dg.ITOA, seed=dg.SEED + 2)
ords = dg.orders(M, NP, NPROD)
d_people, d_prod = people_id.to_device(dev), prod_id.to_device(dev)
d_ocust, d_oprod = ords["cust_id"].to_device(dev), ords["prod_id"].to_device(dev)


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    ctx.profile(True); ctx.profile_read(reset=True)
    fn()
    p = ctx.profile_read(reset=True); ctx.profile(False)
    return ms, r, {k: round(v["total_ms"], 4) for k, v in sorted(p.items(), key=lambda kv: -kv[1]["total_ms"])}


def build():
    ix = eng.index_on([d_ocust], unique=False)
    inf = ix.info(); ix.close()
    return inf
ms, inf, k = timed(build)
print(f"IndexOn(orders.cust_id) {M} rows, duplicates: {ms:.3f} ms", inf); print("   ", k)

io = eng.index_on([d_ocust], unique=False)
ip = eng.index_on([d_prod], unique=True)
for pos in (False, True):
    def j1():
        c = N.join_chain(ctx, [(io, [d_people])], out_mem=N.CPH_MEM_DEVICE, positions=pos)
        n = c.nrows; c.release(); return n
    ms, n, k = timed(j1)
    # bytes: stream keys in (8 B), (lo,cnt) per stream row written + read by the scan (2 x 8), per pair 8 (stream row) + 4 (build row) out
    # [+ 4 perm read in row-id mode]
    algo = NP * (8 + 16 + 4) + n * (12 + (0 if pos else 4))
    print(f"people.Join(ordersIndex, id) {'positions' if pos else 'row ids'}: {ms:.3f} ms, {n} pairs, {n / ms / 1e6:.1f} G pairs/s, "
          f"algorithmic {algo / 1e9:.2f} GB -> {algo / ms / 1e6 / 8000:.3f} of 8 TB/s"); print("   ", k)
    def j2():
        c = N.join_chain(ctx, [(io, [d_people]), (ip, [d_oprod], 1)], out_mem=N.CPH_MEM_DEVICE, positions=pos)
        n = c.nrows; c.release(); return n
    ms, n, k = timed(j2)
    print(f"  .Join(products, orders.prod_id) {'positions' if pos else 'row ids'}: {ms:.3f} ms, {n} rows, {n / ms / 1e6:.1f} G rows/s"); print("   ", k)
# generic probe entry point (cph_join_probe): pairs / bounds only
for want in (True, False):
    def pr():
        m = io.probe([d_people], out_mem=N.CPH_MEM_DEVICE, want_pairs=want); n = m.nmatches; m.release(); return n
    ms, n, k = timed(pr)
    print(f"cph_join_probe want_pairs={want}: {ms:.3f} ms, {n} matches"); print("   ", k)
# small-scale parity of the same shape against the oracle
from oracle import orc
n_s, m_s = 20_000, 200_000
sp = dg.column(dg.SEQ_PERM, n_s, n_s, encoding=dg.FIXED8, seed=5)
so = dg.orders(m_s, n_s, 700)
spr = dg.column(dg.SEQ_PERM, 700, 700, encoding=dg.ITOA, seed=6)
g = N.DeviceIndex(ctx, [so["cust_id"]]); gp = N.DeviceIndex(ctx, [spr], unique=True)
o = orc.OracleIndex([so["cust_id"]]); op = orc.OracleIndex([spr])
c = N.join_chain(ctx, [(g, [sp]), (gp, [so["prod_id"]], 1)])
j = o.join([sp]); j2 = op.join([so["prod_id"]], row_sel=j["build_row"])
pick = j2["probe_idx"].astype(np.int64)
ok = c.nrows == j2["nmatches"] and np.array_equal(c.stream_row, j["probe_idx"][pick]) and np.array_equal(c.build_row(0), j["build_row"][pick]) \
    and np.array_equal(c.build_row(1), j2["build_row"])
print("parity vs oracle at", n_s, m_s, ":", ok)

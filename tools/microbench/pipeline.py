#!/usr/bin/env python3
"""End-to-end README pipeline on the device: 3 CSV texts resident in HBM -> parse -> 2 unique indices -> chained join
-> gather 4 columns -> ToCsv.  Reports per-stage wall times (stream-synchronised)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import ctypes as C
import torch
from csvplus_amd import _native as N, datagen as dg, ingest, pipeline
from csvplus_amd.engine import Engine

M = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
NC, NP = 10_000_000, 100_000
eng = Engine(0); ctx = eng.ctx; dev = eng.device
if len(sys.argv) > 2:   # reserve a device slab up front (MiB): the first repetition then pays no hipMalloc
    ctx.set_option("pool_reserve_mb", int(sys.argv[2]))
    print(f"pool_reserve_mb = {sys.argv[2]}", flush=True)


def to_device_csv(cols, names):
    dcols = [c.to_device(dev) for c in cols]
    arr = (N.cph_strcol * len(cols))(); keep = []
    for i, c in enumerate(dcols):
        sc, k = c.as_c(); arr[i] = sc; keep.append(k)
    hv = (N.cph_strval * len(cols))(); hk = []
    for i, h in enumerate(names):
        b = (C.c_uint8 * len(h)).from_buffer_copy(h.encode()); hk.append(b)
        hv[i].data = C.cast(b, C.c_void_p).value; hv[i].len = len(h)
    out = C.POINTER(N.cph_bytes)()
    ctx._check(ctx.lib.cph_csv_write(ctx.handle, arr, len(cols), hv, N.CPH_MEM_DEVICE, C.byref(out)))
    return out   # kept alive: the text stays in HBM


cust = dg.customers(NC); prod = dg.products(NP); ords = dg.orders(M, NC, NP)
f_c = to_device_csv([cust["id"], cust["name"], cust["surname"]], ["id", "name", "surname"])
f_p = to_device_csv([prod["prod_id"], prod["product"], prod["price"]], ["prod_id", "product", "price"])
f_o = to_device_csv([ords["cust_id"], ords["prod_id"], ords["qty"]], ["cust_id", "prod_id", "qty"])
sizes = [int(f.contents.size) for f in (f_c, f_p, f_o)]
print(f"customers.csv {sizes[0] / 1e9:.3f} GB, products.csv {sizes[1] / 1e6:.1f} MB, orders.csv {sizes[2] / 1e9:.3f} GB ({M} rows)", flush=True)


def parse(f, idx, nf):
    t = ingest.csv_parse(ctx, None, idx, fields_per_record=nf, skip_records=1, out_mem=N.CPH_MEM_DEVICE,
                         device_ptr=int(f.contents.data), size=int(f.contents.size))
    assert t.error_kind == 0
    t.names = [b"c%d" % i for i in idx]
    return pipeline.Table(t)


for rep in range(5):
    tm = {}
    POSITIONS = rep != 3   # rep 3: the original-row-id mode of rounds 1-3 for comparison (what join_to_csv's default picks at this shape)
    if rep in (2, 3):
        ctx.profile(True); ctx.profile_read(reset=True)
    torch.cuda.synchronize(); t00 = time.perf_counter()
    tc, tp, to = parse(f_c, [0, 1, 2], 3), parse(f_p, [0, 1, 2], 3), parse(f_o, [0, 1, 2], 3)
    torch.cuda.synchronize(); tm["parse_ms"] = (time.perf_counter() - t00) * 1e3
    out_cols = [("cust_id", to, "c0"), ("qty", to, "c2"), ("name", tc, "c1"), ("surname", tc, "c2"), ("product", tp, "c1"), ("price", tp, "c2")]
    t1 = time.perf_counter()
    text = pipeline.join_to_csv(ctx, to, [(tc, "c0", "c0"), (tp, "c0", "c1")], out_cols, timings=tm, out_mem=N.CPH_MEM_DEVICE, positions=POSITIONS)
    torch.cuda.synchronize(); total = (time.perf_counter() - t00) * 1e3
    print(f"rep {rep} ({'sorted positions + payload in index order' if POSITIONS else 'original row ids'}): " + ", ".join(f"{k}={v:.2f}" for k, v in tm.items()) + f" | total {total:.1f} ms, output {len(text) / 1e9:.3f} GB "
          f"(text left in HBM) -> {M / total / 1e6:.2f} G joined rows/s end to end", flush=True)
    if rep in (2, 3):
        p = ctx.profile_read(reset=True); ctx.profile(False)
        for k, v in sorted(p.items(), key=lambda kv: -kv[1]["total_ms"])[:14]:
            print(f"    {k:24s} {v['launches']:4d} launches {v['total_ms']:8.3f} ms", flush=True)
    for t in (tc, tp, to):
        t.release()
    text.release()

"""IndexOn with duplicates at 1e8 rows: counted LDS windows (counted_sort.hip) against the classic radix passes (ctx option counted_sort = 0).
  config 3   surname/name#number, variable length, ~8 rows per key (25-bit codes)
  orders     IndexOn(orders.cust_id): 8-byte ids, 10 rows per key (24-bit codes)
usage: counted_sort.py [rows]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from csvplus_amd import datagen as dg, verify as V
from csvplus_amd.engine import Engine, device_view

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
eng = Engine(0); ctx = eng.ctx; dev = eng.device
for name, col in (("config 3", dg.varkeys(n)), ("orders.cust_id", dg.orders(n, n // 10, 1000)["cust_id"])):
    d = col.to_device(dev)
    ref = None
    for opt in (1, 0):
        ctx.set_option("counted_sort", opt)
        eng.index_on([d]).close(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            eng.index_on([d]).close()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 3 * 1e3
        ctx.profile(True); ctx.profile_read(reset=True)
        ix = eng.index_on([d])
        p = ctx.profile_read(reset=True); ctx.profile(False)
        perm = device_view(ix.perm_device_ptr(), ix.nrows, "<i4", ix, dev)
        chk = V.check_index_order(d, perm)
        dig = V.digest_u64(perm)
        print(f"{name} counted_sort={opt}: {ms:.3f} ms  " + " ".join(f"{k}={v['total_ms']:.3f}" for k, v in p.items()),
              "| verified", chk["ok"], "first_dup", ix.first_dup, "digest %016x" % dig, "same perm" if ref in (None, dig) else "PERM DIFFERS")
        ref = dig
        del perm
        ix.close()
    del d
    torch.cuda.empty_cache()
ctx.set_option("counted_sort", 1)

import sys, time
sys.path.insert(0, '/root/repo')
import torch
from csvplus_amd import DeviceIndex, _native as N, datagen as dg
from csvplus_amd.engine import Engine
eng = Engine(0); ctx = eng.ctx
people = dg.customers(1_200_000, encoding=dg.ITOA)
orders = dg.orders(100_000_000, 1_200_000, 8, cust_encoding=dg.ITOA)
d_id = people["id"].to_device(eng.device); d_cust = orders["cust_id"].to_device(eng.device)
gp = DeviceIndex(ctx, [d_id], unique=True)
print(gp.info())
for _ in range(2):
    m = gp.probe([d_cust], out_mem=N.CPH_MEM_DEVICE); m.release()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    m = gp.probe([d_cust], out_mem=N.CPH_MEM_DEVICE); n = m.nmatches; m.release()
torch.cuda.synchronize(); print("JoinOnSmallSingleIndex 1e8 ITOA ids: %.3f ms, %d matches" % ((time.perf_counter() - t0) / 5 * 1e3, n))
# bounds only (Except / has / counts): a duplicate-free index answers through its rank table (8 B per 32 codes) instead of
# the 8-byte-per-code table
ctx.profile(True); ctx.profile_read(reset=True)
for _ in range(3):
    m = gp.probe([d_cust], out_mem=N.CPH_MEM_DEVICE, want_pairs=False); m.release()
p = ctx.profile_read(reset=True); ctx.profile(False)
print("   the same, bounds only: " + " ".join(f"{k}={v['total_ms'] / v['launches']:.3f}" for k, v in p.items()), gp.info()["lookup_built"])
# the chained join of the bench with reference-style (unpadded) ids on both sides
from csvplus_amd import join_chain
prod = dg.products(100_000)
d_pid = prod["prod_id"].to_device(eng.device)
orders2 = dg.orders(100_000_000, 1_200_000, 100_000, cust_encoding=dg.ITOA)
d_c2, d_p2 = orders2["cust_id"].to_device(eng.device), orders2["prod_id"].to_device(eng.device)
gq = DeviceIndex(ctx, [d_pid], unique=True)
for _ in range(2):
    ch = join_chain(ctx, [(gp, [d_c2]), (gq, [d_p2])], out_mem=N.CPH_MEM_DEVICE); ch.release()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    ch = join_chain(ctx, [(gp, [d_c2]), (gq, [d_p2])], out_mem=N.CPH_MEM_DEVICE); n = ch.nrows; ch.release()
torch.cuda.synchronize(); print("chained join 1e8 rows, unpadded ids: %.3f ms, %d rows" % ((time.perf_counter() - t0) / 5 * 1e3, n))
for _ in range(2):
    ch = join_chain(ctx, [(gp, [d_c2]), (gq, [d_p2])], out_mem=N.CPH_MEM_DEVICE, positions=True); ch.release()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    ch = join_chain(ctx, [(gp, [d_c2]), (gq, [d_p2])], out_mem=N.CPH_MEM_DEVICE, positions=True); n = ch.nrows; ch.release()
torch.cuda.synchronize(); print("   reporting sorted positions: %.3f ms, %d rows" % ((time.perf_counter() - t0) / 5 * 1e3, n))

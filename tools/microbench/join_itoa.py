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

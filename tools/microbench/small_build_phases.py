import sys
sys.path.insert(0,'/root/repo')
import torch
from csvplus_amd import DeviceIndex, datagen as dg
from csvplus_amd.engine import Engine
eng = Engine(0); ctx = eng.ctx
for n in (120, 4000, 16384):
    people = dg.customers(n, encoding=dg.ITOA)
    col=[people["id"].to_device(eng.device)]
    for _ in range(3): DeviceIndex(ctx, col).close()
    ctx.set_option("codec_debug",1)
    for _ in range(2): DeviceIndex(ctx, col).close()
    ctx.set_option("codec_debug",0)

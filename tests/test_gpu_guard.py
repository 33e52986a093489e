"""Canary mode of the device pool (cph_ctx_set_option "pool_guard"): every block the ctx allocates
carries 256 guard bytes behind the bytes its user asked for, checked when the block is released.  A representative
slice of the hot path runs under it here on every GPU test run (tools/gpu_guard.sh runs the WHOLE suite under it),
and a positive control shows that an out-of-bounds write is in fact reported."""
import numpy as np
import pytest
import torch

from csvplus_amd import Context, DeviceIndex, StrCol, _native as N, datagen as dg, join_chain
from csvplus_amd.engine import device_view
from tests.helpers import random_keys

pytestmark = pytest.mark.gpu


def test_hot_path_under_pool_guard():
    ctx = Context(0)
    ctx.set_option("pool_guard", 1)
    rng = np.random.default_rng(3)
    cust, prod = dg.customers(30_000)["id"], dg.products(500)["prod_id"]
    ia, ib = DeviceIndex.build_many(ctx, [([cust], True), ([prod], True)])
    for domain, n in ((30_000, 70_001), (60_000, 4097), (30_000, 1)):
        o = dg.orders(n, domain, 500)
        for mem in (N.CPH_MEM_HOST, N.CPH_MEM_DEVICE):
            ch = join_chain(ctx, [(ia, [o["cust_id"]]), (ib, [o["prod_id"]])], out_mem=mem)
            ch.release()
        ia.probe([o["cust_id"]]).release()
    # duplicate keys, several code words, dictionary groups, long keys
    for vals in (random_keys(rng, 20_000, 0, 6, alphabet=np.frombuffer(b"ab", np.uint8)),
                 random_keys(rng, 5000, 10, 40, distinct=2000),
                 [b"%s/%s#%d" % (b"Smith", b"Amelia", i % 977) for i in range(30_000)],
                 random_keys(rng, 2000, 0, 400, alphabet=np.frombuffer(b"xy", np.uint8), distinct=500)):
        col = StrCol.from_values(vals)
        ix = DeviceIndex(ctx, [col])
        ix.probe([StrCol.from_values(vals[:3000])]).release()
        ix.dup_groups()
        ix.select(list(range(0, col.nrows, 2))).close()
        ix.find(vals[0])
        ix.close()
    ia.close()
    ib.close()
    ctx.set_option("pool_guard_check", 0)
    ctx.close()


def test_pool_guard_reports_an_overrun():
    ctx = Context(0)
    ctx.set_option("pool_guard", 1)
    cust = dg.customers(1000)["id"]
    ix = DeviceIndex(ctx, [cust], unique=True)
    ch = join_chain(ctx, [(ix, [dg.orders(1000, 1000, 10)["cust_id"]])], out_mem=N.CPH_MEM_DEVICE)
    p = ch.device_ptrs()["build_row"][0]
    t = device_view(p, 1000 + 2, "<i4", ch, torch.device("cuda", 0))   # two elements past the 4000-byte block
    t[1001] = 7
    torch.cuda.synchronize()
    with pytest.raises(N.CphError) as e:
        ctx.set_option("pool_guard_check", 0)
    assert "pool guard" in str(e.value)
    ch.release()
    ix.close()
    ctx.close()


def test_reserved_slab_serves_the_same_results():
    """cph_ctx_set_option "pool_reserve_mb": blocks carved first-fit out of one slab (and coalesced on release) — the
    hot path on a slab-backed ctx, with the canaries on, gives the oracle's results; requests larger than the slab fall
    through to hipMalloc."""
    from oracle import orc
    ctx = Context(0)
    ctx.set_option("pool_reserve_mb", 64)
    ctx.set_option("pool_guard", 1)
    cust, prod = dg.customers(40_000)["id"], dg.products(300)["prod_id"]
    oa, ob = orc.OracleIndex([cust]), orc.OracleIndex([prod])
    for rep in range(3):
        ia, ib = DeviceIndex.build_many(ctx, [([cust], True), ([prod], True)])
        np.testing.assert_array_equal(ia.perm(), oa.perm)
        o = dg.orders(200_000 + 777 * rep, 80_000, 300)
        ch = join_chain(ctx, [(ia, [o["cust_id"]]), (ib, [o["prod_id"]])])
        j1 = oa.join([o["cust_id"]])
        j2 = ob.join([o["prod_id"]], row_sel=j1["probe_idx"].astype(np.uint32))
        pick = j2["probe_idx"].astype(np.int64)
        np.testing.assert_array_equal(ch.stream_row, j1["probe_idx"][pick])
        np.testing.assert_array_equal(ch.build_row(0), j1["build_row"][pick])
        np.testing.assert_array_equal(ch.build_row(1), j2["build_row"])
        ch.release()
        ia.close()
        ib.close()
    big = dg.orders(30_000_000, 80_000, 300)["cust_id"]     # 240 MB of keys: beyond the 64 MiB slab
    ia = DeviceIndex(ctx, [cust], unique=True)
    m = ia.probe([big], want_pairs=False)
    assert m.nprobe == 30_000_000
    m.release()
    ia.close()
    ctx.set_option("pool_guard_check", 0)
    with pytest.raises(N.CphError):
        ctx.set_option("pool_reserve_mb", 16)      # one slab per ctx
    ctx.close()

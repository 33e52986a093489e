"""The exchange behind the C ABI (cph_dist_*, csrc/dist.hip).

  * RCCL transport with ONE rank through the C entry points on the GPU box (unique id, ncclCommInitRank,
    count all-gather, grouped exchange, broadcast): catches loading / dtype / initialisation problems.
  * The multi-rank control flow (counts, displacements, unequal and empty shards, the identity rule, index
    broadcast) through the in-process loopback transport: the ranks are THREADS, each with its own ctx,
    sharing the one GPU — same code above the transport as a real N-GPU run.  Every rank's gathered result
    is compared with the oracle's join over the WHOLE stream (csvplus.go:553-567: stream order).
N > 1 over real xGMI links is not reachable from here and stays unmeasured.
"""
import threading

import numpy as np
import pytest
import torch

from csvplus_amd import Context, DeviceIndex, _native as N, datagen as dg, join_chain
from csvplus_amd.engine import shard_range
from oracle import orc

pytestmark = pytest.mark.gpu


def dev_array(ptr, n, dtype):
    """Host copy of a device array of the library."""
    if n == 0 or not ptr:
        return np.empty(0, dtype=dtype)
    from csvplus_amd.engine import device_view

    t = device_view(ptr, n, {np.uint32: "<i4", np.uint64: "<i8"}[dtype], None, torch.device("cuda", 0))
    return t.cpu().numpy().view(dtype).copy()


def oracle_whole(cust, prod, ords_cols):
    oa, ob = orc.OracleIndex([cust]), orc.OracleIndex([prod])
    j1 = oa.join([ords_cols[0]])
    j2 = ob.join([ords_cols[1]], row_sel=j1["probe_idx"].astype(np.uint32))
    pick = j2["probe_idx"].astype(np.int64)
    return j1["probe_idx"][pick].astype(np.uint64), j1["build_row"][pick], j2["build_row"]


def run_ranks(world, fn):
    """fn(rank) on `world` threads; re-raises the first failure."""
    errs = [None] * world

    def body(r):
        try:
            fn(r)
        except BaseException as e:   # noqa: BLE001
            errs[r] = e

    ths = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=600)
    for e in errs:
        if e is not None:
            raise e


def test_rccl_single_rank_through_the_c_abi(ctx):
    d = N.Dist.create(ctx, N.Dist.unique_id(ctx), 0, 1)
    assert (d.rank, d.size) == (0, 1)
    # ONE RCCL per process: torch (imported above) has mapped its own librccl, so the library binds to THAT copy
    # instead of opening a second one; no second librccl mapping may appear
    tr = d.transport()
    assert tr.startswith("rccl nranks=1 ") and "the copy the host process had loaded" in tr, tr
    mapped = {ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln}
    assert len(mapped) == 1, mapped
    a = torch.arange(1000, dtype=torch.int64, device="cuda") * 3
    b = torch.arange(1000, dtype=torch.int32, device="cuda") + 7
    torch.cuda.synchronize()
    g = d.allgatherv([a.data_ptr(), b.data_ptr()], [8, 4], 1000)
    ctx.synchronize()
    assert g.total == 1000 and g.counts == [1000] and g.displs == [0]
    np.testing.assert_array_equal(dev_array(g.data_ptrs[0], 1000, np.uint64), a.cpu().numpy().view(np.uint64))
    np.testing.assert_array_equal(dev_array(g.data_ptrs[1], 1000, np.uint32), b.cpu().numpy().view(np.uint32))
    g.release()
    g = d.allgatherv([0, 0], [8, 4], 0)   # empty contribution
    assert g.total == 0
    g.release()
    # a chain result (identity and not), and the root side of an index broadcast
    cust, prod = dg.customers(5000)["id"], dg.products(80)["prod_id"]
    ia, ib = DeviceIndex(ctx, [cust], unique=True), DeviceIndex(ctx, [prod], unique=True)
    for domain in (5000, 10000):   # every row joins / about half of them
        o = dg.orders(30_000, domain, 80)
        cols = [o["cust_id"], o["prod_id"]]
        ch = join_chain(ctx, [(ia, [cols[0]]), (ib, [cols[1]])], probe_base=1000, out_mem=N.CPH_MEM_DEVICE)
        g = d.chain_allgather(ch)
        ctx.synchronize()
        es, ea, eb = oracle_whole(cust, prod, cols)
        assert g.total == len(es) and g.identity == (domain == 5000)
        if g.identity:
            assert g.stream_base == 1000
            rows = g.data_ptrs
        else:
            np.testing.assert_array_equal(dev_array(g.data_ptrs[0], g.total, np.uint64), es + 1000)
            rows = g.data_ptrs[1:]
        np.testing.assert_array_equal(dev_array(rows[0], g.total, np.uint32), ea)
        np.testing.assert_array_equal(dev_array(rows[1], g.total, np.uint32), eb)
        g.release()
        ch.release()
    assert d.index_broadcast(ia, root=0) is ia
    d.close()


def loopback_factory(group, world):
    return lambda ctx, r: N.Dist.loopback(ctx, group, r, world)


@pytest.mark.parametrize("world,domain_factor", [(2, 1), (3, 2), (3, 1)])
def test_loopback_sharded_chain_matches_oracle(world, domain_factor):
    run_sharded_chain(world, domain_factor, loopback_factory(f"chain-{world}-{domain_factor}", world))


def run_sharded_chain(world, domain_factor, make_dist, expect_transport="loopback"):
    """Each rank joins its contiguous row range against replicated indexes; the gathered lists of EVERY rank
    equal the oracle's join over the whole stream.  domain_factor 2: half of the customer ids miss (unequal
    counts, explicit stream rows); rank 1 of 3 additionally gets rows that never join (an empty contribution)."""
    m, nc, npd = 50_001, 4000, 60
    cust, prod = dg.customers(nc)["id"], dg.products(npd)["prod_id"]
    o = dg.orders(m, nc * domain_factor, npd)
    cols = [o["cust_id"], o["prod_id"]]
    empty_rank = 1 if (world == 3 and domain_factor == 2) else None
    if empty_rank is not None:   # that rank's customers all miss: ids from a disjoint domain
        b, e = shard_range(m, empty_rank, world)
        miss = dg.column(dg.UNIFORM, m, nc, encoding=dg.FIXED8, base=50_000_000, seed=99)
        vals = cols[0].values()
        vals[b:e] = miss.values()[b:e]
        from csvplus_amd import StrCol
        cols[0] = StrCol.from_values(vals)
    es, ea, eb = oracle_whole(cust, prod, cols)

    def rank_body(r):
        ctx = Context(0)
        d = make_dist(ctx, r)
        assert d.transport().startswith(expect_transport), d.transport()
        ia, ib = DeviceIndex(ctx, [cust], unique=True), DeviceIndex(ctx, [prod], unique=True)
        b, e = shard_range(m, r, world)
        ch = join_chain(ctx, [(ia, [cols[0].slice(b, e)]), (ib, [cols[1].slice(b, e)])], probe_base=b,
                        out_mem=N.CPH_MEM_DEVICE)
        if empty_rank == r:
            assert ch.nrows == 0
        g = d.chain_allgather(ch)
        ctx.synchronize()
        assert g.total == len(es) and sum(g.counts) == g.total and len(g.counts) == world
        assert g.identity == (domain_factor == 1)
        if g.identity:
            assert g.stream_base == 0 and g.total == m
            rows = g.data_ptrs
        else:
            np.testing.assert_array_equal(dev_array(g.data_ptrs[0], g.total, np.uint64), es)
            rows = g.data_ptrs[1:]
        np.testing.assert_array_equal(dev_array(rows[0], g.total, np.uint32), ea)
        np.testing.assert_array_equal(dev_array(rows[1], g.total, np.uint32), eb)
        g.release()
        ch.release()
        d.close()
        ctx.close()

    run_ranks(world, rank_body)


def gathered_array(g, a, dtype):
    """Host copy of array a of a Gathered, wherever it lives (device memory / the shared host buffer)."""
    if g.total == 0 or not g.data_ptrs[a]:
        return np.empty(0, dtype=dtype)
    if g.mem == N.CPH_MEM_HOST:
        import ctypes as C

        nb = g.total * np.dtype(dtype).itemsize
        return np.frombuffer((C.c_uint8 * nb).from_address(g.data_ptrs[a]), dtype=dtype).copy()
    return dev_array(g.data_ptrs[a], g.total, dtype)


def cut_points(m, world, unequal):
    """Shard boundaries: the even range split, or deliberately uneven ones (with an EMPTY shard when world >= 3)."""
    if not unequal:
        return [shard_range(m, r, world)[0] for r in range(world)] + [m]
    cuts = [0, m // 7] + ([m // 7] if world >= 3 else []) + [m * 5 // 7 + 3 * k for k in range(world)]
    return sorted(cuts[:world]) + [m]


@pytest.mark.parametrize("host", [False, True])
@pytest.mark.parametrize("world,domain_factor,nchunks,unequal,positions,shard_given",
                         [(2, 1, 4, False, True, True), (3, 1, 5, True, False, True), (3, 2, 3, True, True, False),
                          (2, 2, 1, False, False, True), (1, 2, 4, False, True, False)])
def test_loopback_pipelined_join_chain(world, domain_factor, nchunks, unequal, positions, shard_given, host):
    run_pipelined_chain(world, domain_factor, loopback_factory(f"pipe-{world}-{domain_factor}-{nchunks}-{host}", world), nchunks,
                        host=host, positions=positions, unequal=unequal, shard_given=shard_given)


@pytest.mark.parametrize("world,domain_factor,nchunks,unequal,positions",
                         [(2, 1, 4, False, True), (3, 2, 5, True, False), (3, 2, 3, True, True), (2, 2, 1, False, False), (4, 1, 2, True, True)])
def test_loopback_pipelined_join_chain_bit_packed(world, domain_factor, nchunks, unequal, positions):
    """CPH_DIST_PACKED: 12 + 6 bits per row (row ids: 4000 customers + the absent code, 60 products) cross the transport instead
    of 64; every rank's gathered arrays are the ones of the plain format."""
    run_pipelined_chain(world, domain_factor, loopback_factory(f"packed-{world}-{domain_factor}-{nchunks}", world), nchunks,
                        positions=positions, unequal=unequal, packed=True)


def run_pipelined_chain(world, domain_factor, make_dist, nchunks, host=False, positions=False, unequal=False, shard_given=True,
                        expect_transport="loopback", packed=False):
    """cph_dist_join_chain: every rank joins its shard in `nchunks` sub-chunks whose rows are exchanged (xGMI path / shared host
    buffer) while the next chunk is joined; the gathered list of EVERY rank equals the oracle's join over the whole stream
    (csvplus.go:553-567), for even and uneven shards, an empty shard, all rows joining (identity) or half of them."""
    m, nc, npd = 61_003, 4000, 60
    cust, prod = dg.customers(nc)["id"], dg.products(npd)["prod_id"]
    o = dg.orders(m, nc * domain_factor, npd)
    cols = [o["cust_id"], o["prod_id"]]
    es, ea, eb = oracle_whole(cust, prod, cols)
    cuts = cut_points(m, world, unequal)
    shard_rows = [cuts[r + 1] - cuts[r] for r in range(world)]

    def rank_body(r):
        ctx = Context(0)
        d = make_dist(ctx, r)
        assert d.transport().startswith(expect_transport), d.transport()
        ia, ib = DeviceIndex(ctx, [cust], unique=True), DeviceIndex(ctx, [prod], unique=True)
        perms = [ia.perm(), ib.perm()]
        b, e = cuts[r], cuts[r + 1]
        for rep in range(2):   # the second call reuses the exchange stream, the events and the shared host buffer
            g = d.join_chain([(ia, [cols[0].slice(b, e)]), (ib, [cols[1].slice(b, e)])], probe_base=b,
                             shard_rows=shard_rows if shard_given else None, nchunks=nchunks, positions=positions, host=host,
                             packed=packed)
            ctx.synchronize()
            if packed:   # ceil(log2(4000 + 1)) + ceil(log2(60)) bits; a shard's chunk travels in whole 64-row groups
                assert g.stats["packed_bits"] == (18 if world > 1 else 0), g.stats
                rows_sent = sum(-(-(shard_rows[r] // nchunks + (1 if c < shard_rows[r] % nchunks else 0)) // 64) * 64
                                for c in range(nchunks)) if world > 1 else 0
                assert g.stats["bytes_sent"] <= (rows_sent * 18 // 8 + 16 * nchunks) * (world - 1), g.stats
            else:
                assert g.stats["packed_bits"] == 0
            assert g.stats["chunks"] == nchunks and g.stats["pipelined"] == (1 if nchunks > 1 else 0), g.stats
            assert g.mem == (N.CPH_MEM_HOST if host else N.CPH_MEM_DEVICE)
            assert g.total == len(es) and sum(g.counts) == g.total and len(g.counts) == world
            assert g.identity == (domain_factor == 1)
            first = 0
            if g.identity:
                assert g.stream_base == 0 and g.total == m and g.counts == shard_rows and g.narrays == 2
            else:
                assert g.narrays == 3
                np.testing.assert_array_equal(gathered_array(g, 0, np.uint64), es)
                first = 1
            ga, gb = gathered_array(g, first, np.uint32), gathered_array(g, first + 1, np.uint32)
            if positions:
                ga, gb = perms[0][ga], perms[1][gb]
            np.testing.assert_array_equal(ga, ea)
            np.testing.assert_array_equal(gb, eb)
            g.release()
        d.close()
        ctx.close()

    run_ranks(world, rank_body)


@pytest.mark.parametrize("host", [False, True])
def test_loopback_join_chain_one_shot_for_duplicate_keys(host):
    """A chain the dense pipeline cannot carry (an index with duplicate keys: several tuples per stream row) is joined whole
    and exchanged compact — same entry point, stats say so."""
    world, m = 2, 20_000
    keys = dg.varkeys(3000)                                   # duplicates
    probe = dg.varkeys(m, 3000, seed=dg.SEED + 5)
    oj = orc.OracleIndex([keys]).join([probe])

    def rank_body(r):
        ctx = Context(0)
        d = loopback_factory(f"oneshot-{host}", world)(ctx, r)
        ix = DeviceIndex(ctx, [keys])
        b, e = shard_range(m, r, world)
        g = d.join_chain([(ix, [probe.slice(b, e)])], probe_base=b, nchunks=4, host=host)
        ctx.synchronize()
        assert g.stats["chunks"] == 0 and not g.identity and g.narrays == 2 and g.total == len(oj["probe_idx"])
        np.testing.assert_array_equal(gathered_array(g, 0, np.uint64), oj["probe_idx"].astype(np.uint64))
        np.testing.assert_array_equal(gathered_array(g, 1, np.uint32), oj["build_row"])
        g.release()
        d.close()
        ctx.close()

    run_ranks(world, rank_body)


def test_loopback_index_broadcast_option_b():
    run_index_broadcast(3, loopback_factory("bcast", 3))


def run_index_broadcast(world, make_dist):
    """Build side option B: rank 0 sorts, ranks 1..2 receive descriptor + sorted codes + perm and join with it."""
    nc = 20_000
    cust = dg.customers(nc)["id"]
    varkeys = dg.varkeys(30_000)                      # duplicate keys, dictionary-coded groups in the codec
    from csvplus_amd import StrCol
    rng = np.random.default_rng(8)
    longcol = StrCol.from_values([bytes(rng.integers(97, 100, int(rng.integers(0, 300)), dtype=np.uint8)) for _ in range(4000)])
    ol = orc.OracleIndex([longcol])                   # keys beyond one codec window: the windows travel too
    probe = dg.orders(10_000, 2 * nc, 10)["cust_id"]
    ou, ov = orc.OracleIndex([cust]), orc.OracleIndex([varkeys])
    ej = ou.join([probe])

    def rank_body(r):
        ctx = Context(0)
        d = make_dist(ctx, r)
        for col, oix, unique in ((cust, ou, True), (varkeys, ov, False), (longcol, ol, False)):
            mine = DeviceIndex(ctx, [col], unique=unique) if r == 0 else None
            ix = d.index_broadcast(mine, root=0)
            assert ix.nrows == col.nrows
            np.testing.assert_array_equal(ix.perm(), oix.perm)
            assert ix.info()["code_bits"] > 0
            if unique:
                mt = ix.probe([probe])
                np.testing.assert_array_equal(mt.probe_idx, ej["probe_idx"])
                np.testing.assert_array_equal(mt.build_row, ej["build_row"])
                mt.release()
            ix.close()
        d.close()
        ctx.close()

    run_ranks(world, rank_body)

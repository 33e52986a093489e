"""Committed golden vectors for the hot path (tests/golden/hotpath_vectors.json, made by
tests/golden/make_hotpath_vectors.py): the oracle must still reproduce them (CPU), and the GPU path must
reproduce them WITHOUT the oracle in the loop (GPU)."""
import json
from pathlib import Path

import numpy as np
import pytest

from csvplus_amd import datagen as dg

GOLD = json.loads((Path(__file__).parent / "golden" / "hotpath_vectors.json").read_text())


def _tables(inp):
    enc = getattr(dg, inp["enc"])
    assert inp["seed"] == dg.SEED
    cust, prod = dg.customers(inp["nc"], encoding=enc), dg.products(inp["np"])
    ords = dg.orders(inp["m"], inp["nc"] + inp["extra_ids"], inp["np"], cust_encoding=enc)
    return cust, prod, ords


@pytest.mark.parametrize("name", sorted(GOLD))
def test_oracle_reproduces_golden_vectors(name):
    from tests.golden.make_hotpath_vectors import case_vectors
    g = GOLD[name]
    inp = g["inputs"]
    got = case_vectors({"name": name, "nc": inp["nc"], "np": inp["np"], "m": inp["m"], "enc": inp["enc"],
                        "extra_ids": inp["extra_ids"]})
    assert got == g


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(GOLD))
def test_gpu_reproduces_golden_vectors(ctx, name):
    from csvplus_amd import DeviceIndex, join_chain
    from oracle import orc   # only its FNV digest helper: the joins below are the GPU's

    def digest(a):
        return "%016x" % orc.fnv1a64(np.ascontiguousarray(a))

    g = GOLD[name]
    cust, prod, ords = _tables(g["inputs"])
    ia, ib = DeviceIndex(ctx, [cust["id"]], unique=True), DeviceIndex(ctx, [prod["prod_id"]], unique=True)
    assert ia.status == 0 and ib.status == 0
    pa = ia.perm()
    assert digest(pa) == g["customers_perm"] and [int(x) for x in pa[:8]] == g["customers_perm_head"]
    assert digest(ib.perm()) == g["products_perm"]
    m = ia.probe([ords["cust_id"]])
    assert m.nmatches == g["join1_nmatches"]
    cnt = m.cnt
    assert [digest(m.lo[cnt > 0]), digest(cnt)] == g["join1_lo_cnt"]
    assert [digest(m.probe_idx), digest(m.build_row)] == g["join1_pairs"]
    ch = join_chain(ctx, [(ia, [ords["cust_id"]]), (ib, [ords["prod_id"]])])
    assert ch.nrows == g["chain_rows"]
    assert [digest(ch.stream_row.astype(np.uint64)), digest(ch.build_row(0)), digest(ch.build_row(1))] == g["chain"]
    by_name = DeviceIndex(ctx, [cust["surname"], cust["name"]])
    assert digest(by_name.perm()) == g["dup_index_perm"] and by_name.first_dup == g["dup_first_dup"]
    jn = by_name.probe([cust["surname"].slice(0, 100)])
    assert jn.nmatches == g["dup_join_nmatches"]
    assert [digest(jn.probe_idx), digest(jn.build_row)] == g["dup_join_pairs"]

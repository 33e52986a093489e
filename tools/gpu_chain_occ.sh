#!/bin/bash
# chain kernel in positions mode with the compiler held to N waves per SIMD (amdgpu_waves_per_eu): rebuilds chain.o on the box.
# usage: tools/gpu_chain_occ.sh "0 4 5"      (0 = as committed)
export TMPDIR=/tmp
cp csvplus_amd/csrc/chain.hip /tmp/chain.orig
for W in ${1:-"0 4"}; do
  if [ "$W" = "0" ]; then cp /tmp/chain.orig csvplus_amd/csrc/chain.hip
  else sed -E "s/__global__ __launch_bounds__\(kChainThreads\) void k_chain_dense/__global__ __launch_bounds__(kChainThreads) __attribute__((amdgpu_waves_per_eu($W, $W))) void k_chain_dense/" /tmp/chain.orig > csvplus_amd/csrc/chain.hip; fi
  make hip > /tmp/make_occ_$W.log 2>&1 || { echo "build failed for $W"; tail -5 /tmp/make_occ_$W.log; continue; }
  CHAIN_POS=1 timeout 300 python - <<PY
import sys, os
sys.path.insert(0, '.')
from csvplus_amd import datagen as dg
from csvplus_amd.engine import Engine
eng = Engine(0); dev = eng.device
M, NC, NP = 100_000_000, 10_000_000, 100_000
cust = dg.column(dg.SEQ_PERM, NC, NC, encoding=dg.FIXED8, seed=dg.SEED + 1).to_device(dev)
prod = dg.column(dg.SEQ_PERM, NP, NP, encoding=dg.ITOA, seed=dg.SEED + 2).to_device(dev)
o = dg.orders(M, NC, NP)
oc, op = o["cust_id"].to_device(dev), o["prod_id"].to_device(dev)
ia = eng.index_on([cust], unique=True); ib = eng.index_on([prod], unique=True)
steps = [(ia, oc), (ib, op)]
for pos in (True, False):
    r = eng.chained_join(steps, positions=pos); assert r.n == M; r.release()
    eng.ctx.profile(True); eng.ctx.profile_read(reset=True)
    for _ in range(5): eng.chained_join(steps, positions=pos).release()
    p = eng.ctx.profile_read(reset=True); eng.ctx.profile(False)
    print("waves_per_eu=$W %s k_chain_dense %.3f ms" % ("positions" if pos else "row ids  ", p['k_chain_dense']['total_ms'] / 5), flush=True)
PY
done
cp /tmp/chain.orig csvplus_amd/csrc/chain.hip

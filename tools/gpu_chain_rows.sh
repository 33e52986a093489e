#!/bin/bash
# chain-kernel tuning sweep (threads per workgroup x rows in flight per thread): rebuilds chain.o on the box
# for each setting (the checkout there is a scratch copy).  usage: tools/gpu_chain_rows.sh "256,3 128,3 512,3"
export TMPDIR=/tmp
cp csvplus_amd/csrc/chain.hip /tmp/chain.orig
for cfg in ${1:-"256,3 256,4 256,2 256,6"}; do
  T=${cfg%,*}; R=${cfg#*,}
  sed -E "s/constexpr int kChainRows    = [0-9]+;/constexpr int kChainRows    = $R;/; s/constexpr int kChainThreads = [0-9]+;/constexpr int kChainThreads = $T;/" /tmp/chain.orig > csvplus_amd/csrc/chain.hip
  make hip > /tmp/make_$T_$R.log 2>&1 || { echo "build failed for $cfg"; tail -5 /tmp/make_$T_$R.log; continue; }
  timeout 300 python - <<PY
import sys, os
POS = os.environ.get("CHAIN_POS", "0") == "1"
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
r = eng.chained_join(steps, positions=POS); assert r.n == M; r.release()
eng.ctx.profile(True); eng.ctx.profile_read(reset=True)
for _ in range(5): eng.chained_join(steps, positions=POS).release()
p = eng.ctx.profile_read(reset=True)
print(("positions " if POS else "row ids   ") + "threads,rows=$cfg k_chain_dense %.3f ms" % (p['k_chain_dense']['total_ms'] / 5), flush=True)
PY
done
cp /tmp/chain.orig csvplus_amd/csrc/chain.hip

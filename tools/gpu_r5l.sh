#!/bin/bash
# round 5, run l: hash probe sequences followed in rounds (hash_resolve16); sparse variant again
mkdir -p gpurun_out/r5l
timeout 900 python -m pytest tests/test_gpu_hash_probe.py tests/test_gpu_chain.py tests/test_gpu_parity.py tests/test_gpu_hypothesis.py -m gpu -q -x 2>&1 | tail -8
timeout 600 python bench.py --no-cpu-baseline --no-e2e --no-index-1e8 --no-traffic --no-calibration --variants sparse > gpurun_out/r5l/bench.out 2> gpurun_out/r5l/bench.err
echo "bench rc=$?"; tail -2 gpurun_out/r5l/bench.err
tail -1 gpurun_out/r5l/bench.out > gpurun_out/r5l/bench.json
python - <<PY
import json
d = json.load(open("gpurun_out/r5l/bench.json"))
for k, v in d["variants"].items():
    if isinstance(v, dict):
        print(k, v.get("ms_per_step"), "chain", v.get("k_chain_dense_ms"), "verified", v.get("verified"), "hash_bytes", (v.get("customers_index") or {}).get("hash_bytes"), v.get("kernels_ms"))
PY
timeout 300 python tools/microbench/hash_probe.py 2>&1 | tail -20
true

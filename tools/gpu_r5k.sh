#!/bin/bash
# round 5, run k: hash tables by distinct keys at load 0.75 (A/B on the sparse variant), --stream and the N > 1 report on the one-GPU box
mkdir -p gpurun_out/r5k
timeout 900 python -m pytest tests/test_bench_launch.py tests/test_gpu_hash_probe.py tests/test_gpu_chain.py tests/test_gpu_stream.py -m gpu -q -x 2>&1 | tail -15
for pct in 50 75 85; do
timeout 600 python bench.py --no-cpu-baseline --no-e2e --no-index-1e8 --no-traffic --no-calibration --variants sparse --ctx-option hash_load_pct=$pct > gpurun_out/r5k/bench_$pct.out 2> gpurun_out/r5k/bench_$pct.err
echo "bench rc=$?"; tail -2 gpurun_out/r5k/bench_$pct.err
tail -1 gpurun_out/r5k/bench_$pct.out > gpurun_out/r5k/bench_$pct.json
python - <<PY
import json
d = json.load(open("gpurun_out/r5k/bench_$pct.json"))
for k, v in d["variants"].items():
    if isinstance(v, dict):
        print("load $pct:", k, v.get("ms_per_step"), "chain", v.get("k_chain_dense_ms"), "verified", v.get("verified"), "hash_bytes", (v.get("customers_index") or {}).get("hash_bytes"), v.get("kernels_ms"))
PY
done
timeout 900 python bench.py --stream --rows 1000000000 --steps 2 --warmup 1 > gpurun_out/r5k/stream_1e9.json 2> gpurun_out/r5k/stream_1e9.err
echo "stream rc=$?"; tail -2 gpurun_out/r5k/stream_1e9.err; cat gpurun_out/r5k/stream_1e9.json
true

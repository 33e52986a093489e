#!/bin/bash
# round 4, run j: alphabets from a sample (stats_sample): tests, then A/B on the bench step and the 1e8 builds
set -x
mkdir -p gpurun_out/r4j
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_guard.py tests/test_gpu_chain.py -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r4j/pytest.txt
tail -3 gpurun_out/r4j/pytest.txt
FAST="--steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-traffic --no-positions --no-calibration"
for v in 1 0; do
  timeout 600 python bench.py $FAST --ctx-option stats_sample=$v 2>gpurun_out/r4j/bench_$v.err | tail -1 > gpurun_out/r4j/bench_$v.json
  python - <<PY
import json
d = json.load(open("gpurun_out/r4j/bench_$v.json"))
k = d["kernels"]
print("stats_sample=$v ms_per_step", round(d["ms_per_step"], 4), "verified", d.get("verified"), {n: round(v["total_ms"] / 3, 4) for n, v in k.items() if "stats" in n or "split" in n or "encode" in n})
for name, b in d.get("index_on_1e8", {}).items():
    if isinstance(b, dict) and "ms" in b:
        print("  index_on_1e8", name, b["ms"], b.get("verified"), {n: round(v["total_ms"], 3) for n, v in b.get("kernels", {}).items()})
PY
done
true

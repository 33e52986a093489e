#!/bin/bash
# round 4, run l: two-level direct sort: tests, then A/B (1 = partition + windowed scatter, 3 = plain random scatter, 0 = radix)
set -x
mkdir -p gpurun_out/r4l
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_guard.py -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r4l/pytest.txt
tail -3 gpurun_out/r4l/pytest.txt
FAST="--steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-traffic --no-positions --no-calibration"
for v in 1 3 0; do
  timeout 600 python bench.py $FAST --ctx-option direct_sort=$v 2>gpurun_out/r4l/bench_$v.err | tail -1 > gpurun_out/r4l/bench_$v.json
  python - <<PY
import json
d = json.load(open("gpurun_out/r4l/bench_$v.json"))
k = d["kernels"]
print("direct_sort=$v ms_per_step", round(d["ms_per_step"], 4), "verified", d.get("verified"), {n: round(v["total_ms"] / (d["steps"] if v.get("timed_region") else 3), 4) for n, v in k.items()})
for name, b in d.get("index_on_1e8", {}).items():
    if isinstance(b, dict) and "ms" in b:
        print("  index_on_1e8", name, b["ms"], b.get("verified"), b["frac_pass_model"], b["kernels_ms"])
PY
done
true

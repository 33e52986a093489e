#!/bin/bash
# round 4, run o: the encode kernel fills the slots of a full code space itself (direct_sort 1 vs 3)
mkdir -p gpurun_out/r4o
timeout 70 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "direct or config2 or alphabets or build_many or config1" 2>&1 | tail -3 > gpurun_out/r4o/pytest.txt
tail -2 gpurun_out/r4o/pytest.txt
FAST="--steps 40 --warmup 5 --no-cpu-baseline --no-index-1e8 --no-e2e --no-traffic --no-positions --no-calibration"
for v in 1 3; do
  timeout 60 python bench.py $FAST --ctx-option direct_sort=$v 2>gpurun_out/r4o/bench_$v.err | tail -1 > gpurun_out/r4o/bench_$v.json
  python - <<PY
import json
d = json.load(open("gpurun_out/r4o/bench_$v.json"))
print("direct_sort=$v ms_per_step", round(d["ms_per_step"], 4), "verified", d.get("verified"), {n: round(v["total_ms"] / (d["steps"] if v.get("timed_region") else 3), 4) for n, v in d["kernels"].items() if "direct" in n or "encode" in n})
PY
done
true

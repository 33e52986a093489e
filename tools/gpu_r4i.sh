#!/bin/bash
# round 4, run i: two-stream build batches + wave-parallel look-back scan: new tests, then the A/B on the bench step
set -x
mkdir -p gpurun_out/r4i
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_guard.py -m gpu -q -x -k "build_many or scan_lookback or guard or config" 2>&1 | tail -8 > gpurun_out/r4i/pytest.txt
tail -3 gpurun_out/r4i/pytest.txt
FAST="--steps 40 --warmup 5 --no-cpu-baseline --no-index-1e8 --no-e2e --no-traffic --no-verify --no-positions --no-calibration"
for cfg in "1 1" "0 0" "1 0" "0 1" "1 1"; do
  set -- $cfg
  timeout 300 python bench.py $FAST --ctx-option build_side_stream=$1 --ctx-option scan_lookback=$2 2>gpurun_out/r4i/bench_$1$2.err | tail -1 > gpurun_out/r4i/bench_$1$2.json
  python - <<PY
import json
d = json.load(open("gpurun_out/r4i/bench_$1$2.json"))
k = d["kernels"]
print("side=$1 lookback=$2 ms_per_step", round(d["ms_per_step"], 4), "kernel_ms", d["kernel_ms_per_step"], "scan", {n: round(v["total_ms"] / 3, 4) for n, v in k.items() if "scan" in n})
PY
done
true

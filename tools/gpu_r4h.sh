#!/bin/bash
# round 4, run h: the whole GPU suite (new: one-launch scan, cph_dist_join_chain over loopback / NCCL stand-in / forced one-rank
# bench, host-formed codes) + the scan A/B on the bench step
set -x
mkdir -p gpurun_out/r4h
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > gpurun_out/r4h/pytest.txt
tail -3 gpurun_out/r4h/pytest.txt
FAST="--steps 30 --warmup 5 --no-cpu-baseline --no-index-1e8 --no-e2e --no-traffic --no-verify --no-positions --no-calibration"
for v in 1 0 1; do
  timeout 300 python bench.py $FAST --ctx-option scan_lookback=$v 2>gpurun_out/r4h/bench_scan$v.err | tail -1 > gpurun_out/r4h/bench_scan$v.json
  python - <<PY
import json
d = json.load(open("gpurun_out/r4h/bench_scan$v.json"))
k = d["kernels"]
print("scan_lookback=$v ms_per_step", round(d["ms_per_step"], 4), "scan", {n: (v["launches"], round(v["total_ms"], 4)) for n, v in k.items() if "scan" in n})
PY
done
cp gpurun_out/bench_first_attempt_failure.txt gpurun_out/r4h/ 2>/dev/null
true

#!/bin/bash
# round 5, run s: rank table written by the window sort (k_win_place), no k_build_ranktab at the first Join
mkdir -p gpurun_out/r5s
timeout 900 python -m pytest tests/test_gpu_window_sort.py tests/test_gpu_parity.py tests/test_gpu_chain.py tests/test_gpu_stream.py tests/test_gpu_host_build.py -m gpu -q -x 2>&1 | tail -8
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-index-1e8 --no-traffic --no-calibration --variants half,itoa > gpurun_out/r5s/bench.out 2> gpurun_out/r5s/bench.err
echo "bench rc=$?"; tail -2 gpurun_out/r5s/bench.err
tail -1 gpurun_out/r5s/bench.out > gpurun_out/r5s/bench.json
python - <<PY
import json
d = json.load(open("gpurun_out/r5s/bench.json"))
print("ms_per_step", round(d["ms_per_step"], 4), "verified", d["verified"], {k: v["avg_ms"] for k, v in d["kernels"].items()})
for k, v in d["variants"].items():
    if isinstance(v, dict):
        print("   ", k, v.get("ms_per_step"), "chain", v.get("k_chain_dense_ms"), "verified", v.get("verified"), v.get("kernels_ms"))
PY
true

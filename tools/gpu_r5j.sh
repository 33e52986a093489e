#!/bin/bash
# round 5, run j: pre-joined build sides (chain steps keyed by an earlier build table), the L2 gather policy sweep
mkdir -p gpurun_out/r5j
timeout 900 python -m pytest tests/test_gpu_chain_sources.py tests/test_gpu_chain.py tests/test_host_cpp.py tests/test_gpu_split_codec.py -m gpu -q -x 2>&1 | tail -15
timeout 600 python bench.py --no-cpu-baseline --no-e2e --no-index-1e8 --no-traffic --no-calibration --variants side,half > gpurun_out/r5j/bench.out 2> gpurun_out/r5j/bench.err
echo "bench rc=$?"; tail -3 gpurun_out/r5j/bench.err
tail -1 gpurun_out/r5j/bench.out > gpurun_out/r5j/bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5j/bench.json"))
for k, v in d["variants"].items():
    if isinstance(v, dict):
        print(k, v.get("ms_per_step"), "chain", v.get("k_chain_dense_ms"), "verified", v.get("verified"), v.get("kernels_ms"))
PY
hipcc --offload-arch=gfx950 -O3 tools/microbench/l2_gather.hip -o /tmp/l2_gather && timeout 300 /tmp/l2_gather > gpurun_out/r5j/l2_gather.txt 2>&1
cat gpurun_out/r5j/l2_gather.txt
true

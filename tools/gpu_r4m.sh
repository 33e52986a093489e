#!/bin/bash
# round 4, run m: the host encoder on the block pool + AVX2 loop: its tests, then the two end-to-end scopes of the bench line
set -x
mkdir -p gpurun_out/r4m
timeout 600 python -m pytest tests/test_gpu_stream.py tests/test_host_cpp.py -q -x 2>&1 | tail -6 > gpurun_out/r4m/pytest.txt
tail -3 gpurun_out/r4m/pytest.txt
tests/cpp/test_host_encode > gpurun_out/r4m/host_encode_rates.txt 2>&1
cat gpurun_out/r4m/host_encode_rates.txt
timeout 600 python bench.py --steps 5 --no-cpu-baseline --no-index-1e8 --no-traffic --no-calibration --no-positions --no-verify 2>gpurun_out/r4m/bench.err | tail -1 > gpurun_out/r4m/bench.json
python - <<PY
import json
d = json.load(open("gpurun_out/r4m/bench.json"))
print("step", d["ms_per_step"])
for k in ("e2e_pinned_host", "e2e_pinned_host_encoded"):
    print(k, {a: b for a, b in d[k].items() if a != "scope"})
PY
true

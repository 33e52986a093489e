#!/bin/bash
# round 5, run e: IndexOn from host memory through host-formed codes (host_encode.hip: build_from_host_codes)
mkdir -p gpurun_out/r5e
timeout 600 python -m pytest tests/test_gpu_host_build.py -m gpu -q -x 2>&1 | tail -25 > gpurun_out/r5e/pytest.txt
tail -25 gpurun_out/r5e/pytest.txt
nproc
FAST="--steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-positions --no-calibration --no-variants"
timeout 300 python bench.py $FAST 2>gpurun_out/r5e/bench.err | tail -1 > gpurun_out/r5e/bench.json
tail -5 gpurun_out/r5e/bench.err
python tools/bench_summary.py gpurun_out/r5e/bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5e/bench.json"))
print(json.dumps(d["index_on_1e8"].get("e2e_pinned_host"), indent=1)[:2500])
PY
true

#!/bin/bash
# round 5, run g: the driver's bench line (all blocks), the rocprofv3 passes behind profiles/r05/, then the whole GPU suite
set -x
mkdir -p gpurun_out/r5g
timeout 900 python bench.py > gpurun_out/r5g/bench.out 2> gpurun_out/r5g/bench.err
echo "bench rc=$?"
tail -1 gpurun_out/r5g/bench.out > gpurun_out/r5g/bench.json
python tools/show_bench.py gpurun_out/r5g/bench.json 2>&1 | head -80
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5g/bench.json"))
print(json.dumps(d.get("variants"), indent=1)[:6000])
PY
if [ "$1" = "profile" ]; then
timeout 900 bash tools/gpu_profile.sh r05 > gpurun_out/r5g/profile.log 2>&1
tail -40 gpurun_out/r5g/profile.log
fi
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r5g/pytest.txt
tail -3 gpurun_out/r5g/pytest.txt
cp gpurun_out/bench_first_attempt_failure.txt gpurun_out/r5g/ 2>/dev/null
true

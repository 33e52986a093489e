#!/bin/bash
# round 4: the driver's bench line (all blocks), the rocprofv3 passes behind profiles/r04/, then the whole GPU suite
set -x
mkdir -p gpurun_out/r4final
timeout 900 python bench.py > gpurun_out/r4final/bench.out 2> gpurun_out/r4final/bench.err
echo "bench rc=$?"
tail -1 gpurun_out/r4final/bench.out > gpurun_out/r4final/bench.json
python tools/show_bench.py gpurun_out/r4final/bench.json 2>&1 | head -80
if [ "$1" = "profile" ]; then
timeout 900 bash tools/gpu_profile.sh r04 > gpurun_out/r4final/profile.log 2>&1
tail -30 gpurun_out/r4final/profile.log
fi
timeout 200 python tools/microbench/sort_digits.py > gpurun_out/r4final/sort_digits.txt 2>&1
grep rbits gpurun_out/r4final/sort_digits.txt
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r4final/pytest.txt
tail -3 gpurun_out/r4final/pytest.txt
cp gpurun_out/bench_first_attempt_failure.txt gpurun_out/r4final/ 2>/dev/null
true

#!/bin/bash
# round 5, run b: the LDS-window direct sort (direct_sort=1) against the plain scatter (4); parity of the build paths first
mkdir -p gpurun_out/r5b
timeout 300 python -m pytest tests/test_gpu_window_sort.py tests/test_gpu_parity.py tests/test_gpu_chain.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r5b/pytest.txt
tail -5 gpurun_out/r5b/pytest.txt
FAST="--steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-traffic --no-positions --no-calibration --no-variants"
for v in 1 4; do
  timeout 120 python bench.py $FAST --ctx-option direct_sort=$v 2>gpurun_out/r5b/bench_$v.err | tail -1 > gpurun_out/r5b/bench_$v.json
  tail -3 gpurun_out/r5b/bench_$v.err
  python tools/bench_summary.py gpurun_out/r5b/bench_$v.json
done
true

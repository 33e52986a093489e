#!/bin/bash
# round 5, run i: register-resident window partition + split codec from the sample alone: their tests, then the index figures
mkdir -p gpurun_out/r5i
timeout 900 python -m pytest tests/test_gpu_window_sort.py tests/test_gpu_split_codec.py tests/test_gpu_host_build.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -15
timeout 600 python bench.py --no-cpu-baseline --no-e2e --no-variants --no-traffic --no-calibration > gpurun_out/r5i/bench.out 2> gpurun_out/r5i/bench.err
echo "bench rc=$?"; tail -3 gpurun_out/r5i/bench.err
tail -1 gpurun_out/r5i/bench.out > gpurun_out/r5i/bench.json
python tools/show_bench.py gpurun_out/r5i/bench.json 2>&1 | head -30
true

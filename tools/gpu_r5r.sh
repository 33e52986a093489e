#!/bin/bash
# round 5, run r: chains of up to 8 Joins per call (CPH_MAX_CHAIN 4 -> 8), host encoder with streaming stores
mkdir -p gpurun_out/r5r
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_chain_sources.py tests/test_host_cpp.py tests/test_gpu_stream.py tests/test_gpu_dist.py tests/test_gpu_dist_standin.py tests/test_gpu_hypothesis.py -m gpu -q -x 2>&1 | tail -8
timeout 600 python bench.py --no-cpu-baseline --no-index-1e8 --no-traffic --no-calibration --no-variants > gpurun_out/r5r/bench.out 2> gpurun_out/r5r/bench.err
echo "bench rc=$?"; tail -2 gpurun_out/r5r/bench.err
tail -1 gpurun_out/r5r/bench.out > gpurun_out/r5r/bench.json
python tools/show_bench.py gpurun_out/r5r/bench.json 2>&1 | grep "e2e\|value"
true

#!/bin/bash
# round 5, run c: chains whose later key sits on a build side (cph_chain_step.source): parity, the C++ facade's TestLongChain, bench sanity
mkdir -p gpurun_out/r5c
timeout 600 python -m pytest tests/test_gpu_chain_sources.py tests/test_gpu_chain.py tests/test_host_cpp.py -m gpu -q -x 2>&1 | tail -25 > gpurun_out/r5c/pytest.txt
tail -25 gpurun_out/r5c/pytest.txt
./tests/cpp/test_host 2>&1 | tail -20 > gpurun_out/r5c/test_host.txt
tail -20 gpurun_out/r5c/test_host.txt
FAST="--steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-traffic --no-positions --no-calibration --no-variants --no-index-1e8"
timeout 120 python bench.py $FAST 2>gpurun_out/r5c/bench.err | tail -1 > gpurun_out/r5c/bench.json
python tools/bench_summary.py gpurun_out/r5c/bench.json
true

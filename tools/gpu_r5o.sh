#!/bin/bash
# round 5, run o: differential fuzz of this round's paths (tools/fuzz_round5.py) and of the older ones again
mkdir -p gpurun_out/r5o
timeout 400 python tools/fuzz_round5.py 150 501 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5o/fuzz_round5.txt | tail -5
timeout 200 python tools/fuzz_gpu.py 60 502 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5o/fuzz_gpu.txt | tail -3
timeout 200 python tools/fuzz_builds.py 60 503 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5o/fuzz_builds.txt | tail -3
true

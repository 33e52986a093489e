#!/bin/bash
# round 5, run w: a longer fuzz session on the final tree, other seeds
mkdir -p gpurun_out/r5w
timeout 500 python tools/fuzz_round5.py 230 9051 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5w/fuzz_round5.txt | tail -3
timeout 300 python tools/fuzz_gpu.py 110 9052 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5w/fuzz_gpu.txt | tail -2
timeout 300 python tools/fuzz_builds.py 110 9053 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5w/fuzz_builds.txt | tail -2
true

#!/bin/bash
# usage: tools/gpu_round2.sh <tag>   — full GPU test suite, smoke, default bench, rocprofv3 passes
TAG=${1:-r01}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
tail -2 gpurun_out/smoke.log | cut -c1-200
timeout 900 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"
tail -1 gpurun_out/bench_$TAG.json | cut -c1-400
bash tools/gpu_profile.sh $TAG

#!/bin/bash
# One gpurun call: GPU parity tests, smoke, a small and the full bench. Logs under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== rocminfo"; /opt/rocm/bin/rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8
  echo "== nproc $(nproc)  mem $(free -g | awk '/Mem/{print $2}') GiB"
} > gpurun_out/env.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
tail -3 gpurun_out/smoke.log
timeout 300 python bench.py --rows 10000000 --customers 1000000 --products 10000 --steps 3 --warmup 1 --cpu-sample-rows 500000 > gpurun_out/bench_small.log 2>&1; echo "rc=$?" >> gpurun_out/bench_small.log
tail -5 gpurun_out/bench_small.log | cut -c1-1500
timeout 900 python bench.py > gpurun_out/bench_full.log 2>&1; echo "rc=$?" >> gpurun_out/bench_full.log
tail -3 gpurun_out/bench_full.log | cut -c1-3000

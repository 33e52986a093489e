#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 900 python tools/microbench/chain_attrib.py > gpurun_out/chain_attrib.log 2>&1; echo "rc=$?" >> gpurun_out/chain_attrib.log
cat gpurun_out/chain_attrib.log | cut -c1-200
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_nocpu.log 2>&1; tail -1 gpurun_out/bench_nocpu.log | cut -c1-1200

#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_chain.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python tools/microbench/chain_variants.py > gpurun_out/chain_variants.log 2>&1; echo "rc=$?" >> gpurun_out/chain_variants.log
cat gpurun_out/chain_variants.log | cut -c1-300

#!/bin/bash
# round 5, run t: result blocks of per-batch Joins out of the ctx's cache of pinned blocks; the façade tests' run time
mkdir -p gpurun_out/r5t
timeout 300 python tools/microbench/batch_join.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5t/batch_join.txt
( time ./tests/cpp/test_host ) 2>&1 | tail -6
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_chain.py tests/test_gpu_chain_sources.py tests/test_host_cpp.py tests/test_index_ops.py tests/test_gpu_guard.py -m gpu -q -x 2>&1 | tail -4
true

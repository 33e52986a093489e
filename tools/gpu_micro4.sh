#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 900 python tools/microbench/stream_join.py 5e8 2>&1 | grep -v amdgpu.ids | tee gpurun_out/stream_join.log

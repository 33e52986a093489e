#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 900 python tools/microbench/generic_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/generic_probe.log | cut -c1-330

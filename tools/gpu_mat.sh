#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_materialize.py tests/test_csv_ingest.py -m gpu -x -q > gpurun_out/one.txt 2>&1
grep -v "dist-packages\|^  File \"/usr/lib" gpurun_out/one.txt | tail -30
timeout 600 python tools/microbench/pipeline.py 5e7 2>&1 | grep -v amdgpu.ids | tee gpurun_out/pipeline.txt | tail -20

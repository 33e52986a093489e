#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_csv_ingest.py tests/test_materialize.py -m gpu -x -q > gpurun_out/one.txt 2>&1
tail -3 gpurun_out/one.txt
timeout 600 python tools/microbench/csv_ingest.py 5e7 2>&1 | grep -v amdgpu.ids | tee gpurun_out/csv_ingest.txt | tail -3
timeout 600 python tools/microbench/pipeline.py 5e7 2>&1 | grep -v amdgpu.ids | tee gpurun_out/pipeline.txt | tail -18

#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_csv_ingest.py -m gpu -x -q > gpurun_out/csv_tests.txt 2>&1
grep -v "site-packages\|dist-packages" gpurun_out/csv_tests.txt | head -60
timeout 600 python tools/microbench/csv_ingest.py 5e7 2>&1 | tee gpurun_out/csv_ingest.txt | tail -8

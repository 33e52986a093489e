#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_codec_groups.py tests/test_gpu_parity.py tests/test_gpu_hypothesis.py tests/test_index_ops.py tests/test_csv_ingest.py -m gpu -x -q > gpurun_out/one.txt 2>&1
grep -v "dist-packages\|^  File \"/usr/lib" gpurun_out/one.txt | tail -30
bash tools/gpu_index1e8.sh

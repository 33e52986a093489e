#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/one.txt 2>&1
tail -3 gpurun_out/one.txt
timeout 600 python tools/microbench/csv_ingest.py 5e7 2>&1 | grep -v amdgpu.ids | tail -3
timeout 600 python bench.py --no-cpu-baseline --no-index-1e8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'scan', d['kernels'].get('exclusive_scan_u32'))"

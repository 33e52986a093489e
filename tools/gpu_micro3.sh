#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_nocpu.log 2>&1; tail -1 gpurun_out/bench_nocpu.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms_per_step', d['ms_per_step'], 'kernel ms', d['kernel_ms_per_step'], 'value', d['value'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['total_ms']): print(' ', k, v)
print(d['roofline'])
for k,v in d.get('index_on_1e8',{}).items(): print(k, {a:b for a,b in v.items() if a!='info'}); print('   ', v['info'])
"

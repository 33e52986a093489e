#!/bin/bash
# IndexOn at 1e8 rows (unique fixed8 ids; config-3 var-length duplicate keys) with the per-kernel breakdown
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/bench_idx.json 2> gpurun_out/bench_idx.err
tail -1 gpurun_out/bench_idx.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms_per_step', d['ms_per_step'])
for k,v in d.get('index_on_1e8',{}).items():
    print(k, {a:b for a,b in v.items() if a not in ('info','kernels_ms')}); print('   ', v['info']); print('   ', v['kernels_ms'])
" | tee gpurun_out/index1e8.txt

#!/bin/bash
# quick check after touching the chained-join kernel: its parity tests, then the kernel's time at the bench shape
# usage: tools/gpu_chain_quick.sh [tag]
export TMPDIR=/tmp
TAG=${1:-quick}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_stream.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/chain_tests_$TAG.txt 2>&1
tail -5 gpurun_out/chain_tests_$TAG.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-index-1e8 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
python - <<PY
import json
d = json.load(open("gpurun_out/bench_$TAG.json"))
print("ms_per_step", round(d["ms_per_step"], 3), "kernel_ms", d["kernel_ms_per_step"], "roofline", d["roofline"]["frac"])
for k, v in d["kernels"].items():
    print(f"  {k:28s} {v['avg_ms']:9.4f} ms x {v['launches']}")
PY

#!/bin/bash
# round 5, run p: match total reported by k_chain_dense itself, alphabets of variable-length ids from the split decision's sample
mkdir -p gpurun_out/r5p
timeout 900 python -m pytest tests/test_gpu_split_codec.py tests/test_gpu_chain.py tests/test_gpu_chain_sources.py tests/test_gpu_parity.py tests/test_gpu_stream.py tests/test_host_cpp.py -m gpu -q -x 2>&1 | tail -8
for opt in chain_fused_total=1 chain_fused_total=0; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-index-1e8 --no-traffic --no-calibration --variants itoa --ctx-option $opt > gpurun_out/r5p/bench_$opt.out 2> gpurun_out/r5p/bench_$opt.err
echo "bench rc=$?"; tail -2 gpurun_out/r5p/bench_$opt.err
tail -1 gpurun_out/r5p/bench_$opt.out > gpurun_out/r5p/bench_$opt.json
python - <<PY
import json
d = json.load(open("gpurun_out/r5p/bench_$opt.json"))
print("$opt", "ms_per_step", round(d["ms_per_step"], 4), "verified", d["verified"], {k: v["avg_ms"] for k, v in d["kernels"].items()})
for k, v in d["variants"].items():
    if isinstance(v, dict):
        print("   ", k, v.get("ms_per_step"), "chain", v.get("k_chain_dense_ms"), "verified", v.get("verified"), v.get("kernels_ms"))
PY
done
true

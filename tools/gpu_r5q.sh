#!/bin/bash
# round 5, run q: keys coded inside the first partition level of the window sort (k_win_partition<2>)
mkdir -p gpurun_out/r5q
timeout 900 python -m pytest tests/test_gpu_window_sort.py tests/test_gpu_parity.py tests/test_gpu_host_build.py tests/test_gpu_streams.py -m gpu -q -x 2>&1 | tail -8
for opt in direct_fused_encode=1 direct_fused_encode=0; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-traffic --no-calibration --no-variants --ctx-option $opt > gpurun_out/r5q/bench_$opt.out 2> gpurun_out/r5q/bench_$opt.err
echo "bench rc=$?"; tail -2 gpurun_out/r5q/bench_$opt.err
tail -1 gpurun_out/r5q/bench_$opt.out > gpurun_out/r5q/bench_$opt.json
python - <<PY
import json
d = json.load(open("gpurun_out/r5q/bench_$opt.json"))
print("$opt", "ms_per_step", round(d["ms_per_step"], 4), "verified", d["verified"], {k: v["avg_ms"] for k, v in d["kernels"].items()})
u = d["index_on_1e8"]["unique_fixed8_ids"]
print("   unique_fixed8_ids", u["ms"], u["kernels_ms"], "pass", u["frac_pass_model"], "verified", u["verified"], u["byte_terms"])
print("   step_frac", d["roofline"].get("step_frac"), d.get("index_plus_join_1e8", {}).get("unique_fixed8_ids_plus_step"))
PY
done
true

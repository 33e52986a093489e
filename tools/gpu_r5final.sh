#!/bin/bash
# round 5: the driver's command line (bench --gpus 1 --steps 20 --warmup 5), smoke, the rocprofv3 passes behind profiles/r05/, the whole GPU suite,
# one more minute of build fuzz (8-byte padded dense tables: keys coded inside the window sort)
set -x
mkdir -p gpurun_out/r5final
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
/usr/bin/time -v timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5final/bench.out 2> gpurun_out/r5final/bench.err
echo "bench rc=$?"; grep "Elapsed (wall clock)" gpurun_out/r5final/bench.err
tail -1 gpurun_out/r5final/bench.out > gpurun_out/r5final/bench.json
python tools/show_bench.py gpurun_out/r5final/bench.json 2>&1 | head -40
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5final/bench.json"))
for k, v in d["variants"].items():
    if isinstance(v, dict):
        print(k, v.get("ms_per_step"), "chain", v.get("k_chain_dense_ms"), "frac", (v.get("roofline") or {}).get("frac"), "verified", v.get("verified"))
print(json.dumps(d["index_plus_join_1e8"])[:400])
PY
if [ "$1" = "profile" ]; then
timeout 900 bash tools/gpu_profile.sh r05 > gpurun_out/r5final/profile.log 2>&1
tail -12 gpurun_out/r5final/profile.log
fi
timeout 200 python tools/fuzz_builds.py 60 504 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5final/fuzz_builds.txt | tail -2
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r5final/pytest.txt
tail -3 gpurun_out/r5final/pytest.txt
true

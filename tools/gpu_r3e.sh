set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_stream.py tests/test_gpu_dist.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -30 > $O/pytest.txt
timeout 600 python bench.py --no-cpu-baseline --no-index-1e8 --no-e2e --no-traffic > $O/bench.json 2> $O/bench.err
tail -5 $O/pytest.txt; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3e/bench.json'))
print("main ms", d["ms_per_step"], "verified", d.get("verified"))
print(json.dumps(d.get("join_positions"), indent=1))
PY

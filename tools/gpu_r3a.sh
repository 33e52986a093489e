set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/r3a/pytest.txt
timeout 600 python tools/microbench/hash_probe.py > gpurun_out/r3a/hash_probe.txt 2>&1
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-index-1e8 --no-e2e > gpurun_out/r3a/bench_small.json 2> gpurun_out/r3a/bench_small.err
tail -5 gpurun_out/r3a/pytest.txt
cat gpurun_out/r3a/hash_probe.txt

set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_stream.py tests/test_gpu_hash_probe.py -m gpu -x -q --tb=short > $O/pytest_chain.txt 2>&1
tail -5 $O/pytest_chain.txt
timeout 600 python tools/microbench/chain_lean.py > $O/chain_lean.txt 2>&1; grep -v amdgpu $O/chain_lean.txt
BENCH_EXTRA="" timeout 600 bash tools/gpu_pmc_bench.sh > $O/pmc_bench.txt 2>&1; tail -12 $O/pmc_bench.txt
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-index-1e8 --no-e2e --no-traffic > $O/bench.json 2> $O/bench.err; python tools/show_bench.py $O/bench.json 2>/dev/null | head -40

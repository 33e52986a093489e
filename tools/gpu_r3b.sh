set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > $O/pytest.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 300 python tools/microbench/multicol_index.py > $O/multicol.txt 2>&1
timeout 300 python tools/microbench/reference_benchmarks.py 1 100 10000 > $O/refbench.txt 2>&1
timeout 300 python tools/microbench/hash_probe.py > $O/hash_probe.txt 2>&1
tail -5 $O/pytest.txt; tail -3 $O/bench.err; cat $O/multicol.txt; tail -30 $O/refbench.txt

set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > $O/pytest.txt
timeout 300 python tools/microbench/reference_benchmarks.py 1 100 > $O/refbench.txt 2>&1
tail -8 $O/pytest.txt; cat $O/refbench.txt

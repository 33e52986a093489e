set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4d; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --tb=short > $O/pytest.txt 2>&1
tail -8 $O/pytest.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; python tools/show_bench.py $O/bench.json 2>/dev/null | head -60

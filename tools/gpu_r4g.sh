set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4g; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --tb=short > $O/pytest.txt 2>&1
tail -8 $O/pytest.txt
timeout 500 python tools/fuzz_gpu.py 330 20260921 > $O/fuzz1.txt 2>&1; tail -2 $O/fuzz1.txt

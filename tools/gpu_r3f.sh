set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3f; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q --tb=short > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
timeout 300 python tools/microbench/multicol_index.py > $O/multicol.txt 2>&1; grep -v amdgpu $O/multicol.txt
timeout 300 python tools/microbench/join_itoa.py > $O/join_itoa.txt 2>&1; grep -v amdgpu $O/join_itoa.txt | tail -12

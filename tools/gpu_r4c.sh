set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_split_codec.py -m gpu -x -q --tb=short > $O/pytest_split.txt 2>&1
tail -5 $O/pytest_split.txt
timeout 600 python tools/microbench/config3.py 1e8 2 > $O/config3.txt 2>&1; grep -v "amdgpu\|codec_try" $O/config3.txt | tail -5
ROWS=1e8 bash tools/gpu_pmc_config3.sh > $O/pmc.txt 2>&1; grep -v "^+" $O/pmc.txt | head -6

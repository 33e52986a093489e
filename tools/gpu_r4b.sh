set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_split_codec.py -m gpu -x -q --tb=short > $O/pytest_split.txt 2>&1
tail -30 $O/pytest_split.txt
timeout 900 python -m pytest tests/test_gpu_codec_groups.py tests/test_gpu_parity.py tests/test_index_ops.py -m gpu -x -q --tb=short > $O/pytest_more.txt 2>&1
tail -5 $O/pytest_more.txt
timeout 600 python tools/microbench/config3.py > $O/config3.txt 2>&1; grep -v amdgpu $O/config3.txt | tail -30

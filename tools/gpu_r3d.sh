set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_stream.py tests/test_abi.py tests/test_gpu_chain.py -m gpu -x -q 2>&1 | tail -30 > $O/pytest.txt
timeout 600 python tools/microbench/stream_general.py > $O/stream_general.txt 2>&1
tail -8 $O/pytest.txt; cat $O/stream_general.txt

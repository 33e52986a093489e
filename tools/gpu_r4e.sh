set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4e; mkdir -p $O
timeout 900 python -m pytest tests/test_host_cpp.py tests/test_csv_ingest.py tests/test_materialize.py -m gpu -x -q --tb=short > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
./tests/cpp/test_host 2>&1 | tail -14
timeout 600 python tools/microbench/pipeline.py > $O/pipeline.txt 2>&1; grep -v amdgpu $O/pipeline.txt | tail -45

set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4f; mkdir -p $O
timeout 900 python tools/fuzz_gpu.py 240 20260921 > $O/fuzz1.txt 2>&1; tail -3 $O/fuzz1.txt
timeout 600 python tools/fuzz_gpu.py 120 4242 > $O/fuzz2.txt 2>&1; tail -3 $O/fuzz2.txt

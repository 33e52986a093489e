#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/microbench/gather_bw.hip -o /tmp/gather_bw && timeout 300 /tmp/gather_bw > gpurun_out/gather_bw.log 2>&1
cat gpurun_out/gather_bw.log
timeout 900 python tools/microbench/chain_variants.py > gpurun_out/chain_variants.log 2>&1; echo "rc=$?" >> gpurun_out/chain_variants.log
cat gpurun_out/chain_variants.log | cut -c1-400

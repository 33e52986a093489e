#!/bin/bash
# usage: tools/gpu_one.sh <pytest args...>   (output kept in gpurun_out/one.txt)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest "$@" > gpurun_out/one.txt 2>&1
grep -v "dist-packages\|^  File \"/usr/lib" gpurun_out/one.txt | tail -60

#!/bin/bash
# the whole GPU suite with the device pool in canary mode (tests/conftest.py: CPH_POOL_GUARD=1)
export TMPDIR=/tmp
mkdir -p gpurun_out
CPH_POOL_GUARD=1 timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/pool_guard.txt 2>&1
tail -5 gpurun_out/pool_guard.txt

#!/bin/bash
# round 5, run d: stream operations of the step cut (report words in pinned host memory, self-cleaning accumulators):
# the whole GPU suite, the bench step, and a kernel trace that counts the fills / copies per step
mkdir -p gpurun_out/r5d
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r5d/pytest.txt
tail -5 gpurun_out/r5d/pytest.txt
FAST="--steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-traffic --no-positions --no-calibration --no-variants --no-index-1e8"
timeout 120 python bench.py $FAST 2>gpurun_out/r5d/bench.err | tail -1 > gpurun_out/r5d/bench.json
python tools/bench_summary.py gpurun_out/r5d/bench.json
ROOTDIR=$(pwd)
OUT=$ROOTDIR/gpurun_out/r5d/trace
ARGS="--steps 8 --warmup 2 --no-cpu-baseline --no-index-1e8 --no-verify --no-e2e --no-traffic --no-positions --no-calibration --no-variants"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python $ROOTDIR/bench.py $ARGS > $ROOTDIR/gpurun_out/r5d/trace.log 2>&1)
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r5d/kernel_stats.csv
find $OUT -size +5M -delete
head -40 gpurun_out/r5d/kernel_stats.csv | cut -c1-150
true

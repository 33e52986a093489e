#!/bin/bash
# rocprofv3 passes over bench.py (kernel trace + stats; PMC counters in their own runs).
# usage: tools/gpu_profile.sh <tag> [bench args...]
TAG=${1:-r01}; shift
mkdir -p gpurun_out/prof_$TAG
export TMPDIR=/tmp
ROOTDIR=$(pwd)
OUT=$ROOTDIR/gpurun_out/prof_$TAG
ARGS="--extras /dev/null --steps 3 --warmup 1 --no-cpu-baseline --no-index-1e8 --no-verify --no-e2e --no-traffic --no-positions --no-calibration $@"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $ROOTDIR/bench.py $ARGS > $OUT/trace.log 2>&1
echo "trace rc=$?" >> $OUT/trace.log
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -o bench -- python $ROOTDIR/bench.py $ARGS > $OUT/pmc_$C.log 2>&1
  echo "pmc $C rc=$?" >> $OUT/pmc_$C.log
done
cd $ROOTDIR
find $OUT -name "*.csv" | head -30
# keep the merged output small: drop anything above 20 MB
find $OUT -size +20M -delete
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt | head -60

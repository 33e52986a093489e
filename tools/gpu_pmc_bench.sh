#!/bin/bash
# SQ counters of the benchmark step (one --pmc pass, kernel trace only): which kernels wait, which issue
export TMPDIR=/tmp; ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/pmc_bench; mkdir -p $OUT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD \
  --output-format csv -d $OUT -o b -- python $ROOTDIR/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-index-1e8 --no-e2e --no-traffic --no-positions --no-verify --no-calibration $BENCH_EXTRA > $OUT/run.log 2>&1
echo "rc=$?" >> $OUT/run.log
cd $ROOTDIR
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('gpurun_out/pmc_bench/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void cph::', '')[:40]
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
names = ['SQ_WAVE_CYCLES','SQ_WAIT_ANY','SQ_WAIT_INST_ANY','SQ_ACTIVE_INST_ANY','SQ_INSTS_VMEM_RD','SQ_INSTS_VALU','SQ_INSTS_SALU','SQ_INSTS_LDS']
print('%-42s' % 'kernel (per launch)' + ''.join('%14s' % n[3:] for n in names))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0))[:8]:
    print('%-42s' % k + ''.join('%14.3g' % (v.get(n, 0) / max(1, cnt[(k, n)])) for n in names))
PY

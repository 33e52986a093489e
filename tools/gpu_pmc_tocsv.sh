#!/bin/bash
# SQ counters of the ToCsv kernels (one --pmc pass, kernel trace only).  Usage: gpu_pmc_tocsv.sh [rows] [modes]
export TMPDIR=/tmp; ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/pmc_tocsv; rm -rf $OUT; mkdir -p $OUT
ROWS=${1:-5e7}; MODES=${2:-0:0,1:0}
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU \
  --output-format csv -d $OUT -o tocsv -- python $ROOTDIR/tools/microbench/to_csv.py $ROWS $MODES > $OUT/run.log 2>&1
echo "rc=$?" >> $OUT/run.log
cd $ROOTDIR
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('gpurun_out/pmc_tocsv/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void cph::', '')[:40]
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
names = ['SQ_WAVE_CYCLES','SQ_WAIT_ANY','SQ_WAIT_INST_ANY','SQ_ACTIVE_INST_ANY','SQ_ACTIVE_INST_VALU','SQ_INSTS_VALU','SQ_INSTS_SALU','SQ_INSTS_LDS']
print('%-42s' % 'kernel (per launch)' + ''.join('%14s' % n[3:] for n in names))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0))[:10]:
    print('%-42s' % k + ''.join('%14.3g' % (v.get(n, 0) / max(1, cnt[(k, n)])) for n in names))
PY
tail -4 $OUT/run.log

#!/bin/bash
# SQ counters of the config-3 index build with the split codec (two --pmc passes, kernel trace only): ROWS rows (default 1e8)
export TMPDIR=/tmp; ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/pmc_config3; mkdir -p $OUT
ROWS=${ROWS:-1e8}
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU \
  --output-format csv -d $OUT -o a -- python $ROOTDIR/tools/microbench/config3.py $ROWS 1 > $OUT/run_a.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM \
  --output-format csv -d $OUT -o b -- python $ROOTDIR/tools/microbench/config3.py $ROWS 1 > $OUT/run_b.log 2>&1
cd $ROOTDIR
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('gpurun_out/pmc_config3/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void cph::', '')[:34]
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
names = sorted({n for v in acc.values() for n in v})
print('%-36s' % 'kernel (per launch)' + ''.join('%13s' % n[3:16] for n in names))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0))[:8]:
    print('%-36s' % k + ''.join('%13.3g' % (v.get(n, 0) / max(1, cnt[(k, n)])) for n in names))
PY

#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration against known byte counts (tools/microbench/pmc_calib.hip)
export TMPDIR=/tmp
ROOTDIR=$(pwd)
OUT=$ROOTDIR/gpurun_out/pmc_calib
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 tools/microbench/pmc_calib.hip -o /tmp/pmc_calib || exit 1
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -o calib -- /tmp/pmc_calib > $OUT/$C.log 2>&1
done
cd $ROOTDIR
python - <<'PY' | tee gpurun_out/pmc_calibration.txt
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
order = []
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/pmc_calib/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0] + "#" + r["Dispatch_Id"] if False else r["Kernel_Name"].split("(")[0]
            if r["Counter_Name"] == c:
                acc[k][c].append(float(r["Counter_Value"]) * 1024)
                if k not in order: order.append(k)
GiB = 1 << 30
print(f"{'kernel':42s} {'FETCH_SIZE bytes':>20s} {'WRITE_SIZE bytes':>20s}   (per launch; 1 GiB = {GiB})")
for k in order:
    f, w = acc[k].get("FETCH_SIZE", []), acc[k].get("WRITE_SIZE", [])
    print(f"{k:42s} {' / '.join(f'{x:.4g}' for x in f):>20s} {' / '.join(f'{x:.4g}' for x in w):>20s}")
PY

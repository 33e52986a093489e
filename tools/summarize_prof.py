#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (kernel trace + PMC passes) into a per-kernel table:
calls, total/avg duration, and FETCH_SIZE / WRITE_SIZE per launch (KB as reported; the gfx950
corrections of MI355X_MICROARCH.md §HBM are applied by the reader)."""
import csv
import sys
from collections import defaultdict
from pathlib import Path

root = Path(sys.argv[1])


def short(name):
    name = name.split("(")[0]
    for p in ("void cph::", "cph::"):
        if name.startswith(p):
            name = name[len(p):]
    return name[:70]


dur = defaultdict(lambda: [0, 0.0])
launches = defaultdict(list)      # kernel -> [(start_ns, duration_us)] in launch order
for f in root.glob("trace/**/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        dur[k][0] += 1
        dur[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        launches[k].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
# the per-launch durations of the kernel with the largest total (the one `roofline` is quoted on): small enough to be tracked
if dur:
    dom = max(((k, v) for k, v in dur.items() if not k.startswith("k_cal_")), key=lambda kv: kv[1][1], default=(None, None))[0] or max(dur.items(), key=lambda kv: kv[1][1])[0]
    rows = sorted(launches[dom])
    with open(root / "dominant_kernel_launches.csv", "w") as fh:
        fh.write(f"# {dom}\nlaunch,duration_us\n")
        for i, (_, d_us) in enumerate(rows):
            fh.write(f"{i},{d_us:.2f}\n")
pmc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in root.glob("pmc_*/**/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        c = r["Counter_Name"]
        pmc[k][c][0] += 1
        pmc[k][c][1] += float(r["Counter_Value"])
import json

traffic = {}
for k, (n, t) in dur.items():
    fs, ws = pmc[k].get("FETCH_SIZE"), pmc[k].get("WRITE_SIZE")
    traffic[k] = {"calls": n, "avg_us": t / n,
                  "fetch_bytes_per_launch": fs[1] / fs[0] * 1024 if fs else None,     # FETCH_SIZE is in KB
                  "write_bytes_per_launch": ws[1] / ws[0] * 1024 if ws else None}
(root / "traffic.json").write_text(json.dumps(traffic, indent=1, sort_keys=True))
print(f"{'kernel':70s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'FETCH_SIZE/launch':>18s} {'WRITE_SIZE/launch':>18s}")
for k, (n, t) in sorted(dur.items(), key=lambda kv: -kv[1][1]):
    fs = pmc[k].get("FETCH_SIZE")
    ws = pmc[k].get("WRITE_SIZE")
    print(f"{k:70s} {n:6d} {t:12.1f} {t / n:10.2f} {(fs[1] / fs[0] if fs else float('nan')):18.1f} "
          f"{(ws[1] / ws[0] if ws else float('nan')):18.1f}")

#!/usr/bin/env python3
"""Timeline of the launches between two kernels of a rocprofv3 --kernel-trace run: gap in front of every launch, its duration.
usage: kernel_gaps.py <trace dir> [last kernel substring = k_cs_window] [first kernel substring = k_split_count] [steps back = 0]
(the window ends at the LAST launch of `last`, `steps back` launches of it earlier, and starts at the nearest `first` before it)"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void cph::', '').replace('cph::', '')[:44]))
rows.sort()
last = sys.argv[2] if len(sys.argv) > 2 else 'k_cs_window'
first = sys.argv[3] if len(sys.argv) > 3 else 'k_split_count'
back = int(sys.argv[4]) if len(sys.argv) > 4 else 0
idx = [i for i, r in enumerate(rows) if last in r[2]]
end = idx[-1 - back]
start = max(i for i in range(end) if first in rows[i][2])
prev = None
gaps = 0.0
for s, e, n in rows[start:end + 1]:
    gap = (s - prev) / 1e3 if prev else 0.0
    gaps += gap
    print(f"{n:46s} gap {gap:8.1f} us   dur {(e - s) / 1e3:8.1f} us")
    prev = e
print(f"total {(rows[end][1] - rows[start][0]) / 1e3:.1f} us, gaps {gaps:.1f} us")

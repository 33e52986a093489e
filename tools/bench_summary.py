#!/usr/bin/env python3
"""Compact view of bench.py's JSON line(s): tools/bench_summary.py file.json [...]"""
import json
import sys

for path in sys.argv[1:]:
    try:
        d = json.loads(open(path).read().strip().splitlines()[-1])
    except Exception as ex:   # noqa: BLE001
        print(path, "unreadable:", ex)
        continue
    r = d.get("roofline") or {}
    print(f"== {path}: step {d['ms_per_step']:.4f} ms  value {d['value'] / 1e9:.1f} G rows/s  verified={d.get('verified')}  "
          f"dom {r.get('kernel')} {r.get('avg_launch_ms')} ms frac {r.get('frac')} step_frac {r.get('step_frac')} traffic {r.get('traffic')}")
    print("   kernels/step:", {k: round(v["avg_ms"] * v["launches"] / (d["steps"] if v.get("timed_region") else 3), 4) for k, v in (d.get("kernels") or {}).items()})
    for name in ("join_row_ids", "join_positions"):
        if d.get(name):
            print(f"   {name}: {d[name]['ms_per_step']} ms, k_chain_dense {d[name]['k_chain_dense_ms']}")
    for k, v in (d.get("variants") or {}).items():
        if "error" in v:
            print(f"   variant {k}: ERROR {v['error']}")
            continue
        print(f"   variant {k}: step {v['ms_per_step']} ms, k_chain_dense {v['k_chain_dense_ms']} ms, frac {(v.get('roofline') or {}).get('frac')}, verified={v.get('verified')}")
        print("      ", {a: b for a, b in list(v["kernels_ms"].items())[:9]})
    for k, v in (d.get("index_on_1e8") or {}).items():
        if isinstance(v, dict) and "ms" in v:
            print(f"   index_on_1e8.{k}: {v['ms']} ms kernel {v.get('kernel_ms')} frac_pass {v.get('frac_pass_model')} verified={v.get('verified')}")
            print("      ", v.get("kernels_ms"))
        elif isinstance(v, dict):
            for kk, vv in v.items():
                if isinstance(vv, dict) and "ms" in vv:
                    print(f"   index_on_1e8.{k}.{kk}: {vv['ms']} ms  pcie {vv.get('pcie_GBps')} GB/s")
    for k in ("e2e_pinned_host", "e2e_pinned_host_encoded"):
        if d.get(k):
            print(f"   {k}: {d[k].get('ms')} ms  {d[k].get('error', '')}")
    if d.get("cpu_baseline"):
        print("   cpu_baseline:", d["cpu_baseline"]["value"], d["cpu_baseline"]["unit"])

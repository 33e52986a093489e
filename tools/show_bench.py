#!/usr/bin/env python3
"""Condensed view of a bench.py JSON line: tools/show_bench.py gpurun_out/bench_x.json"""
import json
import sys

d = json.load(open(sys.argv[1]))
r = d.get("roofline") or {}
print("value %.3e %s | ms_per_step %.3f | kernel_ms %.3f | frac %.4f useful %s step_frac %s | verified %s" % (
    d["value"], d["unit"], d["ms_per_step"], d["kernel_ms_per_step"], r.get("frac", 0), r.get("useful"), r.get("step_frac"),
    d.get("verified")))
if r.get("traffic"):
    print("traffic %.3e (raw %s) algorithmic %.3e" % (r["traffic"], r.get("traffic_raw"), r.get("algorithmic_bytes_per_launch", 0)))
for k, v in d["kernels"].items():
    print(f"  {k:28s} {v['avg_ms']:9.4f} ms x {v['launches']:4d}  {v['GBps']} GB/s")
for k, v in (d.get("index_on_1e8") or {}).items():
    if "kernel_ms" not in v:
        print(k, {a: ({x: y for x, y in b.items() if x in ("ms", "pcie_GBps")} if isinstance(b, dict) else b) for a, b in v.items() if a != "scope"})
        continue
    print(k, "ms", v["ms"], "kernel_ms", v["kernel_ms"], "pass", v.get("frac_pass_model"), "verified", v.get("verified"))
    print("    ", v["kernels_ms"])
print("timed step output mode:", r.get("output_mode"), "|", (d.get("config") or {}).get("build_row_mode", "")[:60])
for name in ("join_positions", "join_row_ids"):
    jp = d.get(name)
    if jp:
        print("%s ms_per_step %.3f (timed step / this = %.2f) k_chain_dense %.3f ms frac %s over gather ceiling %s verified %s" % (
            name, jp["ms_per_step"], jp.get("timed_step_over_this", jp.get("speedup_vs_row_ids", 0)), jp["k_chain_dense_ms"],
            (jp.get("roofline") or {}).get("frac"), (jp.get("roofline") or {}).get("kernel_over_gather_ceiling"),
            (jp.get("verify") or {}).get("ok")))
gc = r.get("gather_ceiling")
if gc:
    print("gather ceiling %.3f ms (%s G lookups/s), kernel / ceiling %s, copy %s TB/s" % (gc["ms"], gc["Glookups_per_s"], gc.get("kernel_over_ceiling"), r.get("copy_TBps")))
for k in ("e2e_pinned_host", "e2e_pinned_host_encoded", "cpu_baseline"):
    if k in d:
        print(k, {a: b for a, b in d[k].items() if a not in ("scope", "sample", "variants", "extrapolated_full_size")})
for k, v in ((d.get("cpu_baseline") or {}).get("variants") or {}).items():
    print("   ", k, {a: b for a, b in v.items() if a != "sample"})

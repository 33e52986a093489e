#!/usr/bin/env python3
"""bench.py — the hot path on synthetic tables, one JSON line on rank 0.

A "step" = one pass of the hot path over one batch, BASELINE.json's headline shape
("10^8-row 3-col join"; configs[3] at one GPU): with all staged columns already
resident in HBM,

    customers.UniqueIndexOn("id")        (1e7 rows, 8-byte ids)      } cph_index_build_many
    products.UniqueIndexOn("prod_id")    (1e5 rows)                  } (one batch of two builds)
    orders.Join(customers,"cust_id").Join(products,"prod_id")       cph_join_chain_ex (one fused pass)
        over 1e8 orders rows x 3 string columns (cust_id, prod_id, qty)

The Join reports, per joined row and index, the SORTED POSITION of the matching index row — the reference's own row
handle (its Join reads index.impl.rows[first()+i], csvplus.go:553-567; the cgo shim and the C++ facade consume exactly
that).  The mode rounds 1-2 timed, original row ids (= perm[position]), is measured in the same run and reported beside
`value` as `join_row_ids`; `--row-ids` swaps the two.

and, for N > 1, the probe rows are split into N contiguous ranges (strong scaling:
the 1e8 rows are fixed), the build side is replicated, and the joined position lists are
allgatherv'ed over RCCL behind the C ABI (cph_dist_chain_allgather) so that every rank holds the
whole list in emission order.

value = joined rows per second of the whole job (max over ranks of the step time).

Launching: `python bench.py --gpus N` works as a plain command — without a launcher's WORLD_SIZE in the
environment it starts its own N ranks (one per visible GPU) under torch.distributed.run on 127.0.0.1; under a
launcher (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) it is one of the ranks.
For N > 1 the exchange that is timed is ALWAYS cph_dist_* (RCCL behind the C ABI): if the communicator cannot
be created on some rank, every rank exits non-zero — there is no silent fallback to another transport.  (The
one exception is the explicitly requested debug mode CPH_BENCH_SHARE_GPU=1, several ranks on one GPU over gloo,
which RCCL itself refuses; its line says so.)
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s HBM3E peak


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=None, help="orders (probe) rows, whole job (default 1e8; 1e9 with --stream)")
    ap.add_argument("--customers", type=int, default=10_000_000)
    ap.add_argument("--products", type=int, default=100_000)
    ap.add_argument("--exchange", choices=["allgatherv", "packed", "host", "oneshot", "none"], default="allgatherv",
                    help="N > 1: allgatherv = cph_dist_join_chain (sub-chunks of a shard leave over xGMI while the next one is joined; "
                         "every rank ends with the whole list in HBM); host = the same pipeline, each rank copying its chunks into ITS range "
                         "of one pinned host buffer shared by the ranks (no xGMI traffic, N PCIe links in parallel, SURVEY 8e); oneshot = join "
                         "the shard, then cph_dist_chain_allgather (rounds 1-3); none = every rank keeps its shard's list; packed = allgatherv with "
                         "CPH_DIST_PACKED (24 + 17 bits per row on the links instead of 64; packed and unpacked on the device)")
    ap.add_argument("--chunks", type=int, default=0, help="sub-chunks per shard for --exchange allgatherv / host (0: the library picks, 1..8)")
    ap.add_argument("--ctx-option", action="append", default=[], metavar="KEY=INT",
                    help="cph_ctx_set_option on the bench's ctx before anything runs (A/B switches: scan_lookback=0, chain_arith=0 ...)")
    ap.add_argument("--stream", action="store_true",
                    help="BASELINE config 5 shape: every rank streams ITS shard of the orders from pinned host memory through "
                         "cph_stream_join_* (chunks uploaded, joined and downloaded on overlapping HIP streams), results land in the "
                         "node's host memory; --rows defaults to 1e9 here.  PCIe inclusive: not the `value` of the default mode")
    ap.add_argument("--no-alternatives", action="store_true",
                    help="N > 1: skip the same-run figures of the other exchange modes and the measured exchange rate")
    ap.add_argument("--no-n1", action="store_true", help="N > 1: skip rank 0's one-GPU run of the whole stream behind efficiency_vs_n1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-index-1e8", action="store_true", help="skip the extra IndexOn-at-full-size measurements")
    ap.add_argument("--cpu-sample-rows", type=int, default=2_000_000)
    ap.add_argument("--no-verify", action="store_true", help="skip the full-size checks of the timed outputs")
    ap.add_argument("--row-ids", action="store_true",
                    help="the timed step reports ORIGINAL ROW IDS (cph_join_chain, the rounds 1-2 mode) instead of sorted positions; "
                         "by default the step reports sorted positions — the reference's own row handle, what its Join reads "
                         "(index.impl.rows[first()+i], csvplus.go:553-567) and what the cgo shim / the C++ facade consume — and the "
                         "row-id mode is measured beside it (join_row_ids)")
    ap.add_argument("--positions", action="store_true", help="(the default since round 3; accepted for old command lines)")
    ap.add_argument("--no-positions", action="store_true",
                    help="skip the second measurement of the step in the OTHER output mode (row ids by default)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the pinned-host -> pinned-host scope (cph_stream_join_*)")
    ap.add_argument("--no-variants", action="store_true",
                    help="skip the `variants` block: the same step (2 index builds + chained Join of all rows) on other key shapes — "
                         "unpadded Itoa ids, a half-occupied id space, sparse random keys — and with the payload columns laid out in "
                         "index order (cph_index_permute)")
    ap.add_argument("--variants", default="all", help="comma list out of: itoa,half,sparse,side,permute,dup (default all)")
    ap.add_argument("--verify-sample", type=int, default=100_000)
    ap.add_argument("--no-traffic", action="store_true",
                    help="skip the two rocprofv3 --pmc child runs (FETCH_SIZE, WRITE_SIZE) behind roofline.traffic")
    ap.add_argument("--no-calibration", action="store_true",
                    help="skip the in-run box calibration (plain 4-byte gather into a table of the customers row table's size, "
                         "streaming copy) behind roofline.gather_ceiling_ms / copy_TBps")
    ap.add_argument("--extras", default=None, metavar="PATH",
                    help="where the FULL record goes (default gpurun_out/bench_extras.json); stdout carries one compact line")
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous only: every rank joins the process group (gloo when no GPU is visible) and rank 0 "
                         "prints which ranks it saw; no compute (what tests/test_bench_launch.py drives on CPU)")
    args = ap.parse_args()
    if args.rows is None:
        args.rows = 1_000_000_000 if args.stream else 100_000_000
    return args


EXIT_USAGE, EXIT_NO_GPUS, EXIT_NO_RCCL = 2, 2, 3


def fatal(msg, code=EXIT_USAGE):
    print(f"bench.py: {msg}", file=sys.stderr, flush=True)
    sys.exit(code)


def self_launch(args):
    """`python bench.py --gpus N` as a plain command: start N ranks of this script, one per visible GPU, under
    torch.distributed.run (the launcher the driver would use) and hand its exit code on."""
    import socket
    import subprocess

    share_gpu = os.environ.get("CPH_BENCH_SHARE_GPU") == "1"
    if not args.launch_check and not share_gpu:
        import torch

        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            fatal(f"--gpus {args.gpus} needs {args.gpus} visible GPUs, this machine shows {have}: one rank per GPU over RCCL "
                  f"(CPH_BENCH_SHARE_GPU=1 runs the ranks on one GPU as a control-flow check over gloo — not an RCCL measurement)",
                  EXIT_NO_GPUS)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def launch_check(args, rank, world):
    """Every rank joins the process group and contributes its rank; rank 0 prints what it saw."""
    import torch
    import torch.distributed as dist

    backend = "nccl" if (torch.cuda.is_available() and torch.cuda.device_count() >= world
                         and os.environ.get("CPH_BENCH_SHARE_GPU") != "1") else "gloo"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(backend)
        dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))) if backend == "nccl" else torch.device("cpu")
        seen = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(seen, torch.tensor([rank], dtype=torch.int64, device=dev))
        ranks = [int(t.item()) for t in seen]
        dist.barrier()
        dist.destroy_process_group()
    else:
        ranks = [0]
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": args.gpus, "world": world, "ranks_seen": ranks,
                          "backend": backend if world > 1 else "none"}), flush=True)


def measure_traffic(kernel_prefix, args):
    """HBM-side bytes per launch of one kernel: two child runs of this same script under
    `rocprofv3 --kernel-trace --pmc <counter>` (one counter per run, never combined with other trace domains:
    MI355X_MICROARCH.md "rocprofv3 PMC slots").  Returns (dict, note) or (None, reason)."""
    import csv
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    vals = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="cph_pmc_", dir="/tmp")
        cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "pmc", "--",
               sys.executable, str(ROOT / "bench.py"), "--steps", "2", "--warmup", "1", "--rows", str(args.rows),
               "--customers", str(args.customers), "--products", str(args.products), "--no-cpu-baseline",
               "--no-index-1e8", "--no-verify", "--no-e2e", "--no-traffic", "--no-positions", "--no-variants", "--extras", os.devnull] + (["--row-ids"] if args.row_ids else [])
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300,
                           check=True)
            n, tot = 0, 0.0
            for f in Path(d).rglob("*counter_collection.csv"):
                for r in csv.DictReader(open(f)):
                    name = r["Kernel_Name"]
                    for pfx in ("void cph::", "cph::"):
                        if name.startswith(pfx):
                            name = name[len(pfx):]
                    if name.startswith(kernel_prefix) and r["Counter_Name"] == counter:
                        n += 1
                        tot += float(r["Counter_Value"])
            if n == 0:
                return None, f"no {counter} rows for {kernel_prefix}"
            vals[counter] = tot / n * 1024.0   # the counters are reported in KB
        except (subprocess.SubprocessError, OSError, KeyError, ValueError) as e:
            return None, f"{counter} pass failed: {type(e).__name__}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return vals, "measured in this run"


def flush_c_stdio():
    """RCCL announces itself with printf ("RCCL version : ...") into C's stdout buffer, which — stdout being a pipe under the driver —
    is only flushed at exit, i.e. BEHIND the JSON line Python prints: flush it first, so that the JSON line is the last line of stdout."""
    try:
        import ctypes

        ctypes.CDLL(None).fflush(None)
    except Exception:   # noqa: BLE001
        pass


LINE_BUDGET = 6000      # bytes of the ONE stdout line (round 5's 22 KB line left the driver's record with parsed = null)


def _r(x, nd=4):
    return round(x, nd) if isinstance(x, float) else x


def _finite(o):
    """NaN / Infinity are not JSON: they become null."""
    if isinstance(o, float):
        return o if o == o and abs(o) != float("inf") else None
    if isinstance(o, dict):
        return {k: _finite(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_finite(v) for v in o]
    return o


def _cgroup_cpu_max():
    """The container's CPU quota ("1600000 100000" = 16 CPUs per period, "max ..." = none): host-side thread pools and the all-cores CPU
    baseline run inside it whatever nproc says."""
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            return open(path).read().strip()
        except OSError:
            continue
    return None


def compact_line(out, extras_path):
    """The ONE stdout line: the contract's fields, a compact `roofline` and `cpu_baseline`, and the other measurements of the run as
    bare numbers.  Everything else — per-kernel tables, byte models term by term, verification details, notes — stays in the
    full record `out`, which emit() writes to `extras_path`.  Lists are positional; `cols` names the positions."""
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data") if k in out}
    cfg = out.get("config") or {}
    line["config"] = {k: (v[:160] if isinstance(v, str) else v) for k, v in cfg.items()
                      if k in ("workload", "rows", "customers", "products", "rows_this_rank", "exchange", "rccl_nranks", "chunk_rows", "slots",
                               "output", "shard_rows_rank0")}
    if cfg.get("build_row_mode"):
        line["config"]["output"] = "sorted positions" if cfg["build_row_mode"].startswith("sorted") else "original row ids"
    if cfg.get("exchange_transport") and cfg["exchange_transport"] != "none":
        line["config"]["transport"] = cfg["exchange_transport"][:80]
    for k in ("scope", "joined_rows_per_step", "per_rank_ms_per_step", "verified", "efficiency_vs_n1", "exchange_ms", "compute_ms",
              "kernel_ms_per_step", "h2d_GBps_rank0", "d2h_GBps_rank0"):
        if out.get(k) is not None:
            line[k] = out[k]
    rf = out.get("roofline")
    if rf:
        c = {k: rf.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches",
                                    "algorithmic_bytes_per_launch", "useful_bytes_per_launch", "frac_hbm", "step_algorithmic_bytes",
                                    "step_frac", "copy_TBps", "output_mode") if k in rf}
        if rf.get("traffic") is None and rf.get("traffic_note"):
            c["traffic_note"] = rf["traffic_note"][:100]
        if (rf.get("gather_ceiling") or {}).get("ms"):
            c["gather_ceiling_ms"] = rf["gather_ceiling"]["ms"]
        om = rf.get("other_output_mode")
        if om:
            c["other_output_mode"] = [om.get("mode"), _r(om.get("ms_per_step")), _r(om.get("k_chain_dense_ms")), om.get("frac"),
                                      om.get("positions_equal_row_ids_through_perm")]
        vs = out.get("variants")
        if vs:
            c["variants_cols"] = "ms_per_step,kernel_ms,frac,verified"
            c["variants"] = {k: ([v.get("ms_per_step"), v.get("k_chain_dense_ms"), (v.get("roofline") or {}).get("frac"), v.get("verified")]
                                 if "error" not in v else {"error": str(v["error"])[:80]}) for k, v in vs.items()}
        line["roofline"] = c
    cb = out.get("cpu_baseline")
    if cb:
        c = {"value": _r(cb.get("value"), 1), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
             "sample": (cb.get("sample") or "")[:200]}
        if (cb.get("extrapolated_full_size") or {}).get("value"):
            c["extrapolated_full_size"] = _r(cb["extrapolated_full_size"]["value"], 1)
        for name, v in (cb.get("variants") or {}).items():
            c[name] = [_r(v.get("value"), 1), v.get("cores")]
        line["cpu_baseline"] = c
    ix = out.get("index_on_1e8")
    if ix:
        c = {"cols": "ms,kernel_ms,frac_pass_model,frac_compulsory,verified"}
        for name, v in ix.items():
            if isinstance(v, dict) and "ms" in v:
                c[name] = [v["ms"], v.get("kernel_ms"), v.get("frac_pass_model"), v.get("frac_compulsory"), v.get("verified")]
        e2e = ix.get("e2e_pinned_host") or {}
        hc = {name: [v.get("ms"), v.get("strings_uploaded_ms"), v.get("verified")] for name, v in e2e.items() if isinstance(v, dict) and "ms" in v}
        if hc:
            c["from_pinned_host_cols"] = "ms,strings_uploaded_ms,verified"
            c["from_pinned_host"] = hc
        line["index_on_1e8"] = c
    comb = out.get("index_plus_join_1e8")
    if comb:
        line["index_plus_join_1e8"] = {"cols": "ms,frac", **{k: [v["ms"], v["frac"]] for k, v in comb.items() if isinstance(v, dict)}}
    e2e = {k: [_r(out[k].get("ms")), _r(out[k].get("rows_per_s"), 0)] for k in ("e2e_pinned_host", "e2e_pinned_host_encoded")
           if isinstance(out.get(k), dict) and "ms" in out[k]}
    if e2e:
        line["e2e"] = {"cols": "ms,rows_per_s", **e2e}
    tc = out.get("to_csv")
    if tc:
        line["to_csv"] = ({"cols": "ms,TBps,two_pass_ms,verified", "joined_rows_to_text": [tc.get("ms"), tc.get("TBps"), tc.get("two_pass_ms"), tc.get("verified")]}
                          if "error" not in tc else {"error": str(tc["error"])[:80]})
    cp = out.get("csv_parse")
    if cp:
        line["csv_parse"] = ({"cols": "ms,GBps,verified", "orders_text_to_columns": [cp.get("ms"), cp.get("GBps"), cp.get("verified")]}
                             if "error" not in cp else {"error": str(cp["error"])[:80]})
    m = out.get("multi_gpu")
    if m:
        c = {k: m.get(k) for k in ("mode", "chunks", "join_compute_ms", "exchange_ms", "exposed_exchange_ms", "bytes_sent_per_step",
                                   "bytes_received_per_step", "n1_ms_per_step", "efficiency_vs_n1")}
        if m.get("same_run_other_modes"):
            c["other_modes_ms"] = {k: v.get("ms_per_step", v.get("error")) for k, v in m["same_run_other_modes"].items()}
        mr = m.get("measured_exchange_rate") or {}
        c["link_GBps"] = mr.get("GBps_per_peer_link")
        c["recv_GBps"] = mr.get("GBps_received_per_rank")
        if m.get("build_side"):
            c["build_side"] = m["build_side"].get("choice")
        line["multi_gpu"] = c
    ks = out.get("kernels")
    if ks:   # the five largest kernels of a step, avg ms per launch
        top = sorted(ks.items(), key=lambda kv: -kv[1]["total_ms"] / max(1, kv[1]["launches"]))[:5]
        line["kernels_avg_ms"] = {k: v["avg_ms"] for k, v in top}
    if out.get("host"):
        line["host"] = out["host"]
    if out.get("extras_error"):
        line["extras_error"] = str(out["extras_error"])[:200]
    line["extras"] = extras_path
    # the budget is a hard limit: optional blocks go, in this order, until the line fits
    for drop in ("kernels_avg_ms", "e2e", "host", "csv_parse", "to_csv", "index_plus_join_1e8", "multi_gpu", "index_on_1e8"):
        if len(json.dumps(line)) <= LINE_BUDGET:
            break
        line.pop(drop, None)
    if len(json.dumps(line)) > LINE_BUDGET and "roofline" in line:
        line["roofline"].pop("variants", None)
    return line


def emit(out, args):
    """Write the full record beside the repo (gpurun_out/bench_extras.json unless --extras says otherwise), then print the ONE
    compact line — last thing on stdout."""
    path = Path(args.extras) if args.extras else ROOT / "gpurun_out" / "bench_extras.json"
    shown = str(path)
    try:
        path.parent.mkdir(parents=True, exist_ok=True)
        path.write_text(json.dumps(out, indent=1) + "\n")
        try:
            shown = str(path.resolve().relative_to(ROOT))
        except ValueError:
            pass
    except OSError as ex:
        shown = f"not written: {type(ex).__name__}: {ex}"[:120]
    flush_c_stdio()
    sys.stdout.flush()
    print(json.dumps(_finite(compact_line(out, shown)), allow_nan=False), flush=True)


def stream_mode(args, eng, dev, rank, world, share_gpu):
    """BASELINE.json configs[4] (1e9-row orders joined against resident indexes) on N ranks: the build side is replicated (every
    rank builds both indexes in its HBM), the orders are sharded by row range and every rank streams ITS shard from pinned host
    memory through cph_stream_join_* — 2^23-row chunks, upload / join / download of consecutive chunks overlapped on the
    pipeline's HIP streams — and leaves the joined tuples (sorted positions + match bitmap) in pinned host memory: the node's
    host memory then holds the whole result in row order (rank r's block follows rank r-1's), which is what the Go caller of
    INTEGRATION.md iterates over.  No collective on the data path ("weak" in the sense of the contract: units per rank fixed
    by the shard).  A step = one pass of the rank's whole shard; timing as in the default mode (barrier + synchronize on both
    sides, max over ranks).  PCIe inclusive by construction."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from csvplus_amd import datagen as dg
    from csvplus_amd.engine import shard_range
    from csvplus_amd.streaming import PinnedCol, StreamJoin

    begin, end = shard_range(args.rows, rank, world)
    nloc = end - begin
    cust_id = dg.column(dg.SEQ_PERM, args.customers, args.customers, encoding=dg.FIXED8, seed=dg.SEED + 1)
    prod_id = dg.column(dg.SEQ_PERM, args.products, args.products, encoding=dg.ITOA, seed=dg.SEED + 2)
    ia, ib = eng.index_on_many([[cust_id.to_device(dev)], [prod_id.to_device(dev)]], unique=True)
    POS = not args.row_ids
    # the shard is generated and pinned in pieces of 2^26 rows (the generator's temporaries stay small)
    piece = 1 << 26
    chunk = 1 << 23
    pinned, chunks, bounds, h2d = [], [], [], 0
    for p0 in range(0, nloc, piece):
        p1 = min(p0 + piece, nloc)
        o = dg.orders(args.rows, args.customers, args.products, row0=begin + p0, nrows=p1 - p0)
        pc = [PinnedCol(eng.ctx, o["cust_id"]), PinnedCol(eng.ctx, o["prod_id"])]
        h2d += o["cust_id"].nbytes_values() + o["prod_id"].nbytes_values() + o["cust_id"].nbytes_offsets() + o["prod_id"].nbytes_offsets()
        pinned.append(pc)
        for b in range(0, p1 - p0, chunk):
            e = min(b + chunk, p1 - p0)
            chunks.append([c.col.slice(b, e) for c in pc])
            bounds.append(begin + p0 + b)
        del o
    sj = StreamJoin(eng.ctx, [ia, ib], nslots=2, positions=POS)

    def one_pass():
        sub = done = joined = 0
        while done < len(chunks):
            while sub < len(chunks) and sj.pending < 2:
                sj.submit(chunks[sub], probe_base=bounds[sub])
                sub += 1
            joined += sj.next(copy=False)["nmatches"]
            done += 1
        return joined

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(max(args.warmup, 1)):   # (the first pass page-locks the slots' result blocks)
        one_pass()
    sync_all()
    t0 = time.perf_counter()
    joined = 0
    for _ in range(args.steps):
        joined = one_pass()
    sync_all()
    dt = time.perf_counter() - t0
    sj.close()
    per_rank_ms = [dt / args.steps * 1e3]
    total = joined
    if world > 1:
        cdev = "cpu" if share_gpu else dev
        allt = torch.zeros(world, dtype=torch.float64, device=cdev)
        dist.all_gather_into_tensor(allt, torch.tensor([dt], dtype=torch.float64, device=cdev))
        per_rank_ms = [float(x) / args.steps * 1e3 for x in allt.tolist()]
        dt = float(allt.max().item())
        t = torch.tensor([joined], dtype=torch.int64, device=cdev)
        dist.all_reduce(t)
        total = int(t.item())
        dist.destroy_process_group()
    if rank != 0:
        return
    ms = dt / args.steps * 1e3
    d2h = 8 * nloc + nloc // 8
    emit({
        "metric": "joined rows/sec (streaming Join of host-resident orders against HBM-resident indexes, PCIe inclusive)",
        "value": total / (dt / args.steps), "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 1),
        "ms_per_step": ms, "higher_is_better": True, "scaling": "strong" if world > 1 else "weak", "vs_baseline": None,
        "dtype": "u32 codes of byte-string keys", "data": "synthetic",
        "config": {"workload": f"BASELINE config 5 shape: {args.rows:.0e} orders streamed from pinned host memory in 2^23-row chunks x "
                               f"{args.customers:.0e} customers x {args.products:.0e} products, indexes resident in HBM on every rank",
                   "rows": args.rows, "shard_rows_rank0": nloc, "chunk_rows": chunk, "slots": 2,
                   "output": "sorted positions" if POS else "build-row ids", "exchange": "none (results stay in the node's host memory, rank blocks in row order)"},
        "scope": "pcie_inclusive: NOT comparable with the default mode's value (inputs resident in HBM)",
        "joined_rows_per_step": total, "per_rank_ms_per_step": [round(x, 3) for x in per_rank_ms],
        "h2d_GBps_rank0": round(h2d / (per_rank_ms[0] / 1e3) / 1e9, 1), "d2h_GBps_rank0": round(d2h / (per_rank_ms[0] / 1e3) / 1e9, 1),
        "verified": bool(total == args.rows), "verify": {"every_order_joined_once": total == args.rows}}, args)


def main():
    args = parse_args()
    if args.gpus < 1:
        fatal("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))       # plain command: start the ranks ourselves
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        fatal(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ.get('WORLD_SIZE')} ranks: "
              f"use --nproc-per-node {args.gpus}, or run `python bench.py --gpus {args.gpus}` without a launcher")
    if args.launch_check:
        return launch_check(args, int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))
    import numpy as np
    import torch
    import torch.distributed as dist

    from csvplus_amd import _native as N, datagen as dg
    from csvplus_amd.dist import allgatherv_many, connect
    from csvplus_amd.engine import Engine, shard_range

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Debug aid for 1-GPU boxes: CPH_BENCH_SHARE_GPU=1 maps every rank to cuda:0 and uses the gloo
    # backend (RCCL refuses two ranks on one device), so the N>1 control flow can be exercised.
    share_gpu = os.environ.get("CPH_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    if not torch.cuda.is_available() or (not share_gpu and torch.cuda.device_count() <= local_rank):
        fatal(f"rank {rank}: no GPU for local rank {local_rank} ({torch.cuda.device_count() if torch.cuda.is_available() else 0} "
              f"visible); csvplus_amd has no CPU path", EXIT_NO_GPUS)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    eng = Engine(local_rank)
    for kv in args.ctx_option:      # A/B switches of the library (cph_ctx_set_option), e.g. scan_lookback=0
        k, _, v = kv.partition("=")
        eng.ctx.set_option(k, int(v))
    # the exchange runs behind the C ABI (cph_dist_*: RCCL); torch.distributed only ships the communicator id (and
    # provides the barrier / max-over-ranks of the timing contract).  If the communicator cannot be created on ANY rank
    # the whole job stops with a non-zero exit code: a scaling curve is a measurement of cph_dist_* or it is nothing.
    # Debug mode with several ranks on one GPU (CPH_BENCH_SHARE_GPU=1, gloo): the torch transport of csvplus_amd/dist.py.
    cdist, transport, rccl_nranks = None, "none", None
    # CPH_BENCH_FORCE_DIST=1: take the N > 1 code path (communicator, cph_dist_join_chain, the multi_gpu report) with ONE rank
    # — what a one-GPU box can execute of it (tests/test_bench_launch.py)
    force_dist = world == 1 and os.environ.get("CPH_BENCH_FORCE_DIST") == "1" and args.exchange != "none"
    if force_dist:
        cdist = connect(eng.ctx)
        rccl_nranks = cdist.size
        transport = "cph_dist_* (RCCL behind the C ABI): " + cdist.transport()
    if world > 1 and args.exchange != "none":
        if share_gpu:
            transport = "torch.distributed gloo (DEBUG: ranks share one GPU; not an RCCL measurement)"
        else:
            ok, why = 1, ""
            try:
                cdist = connect(eng.ctx)
                if cdist.size != world:
                    raise RuntimeError(f"cph_dist_size {cdist.size} != world {world}")
            except Exception as e:   # noqa: BLE001 — reported below, on every rank
                ok, why = 0, f"{type(e).__name__}: {e}"
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)   # one decision for all ranks
            if not int(flag.item()):
                print(f"bench.py: rank {rank}: cph_dist_create (RCCL behind the C ABI) failed on "
                      f"{'this rank: ' + why if why else 'another rank'}; no fallback transport — stopping",
                      file=sys.stderr, flush=True)
                dist.destroy_process_group()
                sys.exit(EXIT_NO_RCCL)
            rccl_nranks = cdist.size
            transport = "cph_dist_* (RCCL behind the C ABI): " + cdist.transport()

    if args.stream:
        return stream_mode(args, eng, dev, rank, world, share_gpu)

    # ---- synthetic tables (deterministic), staged to HBM before timing ------------
    t0 = time.time()
    begin, end = shard_range(args.rows, rank, world)
    cust_id = dg.column(dg.SEQ_PERM, args.customers, args.customers, encoding=dg.FIXED8, seed=dg.SEED + 1)
    prod_id = dg.column(dg.SEQ_PERM, args.products, args.products, encoding=dg.ITOA, seed=dg.SEED + 2)
    ords = dg.orders(args.rows, args.customers, args.products, row0=begin, nrows=end - begin)
    host_bytes = {k: v.nbytes_values() for k, v in ords.items()}
    d_cust, d_prod = cust_id.to_device(dev), prod_id.to_device(dev)
    d_ord = {k: v.to_device(dev) for k, v in ords.items()}   # all 3 columns live in HBM; qty rides along
    torch.cuda.synchronize(dev)
    gen_s = time.time() - t0
    nloc = end - begin

    POS = not args.row_ids          # the timed step's output mode: sorted positions (default) or original row ids
    step_info = {}
    shard_rows = [shard_range(args.rows, r, world)[1] - shard_range(args.rows, r, world)[0] for r in range(world)]
    xstats = []                     # cph_dist_join_stats of every step (N > 1)

    def step():
        # both build sides as one batch (cph_index_build_many): their host round trips are shared
        ia, ib = eng.index_on_many([[d_cust], [d_prod]], unique=True)
        # the chained join straight through the binding (cph_join_chain, results left in HBM): the timed loop holds
        # no torch views of the result — it needs the row count only.  stream_row is NULL when every order joined
        # (the result row IS the stream row): then only the two build-row arrays exist, and only they are exchanged.
        if cdist is not None and args.exchange in ("allgatherv", "packed", "host"):
            # the shard in sub-chunks, chunk k travelling while chunk k+1 is joined (cph_dist_join_chain); the shard sizes of a
            # range split are known to everybody, so the call's only host wait is for the match totals at its end
            g = cdist.join_chain([(ia, [d_ord["cust_id"]]), (ib, [d_ord["prod_id"]])], probe_base=begin, shard_rows=shard_rows,
                                 nchunks=args.chunks, positions=POS, host=args.exchange == "host", packed=args.exchange == "packed")
            n = g.total
            xstats.append(g.stats)
            g.release()
            if not step_info:
                step_info["info"] = (ia.info(), ib.info())
            ia.close()
            ib.close()
            return n, step_info["info"]
        ch = N.join_chain(eng.ctx, [(ia, [d_ord["cust_id"]]), (ib, [d_ord["prod_id"]])], probe_base=begin,
                          out_mem=N.CPH_MEM_DEVICE, positions=POS)
        if cdist is not None and args.exchange != "none":   # one count exchange + one grouped batch for all arrays (cph_dist_chain_allgather)
            g = cdist.chain_allgather(ch)
            n = g.total
            g.release()
        elif world > 1 and args.exchange != "none":
            from csvplus_amd.engine import device_view
            p = ch.device_ptrs()
            ts = [device_view(q, ch.nrows, "<i4", ch, dev) for q in p["build_row"]]
            if not ch.identity:
                ts.insert(0, device_view(p["stream_row"], ch.nrows, "<i8", ch, dev))
            n = int(allgatherv_many(ts)[0][-1].numel())
        else:
            n = ch.nrows
        if not step_info:
            step_info["info"] = (ia.info(), ib.info())
        ch.release()
        ia.close()
        ib.close()
        return n, step_info["info"]

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    # Per-kernel timing costs stream time (two HIP events = ~10 us per launch, ~25 launches per step), so the timed
    # region times ONLY its dominant kernel (cph_ctx_profile_only); which one that is, and the per-kernel breakdown
    # reported in `kernels`, come from fully profiled steps OUTSIDE the timed region.
    eng.ctx.profile(True)
    eng.ctx.profile_read(reset=True)
    step()
    pick = eng.ctx.profile_read(reset=True)
    # (among the kernels whose byte model moves at least a megabyte per step: at toy sizes a sampling or bookkeeping kernel can be the
    # longest launch of a step by timing noise alone, and a roofline over its few kilobytes rounds to 0 — tests/test_bench_launch.py ran
    # into that once in ten runs.  At the benchmark's size every kernel of the step is far above the bar.)
    # (k_chain_dense's bytes are modelled by bench.py itself — chain_bytes() below — not by the library's launch record)
    modelled = {k: v for k, v in pick.items() if v.get("algo_bytes", 0) >= 1e6 or k == "k_chain_dense"}
    dom_name = max(modelled.items(), key=lambda kv: kv[1]["total_ms"])[0] if modelled else "k_chain_dense"
    eng.ctx.profile_only(dom_name)
    sync_all()
    del xstats[:]
    t0 = time.perf_counter()
    joined = 0
    for _ in range(args.steps):
        n, info = step()
        joined = n
    sync_all()
    dt = time.perf_counter() - t0
    xtimed = list(xstats)
    prof_timed = eng.ctx.profile_read(reset=True)
    breakdown_steps = 3
    eng.ctx.profile(True)
    for _ in range(breakdown_steps):
        step()
    prof = eng.ctx.profile_read(reset=True)
    eng.ctx.profile(False)
    per_rank_ms = [dt / args.steps * 1e3]
    if world > 1:
        cdev = "cpu" if share_gpu else dev
        allt = torch.zeros(world, dtype=torch.float64, device=cdev)
        dist.all_gather_into_tensor(allt, torch.tensor([dt], dtype=torch.float64, device=cdev))
        per_rank_ms = [float(x) / args.steps * 1e3 for x in allt.tolist()]
        dt = float(allt.max().item())          # the job is as slow as its slowest rank
    total_joined = joined if (world == 1 or args.exchange != "none") else None
    if total_joined is None:
        t = torch.tensor([joined], dtype=torch.int64, device="cpu" if share_gpu else dev)
        dist.all_reduce(t)
        total_joined = int(t.item())

    # ---- N > 1: the other exchange modes and the exchange's own rate, IN THIS RUN (every rank takes part) --------------------
    # A scaling curve of the default mode alone reads as a failure of the join when it is the price of the mandated all-gather:
    # the same step with the node's shared host buffer as the meeting point and with no exchange at all (every rank keeps its
    # shard's result) are timed right here, 3 steps each, max over ranks — and one large cph_dist_allgatherv says what a rank
    # actually receives per second over the links (DESIGN §7 assumed 50 GB/s per link until measured).
    alternatives, measured_rate = None, None
    if (world > 1 or force_dist) and not args.no_alternatives:
        alternatives = {}
        cdev = "cpu" if share_gpu else dev

        def max_over_ranks(x):
            if world == 1:
                return x
            allx = torch.zeros(world, dtype=torch.float64, device=cdev)
            dist.all_gather_into_tensor(allx, torch.tensor([x], dtype=torch.float64, device=cdev))
            return float(allx.max().item())

        timed_mode = args.exchange
        for mode in ("allgatherv", "packed", "host", "none"):
            if mode == timed_mode or (mode != "none" and cdist is None):
                continue
            args.exchange = mode
            try:
                del xstats[:]
                step()
                sync_all()
                t1 = time.perf_counter()
                for _ in range(3):
                    step()
                sync_all()
                ms_mode = max_over_ranks((time.perf_counter() - t1) / 3 * 1e3)
                alternatives[mode] = {"ms_per_step": round(ms_mode, 4), "bytes_sent": xstats[0]["bytes_sent"] if xstats else None,
                                      "exposed_exchange_ms": round(sum(x["exposed_exchange_ms"] for x in xstats) / len(xstats), 4) if xstats else None}
            except Exception as ex:   # noqa: BLE001 — an extra figure must not cost the line
                alternatives[mode] = {"error": f"{type(ex).__name__}: {ex}"}
            finally:
                args.exchange = timed_mode
        if cdist is not None:
            try:
                words = 16 << 20   # 64 MB per rank
                buf = torch.zeros(words, dtype=torch.int32, device=dev)
                cdist.allgatherv([buf.data_ptr()], [4], words).release()
                sync_all()
                t1 = time.perf_counter()
                for _ in range(3):
                    cdist.allgatherv([buf.data_ptr()], [4], words).release()
                sync_all()
                ms_x = max_over_ranks((time.perf_counter() - t1) / 3 * 1e3)
                recv = 4 * words * (world - 1)
                measured_rate = {"what": "cph_dist_allgatherv of 64 MB per rank (grouped ncclSend / ncclRecv to every peer), max over ranks",
                                 "ms": round(ms_x, 3), "bytes_received_per_rank": recv,
                                 "GBps_received_per_rank": round(recv / ms_x / 1e6, 1) if recv else None,
                                 "GBps_per_peer_link": round(4 * words / ms_x / 1e6, 1) if recv else None}
                del buf
            except Exception as ex:   # noqa: BLE001
                measured_rate = {"error": f"{type(ex).__name__}: {ex}"}

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = dt / args.steps * 1e3
    value = total_joined / (dt / args.steps)

    # ---- per-kernel roofline (HIP events recorded by the library on the ctx stream) -------------
    # algorithmic bytes the library cannot know (value bytes of the input columns) are added here;
    # the model is documented in DESIGN.md §"Algorithmic bytes".
    ia_info, ib_info = info
    K = breakdown_steps   # the `kernels` table comes from the fully profiled steps behind the timed region
    total_joined_local = joined if world == 1 or args.exchange == "none" else nloc
    # ---- N > 1: where the step went (rank 0's view), and what one GPU needs for the WHOLE stream ----------------
    multi = None
    if world > 1 or force_dist:
        from csvplus_amd.dist import build_side_estimate

        def mean(key):
            return round(sum(x[key] for x in xtimed) / len(xtimed), 4) if xtimed else None

        exposed = mean("exposed_exchange_ms")
        multi = {"mode": args.exchange, "chunks": xtimed[0]["chunks"] if xtimed else None,
                 "join_compute_ms": mean("compute_ms"), "exchange_ms": mean("exchange_ms"), "exposed_exchange_ms": exposed,
                 "bytes_sent_per_step": xtimed[0]["bytes_sent"] if xtimed else None,
                 "bytes_received_per_step": xtimed[0]["bytes_received"] if xtimed else None,
                 "packed_bits_per_row": xtimed[0].get("packed_bits") if xtimed else None,
                 "note": "events on the ctx stream (join of all chunks) and on the exchange stream (first chunk ready -> match totals of "
                         "all ranks received); exposed = the part of the exchange behind the last chunk's join, which nothing hides"}
        n1_ms, n1_all = None, []
        if not args.no_n1 and not share_gpu:
            # strong scaling's reference point, measured here: rank 0 alone runs the step over ALL rows (no exchange)
            full = dg.orders(args.rows, args.customers, args.products)
            d_full = {k: full[k].to_device(dev) for k in ("cust_id", "prod_id")}

            def step1():
                a, b = eng.index_on_many([[d_cust], [d_prod]], unique=True)
                c = N.join_chain(eng.ctx, [(a, [d_full["cust_id"]]), (b, [d_full["prod_id"]])], out_mem=N.CPH_MEM_DEVICE, positions=POS)
                c.release(); a.close(); b.close()

            step1()
            # median of five single steps: one stall of the box (seen: 87 ms in one step out of twenty at 3e5 rows,
            # profiles/r05_n1_outlier.txt — in a cph_dist step as well as in a plain one) must not become the reference point
            n1_all = []
            for _ in range(5):
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                step1()
                torch.cuda.synchronize(dev)
                n1_all.append((time.perf_counter() - t1) * 1e3)
            n1_ms = sorted(n1_all)[len(n1_all) // 2]
            del d_full, full
        multi["same_run_other_modes"] = alternatives
        multi["measured_exchange_rate"] = measured_rate
        multi["n1_ms_per_step"] = round(n1_ms, 4) if n1_ms else None
        multi["n1_ms_single_steps"] = [round(x, 4) for x in n1_all] if n1_ms else None
        multi["efficiency_vs_n1"] = round(n1_ms / (world * dt / args.steps * 1e3), 4) if n1_ms else None
        code_bytes = 4
        link = (measured_rate or {}).get("GBps_per_peer_link")
        multi["build_side"] = dict(build_side_estimate(args.customers, code_bytes, world, 3.0e10, link or 50.0),
                                   note="customers table: every rank builds (A) vs rank 0 builds + cph_dist_index_broadcast (B), "
                                        "critical path per rank; 3e10 rows/s build rate; link rate " +
                                        (f"{link} GB/s measured in this run (measured_exchange_rate)" if link else "50 GB/s ASSUMED (one rank: nothing to measure)"))
    off_c, off_p = cust_id.nbytes_offsets(), prod_id.nbytes_offsets()   # 0 for fixed-width columns
    off_o = ords["cust_id"].nbytes_offsets() + ords["prod_id"].nbytes_offsets()
    cust_bytes, prod_bytes = cust_id.nbytes_values(), prod_id.nbytes_values()
    # Byte models of the fused chain pass, per launch.  Streams: both keys' bytes + offsets in, two
    # 4-byte results out per joined row (the stream row is implicit).  Lookups:
    #   original row ids   contract model: one 4-byte table entry per stream row and step (2 * 4 * rows);
    #                      compulsory ("hbm"): each 4-byte-per-code row table read once
    #   sorted positions   the rank tables are 8 bytes per 32 codes (2.5 MB + 40 KB here: L2 / LDS resident) and are charged
    #                      ONCE, in both models — and not at all for an index that fills its code space (states == rows:
    #                      the position of a key is its code, the kernel looks nothing up).  Round 3 still charged
    #                      2 * 4 * rows here, 0.8 GB that never left the L2.
    stream_bytes = host_bytes["cust_id"] + host_bytes["prod_id"] + off_o
    out_bytes = 8 * total_joined_local

    def rank_table_bytes(info, rows):
        # (both indexes are UniqueIndexOn results: no duplicate keys)
        return 0.0 if info["table_entries"] == rows else 0.25 * info["table_entries"]

    def chain_bytes(positions):
        if positions:
            t = rank_table_bytes(ia_info, args.customers) + rank_table_bytes(ib_info, args.products)
            return stream_bytes + t + out_bytes, stream_bytes + t + out_bytes
        return (stream_bytes + 2 * 4 * nloc + out_bytes,
                stream_bytes + 4 * (ia_info["table_entries"] + ib_info["table_entries"]) + out_bytes)

    algo_chain, hbm_chain = chain_bytes(POS)
    # (round 5) one k_encode_build launch per step = the products table only: the customers' keys are coded inside the first
    # partition level of their direct sort, whose library-side byte count already holds the 8 bytes of key per row
    cust_fused = prof.get("k_encode_build", {}).get("launches", 0) <= K and "k_win_partition" in prof
    extra = {
        "k_col_stats": K * (cust_bytes + off_c + prod_bytes + off_p),
        "k_encode_build": K * ((0 if cust_fused else cust_bytes + off_c + ia_info["key_bytes"] * args.customers)
                               + (prod_bytes + off_p + ib_info["key_bytes"] * args.products)),
        "k_chain_dense": K * algo_chain,
    }
    # compulsory bytes (SURVEY.md §8d "useful"): every input read once, each table read once (not one entry per
    # probe), every output written once
    useful = {"k_chain_dense": K * hbm_chain}
    kernels = {}
    for name, st in prof.items():
        b = st["algo_bytes"] + extra.get(name, 0.0)
        kernels[name] = {"launches": st["launches"], "total_ms": round(st["total_ms"], 4),
                         "avg_ms": round(st["total_ms"] / max(1, st["launches"]), 5),
                         "algo_GB": round(b / 1e9, 4),
                         "GBps": round(b / 1e9 / (st["total_ms"] / 1e3), 1) if st["total_ms"] > 0 else None}
    # the dominant kernel: its launches INSIDE the timed region (the only ones timed there)
    if dom_name in kernels and dom_name in prof_timed and prof_timed[dom_name]["launches"]:
        st = prof_timed[dom_name]
        per_launch_bytes = kernels[dom_name]["algo_GB"] * 1e9 / max(1, kernels[dom_name]["launches"])
        kernels[dom_name] = {"launches": st["launches"], "total_ms": round(st["total_ms"], 4),
                             "avg_ms": round(st["total_ms"] / st["launches"], 5),
                             "algo_GB": round(per_launch_bytes * st["launches"] / 1e9, 4),
                             "GBps": round(per_launch_bytes * st["launches"] / 1e9 / (st["total_ms"] / 1e3), 1),
                             "timed_region": True}
    dom = (dom_name, kernels[dom_name]) if dom_name in kernels else (None, None)
    roofline = None
    if dom[0]:
        a = dom[1]["GBps"] or 0.0
        roofline = {"bound": "hbm", "kernel": dom[0], "achieved": a, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(a / HBM_PEAK_GBPS, 4), "traffic": None,
                    "avg_launch_ms": dom[1]["avg_ms"], "launches": dom[1]["launches"],
                    "algorithmic_bytes_per_launch": round(dom[1]["algo_GB"] * 1e9 / dom[1]["launches"])}
        if dom[0] in useful and dom[1]["total_ms"] > 0:
            # `useful` was summed over the K breakdown steps; a step of the N > 1 path launches the kernel once per sub-chunk
            ub = useful[dom[0]] / K / max(1.0, dom[1]["launches"] / args.steps if dom[1].get("timed_region") else dom[1]["launches"] / K)
            roofline["useful_bytes_per_launch"] = round(ub)
            roofline["useful"] = round(ub / 1e9 / (dom[1]["avg_ms"] / 1e3) / HBM_PEAK_GBPS, 4)
            roofline["frac_hbm"] = roofline["useful"]
            roofline["frac_note"] = ("frac: the contract's byte model of the timed output mode (bytes_model); frac_hbm: compulsory bytes "
                                     "only (inputs and lookup tables read once, outputs written once).  Reporting positions the two "
                                     "coincide: the rank tables are charged once")
            roofline["bytes_model"] = {"streams_in": round(stream_bytes), "results_out": round(out_bytes),
                                       "lookups": round(algo_chain - stream_bytes - out_bytes),
                                       "lookups_compulsory": round(hbm_chain - stream_bytes - out_bytes)}
        # the whole step (every kernel + host gaps) against the same peak
        step_bytes = sum(v["algo_GB"] * 1e9 / (args.steps if v.get("timed_region") else K) for v in kernels.values())
        roofline["step_algorithmic_bytes"] = round(step_bytes)
        roofline["step_frac"] = round(step_bytes / 1e9 / (ms_per_step / 1e3) / HBM_PEAK_GBPS, 4)
        # The ceiling of THIS box for the kernel's dominant access pattern, measured now: the chained join makes one
        # random 4-byte lookup per stream row into the customers row table (4 bytes per code: too big for any L2) and
        # one into the small products table; a plain gather out[i] = table[idx[i]] of as many lookups into a table of
        # the same size — no keys to decode, 4 B index in, 4 B value out — is the floor of the customers step alone
        # (cph_calibrate; the kernel also streams ~25 B of key bytes per row and does the second lookup).
        if world == 1 and not args.no_calibration and roofline["kernel"] == "k_chain_dense":
            tb = max(4 * ia_info["table_entries"], 1 << 20)
            g_ms = eng.ctx.calibrate("gather", tb, nloc, 5)
            c_bytes = 1 << 30
            c_ms = eng.ctx.calibrate("copy", c_bytes, 0, 5)
            calib = {"table_bytes": tb, "lookups": nloc, "ms": round(g_ms, 4), "Glookups_per_s": round(nloc / g_ms / 1e6, 1),
                     "what": "plain 4-byte gather, same number of lookups, table of the customers ROW table's size, measured in this "
                             "run (cph_calibrate): the floor of the customers step of k_chain_dense when it reports original row ids"}
            if not POS:
                calib["kernel_over_ceiling"] = round(dom[1]["avg_ms"] / g_ms, 3)
                roofline["gather_ceiling_ms"] = round(g_ms, 4)
            roofline["gather_ceiling"] = calib
            roofline["copy_TBps"] = round(2 * c_bytes / (c_ms / 1e3) / 1e12, 3)
            roofline["copy_note"] = "streaming copy of 1 GiB, read + written bytes per second, measured in this run"
        roofline["output_mode"] = "sorted positions" if POS else "original row ids"
    # HBM-side traffic of the dominant kernel: PMC counters cannot be collected inside this process, so
    # two child runs of this script under `rocprofv3 --pmc` (one counter each) measure them NOW, on this box.
    # gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE tallies a 128-byte request of a wide coalesced
    # stream as 64 bytes; profiles/r02_pmc_calibration.txt (tools/microbench/pmc_calib.hip) measures the factor
    # for this kernel's access classes — streamed 4/8-byte-per-lane reads are under-counted 2x, random 4-byte
    # gathers and WRITE_SIZE are exact — so corrected = FETCH + (streamed input bytes that reached HBM once) + WRITE.
    if roofline and world == 1 and not args.no_traffic:
        vals, note = measure_traffic(roofline["kernel"], args)
        if vals:
            streamed_in = (host_bytes["cust_id"] + host_bytes["prod_id"] + off_o) if roofline["kernel"] == "k_chain_dense" else 0
            raw = vals["FETCH_SIZE"] + vals["WRITE_SIZE"]
            roofline["traffic"] = round(raw + streamed_in / 2)
            roofline["traffic_raw"] = {"FETCH_SIZE": round(vals["FETCH_SIZE"]), "WRITE_SIZE": round(vals["WRITE_SIZE"])}
            roofline["traffic_note"] = ("bytes per launch; FETCH_SIZE + WRITE_SIZE from two rocprofv3 --pmc child runs of "
                                        "this command, + half of the streamed input bytes (gfx950 counts a coalesced "
                                        "stream's 128-byte requests as 64: profiles/r02_pmc_calibration.txt)")
        else:
            roofline["traffic_note"] = f"not measured: {note}"
    build_names = ("k_col_stats", "k_encode_build", "k_radix_hist_u32", "k_radix_hist_u64", "k_radix_scatter_u32",
                   "k_radix_scatter_u64", "exclusive_scan_u32", "k_first_dup", "k_build_table", "k_gather_u64", "k_split_count",
                   "k_direct_scatter", "k_direct_finish", "k_win_partition", "k_win_place", "k_cs_hist", "k_cs_scan", "k_cs_partition", "k_cs_window")
    build_ms = sum(kernels[k]["total_ms"] for k in build_names if k in kernels)
    build_gb = sum(kernels[k]["algo_GB"] for k in build_names if k in kernels)
    kernel_ms_per_step = sum(v["total_ms"] / (args.steps if v.get("timed_region") else K) for v in kernels.values())

    out = {
        "metric": "joined rows/sec (IndexOn build + chained Join, 1e8-row 3-col orders)",
        "value": value, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u32", "dtype_note": "byte-string keys (u8) re-coded to order-preserving u32 codes; integer work only",
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[3]: orders(1e8 x {cust_id,prod_id,qty}).Join(UniqueIndexOn customers.id 1e7).Join(UniqueIndexOn products.prod_id 1e5)",
                   "sharding": "probe rows split into n_gpus contiguous ranges, build side replicated",
                   "rows": args.rows, "customers": args.customers, "products": args.products,
                   "rows_this_rank": nloc, "exchange": args.exchange if (world > 1 or force_dist) else "none (1 GPU)",
                   "exchange_transport": transport, "rccl_nranks": rccl_nranks,
                   "inputs": "resident in HBM before the timed region",
                   "build_row_mode": ("sorted positions in each index (cph_join_chain_ex CPH_CHAIN_POSITIONS; the reference's row handle: "
                                      "csvplus.go:553-567 reads index.impl.rows[first()+i])" if POS else "original row ids (--row-ids)")},
        "joined_rows_per_step": total_joined,
        "per_rank_ms_per_step": [round(x, 4) for x in per_rank_ms],
        "exchange_ms": multi["exchange_ms"] if multi else None,
        "compute_ms": (round(ms_per_step - (multi["exposed_exchange_ms"] or 0.0), 4) if multi else round(ms_per_step, 4)),
        "efficiency_vs_n1": multi["efficiency_vs_n1"] if multi else None,
        "multi_gpu": multi,
        "index_build": {"GBps_algorithmic": round(build_gb / (build_ms / 1e3), 1) if build_ms else None,
                        "kernel_ms_per_step": round(build_ms / K, 4),
                        "customers": ia_info, "products": ib_info},
        "kernel_ms_per_step": round(kernel_ms_per_step, 4),
        "kernels": kernels,
        "kernels_note": f"per-kernel times from {K} fully profiled steps run behind the timed region (two HIP events per launch "
                        f"slow a step by ~10 %); {dom_name} is the one kernel timed inside the timed region itself",
        "roofline": roofline,
        "host": {"nproc": os.cpu_count(), "cgroup_cpu_max": _cgroup_cpu_max(), "gpu": torch.cuda.get_device_name(dev), "datagen_s": round(gen_s, 1)},
    }

    # Everything from here to the print is measured and reported BESIDE the contract fields above.  A failure in one of these
    # blocks (a verification helper, an extra measurement, the CPU baseline) must not cost the run its line: the line is printed with
    # what was gathered, `extras_error` says what broke, and `verified` is false unless the checks had already passed.
    try:
        if os.environ.get("CPH_BENCH_FAIL_EXTRAS") == "1":   # test hook (tests/test_bench_launch.py)
            raise RuntimeError("CPH_BENCH_FAIL_EXTRAS")
        # ---- full-size verification of what was just timed (outside the timed region) ----------------
        gpu_prefix = None
        if world == 1 and not args.no_verify:
            from csvplus_amd import verify as V

            t0 = time.perf_counter()
            ia, ib = eng.index_on_many([[d_cust], [d_prod]], unique=True)
            from csvplus_amd.engine import device_view
            res = eng.chained_join([(ia, d_ord["cust_id"]), (ib, d_ord["prod_id"])], probe_base=begin, positions=POS)
            all_joined = res.n == nloc and res.stream_row is None
            ver = {"joined_rows": res.n, "every_stream_row_joined_once": all_joined, "output_mode": "sorted positions" if POS else "original row ids"}
            # what was timed reports sorted positions: the row a position names is perm[position] (cph_index_perm) — the checks
            # below (key equality at the emitted row, digests, the oracle prefix) run on those rows
            brows = list(res.build_rows)
            if POS and res.n:
                for k, ix in enumerate((ia, ib)):
                    pk = device_view(ix.perm_device_ptr(), ix.nrows, "<i4", ix, dev)
                    ver[f"digest_positions_{k}"] = f"{V.digest_u64(res.build_rows[k]):016x}"
                    brows[k] = pk[res.build_rows[k].long()]
            if all_joined:
                rows = V.sample_rows(nloc, args.verify_sample)
                idx = torch.from_numpy(rows).to(dev)
                b0 = brows[0][idx].cpu().numpy()
                b1 = brows[1][idx].cpu().numpy()
                ver["sample_rows"] = int(rows.size)
                # csvplus.go:553-567: the emitted build row is the one whose key equals the stream row's key
                ver["cust_key_mismatches"] = V.check_join_sample(ords["cust_id"], cust_id, b0, rows)
                ver["prod_key_mismatches"] = V.check_join_sample(ords["prod_id"], prod_id, b1, rows)
                ver["digest_cust_rows"] = f"{V.digest_u64(brows[0]):016x}"
                ver["digest_prod_rows"] = f"{V.digest_u64(brows[1]):016x}"
                ns = min(args.cpu_sample_rows, nloc)
                gpu_prefix = (brows[0][:ns].cpu().numpy().view(np.uint32).copy(),
                              brows[1][:ns].cpu().numpy().view(np.uint32).copy())
            # the two indexes of the step: perm is a permutation, keys ascend through it, ties keep input order
            for name, ix, col in (("customers", ia, d_cust), ("products", ib, d_prod)):
                perm = device_view(ix.perm_device_ptr(), ix.nrows, "<i4", ix, dev)
                ver[f"index_{name}"] = V.check_index_order(col, perm)
            ok = all_joined and ver.get("cust_key_mismatches") == 0 and ver.get("prod_key_mismatches") == 0 \
                and all(ver[f"index_{n_}"].get("ok") for n_ in ("customers", "products"))
            res.release(); ia.close(); ib.close()
            ver["seconds"] = round(time.perf_counter() - t0, 2)
            out["verified"] = bool(ok)
            out["verify"] = ver

        # ---- the same step in the OTHER output mode ------------------------------------------------------------------
        # The reference's Join reads index.impl.rows[first() + i]: rows of an Index are kept in sorted order (csvplus.go:736,
        # :553-567), so the position in the sorted index IS its row handle — it is what the cgo shim and the
        # C++ facade consume; the original row id is one more indirection (perm[position]) that this ABI ALSO offers
        # (cph_join_chain).  Positions let a duplicate-free index over a dense code space answer from presence bits + a running
        # count per 32 codes (2.5 MB for the 1e7 customers: L2 resident) instead of the 40 MB row table (one Infinity-Fabric
        # sector per probe row).  The timed step reports positions (row ids with --row-ids); the other mode is measured here the
        # same way (same builds, same inputs, K steps between synchronisations) and reported beside `value`.  The two results are
        # compared at full size: perm[position] == row id for all rows of both steps.
        if world == 1 and not args.no_positions:
            OTHER = not POS                   # the other mode reports positions?
            other_name = "join_positions" if OTHER else "join_row_ids"

            def step_other():
                ia, ib = eng.index_on_many([[d_cust], [d_prod]], unique=True)
                ch = N.join_chain(eng.ctx, [(ia, [d_ord["cust_id"]]), (ib, [d_ord["prod_id"]])], probe_base=begin,
                                  out_mem=N.CPH_MEM_DEVICE, positions=OTHER)
                n = ch.nrows
                ch.release(); ia.close(); ib.close()
                return n

            for _ in range(max(1, args.warmup)):
                step_other()
            eng.ctx.profile_only("k_chain_dense")
            eng.ctx.profile_read(reset=True)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                jp = step_other()
            torch.cuda.synchronize(dev)
            dtp = time.perf_counter() - t0
            pp = eng.ctx.profile_read(reset=True)
            eng.ctx.profile(True)
            step_other()
            pb = eng.ctx.profile_read(reset=True)
            eng.ctx.profile(False)
            ms_o = dtp / args.steps * 1e3
            kd = pp.get("k_chain_dense", {"launches": 0, "total_ms": 0.0})
            kd_ms = kd["total_ms"] / max(1, kd["launches"])
            # the OTHER mode's own byte model (chain_bytes above): row ids pay one 4-byte entry per row and step, positions the
            # rank tables once
            algo_o, hbm_o = chain_bytes(OTHER)
            blk = {"mode": "sorted positions" if OTHER else "original row ids",
                   "ms_per_step": round(ms_o, 4), "value": jp / (dtp / args.steps), "unit": "rows/s", "joined_rows_per_step": jp,
                   "timed_step_over_this": round(ms_per_step / ms_o, 3),
                   "k_chain_dense_ms": round(kd_ms, 4),
                   "kernels_ms": {k: round(v["total_ms"], 4) for k, v in pb.items()},
                   "what": "the same step (2 index builds + chained Join of the same rows) in the other output mode; reported beside "
                           "`value`, not as it"}
            if algo_o and kd_ms:
                blk["roofline"] = {"kernel": "k_chain_dense (%s)" % ("positions" if OTHER else "row ids"),
                                   "algorithmic_bytes_per_launch": int(algo_o), "bytes_model": "chain_bytes(%s)" % ("positions" if OTHER else "row ids"),
                                   "achieved": round(algo_o / 1e9 / (kd_ms / 1e3), 1), "unit": "GB/s", "peak": HBM_PEAK_GBPS,
                                   "frac": round(algo_o / 1e9 / (kd_ms / 1e3) / HBM_PEAK_GBPS, 4),
                                   "frac_hbm": round(hbm_o / 1e9 / (kd_ms / 1e3) / HBM_PEAK_GBPS, 4)}
                if roofline.get("step_algorithmic_bytes"):
                    blk["roofline"]["step_frac"] = round(roofline["step_algorithmic_bytes"] / 1e9 / (ms_o / 1e3) / HBM_PEAK_GBPS, 4)
                if not OTHER and (roofline.get("gather_ceiling") or {}).get("ms"):   # the row-id kernel against this box's gather floor
                    blk["roofline"]["kernel_over_gather_ceiling"] = round(kd_ms / roofline["gather_ceiling"]["ms"], 3)
            if not args.no_verify:
                from csvplus_amd.engine import device_view
                ia, ib = eng.index_on_many([[d_cust], [d_prod]], unique=True)
                r_rows = eng.chained_join([(ia, d_ord["cust_id"]), (ib, d_ord["prod_id"])], probe_base=begin)
                r_pos = eng.chained_join([(ia, d_ord["cust_id"]), (ib, d_ord["prod_id"])], probe_base=begin, positions=True)
                same = r_rows.n == r_pos.n and (r_rows.stream_row is None) == (r_pos.stream_row is None)
                bad = []
                for k, ix in enumerate((ia, ib)):
                    perm = device_view(ix.perm_device_ptr(), ix.nrows, "<i4", ix, dev)
                    bad.append(int((perm[r_pos.build_rows[k].long()] != r_rows.build_rows[k]).sum().item()) if same else -1)
                blk["verify"] = {"rows": r_pos.n, "perm_of_position_differs_from_row_id": bad, "ok": bool(same and not any(bad))}
                out["verified"] = bool(out.get("verified")) and blk["verify"]["ok"]
                r_rows.release(); r_pos.release(); ia.close(); ib.close()
            out[other_name] = blk
            if roofline is not None:   # a compact copy where the driver's record keeps it (the roofline object)
                roofline["other_output_mode"] = {
                    "mode": blk["mode"], "k_chain_dense_ms": blk["k_chain_dense_ms"], "frac": (blk.get("roofline") or {}).get("frac"),
                    "ms_per_step": blk["ms_per_step"], "value": blk["value"],
                    "positions_equal_row_ids_through_perm": (blk.get("verify") or {}).get("ok"),
                    "note": "the same step in the other output mode of the ABI (details under %s); fractions on THIS object's byte "
                            "model.  Rounds 1-2 timed the row-id mode (cph_join_chain); since round 3 the timed step reports sorted "
                            "positions (cph_join_chain_ex, CPH_CHAIN_POSITIONS): the reference's own row handle" % other_name}

        # ---- the same step on other data (reported beside `value`, each timed and verified like it) --------------------
        # The timed step's customers are zero-padded 8-byte ids that fill their id space: the kernel's best case (aligned 8-byte
        # loads, arithmetic codes, position == code).  `variants` runs the SAME step — both builds + the chained Join of all
        # rows, sorted positions out — where that does not hold:
        #   itoa     customers' ids / orders' cust_id as unpadded decimal strings (the reference's own fixture format,
        #            csvplus_test.go:1241, 1321-1324): variable-length values behind 32-bit offsets, a LUT walk, a rank table
        #   half     8-byte ids over a HALF-occupied id space (1e7 ids drawn from [0, 2e7)): no identity, every key goes through the
        #            rank table of a code space twice the index
        #   sparse   12 random [a-z0-9] characters: a 62-bit code space, the radix sort and the hash probe
        #   dup      people.Join(IndexOn(orders.cust_id) over all orders rows).Join(products keyed by the orders row): a NON-unique build
        #            side — probe -> scan -> k_expand and the pre-joined second step (csvplus_test.go:252-285, 1161-1186)
        #   permute  the timed step + cph_index_permute of customers(name, surname) and products(product, price): what a consumer of
        #            positions pays per build to have its payload rows in index order (csvplus.go:736 moves the rows themselves)
        if world == 1 and not args.no_variants:
            from csvplus_amd import verify as V
            from csvplus_amd.engine import device_view

            want = {"itoa", "half", "sparse", "side", "permute", "dup"} if args.variants == "all" else set(args.variants.split(","))
            variants = {}

            def run_variant(name, what, v_cust, v_ocust, extra_build=None, sample_check=True, side_key=None):
                """v_cust: the customers' id column (host), v_ocust: the orders' cust_id column (host, all rows).  side_key: a HOST
                column of the customers table the products step reads its key from (cph_chain_step.source = 1) instead of
                orders.prod_id."""
                torch.cuda.empty_cache()
                dc, do_ = v_cust.to_device(dev), v_ocust.to_device(dev)
                d_side = side_key.to_device(dev) if side_key is not None else None
                keep = []

                def chain_of(a, b):
                    return [(a, [do_]), (b, [d_side], 1)] if d_side is not None else [(a, [do_]), (b, [d_ord["prod_id"]])]

                def vstep():
                    a, b = eng.index_on_many([[dc], [d_prod]], unique=True)
                    if extra_build:
                        keep[:] = extra_build(a, b)
                    c = N.join_chain(eng.ctx, chain_of(a, b), probe_base=begin, out_mem=N.CPH_MEM_DEVICE, positions=True)
                    n_ = c.nrows
                    inf_ = (a.info(), b.info()) if "info" not in vstep.__dict__ else vstep.info
                    vstep.info = inf_
                    c.release()
                    for x in keep:
                        x.release()
                    keep[:] = []
                    a.close(); b.close()
                    return n_

                for _ in range(max(1, args.warmup)):
                    vstep()
                eng.ctx.profile_only("k_chain_dense")
                eng.ctx.profile_read(reset=True)
                torch.cuda.synchronize(dev)
                t0_ = time.perf_counter()
                for _ in range(args.steps):
                    nj = vstep()
                torch.cuda.synchronize(dev)
                dt_ = (time.perf_counter() - t0_) / args.steps
                pk = eng.ctx.profile_read(reset=True)
                eng.ctx.profile(True)
                vstep()
                pb_ = eng.ctx.profile_read(reset=True)
                eng.ctx.profile(False)
                kd_ = pk.get("k_chain_dense", {"launches": 0, "total_ms": 0.0})
                kd_ms_ = kd_["total_ms"] / kd_["launches"] if kd_["launches"] else None
                ia_i, ib_i = vstep.info
                # byte model of the chain pass on THIS data: both key columns' value bytes + offsets in, two 4-byte
                # positions out per joined row, each lookup structure charged ONCE at its size (rank table: 8 B per 32 codes of a
                # code space the index does not fill; hash table: its sectors), as in roofline.bytes_model of the timed step
                s_in = v_ocust.nbytes_values() + v_ocust.nbytes_offsets() + host_bytes["prod_id"] + ords["prod_id"].nbytes_offsets()
                if side_key is not None:   # per joined row: perm[position] (4 B), two offsets (8 B) and the key bytes of the customer row it matched
                    s_in = v_ocust.nbytes_values() + v_ocust.nbytes_offsets() + nloc * (4.0 + 8.0 + side_key.nbytes_values() / side_key.nrows)
                look = 0.0
                for inf_, rows_ in ((ia_i, args.customers), (ib_i, args.products)):
                    if inf_["hash_bytes"]:
                        look += inf_["hash_bytes"]
                    elif inf_["table_entries"] and inf_["table_entries"] != rows_:
                        look += 0.25 * inf_["table_entries"]
                algo = s_in + look + 8.0 * nj
                kname = "k_chain_dense"
                if side_key is not None and "k_chain_prejoined" in pb_:
                    # pre-joined build sides: the chain is three kernels — the table pass over the customers' side column,
                    # the fused pass over the stream-keyed step, the gather pass — timed together (one fully profiled step); bytes: the
                    # stream's key column in, one 4-byte pre-joined entry per row (contract model, as for row ids), two positions out,
                    # + the table pass (side column in, 4 B out, perm 4 B + 4 B out for the sorted order) per customers row
                    kname = "k_chain_prejoin_table + k_chain_dense + k_chain_prejoined"
                    kd_ms_ = sum(pb_[k]["total_ms"] for k in ("k_chain_prejoin_table", "k_chain_dense", "k_chain_prejoined") if k in pb_)
                    s_in = v_ocust.nbytes_values() + v_ocust.nbytes_offsets() + 4.0 * nloc
                    look = side_key.nbytes_values() + side_key.nbytes_offsets() + 12.0 * side_key.nrows
                    algo = s_in + look + 8.0 * nj
                blk = {"what": what, "ms_per_step": round(dt_ * 1e3, 4), "value": nj / dt_, "unit": "rows/s", "joined_rows_per_step": nj,
                       "timed_step_over_this": round(ms_per_step / (dt_ * 1e3), 3),
                       "k_chain_dense_ms": round(kd_ms_, 4) if kd_ms_ else None,
                       "kernels_ms": {k: round(v["total_ms"], 4) for k, v in sorted(pb_.items(), key=lambda kv: -kv[1]["total_ms"])},
                       "customers_index": ia_i,
                       "roofline": {"kernel": kname if kd_ms_ else None, "algorithmic_bytes_per_launch": round(algo),
                                    "bytes_model": {"streams_in": round(s_in), "lookup_structures_once": round(look), "results_out": 8 * nj},
                                    "achieved": round(algo / 1e9 / (kd_ms_ / 1e3), 1) if kd_ms_ else None, "unit": "GB/s", "peak": HBM_PEAK_GBPS,
                                    "frac": round(algo / 1e9 / (kd_ms_ / 1e3) / HBM_PEAK_GBPS, 4) if kd_ms_ else None}}
                if not args.no_verify:
                    a, b = eng.index_on_many([[dc], [d_prod]], unique=True)
                    if d_side is not None:
                        ch_ = N.join_chain(eng.ctx, chain_of(a, b), probe_base=begin, out_mem=N.CPH_MEM_DEVICE, positions=True)
                        p_ = ch_.device_ptrs()
                        from csvplus_amd.engine import ChainResult
                        res_ = ChainResult(None if ch_.identity else device_view(p_["stream_row"], ch_.nrows, "<i8", ch_, dev),
                                           [device_view(q_, ch_.nrows, "<i4", ch_, dev) for q_ in p_["build_row"]], ch_.nrows, keep=(ch_,), stream_base=begin)
                    else:
                        res_ = eng.chained_join([(a, do_), (b, d_ord["prod_id"])], probe_base=begin, positions=True)
                    allj = res_.n == nloc and res_.stream_row is None
                    ver_ = {"joined_rows": res_.n, "every_stream_row_joined_once": allj}
                    ok_ = allj
                    if allj and sample_check:
                        rows_ = V.sample_rows(nloc, args.verify_sample)
                        idx_ = torch.from_numpy(rows_).to(dev)
                        pk_ = device_view(a.perm_device_ptr(), a.nrows, "<i4", a, dev)
                        b0_ = pk_[res_.build_rows[0][idx_].long()].cpu().numpy()
                        ver_["sample_rows"] = int(rows_.size)
                        ver_["cust_key_mismatches"] = V.check_join_sample(v_ocust, v_cust, b0_, rows_)
                        ver_["digest_positions_0"] = f"{V.digest_u64(res_.build_rows[0]):016x}"
                        ok_ = ok_ and ver_["cust_key_mismatches"] == 0
                        if side_key is not None:   # the product a row reports carries the key of the CUSTOMER row it matched
                            pb2_ = device_view(b.perm_device_ptr(), b.nrows, "<i4", b, dev)
                            b1_ = pb2_[res_.build_rows[1][idx_].long()].cpu().numpy()
                            ver_["side_key_mismatches"] = V.check_join_sample(side_key, prod_id, b1_, b0_.astype(np.int64) & 0xFFFFFFFF)
                            ok_ = ok_ and ver_["side_key_mismatches"] == 0
                            del pb2_
                        del pk_, idx_
                    pm_ = device_view(a.perm_device_ptr(), a.nrows, "<i4", a, dev)
                    ver_["index_customers"] = V.check_index_order(dc, pm_)
                    ok_ = ok_ and bool(ver_["index_customers"].get("ok"))
                    del pm_
                    res_.release(); a.close(); b.close()
                    blk["verified"] = bool(ok_)
                    blk["verify"] = ver_
                del dc, do_
                torch.cuda.empty_cache()
                return blk

            def guarded(name, fn):
                try:
                    variants[name] = fn()
                except Exception as ex:   # noqa: BLE001 — a variant never takes the headline down; the error is the record
                    variants[name] = {"error": f"{type(ex).__name__}: {ex}"}

            if "itoa" in want:
                guarded("itoa_ids", lambda: run_variant(
                    "itoa_ids", "customers.id / orders.cust_id as unpadded decimal strings (strconv.Itoa, the reference's fixture format: "
                    "csvplus_test.go:1241, 1321-1324): 1-8 byte values behind 32-bit offsets",
                    dg.column(dg.SEQ_PERM, args.customers, args.customers, encoding=dg.ITOA, seed=dg.SEED + 1),
                    dg.column(dg.UNIFORM, nloc, args.customers, encoding=dg.ITOA, seed=dg.SEED + 3, row0=begin)))
            if "half" in want:
                guarded("half_occupied_ids", lambda: run_variant(
                    "half_occupied_ids", "8-byte zero-padded ids drawn from an id space TWICE the table (%d ids out of [0, %d)): no identity "
                    "lookup, every key goes through the rank table" % (args.customers, 2 * args.customers),
                    dg.column(dg.SEQ_PERM, args.customers, 2 * args.customers, encoding=dg.FIXED8, seed=dg.SEED + 11),
                    dg.column(dg.FK_SUBSET, nloc, 2 * args.customers, encoding=dg.FIXED8, base=args.customers, seed=dg.SEED + 11, row0=begin)))
            if "sparse" in want:
                guarded("sparse_random_keys", lambda: run_variant(
                    "sparse_random_keys", "customers keyed by 12 random [a-z0-9] characters (62-bit code space): radix-sorted index, hash probe",
                    dg.column(dg.RANDKEY, args.customers, 0, seed=dg.SEED + 12),
                    dg.column(dg.RANDKEY, nloc, 0, base=args.customers, seed=dg.SEED + 12, row0=begin)))
            if "side" in want:
                guarded("build_side_key", lambda: run_variant(
                    "build_side_key", "orders.Join(customers, cust_id).Join(products, fav_prod) with fav_prod a column of the CUSTOMERS table "
                    "(cph_chain_step.source = 1; the shape of the reference's people.Join(orders).Join(products), csvplus_test.go:280-285): "
                    "the build sides are joined with each other first (one pass over the customers' fav_prod column), a stream row then takes "
                    "ONE 4-byte gather for that step (chain.hip: run_prejoined; round 5's first version gathered and coded the key per stream "
                    "row: 7.0 ms)", cust_id, ords["cust_id"],
                    side_key=dg.column(dg.UNIFORM, args.customers, args.products, encoding=dg.ITOA, seed=dg.SEED + 21)))
            if "permute" in want:
                from csvplus_amd.materialize import permute_col

                cust_pay = [dg.column(k_, args.customers, args.customers, seed=dg.SEED + 1).to_device(dev) for k_ in (dg.NAME, dg.SURNAME)]
                prod_pay = [dg.column(k_, args.products, args.products, seed=dg.SEED + 2).to_device(dev) for k_ in (dg.PRODUCT, dg.PRICE)]

                def lay_out(a, b):
                    return [permute_col(eng.ctx, a, c_) for c_ in cust_pay] + [permute_col(eng.ctx, b, c_) for c_ in prod_pay]

                guarded("step_plus_permute", lambda: run_variant(
                    "step_plus_permute", "the timed step + cph_index_permute of customers(name, surname) and products(product, price) inside "
                    "every step: the payload rows in index order, what a consumer of sorted positions needs per build (csvplus.go:736 moves "
                    "the rows; :553-567 reads index.impl.rows[i])", cust_id, ords["cust_id"], extra_build=lay_out))

                def to_csv_block():
                    """The step AFTER the timed path: orders.Join(customers).Join(products).ToCsv(cust_id, qty, name, surname,
                    product, price) — mergeRows (csvplus.go:571-583) folded into the writer (:379-406) over the joined rows of the timed
                    chain, text left in HBM.  One pass (slot tables + decoupled look-back, materialize.hip) beside the two-pass writer; the
                    two texts must be the same bytes and the first rows equal to the oracle's Writer."""
                    from csvplus_amd.materialize import csv_write

                    torch.cuda.empty_cache()
                    ia_, ib_ = eng.index_on_many([[d_cust], [d_prod]], unique=True)
                    ch_ = N.join_chain(eng.ctx, [(ia_, [d_ord["cust_id"]]), (ib_, [d_ord["prod_id"]])], probe_base=begin, out_mem=N.CPH_MEM_DEVICE,
                                       positions=True)
                    pay_ = lay_out(ia_, ib_)
                    try:
                        n_, p_ = ch_.nrows, ch_.device_ptrs()
                        if n_ != nloc or p_["stream_row"]:
                            raise RuntimeError("to_csv block expects every order to join once")
                        cols_ = [d_ord["cust_id"], d_ord["qty"]] + [c_.as_device_strcol() for c_ in pay_]
                        ids_ = [None, None] + [(p_["build_row"][0], 32, n_)] * 2 + [(p_["build_row"][1], 32, n_)] * 2
                        names_ = ["cust_id", "qty", "name", "surname", "product", "price"]

                        def timed(mode, reps):
                            eng.ctx.set_option("csv_onepass", mode)
                            try:
                                csv_write(eng.ctx, cols_, names_, out_mem=N.CPH_MEM_DEVICE, row_ids=ids_, nrows=n_).release()
                                eng.ctx.profile(True)
                                eng.ctx.profile_read(reset=True)
                                torch.cuda.synchronize(dev)
                                t0_ = time.perf_counter()
                                for _ in range(reps):
                                    csv_write(eng.ctx, cols_, names_, out_mem=N.CPH_MEM_DEVICE, row_ids=ids_, nrows=n_).release()
                                torch.cuda.synchronize(dev)
                                ms_ = (time.perf_counter() - t0_) / reps * 1e3
                                pk_ = eng.ctx.profile_read(reset=True)
                                eng.ctx.profile(False)
                                text_ = csv_write(eng.ctx, cols_, names_, out_mem=N.CPH_MEM_DEVICE, row_ids=ids_, nrows=n_)
                                return ms_, {k: round(v["total_ms"] / reps, 4) for k, v in sorted(pk_.items(), key=lambda kv: -kv[1]["total_ms"])[:6]}, text_
                            finally:
                                eng.ctx.set_option("csv_onepass", 1)

                        ms1, k1, t1 = timed(1, 3)
                        ms0, k0, t0x = timed(0, 2)
                        size_ = len(t1)
                        blk = {"what": "ToCsv of the timed chain's %d joined rows, 6 columns (2 stream, 2 + 2 gathered through sorted positions), text left "
                                       "in HBM; bytes per second of TEXT written" % n_, "text_bytes": size_,
                               "ms": round(ms1, 3), "TBps": round(size_ / ms1 / 1e9, 3), "kernels_ms": k1,
                               "two_pass_ms": round(ms0, 3), "two_pass_TBps": round(size_ / ms0 / 1e9, 3), "two_pass_kernels_ms": k0,
                               "one_pass_taken": "k_csv_onepass" in k1}
                        if not args.no_verify:
                            from oracle import orc   # the checker

                            def dev_bytes(t_, off, length):   # bytes [off, off + length) of a text in HBM (the blocks end in >= 16 spare bytes)
                                a0 = off // 8 * 8
                                w_ = device_view(t_.data_ptr + a0, (off + length - a0 + 7) // 8, "<i8", t_, dev).cpu().numpy().tobytes()
                                return w_[off - a0: off - a0 + length]

                            same = len(t0x) == size_
                            if same:
                                v1 = device_view(t1.data_ptr, size_ // 8, "<i8", t1, dev)
                                v0 = device_view(t0x.data_ptr, size_ // 8, "<i8", t0x, dev)
                                same = bool(torch.equal(v1, v0)) and dev_bytes(t1, size_ - 16, 16) == dev_bytes(t0x, size_ - 16, 16)
                                del v1, v0
                            # the first rows against the oracle's restatement of the Writer
                            k_ = min(50_000, n_)
                            # sorted position -> original table row through the index's permutation; the values from the host tables
                            perm_a = device_view(ia_.perm_device_ptr(), ia_.nrows, "<i4", ia_, dev)
                            perm_b = device_view(ib_.perm_device_ptr(), ib_.nrows, "<i4", ib_, dev)
                            pos_a = device_view(p_["build_row"][0], n_, "<i4", ch_, dev)[:k_].long() & 0xFFFFFFFF
                            pos_b = device_view(p_["build_row"][1], n_, "<i4", ch_, dev)[:k_].long() & 0xFFFFFFFF
                            ra = (perm_a[pos_a].long() & 0xFFFFFFFF).cpu().tolist()
                            rb = (perm_b[pos_b].long() & 0xFFFFFFFF).cpu().tolist()
                            del perm_a, perm_b, pos_a, pos_b
                            from csvplus_amd import StrCol
                            hc = [dg.column(k2, args.customers, args.customers, seed=dg.SEED + 1) for k2 in (dg.NAME, dg.SURNAME)]
                            hp = [dg.column(k2, args.products, args.products, seed=dg.SEED + 2) for k2 in (dg.PRODUCT, dg.PRICE)]
                            want_ = orc.csv_write([ords["cust_id"].slice(0, k_), ords["qty"].slice(0, k_)] +
                                                  [StrCol.from_values([c2.value(i) for i in ra]) for c2 in hc] +
                                                  [StrCol.from_values([c2.value(i) for i in rb]) for c2 in hp], names_)
                            head_ok = dev_bytes(t1, 0, len(want_)) == want_
                            blk["verify"] = {"one_pass_equals_two_pass_bytes": same, "oracle_prefix_rows": k_, "oracle_prefix_bytes_equal": head_ok}
                            blk["verified"] = bool(same and head_ok)
                        t1.release()
                        t0x.release()
                        return blk
                    finally:
                        for c_ in pay_:
                            c_.release()
                        ch_.release()
                        ia_.close()
                        ib_.close()

                try:
                    out["to_csv"] = to_csv_block()
                except Exception as ex:   # noqa: BLE001 — an extra never takes the headline down; the error is the record
                    out["to_csv"] = {"error": f"{type(ex).__name__}: {ex}"}

                def csv_parse_block():
                    """The step IN FRONT of the timed path: orders.csv — a header line + one record of 3 fields per orders row,
                    written into HBM by cph_csv_write — parsed back into columns by cph_csv_parse (Reader.Iterate, csvplus.go:1080-1146).
                    Checked at full size through the round trip parse(write(columns)) == columns: every value's bytes and length."""
                    from csvplus_amd import ingest
                    from csvplus_amd.materialize import csv_write

                    torch.cuda.empty_cache()
                    names_ = ["cust_id", "prod_id", "qty"]
                    cols_ = [d_ord[k2] for k2 in names_]
                    text_ = csv_write(eng.ctx, cols_, names_, out_mem=N.CPH_MEM_DEVICE)
                    tab_ = None
                    try:
                        size_ = len(text_)

                        def parse():
                            return ingest.csv_parse(eng.ctx, None, [0, 1, 2], fields_per_record=3, skip_records=1, out_mem=N.CPH_MEM_DEVICE,
                                                    device_ptr=text_.data_ptr, size=size_)

                        parse().release()
                        eng.ctx.profile(True)
                        eng.ctx.profile_read(reset=True)
                        torch.cuda.synchronize(dev)
                        t0_ = time.perf_counter()
                        for _ in range(3):
                            parse().release()
                        torch.cuda.synchronize(dev)
                        ms_ = (time.perf_counter() - t0_) / 3 * 1e3
                        pk_ = eng.ctx.profile_read(reset=True)
                        eng.ctx.profile(False)
                        tab_ = parse()
                        blk = {"what": "cph_csv_parse of orders.csv in HBM (header + %d records x 3 fields, written by cph_csv_write), 3 columns out, left in HBM; "
                                       "bytes per second of TEXT read" % nloc, "text_bytes": size_, "ms": round(ms_, 3), "GBps": round(size_ / ms_ / 1e6, 1),
                               "records": tab_.nrecords, "kernels_ms": {k: round(v["total_ms"] / 3, 4) for k, v in sorted(pk_.items(), key=lambda kv: -kv[1]["total_ms"])[:5]},
                               "fast_path_taken": "k_csv_fast_copy" in pk_}
                        if not args.no_verify:
                            ok_ = tab_.nrecords == nloc and tab_.error_kind == 0
                            ver_ = {"records": tab_.nrecords, "error_kind": tab_.error_kind}
                            for k2, oc, pc in zip(names_, cols_, tab_.columns):
                                nb = ords[k2].nbytes_values()   # (the host twin of the device column knows its size)
                                pdat = device_view(pc.data.data_ptr(), nb // 8, "<i8", tab_, dev)
                                odat = oc.data[: nb // 8 * 8].view(torch.int64)
                                same_bytes = bool(torch.equal(pdat, odat))
                                poff = device_view(pc.offsets.data_ptr(), nloc + 1, "<i4" if pc.offset_bits == 32 else "<i8", tab_, dev).long()
                                if pc.offset_bits == 32:
                                    poff = poff & 0xFFFFFFFF
                                if oc.fixed_width:
                                    same_lens = bool((poff == torch.arange(nloc + 1, device=dev, dtype=torch.int64) * oc.fixed_width).all().item())
                                else:
                                    ooff = oc.offsets.view(torch.int32 if oc.offset_bits == 32 else torch.int64).long()
                                    if oc.offset_bits == 32:
                                        ooff = ooff & 0xFFFFFFFF
                                    same_lens = bool(torch.equal(poff - poff[0], ooff - ooff[0]))
                                    del ooff
                                ver_[k2] = {"value_bytes_equal": same_bytes, "lengths_equal": same_lens, "bytes": nb}
                                ok_ = ok_ and same_bytes and same_lens
                                del pdat, odat, poff
                            blk["verify"] = ver_
                            blk["verified"] = bool(ok_)
                        return blk
                    finally:
                        if tab_ is not None:
                            tab_.release()
                        text_.release()

                try:
                    out["csv_parse"] = csv_parse_block()
                except Exception as ex:   # noqa: BLE001
                    import traceback
                    out["csv_parse"] = {"error": f"{type(ex).__name__}: {ex}", "traceback": traceback.format_exc()[-800:]}
                del cust_pay, prod_pay
            if "dup" in want:
                def dup_build_side():
                    """people(1e7).Join(IndexOn(orders.cust_id) over ALL orders rows, "id").Join(UniqueIndexOn(products.prod_id), prod_id of the
                    ORDERS row): the reference's TestLongChain / BenchmarkJoinOnBiggerMultiIndex shape (csvplus_test.go:252-285, 1161-1186;
                    csvplus.go:559 emits EVERY equal index row, in ascending index position) — a non-unique build side 10x the stream, so the
                    Join is probe -> scan -> k_expand, and the second Join reads its key from the orders row the first one
                    matched (cph_chain_step.source = 1), answered from the orders table joined with the products index once."""
                    torch.cuda.empty_cache()
                    d_people = cust_id.to_device(dev)        # people.id: the same 1e7 distinct 8-byte ids

                    def chain_of(io, ip):
                        return [(io, [d_people]), (ip, [d_ord["prod_id"]], 1)]

                    def dstep():
                        io, ip = eng.index_on_many([[d_ord["cust_id"]], [d_prod]], unique=[False, True])
                        c = N.join_chain(eng.ctx, chain_of(io, ip), out_mem=N.CPH_MEM_DEVICE, positions=True)
                        n_ = c.nrows
                        dstep.info = (io.info(), ip.info())
                        c.release(); io.close(); ip.close()
                        return n_

                    for _ in range(max(1, args.warmup)):
                        dstep()
                    torch.cuda.synchronize(dev)
                    t0_ = time.perf_counter()
                    for _ in range(args.steps):
                        nj = dstep()
                    torch.cuda.synchronize(dev)
                    dt_ = (time.perf_counter() - t0_) / args.steps
                    eng.ctx.profile(True)
                    eng.ctx.profile_read(reset=True)
                    dstep()
                    pb_ = eng.ctx.profile_read(reset=True)
                    eng.ctx.profile(False)
                    # the Join alone, indexes resident (what a caller that keeps its Index pays per Join)
                    io, ip = eng.index_on_many([[d_ord["cust_id"]], [d_prod]], unique=[False, True])
                    N.join_chain(eng.ctx, chain_of(io, ip), out_mem=N.CPH_MEM_DEVICE, positions=True).release()
                    torch.cuda.synchronize(dev)
                    t0_ = time.perf_counter()
                    for _ in range(args.steps):
                        N.join_chain(eng.ctx, chain_of(io, ip), out_mem=N.CPH_MEM_DEVICE, positions=True).release()
                    torch.cuda.synchronize(dev)
                    join_ms = (time.perf_counter() - t0_) / args.steps * 1e3
                    t0_ = time.perf_counter()
                    for _ in range(args.steps):
                        N.join_chain(eng.ctx, [(io, [d_people])], out_mem=N.CPH_MEM_DEVICE, positions=True).release()
                    torch.cuda.synchronize(dev)
                    join1_ms = (time.perf_counter() - t0_) / args.steps * 1e3
                    join_names = ("k_probe_table", "k_probe_rank", "k_probe", "k_probe_fast", "exclusive_scan_u64", "k_expand", "k_chain_prejoin_table",
                                  "k_prejoin_tuples", "k_sum_counts", "k_compose")
                    jk_ms = sum(pb_[k]["total_ms"] for k in join_names if k in pb_)
                    # byte model of the two Joins (per launch set): stream keys in; (lo, cnt) per stream row written and read back by the scan /
                    # expand; per pair 8 B stream row + 4 B position out; the orders table's prod_id column + offsets in and 4 B out per
                    # orders row (table pass); per tuple position 4 + perm 4 + table entry 4 in, 4 out
                    npeople = args.customers
                    s_in = cust_id.nbytes_values() + cust_id.nbytes_offsets() + 16.0 * npeople
                    pairs = 12.0 * nj
                    tpass = host_bytes["prod_id"] + ords["prod_id"].nbytes_offsets() + 4.0 * nloc
                    tup = 16.0 * nj
                    algo = s_in + pairs + tpass + tup
                    blk = {"what": "people(%d).Join(IndexOn(orders.cust_id) over %d rows, id).Join(UniqueIndexOn(products.prod_id), prod_id of the ORDERS row): "
                                   "a NON-unique build side (csvplus_test.go:252-285, 1161-1186; csvplus.go:559); step = both builds + the chain" % (npeople, nloc),
                           "ms_per_step": round(dt_ * 1e3, 4), "value": nj / dt_, "unit": "rows/s", "joined_rows_per_step": nj,
                           "timed_step_over_this": round(ms_per_step / (dt_ * 1e3), 3),
                           "k_chain_dense_ms": round(jk_ms, 4),
                           "join_only_ms": round(join_ms, 4), "first_join_only_ms": round(join1_ms, 4),
                           "first_join_pairs_per_s": nj / (join1_ms / 1e3),
                           "kernels_ms": {k: round(v["total_ms"], 4) for k, v in sorted(pb_.items(), key=lambda kv: -kv[1]["total_ms"])},
                           "customers_index": dstep.info[0],
                           "roofline": {"kernel": "probe + scan + k_expand + k_chain_prejoin_table + k_prejoin_tuples (the two Joins' kernels, summed)",
                                        "algorithmic_bytes_per_launch": round(algo),
                                        "bytes_model": {"stream_keys_and_bounds": round(s_in), "pairs_out": round(pairs), "table_pass": round(tpass),
                                                        "tuples": round(tup)},
                                        "achieved": round(algo / 1e9 / (jk_ms / 1e3), 1) if jk_ms else None, "unit": "GB/s", "peak": HBM_PEAK_GBPS,
                                        "frac": round(algo / 1e9 / (jk_ms / 1e3) / HBM_PEAK_GBPS, 4) if jk_ms else None}}
                    if not args.no_verify:
                        ch_ = N.join_chain(eng.ctx, chain_of(io, ip), out_mem=N.CPH_MEM_DEVICE, positions=True)
                        p_ = ch_.device_ptrs()
                        n_ = ch_.nrows
                        srow = device_view(p_["stream_row"], n_, "<i8", ch_, dev) if p_["stream_row"] else None
                        pos0 = device_view(p_["build_row"][0], n_, "<i4", ch_, dev).long() & 0xFFFFFFFF
                        pos1 = device_view(p_["build_row"][1], n_, "<i4", ch_, dev).long() & 0xFFFFFFFF
                        ver_ = {"joined_rows": n_, "every_order_joined_once": n_ == nloc}
                        ok_ = n_ == nloc and srow is not None
                        if ok_:
                            # emission order (csvplus.go:553-567): stream rows ascend; inside one stream row the index positions ascend by one;
                            # every index position is emitted exactly once (the positions of all pairs are a permutation of 0..n-1)
                            same = srow[1:] == srow[:-1]
                            ver_["stream_rows_ascend"] = bool((srow[1:] >= srow[:-1]).all().item())
                            ver_["positions_consecutive_inside_a_stream_row"] = bool((pos0[1:][same] == pos0[:-1][same] + 1).all().item())
                            seen = torch.zeros(n_, dtype=torch.bool, device=dev)
                            seen[pos0] = True
                            ver_["every_index_position_once"] = bool(seen.all().item())
                            del seen, same
                            rows_ = V.sample_rows(n_, args.verify_sample)
                            idx_ = torch.from_numpy(rows_).to(dev)
                            permo = device_view(io.perm_device_ptr(), io.nrows, "<i4", io, dev)
                            permp = device_view(ip.perm_device_ptr(), ip.nrows, "<i4", ip, dev)
                            orow = (permo[pos0[idx_]].long() & 0xFFFFFFFF)
                            prow = (permp[pos1[idx_]].long() & 0xFFFFFFFF).cpu().numpy()
                            srs = srow[idx_].cpu().numpy()
                            orow_h = orow.cpu().numpy()
                            ver_["sample_rows"] = int(rows_.size)
                            # the orders row a pair names carries the person's id; the product it names carries that orders row's prod_id
                            ver_["cust_key_mismatches"] = V.check_join_sample(cust_id, ords["cust_id"], orow_h, srs)
                            ver_["prod_key_mismatches"] = V.check_join_sample(ords["prod_id"], prod_id, prow, orow_h)
                            ver_["digest_positions_0"] = f"{V.digest_u64(pos0):016x}"
                            ok_ = (ver_["stream_rows_ascend"] and ver_["positions_consecutive_inside_a_stream_row"] and ver_["every_index_position_once"]
                                   and ver_["cust_key_mismatches"] == 0 and ver_["prod_key_mismatches"] == 0)
                            del permo, permp, idx_
                        ver_["index_orders"] = V.check_index_order(d_ord["cust_id"], device_view(io.perm_device_ptr(), io.nrows, "<i4", io, dev))
                        ok_ = ok_ and bool(ver_["index_orders"].get("ok"))
                        del srow, pos0, pos1
                        ch_.release()
                        blk["verified"] = bool(ok_)
                        blk["verify"] = ver_
                    io.close(); ip.close()
                    del d_people
                    torch.cuda.empty_cache()
                    return blk

                guarded("dup_build_side", dup_build_side)
            out["variants"] = variants
            out["variants_note"] = ("the SAME step as `value` (both index builds + the chained Join of all %d rows, sorted positions out) on other key "
                                    "shapes, each timed over %d steps between synchronisations and verified like `value`; reported beside it" % (args.rows, args.steps))
            if roofline is not None:   # a compact copy inside the object the driver's record keeps
                roofline["variants"] = {k: ({"ms_per_step": v.get("ms_per_step"), "k_chain_dense_ms": v.get("k_chain_dense_ms"),
                                             "frac": (v.get("roofline") or {}).get("frac"), "verified": v.get("verified")}
                                            if "error" not in v else v) for k, v in variants.items()}
            if all("error" not in v for v in variants.values()) and not args.no_verify:
                out["verified"] = bool(out.get("verified")) and all(v.get("verified") for v in variants.values())

        # ---- end-to-end C-ABI scope: pinned host SoA in -> pinned host row ids out (PCIe inclusive) ----
        # cph_stream_join_*: 2^24-row chunks of the same orders table, H2D / kernel / D2H of consecutive chunks
        # overlapped on the pipeline's HIP streams.  Reported beside `value`, never part of it.
        if world == 1 and not args.no_e2e:
            from csvplus_amd.streaming import PinnedCol, StreamJoin

            ia = eng.index_on([d_cust], unique=True)
            ib = eng.index_on([d_prod], unique=True)
            pc = [PinnedCol(eng.ctx, ords["cust_id"]), PinnedCol(eng.ctx, ords["prod_id"])]
            chunk = 1 << 23
            bounds = [(b, min(b + chunk, nloc)) for b in range(0, nloc, chunk)]
            chunks = [[c.col.slice(b, e) for c in pc] for b, e in bounds]
            nslots, inflight = 2, 2   # two slots, both in flight: the best of 1-4 slots on every box measured (profiles/r03_stream_join_pcie.txt)
            best = None
            # ONE pipeline for all repetitions: the first pass page-locks the slots' result blocks and sizes their device
            # buffers (a long-running caller pays that once), the best of the following passes is reported
            sj = StreamJoin(eng.ctx, [ia, ib], nslots=nslots, positions=POS)
            for rep in range(4):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                sub = done = 0
                joined_e2e = 0
                while done < len(chunks):
                    while sub < len(chunks) and sj.pending < inflight:
                        sj.submit(chunks[sub], probe_base=bounds[sub][0])
                        sub += 1
                    r = sj.next(copy=False)
                    joined_e2e += r["nmatches"]
                    done += 1
                dt_e2e = time.perf_counter() - t0
                if rep > 0 and (best is None or dt_e2e < best):
                    best = dt_e2e
            sj.close()
            h2d = host_bytes["cust_id"] + host_bytes["prod_id"] + off_o
            d2h = 8 * nloc + nloc // 8
            out["e2e_pinned_host"] = {
                "scope": "pinned host key columns in -> pinned host " + ("sorted positions" if POS else "build-row ids") + " + match bitmap out (cph_stream_join_*), "
                         "indexes already built; PCIe inclusive; one pipeline reused, first pass (page-locking of the result blocks) not counted",
                "rows": nloc, "chunk_rows": chunk, "slots": nslots, "in_flight": inflight, "ms": round(best * 1e3, 2),
                "rows_per_s": nloc / best, "joined": joined_e2e,
                "h2d_GBps": round(h2d / best / 1e9, 1), "d2h_GBps": round(d2h / best / 1e9, 1)}
            # The same scope with the key codes formed ON THE HOST (cph_host_encoder_*: a pool of worker threads walks the codec's
            # table over the pinned strings) and shipped as 4 bytes per row and step (cph_stream_join_submit_codes) instead of
            # the 17 bytes of key strings: the host encode of chunk k+1 overlaps the transfers and kernels of chunk k.  The
            # encode time is INSIDE the timed loop; it is also timed alone (host_encode_ms: all chunks, nothing else running).
            try:
                from csvplus_amd.streaming import HostEncoder, PinnedArray

                encs = [HostEncoder(ia), HostEncoder(ib)]
                code_bufs = [[PinnedArray(eng.ctx, e - b) for _ in range(2)] for b, e in bounds]
                t0 = time.perf_counter()
                for ci, ch_cols in enumerate(chunks):
                    for k in range(2):
                        encs[k].run([ch_cols[k]], code_bufs[ci][k].array)
                enc_alone = time.perf_counter() - t0
                sj = StreamJoin(eng.ctx, [ia, ib], nslots=nslots, positions=POS)
                best_c = None
                import queue
                import threading

                for rep in range(4):
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    # a host application's shape: one thread forms the codes of chunk after chunk (cph_host_encoder_run hands the rows to
                    # its worker pool; ctypes drops the GIL), the other feeds the join pipeline with whatever chunk is ready
                    ready = queue.Queue()
                    enc_err = []

                    def producer():
                        try:
                            for ci in range(len(chunks)):
                                for k in range(2):
                                    encs[k].run([chunks[ci][k]], code_bufs[ci][k].array)
                                ready.put(ci)
                        except Exception as ex_:   # noqa: BLE001 — re-raised by the consumer
                            enc_err.append(ex_)
                            ready.put(None)

                    th = threading.Thread(target=producer)
                    th.start()
                    sub = done = 0
                    joined_c = 0
                    while done < len(chunks):
                        while sub < len(chunks) and sj.pending < inflight:
                            ci = ready.get()
                            if ci is None:
                                raise enc_err[0]
                            sj.submit_codes([p.array for p in code_bufs[ci]], bounds[ci][1] - bounds[ci][0], probe_base=bounds[ci][0])
                            sub += 1
                        r = sj.next(copy=False)
                        joined_c += r["nmatches"]
                        done += 1
                    th.join()
                    dt_c = time.perf_counter() - t0
                    if rep > 0 and (best_c is None or dt_c < best_c):
                        best_c = dt_c
                sj.close()
                out["e2e_pinned_host_encoded"] = {
                    "scope": "as e2e_pinned_host, but the stream's keys cross PCIe as 4-byte codes formed on the host (cph_host_encoder_run: "
                             f"{encs[0].threads} threads, on a producer thread that runs beside the submit / next loop) — host encode time "
                             "included; key strings in pinned host memory in -> "
                             "pinned host " + ("sorted positions" if POS else "build-row ids") + " + match bitmap out",
                    "rows": nloc, "ms": round(best_c * 1e3, 2), "rows_per_s": nloc / best_c, "joined": joined_c,
                    "host_encode_ms_alone": round(enc_alone * 1e3, 2), "host_threads": encs[0].threads,
                    "h2d_GBps": round(8 * nloc / best_c / 1e9, 1), "d2h_GBps": round(d2h / best_c / 1e9, 1),
                    "equals_string_pipeline": joined_c == joined_e2e}
                for e_ in encs:
                    e_.close()
                for p_ in sum(code_bufs, []):
                    p_.free()
            except Exception as ex:   # noqa: BLE001 — reported, never fatal for the headline
                out["e2e_pinned_host_encoded"] = {"error": f"{type(ex).__name__}: {ex}"}
            for c in pc:
                c.free()
            ia.close(); ib.close()

        # ---- IndexOn at 1e8 rows (the other half of BASELINE's metric; reported, not part of `value`) ---
        if world == 1 and not args.no_index_1e8:
            def time_index(col, unique, reps):
                d = col.to_device(dev)
                eng.index_on([d], unique=unique).close()   # warm-up (pool, LDS attributes)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(reps):
                    ix = eng.index_on([d], unique=unique)
                    inf = ix.info()
                    ix.close()
                torch.cuda.synchronize(dev)
                wall = (time.perf_counter() - t0) / reps
                eng.ctx.profile(True)                      # the per-kernel breakdown: one more build, outside the timing
                eng.ctx.profile_read(reset=True)
                eng.index_on([d], unique=unique).close()
                p = eng.ctx.profile_read(reset=True)
                eng.ctx.profile(False)
                check = None
                if not args.no_verify:
                    from csvplus_amd import verify as V
                    from csvplus_amd.engine import device_view
                    ix = eng.index_on([d], unique=unique)
                    check = V.check_index_order(d, device_view(ix.perm_device_ptr(), ix.nrows, "<i4", ix, dev))
                    ix.close()
                    torch.cuda.empty_cache()
                kms = sum(v["total_ms"] for v in p.values())
                n = col.nrows
                # ALGORITHMIC bytes of the build = what the kernels that RAN move (pass model): the library
                # records the bytes of every sort / scan / scan-for-duplicates launch itself (radix histogram: K bytes per
                # key, 1 byte with the digit stream; scatter: 2K+8 [+1 digit byte], first pass 2K+4; first_dup: K*w; a
                # direct table or hash table only if a kernel built one — IndexOn alone builds none); added here are the
                # passes over the SOURCE column, whose bytes the library cannot know: statistics (k_col_stats, and
                # k_group_stats when dictionary windows are completed over all rows) and the encode, which also writes the
                # n*K code bytes.  Every term is printed so that the fraction can be recomputed from the line alone.
                K_ = inf["key_bytes"] * inf["code_words"]
                src = col.nbytes_values() + col.nbytes_offsets()
                # (round 4: k_split_stats = the exact statistics pass of the delimiter-split codec; its sample passes k_split_count /
                # k_split_sample read 2^18 rows and are not charged)
                src_readers = {k: p[k]["launches"] for k in ("k_col_stats", "k_group_stats", "k_split_stats", "k_encode_build") if k in p}
                lib_bytes = {k: v["algo_bytes"] for k, v in p.items() if v["algo_bytes"] > 0}
                # (round 5: the direct sort of fixed-width 8-byte ids codes the keys inside its first partition level — no k_encode_build
                # launch, no code array; the library charges that pass with the 8 bytes of key it reads per row)
                fused_encode = "k_encode_build" not in p and "k_win_partition" in p
                algo = src * sum(src_readers.values()) + n * K_ * src_readers.get("k_encode_build", 0 if fused_encode else 1) + sum(lib_bytes.values())
                compulsory = src + n * (K_ + 4)    # the column read once, sorted codes + perm written once
                return {"rows": n, "ms": round(wall * 1e3, 3), "kernel_ms": round(kms, 3), "rows_per_s": n / wall,
                        "GBps_algorithmic": round(algo / 1e9 / wall, 1),
                        "frac_pass_model": round(algo / 1e9 / wall / HBM_PEAK_GBPS, 4),
                        "frac_compulsory": round(compulsory / 1e9 / wall / HBM_PEAK_GBPS, 4),
                        "algorithmic_bytes": round(algo), "compulsory_bytes": round(compulsory),
                        "byte_terms": {"source_bytes": src, "source_reads": src_readers, "code_bytes_written": n * K_,
                                       "library_kernels": {k: round(v) for k, v in lib_bytes.items()}},
                        "verified": (check or {}).get("ok"), "verify": check, "info": inf,
                        "kernels_ms": {k: round(v["total_ms"], 3) for k, v in p.items()}}

            n8 = args.rows
            out["index_on_1e8"] = {
                "unique_fixed8_ids": time_index(dg.column(dg.SEQ_PERM, n8, n8, encoding=dg.FIXED8, seed=7), True, 3),
                "varlen_dup_keys_config3": time_index(dg.varkeys(n8), False, 2),
            }

            # IndexOn(1e8) + Join(1e8): what the north star's 40 % is quoted on — bytes and milliseconds of the 1e8-row
            # index build and of the bench step (two build-side indexes + the chained Join of 1e8 stream rows) SUMMED
            if roofline and "step_algorithmic_bytes" in roofline:
                comb = {}
                for name, r in out["index_on_1e8"].items():
                    if not isinstance(r, dict) or "algorithmic_bytes" not in r:
                        continue
                    bts = r["algorithmic_bytes"] + roofline["step_algorithmic_bytes"]
                    ms = r["ms"] + ms_per_step
                    comb[name + "_plus_step"] = {"ms": round(ms, 3), "algorithmic_bytes": bts,
                                                 "GBps": round(bts / 1e9 / (ms / 1e3), 1),
                                                 "frac": round(bts / 1e9 / (ms / 1e3) / HBM_PEAK_GBPS, 4),
                                                 "index_ms": r["ms"], "step_ms": round(ms_per_step, 3)}
                    for oname in ("join_positions", "join_row_ids"):   # the same sum with the step in the other output mode
                        if out.get(oname):
                            msp = r["ms"] + out[oname]["ms_per_step"]
                            comb[name + "_plus_" + oname + "_step"] = {"ms": round(msp, 3), "algorithmic_bytes": bts,
                                                                        "GBps": round(bts / 1e9 / (msp / 1e3), 1),
                                                                        "frac": round(bts / 1e9 / (msp / 1e3) / HBM_PEAK_GBPS, 4),
                                                                        "index_ms": r["ms"], "step_ms": out[oname]["ms_per_step"]}
                comb["note"] = ("IndexOn over 1e8 rows (index_on_1e8.*: wall ms, pass-model bytes) + one bench step (2 index builds + "
                                "chained Join of 1e8 rows: ms_per_step, roofline.step_algorithmic_bytes), summed; frac = bytes / ms / 8 TB/s")
                out["index_plus_join_1e8"] = comb

            # IndexOn end to end through the C ABI as a cgo caller sees createIndex (csvplus.go:707-738): key column in PINNED
            # HOST memory in -> host perm out (cph_index_build on host columns + cph_index_perm(HOST)); PCIe inclusive
            if not args.no_e2e:
                from csvplus_amd.streaming import PinnedCol

                def time_index_host(col, unique, reps):
                    pc = PinnedCol(eng.ctx, col)

                    def timed():
                        ix = N.DeviceIndex(eng.ctx, [pc.col], unique=unique)   # warm-up: device pool, pinned blocks, the host worker pool
                        ix.perm_host_view()
                        ix.close()
                        torch.cuda.synchronize(dev)
                        t0 = time.perf_counter()
                        for _ in range(reps):
                            ix = N.DeviceIndex(eng.ctx, [pc.col], unique=unique)
                            pv = ix.perm_host_view()
                            first, last = int(pv[0]), int(pv[-1])
                            path = ix.info()["build_path"]
                            ix.close()
                        wall_ = (time.perf_counter() - t0) / reps
                        dig = None
                        if not args.no_verify:   # (outside the timing) the permutation's digest
                            ix = N.DeviceIndex(eng.ctx, [pc.col], unique=unique)
                            dig = V.digest_u64(np.asarray(ix.perm_host_view()))
                            ix.close()
                        return wall_, path, first, last, dig, None

                    from csvplus_amd import verify as V
                    wall, path, first, last, dig, _tb = timed()
                    h2d = col.nbytes_values() + col.nbytes_offsets()
                    d2h = 4 * col.nrows
                    blk = {"rows": col.nrows, "ms": round(wall * 1e3, 2), "rows_per_s": col.nrows / wall,
                           "build_path": {0: "strings uploaded, device encode", 2: "host-formed codes (4 B/row uploaded)"}.get(path, path),
                           "h2d_bytes": 4 * col.nrows if path == 2 else h2d, "d2h_bytes": d2h,
                           "pcie_GBps": round(((4 * col.nrows if path == 2 else h2d) + d2h) / wall / 1e9, 1),
                           "perm_first_last": [first, last]}
                    if path == 2:   # the A/B inside the same run: the same call with the strings uploaded (ctx option host_build = 0)
                        eng.ctx.set_option("host_build", 0)
                        wall0, path0, f0, l0, dig0, _ = timed()
                        eng.ctx.set_option("host_build", 1)
                        blk["strings_uploaded_ms"] = round(wall0 * 1e3, 2)
                        if dig is not None:
                            blk["verified"] = bool(dig == dig0 and (first, last) == (f0, l0))   # the same permutation either way
                            blk["perm_digest"] = f"{dig:016x}"
                    elif not col.fixed_width and col.nrows >= (1 << 22):
                        # variable-length keys the library chose to upload as strings (ctx option host_split = 1 goes by the cgroup's CPU
                        # quota): the host twin of the split codec forced, for the record — throttled hosts lose, see DESIGN §9
                        try:
                            eng.ctx.set_option("host_split", 2)
                            wall2, path2, f2, l2, dig2, _ = timed()
                            if path2 == 2:
                                blk["host_split_forced_ms"] = round(wall2 * 1e3, 2)
                                if dig is not None:
                                    blk["host_split_forced_same_perm"] = bool(dig2 == dig)
                                try:
                                    blk["cgroup_cpu_max"] = open("/sys/fs/cgroup/cpu.max").read().strip()
                                except OSError:
                                    pass
                        finally:
                            eng.ctx.set_option("host_split", 1)
                    pc.free()
                    return blk

                out["index_on_1e8"]["e2e_pinned_host"] = {
                    "scope": "key column in pinned host memory -> host perm (cph_index_build on host columns + cph_index_perm(HOST)); "
                             "PCIe inclusive.  Round 5: ONE key column of <= 8 byte positions is coded by host threads in 2^22-row chunks, "
                             "each uploaded (4 B/row) while the next is coded, the device only sorts (build_path).  Round 6: config 3's variable-length keys "
                             "the same way through the split codec's host twin (ctx option host_split); any other key uploads its strings",
                    "unique_fixed8_ids": time_index_host(dg.column(dg.SEQ_PERM, n8, n8, encoding=dg.FIXED8, seed=7), True, 2),
                    "varlen_dup_keys_config3": time_index_host(dg.varkeys(n8), False, 2)}

        # ---- CPU baseline: the oracle (C restatement of the reference), bounded sample ----------------
        if world == 1 and not args.no_cpu_baseline:
            from oracle import orc

            ns = min(args.cpu_sample_rows, nloc)
            s_cust = ords["cust_id"].slice(0, ns)
            s_prod = ords["prod_id"].slice(0, ns)
            t0 = time.perf_counter()
            oa = orc.OracleIndex([cust_id])
            ob = orc.OracleIndex([prod_id])
            assert oa.first_dup() is None and ob.first_dup() is None
            t_build = time.perf_counter() - t0
            t0 = time.perf_counter()
            j1 = oa.join([s_cust])
            sel = j1["probe_idx"].astype(np.uint32)
            j2 = ob.join([s_prod], row_sel=sel)
            t_probe = time.perf_counter() - t0
            est = t_build + t_probe * (args.rows / ns)
            if gpu_prefix is not None:
                # bit-exact comparison of the first `ns` result rows of the timed configuration with the oracle
                eq = bool(j1["nmatches"] == ns and j2["nmatches"] == ns
                          and np.array_equal(j1["build_row"].astype(np.uint32), gpu_prefix[0][:ns])
                          and np.array_equal(j2["build_row"].astype(np.uint32), gpu_prefix[1][:ns]))
                out.setdefault("verify", {})["oracle_prefix_rows"] = ns
                out["verify"]["oracle_prefix_bit_exact"] = eq
                out["verified"] = bool(out.get("verified")) and eq
            lean_1 = {
                "value": args.rows / est, "unit": "rows/s", "cores": 1,
                "sample": f"oracle (C restatement, SoA strings, comparison sort + binary-search probe; 1 thread): both "
                          f"index builds in full ({t_build:.2f} s) + chained probe of the first {ns} of {args.rows} "
                          f"orders rows ({t_probe:.2f} s, {j2['nmatches']} joined), probe time scaled to all rows",
                "build_s": round(t_build, 3), "probe_sample_s": round(t_probe, 3),
            }
            # (i) the reference's own cost model — one hash map per row, comparison sort through map lookups, a merged
            #     map per match (oracle/faithful.cpp), single thread like the reference — on a bounded sample: the
            #     first fn customers / fm orders of tables of the same shape, extrapolated with n*log(n) (index) and
            #     m*log(n) (probe).  The extrapolation is labelled; the measured sample stands beside it.
            import math
            fn, fm = min(4_000_000, args.customers), min(2_000_000, args.rows)   # ~17 s on one core (round 4: 1e6 / 5e5, 4 s)
            f_cust = dg.customers(fn)
            f_prod = dg.products(args.products)
            f_ords = dg.orders(fm, fn, args.products)
            fr = orc.faithful_chain_join(f_cust, "id", f_prod, "prod_id", f_ords, "cust_id", "prod_id")
            assert fr["joined"] == fm
            scale_ix = (args.customers * math.log2(max(2, args.customers))) / (fn * math.log2(max(2, fn)))
            scale_pr = (args.rows / fm) * (math.log2(max(2, args.customers)) / math.log2(max(2, fn)))
            f_est = fr["index_s"] * scale_ix + fr["join_s"] * scale_pr
            # (ii) the lean algorithm on all host cores
            mt_n = min(20_000_000, nloc)
            r1, jn1, s1, p1, threads = orc.lean_mt_join(cust_id, ords["cust_id"].slice(0, mt_n))
            r2, jn2, s2, p2, _ = orc.lean_mt_join(prod_id, ords["prod_id"].slice(0, mt_n))
            assert jn1 == mt_n and jn2 == mt_n
            if gpu_prefix is not None:
                k = min(mt_n, ns)
                assert np.array_equal(r1[:k], gpu_prefix[0][:k]) and np.array_equal(r2[:k], gpu_prefix[1][:k])
            mt_est = s1 + s2 + (p1 + p2) * (args.rows / mt_n)
            f_meas = fr["index_s"] + fr["join_s"]
            out["cpu_baseline"] = {
                # MEASURED: the whole job at sample size (index builds + chained Join), joined rows per second of that run;
                # the full-size figure is an extrapolation and stands apart
                "value": fm / f_meas, "unit": "rows/s", "cores": 1, "kind": "port",
                "sample": f"map-per-row restatement of csvplus.go (Row = hash map, sort.Sort through Less, mergeRows per match; "
                          f"oracle/faithful.cpp, 1 thread like the reference), MEASURED on a sample of the workload: UniqueIndexOn over "
                          f"{fn} customers + {args.products} products ({fr['index_s']:.2f} s) + chained Join of {fm} orders "
                          f"({fr['join_s']:.2f} s) = {fm} joined rows in {f_meas:.2f} s; not the Go binary (no Go toolchain here)",
                "measured_sample": {"customers": fn, "orders": fm, "row_maps_s": round(fr["rows_s"], 3),
                                    "index_s": round(fr["index_s"], 3), "join_s": round(fr["join_s"], 3)},
                "extrapolated_full_size": {
                    "value": args.rows / f_est, "unit": "rows/s", "seconds": round(f_est, 1),
                    "how": f"index time x{scale_ix:.1f} (n*log2 n), join time x{scale_pr:.1f} (m*log2 n) to {args.customers} customers / "
                           f"{args.rows} orders: an estimate, not a measurement"},
                "variants": {
                    "lean_soa_1_thread": lean_1,
                    "lean_soa_all_cores": {
                        "value": args.rows / mt_est, "unit": "rows/s", "cores": threads,
                        "sample": f"sort of (key,row) pairs + binary-search probe on {threads} threads (OpenMP, libstdc++ "
                                  f"parallel stable_sort): both index builds in full ({s1 + s2:.2f} s) + both probes of the first "
                                  f"{mt_n} orders rows ({p1 + p2:.2f} s), probe time scaled to all rows; nproc={os.cpu_count()}, cgroup cpu.max={_cgroup_cpu_max()} (the threads share that quota)"},
                },
            }
    except Exception as ex:   # noqa: BLE001
        import traceback
        out["extras_error"] = f"{type(ex).__name__}: {ex}"
        out["extras_traceback"] = traceback.format_exc()[-1500:]
        if "cpu_baseline" not in out:   # (the checks run in front of it: without them nothing is claimed)
            out["verified"] = bool(out.get("verified")) and "verify" in out and "index_on_1e8" in out
        print(f"bench.py: an extra block failed ({out['extras_error']}); the line is printed without it", file=sys.stderr, flush=True)
    # the communicators go first, C's stdout buffer is flushed, THEN the line: nothing a library prints can land behind it.  The
    # other ranks left long ago; should a teardown wait for them all the same, the line is printed after 15 s without it
    def close_communicators():
        try:
            if cdist is not None:
                cdist.close()
            if world > 1:
                dist.destroy_process_group()
        except Exception as ex:   # noqa: BLE001
            print(f"bench.py: closing the communicators: {type(ex).__name__}: {ex}", file=sys.stderr, flush=True)

    if cdist is not None or world > 1:
        import threading

        closer = threading.Thread(target=close_communicators, daemon=True)
        closer.start()
        closer.join(15.0)
        if closer.is_alive():
            print("bench.py: the communicators did not close within 15 s; printing the line first", file=sys.stderr, flush=True)
    emit(out, args)
    if (cdist is not None or world > 1) and closer.is_alive():
        sys.stdout.flush()
        os._exit(0)   # (a hung teardown must not keep the process, and with it the driver, waiting)


if __name__ == "__main__":
    main()

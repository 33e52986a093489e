"""bench.py as a plain command: `python bench.py --gpus N` starts its own N ranks (torch.distributed.run on
127.0.0.1) when no launcher did, refuses with a clear message when the machine has fewer GPUs than ranks, and
never falls back to another exchange transport.  The CPU tests drive the launch path with --launch-check
(rendezvous only, gloo: every rank reports in, no compute); the GPU tests run the real thing on the 1-GPU box."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
BENCH = str(ROOT / "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(kw)
    return env


def _json_line(out: str) -> dict:
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def _full(d: dict) -> dict:
    """The full record behind the compact stdout line (bench.py: emit): the line names the file."""
    p = Path(d["extras"])
    p = p if p.is_absolute() else ROOT / p
    return json.loads(p.read_text())


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("n", [2, 3])
def test_plain_command_starts_its_own_ranks(n):
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--launch-check"], env=_env(), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d == {"launch_check": True, "n_gpus": n, "world": n, "ranks_seen": list(range(n)), "backend": "gloo"}


def test_output_mode_flags_are_accepted():
    """--row-ids (the rounds 1-2 output mode) and the old --positions spelling parse; the self-launched ranks get them too."""
    for flag in ("--row-ids", "--positions"):
        r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check", flag], env=_env(), capture_output=True, text=True,
                           timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        assert _json_line(r.stdout)["ranks_seen"] == [0, 1]


def test_under_a_launcher_it_is_one_of_the_ranks():
    """The driver's way: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N."""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), BENCH, "--gpus", "2", "--launch-check"],
                       env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert _json_line(r.stdout)["ranks_seen"] == [0, 1]


def test_world_size_mismatch_is_an_error_message_not_an_assert():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check"], env=_env(WORLD_SIZE="3", RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE=3" in r.stderr and "Traceback" not in r.stderr, r.stderr[-2000:]


def test_more_ranks_than_gpus_is_refused_clearly():
    """No GPU here (CPU suite) / one GPU on the GPU box: `--gpus 2` must say so and exit 2, not assert or hang."""
    import torch

    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have >= 2:
        pytest.skip("this machine really has 2 GPUs")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-2000:])
    assert f"needs 2 visible GPUs, this machine shows {have}" in r.stderr and "Traceback" not in r.stderr


def test_compact_line_of_the_kept_full_records_fits_the_budget():
    """bench.compact_line on the full records kept under profiles/ (round 5's 22 KB line among them): under the budget, strict JSON,
    `roofline` and `cpu_baseline` with the contract's keys."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", BENCH)
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    seen = 0
    for f in sorted((ROOT / "profiles").glob("r0[56]*bench*.json")):
        full = json.loads(f.read_text())
        if "metric" not in full or "extras" in full:     # (a compact line kept beside its full record)
            continue
        line = json.dumps(b._finite(b.compact_line(full, "gpurun_out/bench_extras.json")), allow_nan=False)
        assert len(line) < b.LINE_BUDGET, (f.name, len(line))
        d = json.loads(line)
        assert {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config"} <= set(d), f.name
        if "roofline" in full and full["roofline"]:
            assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(d["roofline"]), f.name
        if "cpu_baseline" in full:
            assert {"value", "unit", "cores", "kind", "sample"} <= set(d["cpu_baseline"]) and len(d["cpu_baseline"]["sample"]) <= 200
        for blk, key in (("to_csv", "joined_rows_to_text"), ("csv_parse", "orders_text_to_columns")):   # round 6: the rows either side of the path
            if blk in full and "error" not in full[blk]:
                assert d[blk][key][0] == full[blk]["ms"] and d[blk][key][-1] is full[blk].get("verified"), (f.name, blk)
        seen += 1
    assert seen >= 3
    # a record bloated far past the budget still yields a line inside it (optional blocks are dropped, the contract fields stay)
    full = json.loads((ROOT / "profiles" / "r05e_bench.json").read_text())
    full["variants"] = {f"variant_{i}": dict(full["variants"]["itoa_ids"]) for i in range(200)}
    line = json.dumps(b.compact_line(full, "x"))
    assert len(line) < b.LINE_BUDGET and json.loads(line)["roofline"]["frac"] > 0 and "cpu_baseline" in json.loads(line)


SMALL = ["--rows", "300000", "--customers", "20000", "--products", "700", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
         "--no-index-1e8", "--no-e2e", "--no-traffic"]


@pytest.mark.gpu
def test_one_gpu_line_has_the_contract_fields():
    cmd = [sys.executable, BENCH, "--gpus", "1", *SMALL, "--verify-sample", "5000"]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        # Seen once in round 3 on a box that had just been handed out, never reproduced (rounds 3-4: every gpurun call of the
        # suite since, profiles/r04_bench_first_attempt.txt).  The evidence must not vanish again: the failing attempt's
        # stderr is written where gpurun brings it back (gpurun_out/), then the bench is tried once more.
        out = Path(BENCH).resolve().parent / "gpurun_out"
        out.mkdir(exist_ok=True)
        (out / "bench_first_attempt_failure.txt").write_text(f"returncode {r.returncode}\n--- stderr ---\n{r.stderr}\n--- stdout ---\n{r.stdout}\n")
        print("first attempt failed:", r.returncode, r.stderr[-3000:], file=sys.stderr)
        r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["joined_rows_per_step"] == 300000 and d["verified"] is True
    assert d["config"]["rccl_nranks"] is None and len(d["per_rank_ms_per_step"]) == 1
    f = _full(d)   # the full record: per-kernel table, verification details, variants with their byte models
    assert f["value"] == d["value"] and f["kernels"] and f["verify"]["joined_rows"] == 300000 and set(f["variants"]) == set(d["roofline"]["variants"])


def _no_constants(x):
    raise AssertionError(f"{x} in the bench line")


@pytest.mark.gpu
def test_the_stdout_line_is_small_and_last(tmp_path):
    """Round 5's line had grown to 22 KB and the driver's record of it came back with parsed = null.  The line is the contract's
    fields + compact `roofline` / `cpu_baseline` + bare numbers, under 6000 bytes, strict JSON, the LAST line of stdout — with
    every optional block switched on (variants, IndexOn at full test size, the end-to-end scopes, the CPU baseline)."""
    extras = tmp_path / "full.json"
    cmd = [sys.executable, BENCH, "--gpus", "1", "--rows", "300000", "--customers", "20000", "--products", "700", "--steps", "2", "--warmup", "1",
           "--no-traffic", "--cpu-sample-rows", "50000", "--verify-sample", "5000", "--extras", str(extras)]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    line = lines[-1]
    assert line.startswith("{") and len(line) < 6000, len(line)
    d = json.loads(line, parse_constant=_no_constants)
    assert d["roofline"]["frac"] > 0 and d["roofline"]["kernel"] == "k_chain_dense" and d["roofline"]["avg_launch_ms"] > 0
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["peak"] == 8000.0 and d["roofline"]["unit"] == "GB/s"
    cb = d["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] == 1 and cb["kind"] == "port" and 0 < len(cb["sample"]) <= 200
    assert set(d["roofline"]["variants"]) >= {"itoa_ids", "half_occupied_ids", "sparse_random_keys", "build_side_key", "step_plus_permute", "dup_build_side"}
    assert all(v[3] is True for v in d["roofline"]["variants"].values()), d["roofline"]["variants"]
    assert d["verified"] is True and d["extras"] == str(extras)
    assert d["to_csv"]["joined_rows_to_text"][0] > 0 and d["to_csv"]["joined_rows_to_text"][3] is True, d["to_csv"]   # ToCsv of the joined rows: ms, verified
    assert d["csv_parse"]["orders_text_to_columns"][0] > 0 and d["csv_parse"]["orders_text_to_columns"][2] is True, d["csv_parse"]   # parse(write(columns)) == columns
    f = json.loads(extras.read_text())
    assert f["cpu_baseline"]["measured_sample"] and f["index_on_1e8"]["varlen_dup_keys_config3"]["verified"] is True
    assert f["to_csv"]["one_pass_taken"] and f["to_csv"]["verify"]["one_pass_equals_two_pass_bytes"] and f["to_csv"]["verify"]["oracle_prefix_bytes_equal"]


@pytest.mark.gpu
def test_shared_gpu_debug_mode_still_runs_and_says_what_it_is():
    """Two ranks on the one GPU (gloo): the control flow of the sharded step, labelled as NOT an RCCL measurement."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", *SMALL], env=_env(CPH_BENCH_SHARE_GPU="1"), capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["joined_rows_per_step"] == 300000
    assert "DEBUG" in d["config"]["transport"] and d["config"]["rccl_nranks"] is None
    assert len(d["per_rank_ms_per_step"]) == 2


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", ["allgatherv", "host"])
def test_multi_gpu_code_path_with_one_rank(exchange):
    """CPH_BENCH_FORCE_DIST=1: what a one-GPU box can execute of the N > 1 step — the RCCL communicator behind the C ABI
    (nranks=1), cph_dist_join_chain with 3 sub-chunks (xGMI mode / shared host buffer), the multi_gpu report."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", *SMALL, "--exchange", exchange, "--chunks", "3", "--no-verify"],
                       env=_env(CPH_BENCH_FORCE_DIST="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d["joined_rows_per_step"] == 300000 and d["config"]["rccl_nranks"] == 1 and d["config"]["exchange"] == exchange
    assert d["multi_gpu"]["mode"] == exchange and d["multi_gpu"]["chunks"] == 3 and len(r.stdout.splitlines()[-1]) < 6000
    m = _full(d)["multi_gpu"]
    assert m["mode"] == exchange and m["chunks"] == 3 and m["join_compute_ms"] > 0 and m["exchange_ms"] > 0
    assert m["n1_ms_per_step"] > 0 and 0 < d["efficiency_vs_n1"] < 3 and d["exchange_ms"] == m["exchange_ms"] and d["compute_ms"] > 0
    assert m["build_side"]["choice"] == "replicated"
    assert m["bytes_sent_per_step"] == (300000 * 8 if exchange == "host" else 0)
    # the other exchange modes and the exchange's own rate, measured in the same run
    others = m["same_run_other_modes"]
    assert set(others) == {"allgatherv", "packed", "host", "none"} - {exchange}
    assert all(v.get("ms_per_step", 0) > 0 for v in others.values()), others
    assert m["measured_exchange_rate"]["ms"] > 0 and m["measured_exchange_rate"]["bytes_received_per_rank"] == 0   # (one rank)
    assert len(m["n1_ms_single_steps"]) == 5


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [1, 2])
def test_stream_mode_config5_shape(ranks):
    """--stream: every rank streams its shard of the orders from pinned host memory through cph_stream_join_* (BASELINE config 5's
    shape at test size); two ranks share the one GPU here (gloo): the control flow of the N-rank line."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(ranks), "--stream", "--rows", "600000", "--customers", "20000", "--products", "700",
                        "--steps", "2", "--warmup", "1"], env=_env(CPH_BENCH_SHARE_GPU="1") if ranks > 1 else _env(), capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == ranks and d["joined_rows_per_step"] == 600000 and d["verified"] is True
    assert d["scope"].startswith("pcie_inclusive") and len(d["per_rank_ms_per_step"]) == ranks and d["value"] > 0


@pytest.mark.gpu
def test_a_failing_extra_block_does_not_cost_the_line():
    """Everything bench.py reports beside the contract fields (verification, variants, end-to-end scopes, CPU baseline) runs
    inside one try: when one of them raises, the line is still printed — with `extras_error`, and without a verified claim."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", *SMALL], env=_env(CPH_BENCH_FAIL_EXTRAS="1"), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d["extras_error"].startswith("RuntimeError: CPH_BENCH_FAIL_EXTRAS") and d["verified"] is False
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["roofline"]["frac"] > 0 and "cpu_baseline" not in d


@pytest.mark.gpu
def test_one_rank_run_of_the_multi_gpu_path_at_bench_size():
    """The N > 1 step (communicator, cph_dist_join_chain, match totals over the exchange stream) with ONE rank at the bench's own size must
    cost little more than the plain step: round 5 measured 1.09 ms against 0.76 ms (8 sub-chunks of a shard that has nobody to send to);
    the library now sizes its sub-chunks by the shard (dist.hip) — efficiency_vs_n1 = n1 step / this step >= 0.88."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-index-1e8", "--no-e2e",
                        "--no-traffic", "--no-verify", "--no-variants", "--no-positions", "--no-calibration"],
                       env=_env(CPH_BENCH_FORCE_DIST="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d["joined_rows_per_step"] == 100_000_000 and d["config"]["rccl_nranks"] == 1 and d["multi_gpu"]["chunks"] == 1
    assert d["efficiency_vs_n1"] >= 0.88, (d["efficiency_vs_n1"], d["ms_per_step"], d["multi_gpu"])

#!/usr/bin/env python3
"""Join against an index whose key codes are too sparse for a direct-address table: 1e7 build rows x 1e8 probe rows of
random 12-character alphanumerics, through cph_join_probe and through the fused chain kernel, with
the hash table (default) and with it switched off (ctx option join_hash = 0: the sorted search it replaces).

  [a-z0-9]     36^12 < 2^63  -> one code word  (hash entries carry the code: kHashK1)
  [A-Za-z0-9]  62^12 ~ 2^71  -> two code words (both in the entry: kHashK3)
  16 random bytes            -> three code words (all in the entry: kHashK3)
  28 random bytes            -> four+ code words (64-bit tag + verification against the sorted codes: kHashTag)

Results are spot-checked against numpy: the build row a probe row reports carries the probe row's bytes."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np
import torch

from csvplus_amd import Context, DeviceIndex, StrCol, _native as N, join_chain
from csvplus_amd.engine import Engine, device_view

_a = [a for a in sys.argv[1:] if not a.startswith("--")]
NB = int(_a[0]) if len(_a) > 0 else 10_000_000
MP = int(_a[1]) if len(_a) > 1 else 100_000_000
eng = Engine(0)
dev = eng.device
A36 = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz0123456789", dtype=np.uint8)
A62 = np.frombuffer(b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789", dtype=np.uint8)
A256 = np.arange(256, dtype=np.uint8)


def fixed_col(mat):
    n, w = mat.shape
    offs = (np.arange(n + 1, dtype=np.uint64) * w).astype(np.uint32 if n * w < 2**32 else np.uint64)
    return StrCol(np.ascontiguousarray(mat).reshape(-1), offs, n, offs.dtype.itemsize * 8, fixed_width=w)


def timed(label, ctx, fn, reps=3):
    fn()
    ctx.profile(True); ctx.profile_read(reset=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    p = ctx.profile_read(reset=True); ctx.profile(False)
    ks = ", ".join(f"{k}={v['total_ms'] / reps:.3f}" for k, v in sorted(p.items(), key=lambda kv: -kv[1]['total_ms'])[:6])
    print(f"  {label:46s} wall {dt * 1e3:8.3f} ms | {ks}", flush=True)
    return dt


def run(name, alphabet, width):
    rng = np.random.default_rng(width * 1000 + len(alphabet))
    build = alphabet[rng.integers(0, len(alphabet), (NB, width), dtype=np.uint8 if len(alphabet) <= 256 else np.uint16)]
    sel = rng.integers(0, NB, MP)
    probe = build[sel]
    miss = rng.random(MP) < 0.05                   # 5 % of the probe rows get a fresh random key (almost surely absent)
    probe[miss] = alphabet[rng.integers(0, len(alphabet), (int(miss.sum()), width))]
    d_build, d_probe = fixed_col(build).to_device(dev), fixed_col(probe).to_device(dev)
    print(f"{name}: build {NB} x probe {MP}, {width}-byte keys", flush=True)
    for label, hash_on in (("hash table", 1), ("sorted search (join_hash=0)", 0)):
        ctx = Context(0)
        ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        ctx.set_option("join_hash", hash_on)
        t0 = time.perf_counter()
        ix = DeviceIndex(ctx, [d_build])
        torch.cuda.synchronize(); t_ix = time.perf_counter() - t0
        t0 = time.perf_counter()
        ix.prepare_join()
        ctx.synchronize(); t_prep = time.perf_counter() - t0
        inf = ix.info()
        print(f" {label}: IndexOn {t_ix * 1e3:.2f} ms (first call), prepare_join {t_prep * 1e3:.3f} ms, words={inf['code_words']} "
              f"bits={inf['code_bits']} unique={ix.first_dup is None} hash_mode={inf['hash_mode']} hash_MB={inf['hash_bytes'] / 1e6:.0f}", flush=True)
        reps = 3 if hash_on else 2
        timed("cph_join_probe bounds only", ctx, lambda: ix.probe([d_probe], want_pairs=False, out_mem=N.CPH_MEM_DEVICE).release(), reps)
        if hash_on and inf["code_words"] > 1 and "--rows-ab" in sys.argv:   # generic kernel: rows per phase, A/B on this box
            ctx.set_option("probe_hash_rows", 2)
            timed("cph_join_probe bounds only (2 rows per phase)", ctx, lambda: ix.probe([d_probe], want_pairs=False, out_mem=N.CPH_MEM_DEVICE).release(), reps)
            ctx.set_option("probe_hash_rows", 4)
        timed("cph_join_probe + pairs", ctx, lambda: ix.probe([d_probe], out_mem=N.CPH_MEM_DEVICE).release(), reps)
        if ix.first_dup is None and inf["code_words"] == 1:
            timed("cph_join_chain (1 step, fused kernel)", ctx, lambda: join_chain(ctx, [(ix, [d_probe])], out_mem=N.CPH_MEM_DEVICE).release(), reps)
        # spot check
        m = ix.probe([d_probe], out_mem=N.CPH_MEM_DEVICE)
        p = m.device_ptrs()
        pidx = device_view(p["probe_idx"], m.nmatches, "<i8", m, dev)
        brow = device_view(p["build_row"], m.nmatches, "<i4", m, dev)
        k = torch.randint(0, m.nmatches, (200_000,), device=dev)
        pi, br = pidx[k].cpu().numpy(), brow[k].cpu().numpy().view(np.uint32)
        ok = bool((probe[pi] == build[br]).all())
        exp_hits = int(MP - miss.sum())
        print(f"  matches {m.nmatches} (>= {exp_hits} expected), sampled key equality: {ok}", flush=True)
        assert ok and m.nmatches >= exp_hits
        m.release(); ix.close(); ctx.close()


run("[a-z0-9] x 12 (one word)", A36, 12)
run("[A-Za-z0-9] x 12 (two words)", A62, 12)
if NB <= 10_000_000:
    run("16 random bytes (three words)", A256, 16)
    run("28 random bytes (tags + verification)", A256, 28)

"""RCCL smoke on the 1-GPU box: a one-rank "nccl" process group goes through the same collectives the sharded
join uses (count all_gather, all_gather_into_tensor of int64 / int32 row ids, MAX all_reduce of the timing, barrier)
— catches dtype / initialisation problems the gloo CPU tests cannot.  The multi-rank exchange itself is covered
by tests/test_dist_cpu.py (gloo, 2 and 3 ranks)."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent

CHILD = r"""
import os, sys
sys.path.insert(0, os.environ["CPH_ROOT"])
import torch, torch.distributed as dist
from csvplus_amd.dist import allgatherv
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
for dtype in (torch.int64, torch.int32):
    t = torch.arange(1000, dtype=dtype, device=dev) * 3
    out, counts = allgatherv(t, single_rank_shortcut=False)
    assert counts == [1000] and torch.equal(out, t), dtype
    out, counts = allgatherv(t[:0], single_rank_shortcut=False)
    assert counts == [0] and out.numel() == 0
x = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(x, op=dist.ReduceOp.MAX)
assert float(x.item()) == 1.5
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_SINGLE_OK")
"""


@pytest.mark.gpu
def test_rccl_one_rank_collectives():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               CPH_ROOT=str(ROOT), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_SINGLE_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])

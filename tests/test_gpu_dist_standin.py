"""The RCCL transport of csrc/dist.hip (RcclTransport: count all-gather, grouped ncclSend / ncclRecv with displacements
for unequal shards, ncclAllGather per array for equal ones, ncclBroadcast of an index) with 2 and 3 ranks on the
one-GPU box.  RCCL refuses two ranks on one device, so the ten NCCL entry points come from tests/c/nccl_standin.cpp
(CPH_RCCL_LIBRARY): ranks are threads, data moves by device copies, and the stand-in enforces NCCL's pairing contract —
an unmatched or mis-sized send / receive, or collectives issued in different orders, fail the call.  Above the ten
entry points everything is the product code a real N-GPU run executes; the scenarios are those of tests/test_gpu_dist.py.
Runs in a child process: a process binds to ONE RCCL library (the other tests use the real one)."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
STANDIN = ROOT / "tests" / "c" / "libnccl_standin.so"

CHILD = r"""
import os, sys, threading
sys.path.insert(0, os.environ["CPH_ROOT"])
from csvplus_amd import Context, _native as N
from tests import test_gpu_dist as T

def factory(world):
    box, lock, ready = {}, threading.Lock(), threading.Event()
    def make(ctx, r):
        if r == 0:
            box["id"] = N.Dist.unique_id(ctx)
            ready.set()
        ready.wait(60)
        return N.Dist.create(ctx, box["id"], r, world)   # collective: returns when every rank has joined
    return make

for world, factor in ((2, 1), (2, 2), (3, 2), (3, 1)):
    T.run_sharded_chain(world, factor, factory(world), expect_transport="rccl nranks=%d lib=%s" % (world, os.environ["CPH_RCCL_LIBRARY"]))
    print("chain", world, factor, "ok", flush=True)
T.run_index_broadcast(3, factory(3))
# the pipelined sharded join (cph_dist_join_chain): per-chunk ncclSend / ncclRecv batches with displacements on the exchange
# stream, 3..5 chunks in flight, even / uneven / empty shards, identity and not; and the host-gather variant, whose only
# collectives are the count words
tr = "rccl nranks=%d lib=%s"
for world, factor, nchunks, unequal, positions, given, host in ((2, 1, 4, False, True, True, False), (3, 1, 5, True, False, True, False),
                                                               (3, 2, 3, True, True, False, False), (3, 2, 4, True, False, True, True)):
    T.run_pipelined_chain(world, factor, factory(world), nchunks, host=host, positions=positions, unequal=unequal, shard_given=given,
                          expect_transport=tr % (world, os.environ["CPH_RCCL_LIBRARY"]))
    print("pipelined", world, factor, nchunks, unequal, host, "ok", flush=True)
for world, factor, nchunks, unequal, positions in ((2, 2, 3, False, True), (3, 2, 4, True, False), (3, 1, 2, True, True)):   # CPH_DIST_PACKED
    T.run_pipelined_chain(world, factor, factory(world), nchunks, positions=positions, unequal=unequal, packed=True,
                          expect_transport=tr % (world, os.environ["CPH_RCCL_LIBRARY"]))
    print("packed", world, factor, nchunks, unequal, "ok", flush=True)
print("STANDIN_RANKS_OK", flush=True)
"""


@pytest.mark.gpu
def test_rccl_transport_with_thread_ranks_through_the_nccl_standin():
    if not STANDIN.exists():
        subprocess.check_call(["make", "-C", str(ROOT), "tests/c/libnccl_standin.so"])
    env = dict(os.environ, CPH_ROOT=str(ROOT), CPH_RCCL_LIBRARY=str(STANDIN))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert r.returncode == 0 and "STANDIN_RANKS_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.gpu
def test_nccl_standin_rejects_a_broken_pairing():
    """The checker checks: a rank that sends to a peer that posts no receive gets an error (not a hang), through the
    raw entry points."""
    if not STANDIN.exists():
        subprocess.check_call(["make", "-C", str(ROOT), "tests/c/libnccl_standin.so"])
    child = r'''
import ctypes as C, threading, sys
import torch
lib = C.CDLL(sys.argv[1])
class Uid(C.Structure):
    _fields_ = [("b", C.c_char * 128)]
uid = Uid()
assert lib.ncclGetUniqueId(C.byref(uid)) == 0
lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
lib.ncclSend.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
lib.ncclRecv.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
res = [None, None]
bufs = [torch.zeros(64, dtype=torch.uint8, device="cuda") + r for r in range(2)]
torch.cuda.synchronize()
def body(r, good):
    comm = C.c_void_p()
    assert lib.ncclCommInitRank(C.byref(comm), 2, uid, r) == 0
    lib.ncclGroupStart()
    if r == 0:
        lib.ncclSend(bufs[0].data_ptr(), 64, 1, 1, comm, None)
    elif good:
        lib.ncclRecv(bufs[1].data_ptr(), 64, 1, 0, comm, None)
    else:
        lib.ncclSend(bufs[1].data_ptr(), 64, 1, 0, comm, None)   # both send, nobody receives
    res[r] = lib.ncclGroupEnd()
    lib.ncclCommDestroy(comm)
for good in (True, False):
    ths = [threading.Thread(target=body, args=(r, good)) for r in range(2)]
    [t.start() for t in ths]; [t.join(60) for t in ths]
    if good:
        assert res == [0, 0], res
        assert int(bufs[1][0].item()) == 0
    else:
        assert res[0] != 0 and res[1] != 0, res
    assert lib.ncclGetUniqueId(C.byref(uid)) == 0
print("PAIRING_CHECK_OK")
'''
    r = subprocess.run([sys.executable, "-c", child, str(STANDIN)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "PAIRING_CHECK_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])

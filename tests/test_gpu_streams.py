"""Stream ordering of the C ABI (include/csvplus_hip.h: "all work of this ctx is enqueued on `hip_stream`"): a batch of
builds that uses the ctx's second stream must still see key columns that kernels of the CALLER, queued on the ctx's stream,
are producing when the call starts — and must not take pool blocks those kernels still use."""
import numpy as np
import pytest

from csvplus_amd import Context, DeviceIndex, StrCol, _native as N, datagen as dg
from oracle import orc

pytestmark = pytest.mark.gpu


def test_build_many_reads_columns_produced_asynchronously_on_the_ctx_stream():
    import torch

    dev = torch.device("cuda", 0)
    ctx = Context(0)
    s = torch.cuda.Stream(device=dev)
    ctx.set_stream(s.cuda_stream)
    n = 400_000
    cols = [dg.column(dg.SEQ_PERM, n, n, encoding=dg.FIXED8, seed=31 + k) for k in range(2)]
    want = [orc.OracleIndex([c]).perm for c in cols]
    for rep in range(3):
        with torch.cuda.stream(s):
            # device columns that hold GARBAGE until the stream reaches the copies below; a long-running kernel in front of
            # them keeps the stream busy while the host races ahead into cph_index_build_many
            bufs = [torch.full((c.data.nbytes + 8,), 0x39, dtype=torch.uint8, device=dev) for c in cols]
            busy = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
            for _ in range(6):
                busy.add_(1)
            srcs = [torch.from_numpy(c.data).pin_memory() for c in cols]
            for b, src in zip(bufs, srcs):
                b[: src.numel()].copy_(src, non_blocking=True)
        dcols = [StrCol(b, None, n, 32, N.CPH_MEM_DEVICE, fixed_width=8) for b in bufs]
        res = DeviceIndex.build_many(ctx, [([dcols[0]], True), ([dcols[1]], True)])   # spec 1 runs on the side stream
        for ix, w in zip(res, want):
            assert ix.status == N.CPH_OK, ctx.last_error()
            np.testing.assert_array_equal(ix.perm(), w)
            ix.close()
        del busy
    ctx.close()

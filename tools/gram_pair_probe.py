"""Do two batched Gram launches on two streams cost more than the same two launches one after the other?
python tools/gram_pair_probe.py [batch] [reps]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgcompressionmps_amd import _lib  # noqa: E402
from imgcompressionmps_amd.core.batch import group_streams  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
m, n = 32768, 512
lib = _lib.load()
sets = []
for g in range(2):
    mats = [torch.rand((m, n), device="cuda") - 0.5 for _ in range(batch)]
    out = torch.empty((batch, n, n), dtype=torch.float64, device="cuda")
    nb = lib.ndmps_gram_batched_workspace_bytes(batch, m, n)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    ptrs = (C.c_void_p * batch)(*[a.data_ptr() for a in mats])
    sets.append((mats, out, ws, nb, ptrs))
streams = group_streams(2)
torch.cuda.synchronize()


def launch(g, stream):
    mats, out, ws, nb, ptrs = sets[g]
    _lib.check(lib.ndmps_gram_batched_f32(batch, ptrs, m, n, n, out.data_ptr(), n * n, ws.data_ptr(), nb, stream.cuda_stream))


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    import time
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


serial = timed(lambda: (launch(0, streams[0]), launch(1, streams[0])))
concurrent = timed(lambda: (launch(0, streams[0]), launch(1, streams[1])))
print(f"two Gram launches of {batch} x ({m} x {n}): one stream {serial:.2f} ms, two streams {concurrent:.2f} ms")

"""Batched raw Gram of a lockstep group: plain rows (the volumes viewed as 32768 x 512 matrices) against the same
launch reading the volumes through the permutation tables (what the fused sweep does).
python tools/gram_gather_probe.py [batch] [reps]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgcompressionmps_amd import _lib  # noqa: E402
from imgcompressionmps_amd.core.ndmps import _plan_for  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
shape = (256, 256, 256)
m, n = 32768, 512
lib = _lib.load()
dev = torch.device("cuda", 0)
vols = [torch.rand(shape, device=dev) for _ in range(batch)]
out = torch.empty((batch, n, n), dtype=torch.float64, device=dev)
nb = lib.ndmps_gram_batched_workspace_bytes(batch, m, n)
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
ptrs = (C.c_void_p * batch)(*[a.data_ptr() for a in vols])
row_off, col_off, col_perm = _plan_for(shape, 0).gather_tables(n, dev)
sp = _lib.stream_ptr()


def plain():
    _lib.check(lib.ndmps_gram_batched_f32(batch, ptrs, m, n, n, out.data_ptr(), n * n, ws.data_ptr(), nb, sp))


def gathered():
    _lib.check(lib.ndmps_gram_batched_indexed_f32(batch, ptrs, m, n, row_off.data_ptr(), col_off.data_ptr(), None, out.data_ptr(),
                                                  n * n, ws.data_ptr(), nb, sp))


for name, fn in (("plain", plain), ("gathered", gathered)):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:9s}: {e0.elapsed_time(e1) / reps:.3f} ms per launch of {batch} volumes")

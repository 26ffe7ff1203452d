"""Batched raw Gram of a lockstep group: plain rows (the volumes viewed as 32768 x 512 matrices) against the same
launch reading the volumes through the permutation tables (what the fused sweep does).
python tools/gram_gather_probe.py [batch] [reps] [n]   (n = 512: the raw Gram of a bond cap of 64; 64: of a cap of 32)
The third line reads the rows in ascending order of their offsets (a Gram is a sum over rows: any order will do)."""
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
n = int(sys.argv[3]) if len(sys.argv) > 3 else 512
m = 256 ** 3 // n
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


row_sorted = torch.sort(row_off[:m])[0].contiguous()


def gathered(rows=row_off):
    _lib.check(lib.ndmps_gram_batched_indexed_f32(batch, ptrs, m, n, rows.data_ptr(), col_off.data_ptr(), None, out.data_ptr(),
                                                  n * n, ws.data_ptr(), nb, sp))


def gathered_sorted():
    gathered(row_sorted)


only = sys.argv[4] if len(sys.argv) > 4 else None  # "sorted": that variant alone (PMC passes)
for name, fn in (("plain", plain), ("gathered", gathered), ("sorted rows", gathered_sorted)):
    if only and not name.startswith(only):
        continue
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:11s}: {e0.elapsed_time(e1) / reps:.3f} ms per launch of {batch} volumes")

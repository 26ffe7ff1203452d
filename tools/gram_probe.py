"""Time of the fp64-MFMA Gram kernel on one unfolding, alone and batched:
python tools/gram_probe.py [m] [n] [reps] [batch]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgcompressionmps_amd import _lib  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 1
lib = _lib.load()
mats = [torch.rand((m, n), device="cuda") - 0.5 for _ in range(batch)]
g = torch.empty((batch, n, n), dtype=torch.float64, device="cuda")
sp = _lib.stream_ptr()
if batch == 1:
    nb = lib.ndmps_gram_workspace_bytes(m, n)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")

    def run():
        _lib.check(lib.ndmps_gram_f32(mats[0].data_ptr(), m, n, n, g.data_ptr(), ws.data_ptr(), nb, sp))
else:
    nb = lib.ndmps_gram_batched_workspace_bytes(batch, m, n)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    ptrs = (C.c_void_p * batch)(*[a.data_ptr() for a in mats])

    def run():
        _lib.check(lib.ndmps_gram_batched_f32(batch, ptrs, m, n, n, g.data_ptr(), n * n, ws.data_ptr(), nb, sp))

run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / reps
flops = batch * m * n * (n + 1)  # upper triangle incl. the diagonal, 2 flops per product
err = 0.0
for b in range(min(batch, 3)):
    a64 = mats[b].double()
    ref = a64.T @ a64
    err = max(err, float((g[b] - ref).abs().max() / ref.abs().max()))
print(f"gram {batch} x ({m} x {n}): {us:.1f} us per call ({us / batch:.1f} per matrix), {flops / us / 1e6:.1f} useful TFLOP/s fp64 "
      f"(of 78.6), rel err {err:.1e}")

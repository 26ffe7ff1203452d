"""Time of the fp64-MFMA Gram kernel on one unfolding: python tools/gram_probe.py [m] [n] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgcompressionmps_amd import _lib  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
lib = _lib.load()
a = torch.rand((m, n), device="cuda")
g = torch.empty((n, n), dtype=torch.float64, device="cuda")
nb = lib.ndmps_gram_workspace_bytes(m, n)
ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
sp = _lib.stream_ptr()


def run():
    _lib.check(lib.ndmps_gram_f32(a.data_ptr(), m, n, n, g.data_ptr(), ws.data_ptr(), nb, sp))


run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / reps
tiles = (n + 127) // 128
flops = 2.0 * m * (tiles * (tiles + 1) // 2) * 128 * 128
err = float((g - a.double().T @ a.double()).abs().max() / g.abs().max())
print(f"gram {m} x {n}: {us:.1f} us per call, {flops / us / 1e6:.1f} TFLOP/s fp64 (of 78.6), rel err {err:.1e}")

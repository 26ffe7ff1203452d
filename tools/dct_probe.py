"""Time of the last-axis DCT / inverse of one volume: the O(N log N) kernel against the basis product.
usage: python tools/dct_probe.py [edge]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgcompressionmps_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
lib = _lib.load()
x = torch.rand((n, n, n), dtype=torch.float32, device="cuda:0")
y = torch.empty_like(x)
basis = torch.empty((n, n), dtype=torch.float32, device="cuda:0")
_lib.check(lib.ndmps_dct_basis_f32(basis.data_ptr(), n, _lib.stream_ptr()))
rows = n * n
for name, env in (("fft", None), ("gemm", "1")):
    if env:
        os.environ["NDMPS_DCT_GEMM"] = env
    for fn_name, fn in (("dct", lib.ndmps_dct_last_f32), ("idct", lib.ndmps_idct_last_f32)):
        _lib.check(fn(x.data_ptr(), y.data_ptr(), rows, n, basis.data_ptr(), _lib.stream_ptr()))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            _lib.check(fn(x.data_ptr(), y.data_ptr(), rows, n, basis.data_ptr(), _lib.stream_ptr()))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"{n}^3 {fn_name} {name}: {ms * 1e3:.1f} us  {2 * 4 * n ** 3 / ms / 1e6:.0f} GB/s read+write")

"""Phases of back_rows_step_kernel (scratch; library built with -DNDMPS_BR_STAMPS): 10 ns ticks summed over the launches
that both apply and form, last row chunk, first column tile (slots 9.. of the stamp array)."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from imgcompressionmps_amd import _lib  # noqa: E402
_lib.LIB_PATH = os.path.join(ROOT, "tools", "scratch", "stamps", "libndmps_stamps.so")
lib = _lib.load()
dev = "cuda:0"
n, k = 2048, 128
a = torch.randn((n + 64, n), dtype=torch.float64, device=dev)
g = (a.T @ a).contiguous()
nb = lib.ndmps_syevd_topk_workspace_bytes(n, 1, k)
ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
v = torch.zeros_like(g)
w = torch.zeros(n, dtype=torch.float64, device=dev)
off = lib.ndmps_syevd_topk_stamps_offset(n, 1, k)
_lib.check(lib.ndmps_syevd_topk_values_f64(1, g.data_ptr(), n * n, _lib.i64_array([n]), v.data_ptr(), n * n, w.data_ptr(), n, k, ws.data_ptr(), nb, _lib.stream_ptr()))
torch.cuda.synchronize()
ws[off:off + 128].zero_()
_lib.check(lib.ndmps_syevd_topk_vectors_f64(1, _lib.i64_array([n]), _lib.i64_array([k]), k, ws.data_ptr(), nb, None, _lib.stream_ptr()))
torch.cuda.synchronize()
st = ws[off:off + 128].view(torch.int64).cpu().numpy()[9:15]
print("us per launch (31 launches): loads %.2f, partial sums %.2f, W2 + update %.2f, form %.2f" % tuple(st[:4] / 100.0 / 31))

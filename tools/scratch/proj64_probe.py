"""First projection of a bond cap of 32 (A = 262144 x 64 gathered from the volume, times 64 x 32): the tile kernel with
rows in site order / in memory order (output rows scattered to their places), and the stream (proj64_stream_kernel)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from imgcompressionmps_amd import _lib  # noqa: E402
from imgcompressionmps_amd.core.ndmps import _plan_for  # noqa: E402

batch = 32
lib = _lib.load()
dev = torch.device("cuda:0")
plan = _plan_for((256, 256, 256), 0)
numel = plan.numel
sp = _lib.stream_ptr
vols = [torch.randn(numel, device=dev) for _ in range(batch)]
row_off, col_off, _ = plan.gather_tables(64, dev)
m0, k = numel // 64, 32
W = [torch.randn(64, k, device=dev) for _ in range(batch)]
out = [torch.empty(m0 * k, device=dev) for _ in range(batch)]
out2 = [torch.empty(m0 * k, device=dev) for _ in range(batch)]
ptrs = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
pa, pb, pc, pc2 = ptrs(vols), ptrs(W), ptrs(out), ptrs(out2)
order = torch.argsort(row_off[:m0])
row_sorted = row_off[:m0][order].contiguous()
c_rows = (order * k).contiguous()
c_cols = torch.arange(k, device=dev, dtype=torch.int64)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


a = timed(lambda: _lib.check(lib.ndmps_sgemm_indexed_batched(batch, m0, k, 64, pa, 0, row_off.data_ptr(), col_off.data_ptr(), 1,
                                                             pb, k, pc, k, None, None, sp())))
b = timed(lambda: _lib.check(lib.ndmps_sgemm_indexed_batched(batch, m0, k, 64, pa, 0, row_sorted.data_ptr(), col_off.data_ptr(), 1,
                                                             pb, k, pc2, k, c_rows.data_ptr(), c_cols.data_ptr(), sp())))
out3 = [torch.empty(m0 * k, device=dev) for _ in range(batch)]
pc3 = ptrs(out3)
order32 = order.to(torch.int32).contiguous()
c = timed(lambda: _lib.check(lib.ndmps_sgemm_gathered64_stream_batched(batch, m0, k, pa, row_sorted.data_ptr(), order32.data_ptr(),
                                                                       col_off.data_ptr(), pb, k, pc3, k, sp())))
print(f"stream {c:.3f} ms ({(4.0 * numel * batch * 1.5) / c / 1e9:.2f} TB/s read + write), max diff vs tiles {max(float((x - y).abs().max()) for x, y in zip(out[:2], out3[:2])):.1e}")
print(f"site order {a:.3f} ms, memory order {b:.3f} ms; max diff {max(float((x - y).abs().max()) for x, y in zip(out[:2], out2[:2])):.1e}")

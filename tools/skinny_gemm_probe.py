"""Probe: the two skinny products that touch every voxel of a 256^3 volume, per lockstep group of `batch` -- first
projection of the sweep (A gathered from the volume, 32768 x 512 times 512 x 64) and last product of the chain
(4096 x 64 times 64 x 4096, scattered into the volume) -- plus the plain site-4 projection (4096 x 512 times 512 x 64).
usage: python tools/skinny_gemm_probe.py [batch]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgcompressionmps_amd import _lib  # noqa: E402
from imgcompressionmps_amd.core.ndmps import _plan_for  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
lib = _lib.load()
dev = torch.device("cuda:0")
plan = _plan_for((256, 256, 256), 0)
numel = plan.numel
sp = _lib.stream_ptr


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


def ptrs(ts):
    return (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


vols = [torch.randn(numel, device=dev) for _ in range(batch)]
row_off, col_off, _ = plan.gather_tables(512, dev)
m0 = numel // 512
W = [torch.randn(512, 64, device=dev) for _ in range(batch)]
out = [torch.empty(m0 * 64, device=dev) for _ in range(batch)]
pa, pb, pc = ptrs(vols), ptrs(W), ptrs(out)
r_off, c_off, _ = plan.split_tables(4096, dev)
left = [torch.randn(4096, 64, device=dev) for _ in range(batch)]
R = [torch.randn(64, 4096, device=dev) for _ in range(batch)]
qa, qb, qc = ptrs(left), ptrs(R), ptrs(vols)
carry = [torch.randn(4096, 512, device=dev) for _ in range(batch)]
core = [torch.randn(64, 512, device=dev) for _ in range(batch)]
nxt = [torch.empty(4096 * 64, device=dev) for _ in range(batch)]
sa, sb, sc = ptrs(carry), ptrs(core), ptrs(nxt)

order = torch.argsort(row_off[:m0])
row_sorted = row_off[:m0][order].contiguous()
c_rows = (order * 64).contiguous()
c_cols = torch.arange(64, device=dev, dtype=torch.int64)
order2 = torch.argsort(r_off[:4096])
r_sorted = r_off[:4096][order2].contiguous()
a_rows = (order2 * 64).contiguous()
a_cols = torch.arange(64, device=dev, dtype=torch.int64)
cases = (
    ("last chain product, rows in memory order", 2.0 * numel * 64 * batch, 4.0 * numel * batch,
     lambda: _lib.check(lib.ndmps_sgemm_indexed_batched(batch, 4096, 4096, 64, qa, 64, a_rows.data_ptr(), a_cols.data_ptr(), 1, qb, 4096, qc, 0,
                                                        r_sorted.data_ptr(), c_off.data_ptr(), sp()))),
    ("first projection, rows in memory order", 2.0 * numel * 64 * batch, 4.0 * numel * batch * 1.125,
     lambda: _lib.check(lib.ndmps_sgemm_indexed_batched(batch, m0, 64, 512, pa, 0, row_sorted.data_ptr(), col_off.data_ptr(), 1,
                                                        pb, 64, pc, 64, c_rows.data_ptr(), c_cols.data_ptr(), sp()))),
    ("first projection (gathered A)", 2.0 * numel * 64 * batch, 4.0 * numel * batch * 1.125,
     lambda: _lib.check(lib.ndmps_sgemm_indexed_batched(batch, m0, 64, 512, pa, 0, row_off.data_ptr(), col_off.data_ptr(), 1,
                                                        pb, 64, pc, 64, None, None, sp()))),
    ("last chain product (scattered C)", 2.0 * numel * 64 * batch, 4.0 * numel * batch,
     lambda: _lib.check(lib.ndmps_sgemm_indexed_batched(batch, 4096, 4096, 64, qa, 64, None, None, 0, qb, 4096, qc, 0,
                                                        r_off.data_ptr(), c_off.data_ptr(), sp()))),
    ("site-4 projection (A core^T)", 2.0 * 4096 * 512 * 64 * batch, 4.0 * 4096 * 576 * batch,
     lambda: _lib.check(lib.ndmps_sgemm_batched(batch, 0, 1, 4096, 64, 512, sa, 512, sb, 512, sc, 64, sp()))),
)
for name, flops, nbytes, fn in cases:
    t = timed(fn)
    print(f"{name:34s} batch {batch}: {t:7.3f} ms  {flops / t / 1e9:6.1f} TFLOP/s  {nbytes / t / 1e9:6.2f} TB/s "
          f"({t / batch * 1e3:.1f} us per volume)", flush=True)

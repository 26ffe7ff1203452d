"""Scratch: conv_to_mps with the reference's defaults (DCT mode, no bond cap: exact sweeps) on 12 volumes of 256^3 and of
128^3: chunk sizes by the workspace rule, time per volume, bonds against the loop."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from imgcompressionmps_amd import NDMPS  # noqa: E402
from imgcompressionmps_amd.core import batch  # noqa: E402

dev = torch.device("cuda:0")
for size, n in ((128, 12), (256, 12)):
    xs = [bench.synthetic_mri_device((size,) * 3, 100 + i, dev) for i in range(n)]
    batch.conv_to_mps(xs[:2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    objs = batch.conv_to_mps(xs)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    loop = [NDMPS.from_tensor(x, mode="DCT") for x in xs[:3]]
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{n} x {size}^3 exact, DCT: conv_to_mps {(t1 - t0) / n * 1e3:.1f} ms per volume, loop {(t2 - t1) / 3 * 1e3:.1f}; "
          f"bonds equal: {[o.bond_sizes() for o in objs[:3]] == [o.bond_sizes() for o in loop]}; peak memory "
          f"{torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB", flush=True)
    del objs, loop, xs
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()

"""Scratch: core/batch.conv_to_mps on a list of 128 volumes of 256^3 (chi = 64, Std) against the reference's loop over
from_tensor (first 8 volumes only)."""
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
xs = [bench.synthetic_mri_device((256,) * 3, 100 + i % 16, dev) for i in range(128)]
batch.conv_to_mps(xs[:64], mode="Std", max_bond=64)
torch.cuda.synchronize()
t0 = time.perf_counter()
objs = batch.conv_to_mps(xs, mode="Std", max_bond=64)
torch.cuda.synchronize()
t1 = time.perf_counter()
[NDMPS.from_tensor(x, mode="Std", max_bond=64) for x in xs[:2]]
torch.cuda.synchronize()
t2 = time.perf_counter()
loop = [NDMPS.from_tensor(x, mode="Std", max_bond=64) for x in xs[:8]]
torch.cuda.synchronize()
t3 = time.perf_counter()
print(f"conv_to_mps, 128 x 256^3: {(t1 - t0) * 1e3:.1f} ms = {(t1 - t0) / 128 * 1e3:.2f} ms per volume; "
      f"loop over from_tensor: {(t3 - t2) / 8 * 1e3:.2f} ms per volume; bonds {objs[5].bond_sizes()} / {loop[5].bond_sizes()}")

"""Scratch: compress_list(0.01) on 12 exact MPS of 256^3 volumes: three lanes against the loop."""
import copy
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from imgcompressionmps_amd.core import batch  # noqa: E402

dev = torch.device("cuda:0")
xs = [bench.synthetic_mri_device((256,) * 3, 100 + i, dev) for i in range(12)]
objs = batch.conv_to_mps(xs, mode="Std")
for serial in ("", "1", "", "1"):
    work = copy.deepcopy(objs)
    if serial:
        os.environ["NDMPS_COMPRESS_LIST_SERIAL"] = "1"
    else:
        os.environ.pop("NDMPS_COMPRESS_LIST_SERIAL", None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    batch.compress_list(work, 0.01)
    torch.cuda.synchronize()
    print(f"compress_list(0.01), 12 exact MPS of 256^3, {'loop' if serial else 'three lanes'}: "
          f"{(time.perf_counter() - t0) / 12 * 1e3:.1f} ms per object; bonds {work[0].bond_sizes()}", flush=True)

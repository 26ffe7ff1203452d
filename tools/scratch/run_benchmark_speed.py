"""Scratch: core/batch.run_benchmark (the reference's quality-vs-ratio loop) on 12 volumes of 128^3, three cutoffs: lanes vs loop."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from imgcompressionmps_amd.core import batch  # noqa: E402

dev = torch.device("cuda:0")
xs = [bench.synthetic_mri_device((128,) * 3, 100 + i, dev) for i in range(12)]
out = {}
for serial in ("", "1", "", "1"):
    if serial:
        os.environ["NDMPS_LIST_SERIAL"] = "1"
    else:
        os.environ.pop("NDMPS_LIST_SERIAL", None)
    objs = batch.conv_to_mps(xs, mode="Std")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = batch.run_benchmark(objs, xs, [0.01, 0.03, 0.1], verbose=False)
    torch.cuda.synchronize()
    out[serial] = res
    print(f"run_benchmark, 12 x 128^3, 3 cutoffs, {'loop' if serial else 'three lanes'}: {(time.perf_counter() - t0) * 1e3:.0f} ms", flush=True)
import numpy as np
print("same figures:", all(np.array_equal(np.asarray(out[""][k], dtype=object), np.asarray(out["1"][k], dtype=object)) for k in out[""]))

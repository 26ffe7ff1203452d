"""One volume (or a lockstep group) through from_tensors + to_tensor, for rocprofv3 timelines.
usage: python tools/single_probe.py [n_volumes] [size] [chi] [reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgcompressionmps_amd import NDMPS  # noqa: E402
from oracle.metrics import synthetic_mri  # noqa: E402

nv = int(sys.argv[1]) if len(sys.argv) > 1 else 1
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
chi = int(sys.argv[3]) if len(sys.argv) > 3 else 64
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
xs = [torch.from_numpy(synthetic_mri((size,) * 3, seed=2025 + j)).cuda() for j in range(nv)]


def run():
    objs = NDMPS.from_tensors(xs, max_bond=chi)
    return [o.to_tensor(as_torch=True) for o in objs]


run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    run()
torch.cuda.synchronize()
print(f"{nv} x {size}^3 chi={chi}: {(time.perf_counter() - t0) / reps * 1e3:.2f} ms per pass")

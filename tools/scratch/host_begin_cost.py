"""Scratch: host time of NDMPS.from_tensors_begin for a batch (GPU drained before every call): total, the library calls
inside it, and result().  usage: python tools/scratch/host_begin_cost.py [N] [SIZE] [CHI]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from imgcompressionmps_amd import NDMPS, _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
size = int(sys.argv[2]) if len(sys.argv) > 2 else 128
chi = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = torch.device("cuda:0")
xs = [bench.synthetic_mri_device((size,) * 3, 100 + i, dev) for i in range(n)]
lib = _lib.load()
acc = {}


def wrap(name):
    fn = getattr(lib, name)

    def timed(*a):
        t = time.perf_counter()
        r = fn(*a)
        acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t) * 1e3
        return r

    setattr(lib, name, timed)


for name in ("ndmps_tt_sweep_batched_fused_begin_f32", "ndmps_chain_contract_scatter_batched_f32", "ndmps_tt_sweep_finish",
             "ndmps_minmax_arena_launch_f32", "ndmps_tt_layout", "ndmps_chain_batched_workspace_bytes"):
    wrap(name)
for _ in range(3):
    NDMPS.from_tensors_begin(xs, max_bond=chi, reconstruct=True).result()
torch.cuda.synchronize()
acc.clear()
tb = tr = 0.0
reps = 10
for _ in range(reps):
    t0 = time.perf_counter()
    p = NDMPS.from_tensors_begin(xs, max_bond=chi, reconstruct=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    p.result()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    tb += (t1 - t0) * 1e3
    tr += (t3 - t2) * 1e3
print(f"{n} x {size}^3 chi {chi}: begin {tb / reps:.2f} ms on the host, result (GPU drained) {tr / reps:.2f} ms; inside: "
      + ", ".join(f"{k.replace('ndmps_', '')} {v / reps:.2f}" for k, v in acc.items()))

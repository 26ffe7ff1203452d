"""Scratch: a STREAM of small batches (8 volumes of 256^3, chi = 64 -- what one GPU of eight sees of 64 volumes): ms per
batch for the one-call form, for begin / result on one lane and on two alternating lanes."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from imgcompressionmps_amd.core import batch as hb  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
chi = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dev = torch.device("cuda:0")
xs = [bench.synthetic_mri_device((size,) * 3, 100 + i, dev) for i in range(n)]
steps = 12


def run(mode, lanes):
    pend = []

    def step(k):
        if mode == "sync":
            hb.encode_decode_concurrent(xs, groups=1, max_bond=chi, wait=False)
            return
        pend.append(hb.encode_decode_begin(xs, groups=1, max_bond=chi, lane=k, lanes=lanes))
        while len(pend) > lanes:
            pend.pop(0).result()

    for k in range(4):
        step(k)
    while pend:
        pend.pop(0).result()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step(k)
    while pend:
        pend.pop(0).result()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for mode, lanes in (("sync", 1), ("begin", 1), ("begin", 2), ("begin", 3), ("sync", 1), ("begin", 2)):
    print(f"{n} x {size}^3 chi {chi}: {mode:5s} lanes {lanes}: {run(mode, lanes):.2f} ms per batch", flush=True)

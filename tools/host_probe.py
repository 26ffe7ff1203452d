#!/usr/bin/env python
"""Where does the HOST spend its time in one lockstep group?  cProfile of NDMPS.from_tensors + to_tensor on
`batch` synthetic 256^3 volumes (GPU box).  usage: python tools/host_probe.py [batch] [size] [chi]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgcompressionmps_amd as pkg  # noqa: E402
from imgcompressionmps_amd.core.ndmps import NDMPS  # noqa: E402


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    chi = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    g = torch.Generator(device="cuda").manual_seed(1)
    vols = [torch.rand((size,) * 3, device="cuda", generator=g) for _ in range(batch)]

    def step():
        objs = NDMPS.from_tensors(vols, max_bond=chi)
        return [o.to_tensor(as_torch=True) for o in objs]

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"batch {batch}: host returns after {t_issue * 1e3:.2f} ms, device done after {t_all * 1e3:.2f} ms")
    pr = cProfile.Profile()
    pr.enable()
    step()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)


if __name__ == "__main__":
    main()

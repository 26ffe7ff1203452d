"""Probe: throughput of B independent 256^3 volumes compressed+reconstructed concurrently
(one host thread + one HIP stream per volume) vs one at a time."""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgcompressionmps_amd import NDMPS
from oracle.metrics import synthetic_mri

size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
chi = int(sys.argv[2]) if len(sys.argv) > 2 else 64
x0 = torch.from_numpy(synthetic_mri((size,) * 3, seed=2025)).cuda()
def one(x, stream):
    with torch.cuda.stream(stream):
        o = NDMPS.from_tensor(x, max_bond=chi)
        r = o.to_tensor(as_torch=True)
        stream.synchronize()
    return r
Bs = tuple(int(v) for v in sys.argv[3].split(',')) if len(sys.argv) > 3 else (1, 2, 4, 8, 16)
for B in Bs:
    xs = [x0.clone() for _ in range(B)]
    streams = [torch.cuda.Stream() for _ in range(B)]
    pool = ThreadPoolExecutor(B)
    list(pool.map(one, xs, streams))  # warm-up
    torch.cuda.synchronize()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        list(pool.map(one, xs, streams))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"B={B:2d}: {dt*1e3:8.1f} ms per batch, {B*size**3/dt/1e6:9.1f} Mvoxels/s", flush=True)
    pool.shutdown()

"""Probe: G concurrent groups (host thread + stream each) x B volumes in lockstep per group."""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgcompressionmps_amd import NDMPS
from oracle.metrics import synthetic_mri
x0 = torch.from_numpy(synthetic_mri((256,) * 3, seed=2025)).cuda()
def group(xs, stream):
    with torch.cuda.stream(stream):
        objs = NDMPS.from_tensors(xs, max_bond=64)
        recs = [o.to_tensor(as_torch=True) for o in objs]
        stream.synchronize()
    return recs
for G, B in [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(1, 8)]:
    xs = [[x0.clone() for _ in range(B)] for _ in range(G)]
    streams = [torch.cuda.Stream() for _ in range(G)]
    pool = ThreadPoolExecutor(G)
    list(pool.map(group, xs, streams)); torch.cuda.synchronize()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        list(pool.map(group, xs, streams))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"groups={G} x batch={B}: {dt*1e3:7.1f} ms per round, {G*B*256**3/dt/1e6:8.1f} Mvoxels/s", flush=True)
    pool.shutdown()

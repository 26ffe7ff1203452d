"""Probe: which torch pool streams share a hardware queue (4 groups x 8 volumes on chosen streams)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from concurrent.futures import ThreadPoolExecutor
from imgcompressionmps_amd.core import batch as B
from oracle.metrics import synthetic_mri

dev = torch.device("cuda", 0)
base = [torch.from_numpy(synthetic_mri((256,) * 3, seed=7 + j)).to(dev) for j in range(8)]
streams = [torch.cuda.Stream() for _ in range(16)]
print([hex(s.cuda_stream) for s in streams], flush=True)
xs = [base[j % 8] for j in range(32)]
pool = ThreadPoolExecutor(4)
for pick in [(0, 1, 2, 3), (1, 2, 3, 4), (0, 2, 4, 6), (0, 3, 6, 9), (4, 5, 6, 7), (0, 5, 10, 15), (0, 4, 8, 12), (0, 1, 4, 5)]:
    for g, s in enumerate(pick):
        B._group_streams[(0, g)] = streams[s]
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        B.encode_decode_concurrent(xs, groups=4, max_bond=64, pool=pool)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(pick, f"{dt*1e3:.1f} ms", flush=True)

"""Probe: wall time and allocator traffic of encode_decode_concurrent for (groups, per-group) pairs."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from concurrent.futures import ThreadPoolExecutor
from imgcompressionmps_amd.core import batch as B
from oracle.metrics import synthetic_mri

dev = torch.device("cuda", 0)
base = [torch.from_numpy(synthetic_mri((256,) * 3, seed=7 + j)).to(dev) for j in range(8)]
for groups, per in [(1, 16), (2, 16), (2, 8), (4, 8)]:
    xs = [base[j % 8] for j in range(groups * per)]
    pool = ThreadPoolExecutor(groups)
    for it in range(4):
        s0 = torch.cuda.memory_stats()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        B.encode_decode_concurrent(xs, groups=groups, max_bond=64, pool=pool)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        s1 = torch.cuda.memory_stats()
        print(groups, per, it, f"{dt*1e3:.1f} ms", "dev_allocs", s1["num_device_alloc"] - s0["num_device_alloc"],
              "dev_frees", s1["num_device_free"] - s0["num_device_free"], "retries", s1["num_alloc_retries"] - s0["num_alloc_retries"],
              "reserved GB", round(s1["reserved_bytes.all.current"] / 2**30, 1), flush=True)
    pool.shutdown()

"""Probe: wall-clock share of from_tensors / to_tensor inside the concurrent groups, per host thread."""
import sys, os, time, threading, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from concurrent.futures import ThreadPoolExecutor
from imgcompressionmps_amd import NDMPS
from imgcompressionmps_amd.core import batch as B
from oracle.metrics import synthetic_mri

dev = torch.device("cuda", 0)
base = [torch.from_numpy(synthetic_mri((256,) * 3, seed=7 + j)).to(dev) for j in range(8)]
xs = [base[j % 8] for j in range(64)]
acc = collections.defaultdict(float)
lock = threading.Lock()
orig_ft, orig_tt = NDMPS.from_tensors.__func__, NDMPS.to_tensor
def ft(cls, *a, **k):
    t0 = time.perf_counter(); r = orig_ft(cls, *a, **k); torch.cuda.current_stream().synchronize()
    with lock: acc["from_tensors"] += time.perf_counter() - t0
    return r
def tt(self, *a, **k):
    t0 = time.perf_counter(); r = orig_tt(self, *a, **k)
    with lock: acc["to_tensor(launch)"] += time.perf_counter() - t0
    return r
NDMPS.from_tensors = classmethod(ft); NDMPS.to_tensor = tt
pool = ThreadPoolExecutor(4)
for it in range(5):
    acc.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    B.encode_decode_concurrent(xs, groups=4, max_bond=64, pool=pool)
    dt = time.perf_counter() - t0
    print(f"step {it}: wall {dt*1e3:.1f} ms; per-thread mean: " + ", ".join(f"{k} {v/4*1e3:.1f} ms" for k, v in acc.items()), flush=True)

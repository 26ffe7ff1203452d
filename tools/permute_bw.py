"""Micro-benchmark of the permutation kernels: back-to-back launches between two HIP events."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgcompressionmps_amd import _lib
from imgcompressionmps_amd.core.ndmps import _plan_for
lib = _lib.load()
for shape in [(256, 256, 256), (512, 512, 512), (512, 680), (128, 128, 64, 256)]:
    n = int(np.prod(shape))
    plan = _plan_for(shape, 0)
    x = torch.rand(n, device="cuda"); y = torch.empty_like(x)
    for name, fn in [("encode", lib.ndmps_encode_permute), ("decode", lib.ndmps_decode_permute),
                     ("encode_generic", lib.ndmps_encode_permute_generic), ("decode_generic", lib.ndmps_decode_permute_generic)]:
        reps = 20
        for _ in range(3): fn(plan.handle, x.data_ptr(), y.data_ptr(), 4, None)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): fn(plan.handle, x.data_ptr(), y.data_ptr(), 4, _lib.stream_ptr())
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) / reps * 1e3
        print(f"{str(shape):22s} {name:15s} tiled={lib.ndmps_plan_is_tiled(plan.handle)} {us:8.1f} us  {2*4*n/us/1e6:7.2f} TB/s (r+w)", flush=True)
# reference: plain copy
x = torch.rand(256**3, device="cuda"); y = torch.empty_like(x)
torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): y.copy_(x)
b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) / 20 * 1e3
print(f"torch copy 64 MiB: {us:.1f} us {2*4*256**3/us/1e6:.2f} TB/s")

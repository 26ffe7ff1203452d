"""Diagnostic: where one inner step of blk_diag_kernel spends its cycles (s_memtime stamps)."""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgcompressionmps_amd import _lib
lib = _lib.load()
n = 512
rng = np.random.default_rng(0)
a = rng.standard_normal((2 * n, n)) * np.logspace(0, -3, n)[None, :]
g = torch.from_numpy(a.T @ a).cuda()
ws = torch.empty(8 << 20, dtype=torch.uint8, device="cuda")
for full in (0, 1):
    out = (C.c_uint64 * (8 * (n // 32)))()
    for rep in range(3):
        gg = g.clone()
        _lib.check(lib.ndmps_debug_diag_stamps(gg.data_ptr(), n, ws.data_ptr(), ws.numel(), out, full, None))
    arr = np.array(out[:], dtype=np.float64).reshape(-1, 8)
    steps = arr[0, 6]
    names = ["load", "rot", "(unused)", "apply", "barrier", "store"]
    med = np.median(arr[:, :6], axis=0)
    print(f"full={full} inner_steps={int(steps)} (cycles of s_memtime, median over {arr.shape[0]} workgroups)")
    for nm, v in zip(names, med):
        per = v / steps if nm in ("rot", "apply", "barrier") else v
        print(f"  {nm:6s} total={v:9.0f}  per_inner_step={per:8.1f}" if nm in ("rot", "apply", "barrier") else f"  {nm:6s} total={v:9.0f}")
    print("  sum per inner step", med[1:5].sum() / steps)

"""Where a column of the resident reduction goes (scratch): needs tools/scratch/stamps/libndmps_stamps.so, the library
with eig_tridiag.hip compiled -DNDMPS_TEAM_STAMPS (accumulated shader-clock cycles per phase in the last workgroup).
usage: python tools/scratch/team_stamps.py [orders] [batch]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from imgcompressionmps_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "tools", "scratch", "stamps", "libndmps_stamps.so")
lib = _lib.load()
dev = "cuda:0"
names = ["poll", "barrier1", "sigma", "records", "body", "sums", "meeting", "-"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for n in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "512,1024,2048").split(",")]:
    gen = torch.Generator(device=dev).manual_seed(n)
    a = torch.randn((B, n + 64, n), dtype=torch.float64, device=dev, generator=gen)
    g = torch.bmm(a.transpose(1, 2), a).contiguous()
    k = 64
    nb = lib.ndmps_syevd_topk_workspace_bytes(n, B, k)
    ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
    v = torch.zeros_like(g)
    w = torch.zeros((B, n), dtype=torch.float64, device=dev)
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(lib.ndmps_syevd_topk_values_f64(B, g.data_ptr(), n * n, _lib.i64_array([n] * B), v.data_ptr(), n * n, w.data_ptr(), n, k,
                                                   ws.data_ptr(), nb, _lib.stream_ptr()))
        e1.record()
        torch.cuda.synchronize()
    off = lib.ndmps_syevd_topk_stamps_offset(n, B, k)
    st = ws[off:off + 64].view(torch.int64).cpu().numpy().astype(np.float64)
    cols = n - 128
    tot = st.sum()
    print(f"n={n} B={B}: values {e0.elapsed_time(e1):.3f} ms; cycles per column {tot / cols:.0f}: " +
          ", ".join(f"{nm} {c / cols:.0f}" for nm, c in zip(names, st) if nm != "-"), flush=True)

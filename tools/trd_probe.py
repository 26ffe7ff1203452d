"""Timing of the direct top-k eigen-solver against the block Jacobi on B Gram matrices of order n built from
synthetic volumes (the shape of the three big eigenproblems of a 256^3 / chi = 64 sweep).
usage: python tools/trd_probe.py [B] [n] [k]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgcompressionmps_amd import _lib  # noqa: E402
from oracle.metrics import synthetic_mri  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
k = int(sys.argv[3]) if len(sys.argv) > 3 else 64
if os.environ.get("NDMPS_PROBE_LIB"):  # A/B against another build of the library
    _lib.LIB_PATH = os.environ["NDMPS_PROBE_LIB"]
lib = _lib.load()
dev = "cuda:0"
vols = [torch.from_numpy(synthetic_mri((128, 128, 128), seed=s)).to(dev) for s in range(min(B, 4))]
a = torch.stack([torch.roll(vols[j % len(vols)], j // len(vols) * 8, 0).reshape(n, -1).to(torch.float64) for j in range(B)])
g0 = torch.bmm(a, a.transpose(1, 2)).contiguous()
del a
sizes = _lib.i64_array([n] * B)
ks = _lib.i64_array([k] * B)
v = torch.empty_like(g0)
w = torch.empty((B, n), dtype=torch.float64, device=dev)
sp = _lib.stream_ptr()


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append((e0.elapsed_time(e1), (time.perf_counter() - t0) * 1e3))
    return min(t[0] for t in ts), min(t[1] for t in ts)


nb = lib.ndmps_syevd_topk_workspace_bytes(n, B, k)
ws = torch.empty(nb, dtype=torch.uint8, device=dev)


def trd_values():
    _lib.check(lib.ndmps_syevd_topk_values_f64(B, g0.data_ptr(), n * n, sizes, v.data_ptr(), n * n, w.data_ptr(), n, k,
                                               ws.data_ptr(), nb, sp))


def trd_all():
    trd_values()
    _lib.check(lib.ndmps_syevd_topk_vectors_f64(B, sizes, ks, k, ws.data_ptr(), nb, None, sp))


tv = timed(trd_values)
ta = timed(trd_all)
off = lib.ndmps_syevd_topk_stamps_offset(n, B, k)
st = ws[off:off + 16 * 8].view(torch.int64).cpu().numpy()
names = ["factor+fwd", "solves", "gram1", "chol1", "trsm1", "gram2", "chol2", "trsm2"]
print("invit phases (us, matrix 0): " + ", ".join(f"{nm} {(st[i + 1] - st[i]) / 100:.1f}" for i, nm in enumerate(names)))
w_trd = w.clone()
v_trd = v.clone()
nbj = lib.ndmps_syevj_batched_workspace_bytes(n, B)
wsj = torch.empty(nbj, dtype=torch.uint8, device=dev)
sw = (C.c_int * B)()


def jac_all():
    g = g0.clone()
    _lib.check(lib.ndmps_syevj_batched_values_f64(B, g.data_ptr(), n * n, sizes, v.data_ptr(), n * n, w.data_ptr(), n, 1e-13,
                                                  wsj.data_ptr(), nbj, sw, sp))
    _lib.check(lib.ndmps_syevj_batched_vectors_f64(B, g.data_ptr(), n * n, sizes, v.data_ptr(), n * n, w.data_ptr(), n, ks,
                                                   wsj.data_ptr(), nbj, sp))


tj = timed(jac_all)
dw = float((w_trd - w).abs().max() / w.abs().max())
pa = v_trd.view(B, n, n)[:, :, :k]
pb = v.view(B, n, n)[:, :, :k]
dp = float((pa @ pa.transpose(1, 2) - pb @ pb.transpose(1, 2)).abs().max())
print(f"B={B} n={n} k={k}: tridiag values {tv[0]:.3f} ms, values+vectors {ta[0]:.3f} ms (host {ta[1]:.3f}); "
      f"jacobi values+vectors {tj[0]:.3f} ms ({max(sw)} sweeps); |dw|/w0 {dw:.1e}  |dP| {dp:.1e}")

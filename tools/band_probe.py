"""The two-stage tridiagonalisation (csrc/eig_band.inc) against LAPACK and against the one-stage resident kernel:
accuracy (eigenvalues, residuals, orthogonality) and time of both phases, for B Gram matrices of order n.
usage: python tools/band_probe.py [B] [n] [k] [kind]     kind: volume (default) | random | graded
The semi-bandwidth comes from NDMPS_TRD_BAND (2 or 4); unset = the one-stage path."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgcompressionmps_amd import _lib  # noqa: E402
from oracle.metrics import synthetic_mri  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
k = int(sys.argv[3]) if len(sys.argv) > 3 else 64
kind = sys.argv[4] if len(sys.argv) > 4 else "volume"
lib = _lib.load()
dev = "cuda:0"
rng = np.random.default_rng(7)
if kind == "volume":
    vols = [synthetic_mri((128, 128, 128), seed=s).astype(np.float64) for s in range(min(B, 4))]
    mats = []
    for j in range(B):
        a = np.roll(vols[j % len(vols)], j // len(vols) * 8, 0).reshape(-1, 512)[:, :n]
        mats.append(a.T @ a)
    g = np.stack(mats)
elif kind == "graded":
    g = np.stack([(lambda x: x.T @ x)(rng.standard_normal((2 * n, n)) * np.logspace(0, -6, n)) for _ in range(B)])
else:
    g = np.stack([(lambda x: x.T @ x)(rng.standard_normal((2 * n, n))) for _ in range(B)])
g0 = torch.from_numpy(g).to(dev)
sizes = _lib.i64_array([n] * B)
ks = _lib.i64_array([k] * B)
v = torch.empty_like(g0)
w = torch.empty((B, n), dtype=torch.float64, device=dev)
sp = _lib.stream_ptr()
nb = lib.ndmps_syevd_topk_workspace_bytes(n, B, k)
ws = torch.empty(nb, dtype=torch.uint8, device=dev)


def values():
    _lib.check(lib.ndmps_syevd_topk_values_f64(B, g0.data_ptr(), n * n, sizes, v.data_ptr(), n * n, w.data_ptr(), n, k,
                                               ws.data_ptr(), nb, sp))


def both():
    values()
    _lib.check(lib.ndmps_syevd_topk_vectors_f64(B, sizes, ks, k, ws.data_ptr(), nb, None, sp))


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


status = (C.c_int * B)()
values()
if os.environ.get("NDMPS_BAND_STAMPS"):
    torch.cuda.synchronize()
    off = lib.ndmps_syevd_topk_stamps_offset(n, B, k)
    st = np.frombuffer(ws[off: off + 64].cpu().numpy().tobytes(), dtype=np.int64)
    names = ["S/M/W (+loads)", "Wc, X update", "QR steps", "T, records", "update pass", "matvec, fold, publish", "meeting"]
    panels = (n + int(os.environ["NDMPS_TRD_BAND"]) - 1) // int(os.environ["NDMPS_TRD_BAND"])
    print("band kernel phases of the last workgroup, us per panel:",
          ", ".join(f"{nm} {st[i] / 100.0 / panels:.2f}" for i, nm in enumerate(names)), f"| total {st[:7].sum() / 100.0 / panels:.2f}")
_lib.check(lib.ndmps_syevd_topk_vectors_f64(B, sizes, ks, k, ws.data_ptr(), nb, status, sp))
wv, vv = w.cpu().numpy(), v.cpu().numpy()
worst = {"eig": 0.0, "res": 0.0, "orth": 0.0}
for bi in range(B):
    ref = np.linalg.eigvalsh(g[bi])[::-1]
    s0 = ref[0]
    worst["eig"] = max(worst["eig"], np.abs(wv[bi, :k] - ref[:k]).max() / s0)
    vk = vv[bi][:, :k]
    worst["res"] = max(worst["res"], np.abs(g[bi] @ vk - vk * wv[bi, :k]).max() / s0)
    worst["orth"] = max(worst["orth"], np.abs(vk.T @ vk - np.eye(k)).max())
t_values, t_both = timed(values), timed(both)
print(f"band={os.environ.get('NDMPS_TRD_BAND', '-')} B={B} n={n} k={k} {kind}: status {sorted(set(status))} "
      f"eig {worst['eig']:.1e} res {worst['res']:.1e} orth {worst['orth']:.1e}   values {t_values:.3f} ms  "
      f"values+vectors {t_both:.3f} ms")

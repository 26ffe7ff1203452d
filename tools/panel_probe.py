"""Timing and a quick accuracy check of the tridiagonalisation paths for orders above 512: panel-blocked launches
(csrc/eig_panel.inc, the default) against the one-launch-per-column path (NDMPS_TRD_NO_PANEL=1).
usage: python tools/panel_probe.py [orders, comma separated] [k] [batch]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgcompressionmps_amd import _lib  # noqa: E402

orders = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1024,2048,4096").split(",")]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 128
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
lib = _lib.load()
dev = "cuda:0"
sp = _lib.stream_ptr()


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = (1e30, 1e30)
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, (e0.elapsed_time(e1), (time.perf_counter() - t0) * 1e3))
    return best


for n in orders:
    gen = torch.Generator(device=dev).manual_seed(n)
    a = torch.randn((B, n + 64, n), dtype=torch.float64, device=dev, generator=gen)
    a = a * torch.logspace(0, -5, n, dtype=torch.float64, device=dev)[None, None, :]
    q = torch.linalg.qr(torch.randn((n, n), dtype=torch.float64, device=dev, generator=gen))[0]
    g0 = q @ torch.bmm(a.transpose(1, 2), a) @ q.T
    g0 = (0.5 * (g0 + g0.transpose(1, 2))).contiguous()
    del a
    sizes = _lib.i64_array([n] * B)
    ks = _lib.i64_array([k] * B)
    v = torch.zeros_like(g0)
    w = torch.zeros((B, n), dtype=torch.float64, device=dev)
    nb = lib.ndmps_syevd_topk_workspace_bytes(n, B, k)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)

    def values():
        _lib.check(lib.ndmps_syevd_topk_values_f64(B, g0.data_ptr(), n * n, sizes, v.data_ptr(), n * n, w.data_ptr(), n, k,
                                                   ws.data_ptr(), nb, sp))

    def both():
        values()
        _lib.check(lib.ndmps_syevd_topk_vectors_f64(B, sizes, ks, k, ws.data_ptr(), nb, None, sp))

    ref = torch.linalg.eigvalsh(g0).flip(-1)
    for label, env in (("panel", None), ("columns", "NDMPS_TRD_NO_PANEL")):
        os.environ.pop("NDMPS_TRD_NO_PANEL", None)
        if env:
            os.environ[env] = "1"
        tv = timed(values)
        ta = timed(both)
        dw = float((w[:, :k] - ref[:, :k]).abs().max() / ref.abs().max())
        vk = v.view(B, n, n)[:, :, :k]
        res = float((torch.bmm(g0, vk) - vk * w[:, None, :k]).abs().max() / ref.abs().max())
        orth = float((torch.bmm(vk.transpose(1, 2), vk) - torch.eye(k, dtype=torch.float64, device=dev)).abs().max())
        print(f"n={n} B={B} k={k} {label:8s}: values {tv[0]:8.3f} ms (host {tv[1]:8.3f}), values+vectors {ta[0]:8.3f} ms; "
              f"|dw|/w0 {dw:.1e} residual {res:.1e} orthogonality {orth:.1e}", flush=True)
    os.environ.pop("NDMPS_TRD_NO_PANEL", None)
    print(f"  resident launches given up: {lib.ndmps_syevd_topk_team_fallbacks()}", flush=True)

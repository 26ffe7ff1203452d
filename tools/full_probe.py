"""Every eigenpair of an order-n Gram matrix: the direct solver (tridiagonalisation, all eigenvalues by multi-section,
inverse iteration in column blocks, Cholesky-QR across the chip, back-transformation) against the block Jacobi it
replaces in exact sweeps and compress().
usage: python tools/full_probe.py [orders, comma separated] [--no-jacobi]"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgcompressionmps_amd import _lib  # noqa: E402

orders = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "512,1024,2048,4096").split(",")]
jacobi = "--no-jacobi" not in sys.argv
lib = _lib.load()
dev = "cuda:0"
sp = _lib.stream_ptr()


def timed(fn, reps=2):
    fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


for n in orders:
    gen = torch.Generator(device=dev).manual_seed(n)
    a = torch.randn((n + 64, n), dtype=torch.float64, device=dev, generator=gen)
    a = a * torch.logspace(0, -3, n, dtype=torch.float64, device=dev)[None, :]
    g0 = (a.T @ a).contiguous()
    del a
    sizes, ks = _lib.i64_array([n]), _lib.i64_array([n])
    v = torch.zeros_like(g0)
    w = torch.zeros(n, dtype=torch.float64, device=dev)
    nb = lib.ndmps_syevd_topk_workspace_bytes(n, 1, n)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    status = (C.c_int * 1)()

    def values():
        _lib.check(lib.ndmps_syevd_topk_values_f64(1, g0.data_ptr(), n * n, sizes, v.data_ptr(), n * n, w.data_ptr(), n, n,
                                                   ws.data_ptr(), nb, sp))

    def both():
        values()
        _lib.check(lib.ndmps_syevd_topk_vectors_f64(1, sizes, ks, n, ws.data_ptr(), nb, None, sp))

    tv, ta = timed(values), timed(both)
    _lib.check(lib.ndmps_syevd_topk_vectors_f64(1, sizes, ks, n, ws.data_ptr(), nb, status, sp))
    ref = torch.linalg.eigvalsh(g0).flip(-1)
    dw = float((w - ref).abs().max() / ref.abs().max())
    res = float((g0 @ v - v * w[None, :]).abs().max() / ref.abs().max())
    orth = float((v.T @ v - torch.eye(n, dtype=torch.float64, device=dev)).abs().max())
    line = (f"n={n} all {n} eigenpairs: values {tv:8.2f} ms, values+vectors {ta:8.2f} ms (workspace {nb / 2**20:.0f} MiB, status "
            f"{status[0]}); |dw|/w0 {dw:.1e} residual {res:.1e} orthogonality {orth:.1e}")
    if jacobi:
        nbj = lib.ndmps_syevj_batched_workspace_bytes(n, 1)
        wsj = torch.empty(nbj, dtype=torch.uint8, device=dev)
        sw = (C.c_int * 1)()
        vj, wj = torch.zeros_like(g0), torch.zeros(n, dtype=torch.float64, device=dev)

        def jac():
            g = g0.clone()
            _lib.check(lib.ndmps_syevj_batched_values_f64(1, g.data_ptr(), n * n, sizes, vj.data_ptr(), n * n, wj.data_ptr(), n,
                                                          1e-15, wsj.data_ptr(), nbj, sw, sp))
            _lib.check(lib.ndmps_syevj_batched_vectors_f64(1, g.data_ptr(), n * n, sizes, vj.data_ptr(), n * n, wj.data_ptr(), n,
                                                           ks, wsj.data_ptr(), nbj, sp))

        line += f"; block Jacobi {timed(jac, 1):8.2f} ms ({sw[0]} sweeps)"
    print(line, flush=True)

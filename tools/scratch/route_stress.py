"""Scratch: random orders / ranks / batches through the default routes of the direct solver against LAPACK (eigenvalues,
residuals, orthogonality), with an eye on the boundaries of the routes (orders around 512, 1024, 2048; odd orders; mixed
batches; k from 1 to more than 128)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from imgcompressionmps_amd import _lib  # noqa: E402
import test_gpu_parity as tp  # noqa: E402

lib = _lib.load()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cases = [([513], [64]), ([1023], [128]), ([1025], [100]), ([2047], [128]), ([2049], [64]), ([2050], [128]), ([1031, 1031], [17, 128]),
         ([2048, 2048], [128, 128]), ([1536, 1536, 1536], [50, 60, 70]), ([1024] * 4, [128] * 4), ([1024] * 5, [32] * 5),
         ([2562], [128]), ([3074], [96]), ([1100], [300]), ([1500], [1500]), ([640, 1900], [64, 128]), ([2048], [1])]
for _ in range(8):
    b = int(rng.integers(1, 4))
    sizes = [int(rng.integers(513, 2300)) for _ in range(b)]
    if rng.random() < 0.5:
        sizes = [sizes[0]] * b
    cases.append((sizes, [int(rng.integers(1, 129)) for _ in range(b)]))
worst = 0.0
for sizes, ks in cases:
    mats = []
    for n in sizes:
        a = rng.standard_normal((n + 8, n)) * np.logspace(0, -5, n)[None, :]
        mats.append(a.T @ a)
    kmax = max(ks)
    out = tp._topk(lib, mats, ks, k_max=kmax)
    for g, k, (w, v) in zip(mats, ks, out):
        tp._check_topk(g, w, v, k, k_max=kmax)
    print("ok", sizes, ks, flush=True)
print("all routes agree with LAPACK; fallbacks:", lib.ndmps_syevd_topk_team_fallbacks())

"""One solve of the direct eigen-solver (values + vectors) on a random graded Gram matrix: the command the PMC passes
of tools/pmc_solver.sh profile (a counter pass serialises every dispatch: thousands of launches take minutes).
usage: python tools/solver_once.py n k"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgcompressionmps_amd import _lib  # noqa: E402

n, k = int(sys.argv[1]), int(sys.argv[2])
lib = _lib.load()
dev = "cuda:0"
a = torch.randn((n + 64, n), dtype=torch.float64, device=dev) * torch.logspace(0, -4, n, dtype=torch.float64, device=dev)[None, :]
g = (a.T @ a).contiguous()
v = torch.zeros_like(g)
w = torch.zeros(n, dtype=torch.float64, device=dev)
nb = lib.ndmps_syevd_topk_workspace_bytes(n, 1, k)
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
sizes, ks = _lib.i64_array([n]), _lib.i64_array([k])
print("solving", n, k, flush=True)
_lib.check(lib.ndmps_syevd_topk_values_f64(1, g.data_ptr(), n * n, sizes, v.data_ptr(), n * n, w.data_ptr(), n, k, ws.data_ptr(), nb,
                                           _lib.stream_ptr()))
_lib.check(lib.ndmps_syevd_topk_vectors_f64(1, sizes, ks, k, ws.data_ptr(), nb, None, _lib.stream_ptr()))
torch.cuda.synchronize()
print("done", float(w[0]), flush=True)

"""Timing probe: batched eigen-solver on random Gram-like matrices (use under rocprofv3, with
NDMPS_EIG_DEBUG_ROLE=0/1/2 to time both roles / diag only / apply only of the step kernel)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from imgcompressionmps_amd import _lib
lib = _lib.load()
batch, n = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(0)
a = rng.standard_normal((batch, n, 4 * n)) * (0.9 ** np.arange(4 * n))[None, None, :]
g = torch.from_numpy(a @ a.transpose(0, 2, 1)).cuda()
v = torch.empty_like(g); w = torch.empty((batch, n), dtype=torch.float64, device="cuda")
nb = lib.ndmps_syevj_batched_workspace_bytes(n, batch)
ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
sw = (C.c_int * batch)()
for it in range(2):
    gg = g.clone()
    rc = lib.ndmps_syevj_batched_f64(batch, gg.data_ptr(), n * n, _lib.i64_array([n] * batch), v.data_ptr(), n * n,
                                     w.data_ptr(), n, ws.data_ptr(), nb, sw, None)
    torch.cuda.synchronize()
    print("rc", rc, "sweeps", list(sw)[:4])

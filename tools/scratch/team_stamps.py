"""Phase clocks of trd_team_kernel's last workgroup (build with make EXTRA=-DNDMPS_TEAM_STAMPS): cycles per column."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from imgcompressionmps_amd import _lib
lib = _lib.load()
B, n, k = int(sys.argv[1]), 512, 64
rng = np.random.default_rng(1)
g = np.stack([(lambda x: x.T @ x)(rng.standard_normal((2 * n, n))) for _ in range(B)])
g0 = torch.from_numpy(g).cuda()
v = torch.empty_like(g0); w = torch.empty((B, n), dtype=torch.float64, device="cuda")
nb = lib.ndmps_syevd_topk_workspace_bytes(n, B, k)
ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
sizes = _lib.i64_array([n] * B)
for _ in range(3):
    _lib.check(lib.ndmps_syevd_topk_values_f64(B, g0.data_ptr(), n * n, sizes, v.data_ptr(), n * n, w.data_ptr(), n, k, ws.data_ptr(), nb, _lib.stream_ptr()))
torch.cuda.synchronize()
off = lib.ndmps_syevd_topk_stamps_offset(n, B, k)
names = ["loads of y, column", "dot: sum, barrier", "sigma: sum, barrier", "householder, records, barrier", "tile loop", "fold, barrier, publish y", "publish column, ack, meeting"]
for b in (0, B - 1):
    st = np.frombuffer(ws[off + b * 128: off + b * 128 + 56].cpu().numpy().tobytes(), dtype=np.int64)
    print(f"matrix {b}: cycles per column:", ", ".join(f"{nm} {c / 384:.0f}" for nm, c in zip(names, st)), "| total", round(st.sum() / 384))

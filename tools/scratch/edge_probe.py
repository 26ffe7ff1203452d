import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from imgcompressionmps_amd import NDMPS
from oracle.ndmps_oracle import OracleNDMPS
rng = np.random.default_rng(0)
cases = {
    "zeros": np.zeros((16, 16, 16), np.float32),
    "const": np.full((16, 16, 16), 3.5, np.float32),
    "delta": np.zeros((16, 16, 16), np.float32),
    "rank1": np.einsum("i,j,k->ijk", rng.random(16), rng.random(16), rng.random(16)).astype(np.float32),
    "tiny": (rng.random((16, 16, 16)) * 1e-30).astype(np.float32),
    "huge": (rng.random((16, 16, 16)) * 1e18).astype(np.float32),
    "2d": rng.random((12, 20)).astype(np.float32),
    "1d": rng.random((64,)).astype(np.float32),
    "prime": rng.random((7, 11, 13)).astype(np.float32),
}
cases["delta"][3, 4, 5] = 2.0
for name, x in cases.items():
    for kw in ({}, {"max_bond": 4}, {"mode": "DCT"}):
        try:
            g = NDMPS.from_tensor(x, **kw); rg = g.to_tensor()
            o = OracleNDMPS.from_tensor(x, **kw); ro = o.to_tensor()
            scale = max(np.abs(x).max(), 1e-300)
            print(f"{name:6s} {str(kw):22s} bonds gpu {g.bond_sizes()} oracle {o.bond_sizes()} "
                  f"err_vs_x {np.abs(rg - x).max() / scale:.1e} err_vs_oracle {np.abs(rg - ro).max() / scale:.1e} "
                  f"norm {g.norm_value:.6g}/{o.norm_value:.6g} finite {np.isfinite(rg).all()}")
        except Exception as e:
            print(f"{name:6s} {str(kw):22s} EXC {type(e).__name__}: {str(e)[:100]}")

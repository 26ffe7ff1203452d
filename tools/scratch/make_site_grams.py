"""Dump the Gram matrices of the chi-capped sites of a synthetic volume (fp64, CPU) for solver experiments."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle.metrics import synthetic_mri
from oracle import index_map as oim
size = int(sys.argv[1]); seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2025
x = synthetic_mri((size,)*3, seed=seed).astype(np.float64)
dest = oim.flat_destination(x.shape)
dense = np.empty(x.size); dense[dest.ravel()] = x.ravel()
L = int(round(np.log2(size))); chi = 64
work = dense.reshape(-1, 1); chi_r = 1; out = {}
for i in range(L-1, 0, -1):
    rows = work.size // (8*chi_r); mat = work.reshape(rows, 8*chi_r); n = mat.shape[1]
    if n > rows: break
    G = mat.T @ mat
    w, V = np.linalg.eigh(G); V = V[:, ::-1]
    k = min(chi, n)
    if n > chi: out[f"G{i}"] = G
    work = mat @ V[:, :k]; chi_r = k
np.savez(f"/tmp/grams_{size}_{seed}.npz", **out)
print({k: v.shape for k, v in out.items()})

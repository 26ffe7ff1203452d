"""Probe: the reference's literal flow at 256^3 -- exact sweep (no bond cap: eigenproblems up to
4096 x 4096), then compress(cutoff), then to_tensor -- timing on the GPU."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imgcompressionmps_amd import NDMPS
from oracle.metrics import synthetic_mri
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
# third argument "f64": fp64 storage (the reference's own element type)
store = torch.float64 if "f64" in sys.argv[2:] else torch.float32
x = torch.from_numpy(synthetic_mri((size,) * 3, seed=2025)).cuda().to(store)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    o = NDMPS.from_tensor(x, dtype=store)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    exact_bonds = o.bond_sizes()
    print(f"exact from_tensor {size}^3 ({str(store)[6:]} storage): {t1 - t0:.3f} s bonds={exact_bonds} elems={o.number_elements_in_MPS()}", flush=True)
    r = o.to_tensor(as_torch=True); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"  to_tensor (exact MPS): {t2 - t1:.3f} s  max|err|={float((r - x).abs().max()):.2e}", flush=True)
    o.compress(0.01); torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"  compress(0.01): {t3 - t2:.3f} s bonds={o.bond_sizes()}", flush=True)
    r = o.to_tensor(as_torch=True); torch.cuda.synchronize(); t4 = time.perf_counter()
    print(f"  to_tensor: {t4 - t3:.3f} s  rel err={float((r - x).norm() / x.norm()):.3e}", flush=True)

if "oracle" in sys.argv[2:]:
    from oracle.ndmps_oracle import OracleNDMPS
    from oracle.metrics import compute_ssim_by_dim
    xh = x.cpu().numpy()
    t0 = time.perf_counter()
    ref = OracleNDMPS.from_tensor(xh, materialise_map=False)
    t1 = time.perf_counter()
    print(f"oracle exact from_tensor: {t1 - t0:.1f} s bonds={ref.bond_sizes()}  exact bonds equal: {exact_bonds == ref.bond_sizes()}", flush=True)
    ref.compress(0.01)
    t2 = time.perf_counter()
    print(f"oracle compress(0.01): {t2 - t1:.1f} s bonds={ref.bond_sizes()}", flush=True)
    rr = ref.to_tensor()
    t3 = time.perf_counter()
    print(f"oracle to_tensor: {t3 - t2:.1f} s", flush=True)
    rg = r.cpu().numpy().astype(np.float64)
    print("bonds equal:", o.bond_sizes() == ref.bond_sizes())
    print("rel Frobenius GPU vs oracle:", float(np.linalg.norm(rg - rr) / np.linalg.norm(rr)))
    x64 = xh.astype(np.float64)
    print("SSIM gpu / oracle:", compute_ssim_by_dim(x64, rg), compute_ssim_by_dim(x64, rr))

"""Oracle (test infrastructure): CPU restatement of the reference quality metrics
/root/reference/src/imgcompressionmps/utils/metrics.py:11-146.

SSIM restates ``skimage.metrics.structural_similarity`` as the reference calls it
(metrics.py:32: float inputs, explicit ``data_range``, ``win_size``; skimage defaults
K1=0.01, K2=0.03, uniform window, sample covariance, border crop (win-1)//2) with
``scipy.ndimage.uniform_filter``.  skimage itself is not importable on the GPU box.
Pinned by fixtures produced with the reference's metrics.py under
/opt/conda/bin/python3.9 (skimage 0.18.3), tests/golden/make_golden_metrics.py.
"""
import numpy as np
from scipy.ndimage import uniform_filter


def _ssim_uniform(im1, im2, data_range, win_size):
    im1 = np.asarray(im1, dtype=np.float64)
    im2 = np.asarray(im2, dtype=np.float64)
    if np.any(np.asarray(im1.shape) - win_size < 0):
        raise ValueError("win_size exceeds image extent.")
    npix = win_size ** im1.ndim
    cov_norm = npix / (npix - 1)
    ux = uniform_filter(im1, size=win_size)
    uy = uniform_filter(im2, size=win_size)
    uxx = uniform_filter(im1 * im1, size=win_size)
    uyy = uniform_filter(im2 * im2, size=win_size)
    uxy = uniform_filter(im1 * im2, size=win_size)
    vx = cov_norm * (uxx - ux * ux)
    vy = cov_norm * (uyy - uy * uy)
    vxy = cov_norm * (uxy - ux * uy)
    c1 = (0.01 * data_range) ** 2
    c2 = (0.03 * data_range) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux ** 2 + uy ** 2 + c1) * (vx + vy + c2))
    pad = (win_size - 1) // 2
    sl = tuple(slice(pad, n - pad) for n in s.shape)
    return float(np.mean(s[sl], dtype=np.float64))


def compute_ssim_2d(original, compressed):
    """metrics.py:11-32."""
    original = np.asarray(original)
    compressed = np.clip(np.asarray(compressed), 0, None)
    data_range = max(original.max(), compressed.max()) - min(original.min(), compressed.min())
    win = min(7, min(original.shape))
    if win % 2 == 0:
        win -= 1
    return _ssim_uniform(original, compressed, data_range, win)


def ssim_3d_axis(original, compressed, axis=0):
    """metrics.py:35-65."""
    if original.shape != compressed.shape:
        raise ValueError("Shape mismatch between 3D arrays.")
    if axis >= original.ndim or axis < -original.ndim:
        raise ValueError(f"Invalid axis {axis} for 3D SSIM.")
    comp = np.clip(compressed, 0, None)
    return [
        compute_ssim_2d(np.take(original, i, axis=axis), np.take(comp, i, axis=axis))
        for i in range(original.shape[axis])
    ]


def avg_ssim_3d(original, compressed):
    """metrics.py:68-85."""
    if original.shape != compressed.shape:
        raise ValueError("Shape mismatch between 3D volumes.")
    return np.mean([np.mean(ssim_3d_axis(original, compressed, a)) for a in range(3)])


def avg_ssim_4d(original, compressed):
    """metrics.py:88-105."""
    if original.shape != compressed.shape:
        raise ValueError("Shape mismatch between 4D volumes.")
    return np.mean([avg_ssim_3d(original[..., t], compressed[..., t])
                    for t in range(original.shape[-1])])


def compute_ssim_by_dim(a, b):
    """metrics.py:108-129."""
    if a.ndim == 4:
        return avg_ssim_4d(a, b)
    if a.ndim == 3:
        return avg_ssim_3d(a, b)
    if a.ndim == 2:
        return compute_ssim_2d(a, b)
    raise ValueError(f"Unsupported tensor dimension for SSIM: {a.ndim}")


def compute_psnr(original, compressed):
    """metrics.py:132-146."""
    mse = np.mean((original - compressed) ** 2)
    if mse == 0:
        return np.inf
    return 10 * np.log10((np.max(original) ** 2) / mse)


def compute_overlap(mps1, mps2):
    """metrics.py:149-160."""
    return (mps1.mps @ mps2.mps) / (mps1.norm_value * mps2.norm_value)


def synthetic_mri(shape, seed=2025, n_blobs=12, noise=0.01, dtype=np.float32):
    """Seeded synthetic "MRI" volume of SURVEY 8(d): anisotropic Gaussian blobs + one
    ellipsoidal shell + white noise, min-max scaled to [0, 1].  Works for 2-D..4-D
    (4-D = 3-D phantom x smooth temporal modulation along the last axis)."""
    rng = np.random.default_rng(seed)
    shape = tuple(shape)
    sp = shape[:3] if len(shape) == 4 else shape
    axes = [np.linspace(-1.0, 1.0, n, dtype=np.float64) for n in sp]
    vol = np.zeros(sp, dtype=np.float64)
    for _ in range(n_blobs):
        amp = rng.uniform(0.3, 1.0)
        factors = []
        for a in axes:
            c = rng.uniform(-0.6, 0.6)
            w = rng.uniform(0.08, 0.45)
            factors.append(np.exp(-0.5 * ((a - c) / w) ** 2))
        blob = factors[0]
        for f in factors[1:]:
            blob = np.multiply.outer(blob, f)
        vol += amp * blob
    r2 = np.zeros(sp, dtype=np.float64)
    for j, a in enumerate(axes):
        e = (a / rng.uniform(0.75, 0.95)) ** 2
        r2 += e.reshape((1,) * j + (-1,) + (1,) * (len(sp) - 1 - j))
    vol += 0.8 * np.exp(-0.5 * ((np.sqrt(r2) - 1.0) / 0.04) ** 2)
    if len(shape) == 4:
        t = np.linspace(0.0, 1.0, shape[3])
        mod = 1.0 + 0.25 * np.sin(2 * np.pi * (t * rng.uniform(1, 3) + rng.uniform()))
        vol = vol[..., None] * mod
    vol += noise * rng.standard_normal(vol.shape)
    vol -= vol.min()
    vol /= vol.max()
    return vol.astype(dtype)

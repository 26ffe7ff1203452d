"""Golden SSIM / PSNR values from the REFERENCE's own metrics.py
(/root/reference/src/imgcompressionmps/utils/metrics.py:11-146).

Needs scikit-image, which only /opt/conda/bin/python3.9 has in the build container
(skimage 0.18.3; the reference pins 0.24.0 -- same uniform-window / sample-covariance
algorithm for float inputs with explicit data_range).  Build container only:
    /opt/conda/bin/python3.9 tests/golden/make_golden_metrics.py   -> metrics.npz
"""
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference/src")
import skimage  # noqa: E402
from imgcompressionmps.utils.metrics import compute_psnr, compute_ssim_by_dim, ssim_3d_axis  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    rng = np.random.default_rng(2025)
    out = {"skimage_version": np.array(skimage.__version__)}
    for name, shape in [("2d", (40, 56)), ("3d", (12, 20, 16)), ("4d", (9, 10, 8, 3)), ("2d_small", (5, 30))]:
        a = rng.random(shape)
        b = a + 0.05 * rng.standard_normal(shape)
        out[name + "/a"] = a
        out[name + "/b"] = b
        out[name + "/ssim"] = np.float64(compute_ssim_by_dim(a, b))
        out[name + "/psnr"] = np.float64(compute_psnr(a, b))
        if len(shape) == 3:  # the per-slice lists of ssim_3d_axis (metrics.py:35-65), one per axis
            for axis in range(3):
                out[f"{name}/ssim_axis{axis}"] = np.asarray(ssim_3d_axis(a, b, axis), dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "metrics.npz"), **out)
    print("wrote metrics.npz with skimage", skimage.__version__)


if __name__ == "__main__":
    main()

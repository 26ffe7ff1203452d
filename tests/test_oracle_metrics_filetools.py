"""Oracle SSIM/PSNR and quantise helpers against fixtures produced by the reference's
own utils/metrics.py (skimage 0.18.3) and utils/filetools.py (tests/golden/make_*.py)."""
import os

import numpy as np
import pytest

from oracle import filetools as ft
from oracle import metrics as om


def test_ssim_psnr_against_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "metrics.npz"))
    for name in ("2d", "3d", "4d", "2d_small"):
        a, b = g[name + "/a"], g[name + "/b"]
        assert om.compute_ssim_by_dim(a, b) == pytest.approx(float(g[name + "/ssim"]), abs=1e-12)
        assert om.compute_psnr(a, b) == pytest.approx(float(g[name + "/psnr"]), rel=1e-13)
    with pytest.raises(ValueError):
        om.compute_ssim_by_dim(np.zeros(4), np.zeros(4))
    assert om.compute_psnr(a, a) == np.inf
    for axis in range(3):  # the per-slice lists of the reference's ssim_3d_axis (metrics.py:35-65)
        got = om.ssim_3d_axis(g["3d/a"], g["3d/b"], axis)
        assert np.abs(np.array(got) - g[f"3d/ssim_axis{axis}"]).max() <= 1e-12


def test_filetools_against_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "filetools.npz"))
    for name in ("a", "b", "c"):
        x = g[name + "/x"]
        for dt in (np.uint8, np.uint16):
            n = np.dtype(dt).name
            q = ft.scale_to_dtype(x, dt)
            assert q.dtype == dt and np.array_equal(q, g[f"{name}/{n}/q"])
            assert np.array_equal(ft.scale_back(q, x.min(), x.max(), dt), g[f"{name}/{n}/back"])
        # fp32-representable input widened to float64 (the values the GPU path's cores hold)
        wide = g[name + "/x32"].astype(np.float64)
        for dt in (np.uint8, np.uint16):
            n = np.dtype(dt).name
            q = ft.scale_to_dtype(wide, dt)
            assert np.array_equal(q, g[f"{name}/x32/{n}/q"])
            assert np.array_equal(ft.scale_back(q, wide.min(), wide.max(), dt), g[f"{name}/x32/{n}/back"])
    assert [ft.get_num_bits(d) for d in (np.uint8, np.uint16, np.int32, np.float32, np.float64)] == g["bits"].tolist()
    with pytest.raises(ValueError):
        ft.get_num_bits(np.bool_)


def test_synthetic_mri_is_seeded_and_in_range():
    a = om.synthetic_mri((16, 16, 16), seed=2025)
    b = om.synthetic_mri((16, 16, 16), seed=2025)
    assert a.dtype == np.float32 and np.array_equal(a, b)
    assert a.min() == 0.0 and a.max() == 1.0
    assert om.synthetic_mri((8, 8, 8, 4)).shape == (8, 8, 8, 4)
    assert om.synthetic_mri((32, 32)).shape == (32, 32)

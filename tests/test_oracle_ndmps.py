"""The oracle restatement against the reference's own properties
(/root/reference/tests/core/test_ndmps.py:6-79), same seed, shapes and tolerances.
These are the only pins the reference holds at the quimb boundary."""
import copy
import math

import numpy as np
import pytest

from oracle.ndmps_oracle import OracleNDMPS as NDMPS
from oracle import mps as omps


@pytest.fixture(scope="module")
def rng():
    return np.random.default_rng(2025)


@pytest.fixture(scope="module", params=[(512, 680), (8, 512, 680)], ids=lambda s: f"shape={s}")
def tensor(request, rng):
    return rng.random(request.param)


@pytest.fixture(params=["Std", "DCT"])
def mode(request):
    return request.param


_BUILT = {}


@pytest.fixture
def ndmps_obj(tensor, mode):
    # encode once per (shape, mode); every test works on its own deep copy
    key = (tensor.shape, mode)
    if key not in _BUILT:
        _BUILT[key] = NDMPS.from_tensor(tensor, norm=False, mode=mode)
    return copy.deepcopy(_BUILT[key])


def test_roundtrip_exact(ndmps_obj, tensor):
    assert np.allclose(ndmps_obj.to_tensor(), tensor, atol=1e-10)


def test_norm_option(tensor):
    obj = NDMPS.from_tensor(tensor, norm=True)
    assert math.isclose(obj.norm_value, 1.0, rel_tol=1e-12)


def test_compression_reduces_elements(ndmps_obj):
    before = ndmps_obj.number_elements_in_MPS()
    ndmps_obj.compress(cutoff=0.1)
    assert ndmps_obj.number_elements_in_MPS() < before


def test_boundary_and_norm_refresh(ndmps_obj):
    ndmps_obj.mps.arrays[0][:] *= 10
    ndmps_obj.update_boundary_list()
    ndmps_obj.update_norm()
    lo, hi = ndmps_obj.boundary_list[0]
    assert lo <= np.min(ndmps_obj.mps.arrays[0]) and hi >= np.max(ndmps_obj.mps.arrays[0])
    assert math.isclose(ndmps_obj.norm_value ** 2, ndmps_obj.mps @ ndmps_obj.mps, rel_tol=1e-12)


def test_disk_compression_ratio(ndmps_obj):
    ndmps_obj.compress(cutoff=0.4)
    r = ndmps_obj.compression_ratio_on_disk(dtype=np.uint16, replace=False)
    assert 0 < r < 1


def test_continuous_compress_prints(ndmps_obj, capsys):
    ndmps_obj.continuous_compress(cutoff=0.05, print_ratio=True)
    assert capsys.readouterr().out.count("Compression ratio at") == 20


# ---- extra self-consistency of the quimb restatement ------------------------
def test_from_dense_gauge_and_bonds():
    rng = np.random.default_rng(7)
    dims = [4, 3, 5, 2]
    x = rng.standard_normal(dims)
    cores, _ = omps.mps_from_dense(x, dims)
    assert [c.shape for c in cores] == [(1, 4, 4), (4, 3, 10), (10, 5, 2), (2, 2, 1)]
    for c in cores[1:]:  # right-isometric sites (absorb left)
        m = c.reshape(c.shape[0], -1)
        assert np.allclose(m @ m.T, np.eye(c.shape[0]), atol=1e-12)
    assert np.allclose(omps.mps_to_dense(cores), x, atol=1e-12)
    assert math.isclose(omps.mps_overlap(cores, cores), float(np.sum(x * x)), rel_tol=1e-12)


def test_compress_bond_is_truncated_two_site_svd():
    rng = np.random.default_rng(11)
    t1 = rng.standard_normal((3, 4, 6))
    t2 = rng.standard_normal((6, 5, 2))
    two = t1.reshape(12, 6) @ t2.reshape(6, 10)
    a, b, s = omps.compress_bond(t1, t2, cutoff=0.0)
    assert a.shape == (3, 4, 6) and np.allclose(a.reshape(12, -1) @ b.reshape(-1, 10), two)
    a, b, s = omps.compress_bond(t1, t2, cutoff=0.5)
    k = int(np.sum(s > 0.5 * s[0]))
    assert a.shape[2] == k == b.shape[0]
    u, sv, vh = np.linalg.svd(two, full_matrices=False)
    assert np.allclose(a.reshape(12, k) @ b.reshape(k, 10), (u[:, :k] * sv[:k]) @ vh[:k])
    a, b, s = omps.compress_bond(t1, t2, cutoff=0.0, max_bond=2)
    assert a.shape[2] == 2
    a, b, s = omps.compress_bond(t1, t2, cutoff=10.0)  # keeps at least one
    assert a.shape[2] == 1


def test_closed_form_permutation_variant_matches_materialised():
    x = np.random.default_rng(3).random((12, 8, 20, 6))
    a = NDMPS.from_tensor(x, mode="DCT", max_bond=7)
    b = NDMPS.from_tensor(x, mode="DCT", max_bond=7, materialise_map=False)
    assert a.bond_sizes() == b.bond_sizes() == [7]
    assert np.array_equal(a.to_tensor(), b.to_tensor())


def test_single_site_and_unknown_mode():
    x = np.random.default_rng(5).random((7, 12))  # prime dim -> L = 1, no bonds
    o = NDMPS.from_tensor(x)
    assert o.bond_sizes() == [] and np.allclose(o.to_tensor(), x)
    o.compress(0.3)
    assert np.allclose(o.to_tensor(), x)
    assert NDMPS.from_tensor(x, mode="weird").to_tensor() is None

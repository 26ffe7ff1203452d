"""GPU tests of fp64 storage (``NDMPS.from_tensor(x, dtype=torch.float64)``): the reference's own element type
(core/ndmps.py:56) and therefore the reference's own tolerances -- /root/reference/tests/core/test_ndmps.py:35-38
(round trip, atol 1e-10) and :41-44, :63-66 (norms, rel 1e-12) -- on the reference's own shapes and data
(``default_rng(2025).random(shape)``).

Against the oracle (truncated results: "parity unpinned" at the quimb boundary, see tests/test_gpu_parity.py):
  * truncated reconstruction: relative Frobenius <= 1e-9 (singular values come from fp64 Gram matrices, so the kept
    subspace carries ~eps s_0^2 / gap);
  * singular values: 1e-10 relative to s_0 for values above 1e-3 s_0;
  * compress(cutoff): equal bonds, relative Frobenius <= 1e-9.
Kernel checks against NumPy / SciPy fp64: Gram 1e-13, DCT 1e-13, quantisation bit-exact.
"""
import copy
import ctypes as C
import math

import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu

from imgcompressionmps_amd import NDMPS, _lib  # noqa: E402
from imgcompressionmps_amd.utils import filetools as hft  # noqa: E402
from oracle import filetools as oft  # noqa: E402
from oracle import index_map as oim  # noqa: E402
from oracle import mps as omps  # noqa: E402
from oracle.metrics import synthetic_mri  # noqa: E402
from oracle.ndmps_oracle import OracleNDMPS  # noqa: E402

DEV = "cuda:0"
F64 = torch.float64


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a HIP device; the product has no CPU path")
    _lib.load()


def sp():
    return _lib.stream_ptr()


# ------------------------------------------------ the reference's own properties at the reference's own tolerances
REF_SHAPES = [(512, 680), (8, 512, 680)]


@pytest.fixture(scope="module")
def rng():
    return np.random.default_rng(2025)


@pytest.fixture(scope="module", params=REF_SHAPES, ids=lambda s: f"shape={s}")
def tensor(request, rng):
    return rng.random(request.param)


@pytest.fixture(params=["Std", "DCT"])
def mode(request):
    return request.param


_BUILT = {}


@pytest.fixture
def ndmps_obj(tensor, mode):
    key = (tensor.shape, mode)
    if key not in _BUILT:
        _BUILT[key] = NDMPS.from_tensor(tensor, norm=False, mode=mode, dtype=F64)
    return copy.deepcopy(_BUILT[key])


def test_roundtrip_exact_at_the_reference_tolerance(ndmps_obj, tensor):
    """tests/core/test_ndmps.py:35-38, verbatim tolerance."""
    out = ndmps_obj.to_tensor()
    assert out.dtype == np.float64 and out.shape == tensor.shape
    assert np.allclose(out, tensor, atol=1e-10), np.abs(out - tensor).max()


def test_norm_option_at_the_reference_tolerance(tensor):
    """tests/core/test_ndmps.py:41-44."""
    obj = NDMPS.from_tensor(tensor, norm=True, dtype=F64)
    assert math.isclose(obj.norm_value, 1.0, rel_tol=1e-12)
    obj.update_norm()  # the full overlap contraction, as the reference evaluates it
    assert math.isclose(obj.norm_value, 1.0, rel_tol=1e-12)


def test_compression_reduces_elements(ndmps_obj):
    before = ndmps_obj.number_elements_in_MPS()
    ndmps_obj.compress(cutoff=0.1)
    assert ndmps_obj.number_elements_in_MPS() < before
    assert all(c.dtype == F64 for c in ndmps_obj.mps.cores)


def test_boundary_and_norm_refresh_at_the_reference_tolerance(ndmps_obj):
    """tests/core/test_ndmps.py:53-66."""
    ndmps_obj.mps.arrays[0][:] *= 10
    ndmps_obj.update_boundary_list()
    ndmps_obj.update_norm()
    new_min, new_max = ndmps_obj.boundary_list[0]
    assert new_min <= np.min(ndmps_obj.mps.arrays[0]) and new_max >= np.max(ndmps_obj.mps.arrays[0])
    assert math.isclose(ndmps_obj.norm_value ** 2, ndmps_obj.mps @ ndmps_obj.mps, rel_tol=1e-12)
    dense = ndmps_obj.mps.to_dense().cpu().numpy()
    assert dense.dtype == np.float64
    assert math.isclose(ndmps_obj.norm_value ** 2, float(np.sum(dense * dense)), rel_tol=1e-12)


def test_disk_compression_ratio_and_continuous_compress(ndmps_obj, capsys):
    ndmps_obj.continuous_compress(cutoff=0.05, print_ratio=True)
    assert capsys.readouterr().out.count("Compression ratio at") == 20
    ndmps_obj.compress(cutoff=0.4)
    r = ndmps_obj.compression_ratio_on_disk(dtype=np.uint16, replace=False)
    assert 0 < r < 1


# ----------------------------------------------------------------------------------- against the oracle
@pytest.mark.parametrize("shape,chi,mode", [((32, 32), 8, "Std"), ((64, 64, 64), 16, "Std"), ((64, 64, 64), 32, "DCT"),
                                            ((48, 40, 36), 12, "Std"), ((16, 16, 8, 12), 10, "Std"),
                                            ((128, 128, 128), 32, "Std")], ids=str)
def test_truncated_sweep_f64_matches_oracle(shape, chi, mode):
    x = synthetic_mri(shape, seed=2025).astype(np.float64)
    gpu = NDMPS.from_tensor(x, mode=mode, max_bond=chi, dtype=F64)
    ref = OracleNDMPS.from_tensor(x, mode=mode, max_bond=chi)
    assert gpu.bond_sizes() == ref.bond_sizes()
    rg, rr = gpu.to_tensor(), ref.to_tensor()
    assert rg.dtype == np.float64
    rel = np.linalg.norm(rg - rr) / np.linalg.norm(rr)
    assert rel <= 1e-9, rel
    assert math.isclose(gpu.norm_value, ref.norm_value, rel_tol=1e-11)
    if mode == "Std":  # orthogonal projection: truncation error adds in quadrature to the kept norm
        assert math.isclose(np.linalg.norm(x - rg) ** 2 + gpu.norm_value ** 2, np.linalg.norm(x) ** 2, rel_tol=1e-11)


def test_sweep_spectra_f64_match_oracle():
    x = synthetic_mri((64, 64, 64), seed=7).astype(np.float64)
    gpu = NDMPS.from_tensor(x, max_bond=24, dtype=F64)
    dense = np.empty(8 ** 6)
    dense[oim.flat_destination((64, 64, 64)).reshape(-1)] = x.reshape(-1)
    _, spectra = omps.mps_from_dense(dense, [8] * 6, max_bond=24)
    for i in range(1, 6):
        s_ref, s_gpu = spectra[i], gpu.sweep_spectra[i]
        m = min(len(s_ref), len(s_gpu), 24)
        big = s_ref[:m] > 1e-3 * s_ref[0]
        assert np.abs(s_gpu[:m] - s_ref[:m])[big].max() <= 1e-10 * s_ref[0], i


def test_exact_sweep_f64_keeps_what_the_reference_keeps():
    """cutoff=1e-10 as core/ndmps.py:74 passes it (quimb default): the exact sweep of a noisy volume keeps every
    direction (the fp32 path's 1e-6 floor drops a few at the widest bond); low-rank bonds are found like the oracle's."""
    x = synthetic_mri((64, 64, 64), seed=3).astype(np.float64)
    gpu = NDMPS.from_tensor(x, dtype=F64)
    ref = OracleNDMPS.from_tensor(x)
    assert gpu.bond_sizes() == ref.bond_sizes()
    assert np.allclose(gpu.to_tensor(), x, atol=1e-10)
    # an exactly low-rank tensor: rank-3 in every unfolding
    r = np.random.default_rng(5)
    low = sum(np.einsum("i,j,k->ijk", r.standard_normal(32), r.standard_normal(32), r.standard_normal(32)) for _ in range(3))
    g2 = NDMPS.from_tensor(low, dtype=F64)
    o2 = OracleNDMPS.from_tensor(low)
    assert g2.bond_sizes() == o2.bond_sizes()
    assert np.allclose(g2.to_tensor(), low, atol=1e-10 * np.abs(low).max())


@pytest.mark.parametrize("cutoff", [0.02, 0.1, 0.3])
def test_compress_f64_matches_oracle(cutoff):
    x = synthetic_mri((64, 64, 64), seed=11).astype(np.float64)
    gpu = NDMPS.from_tensor(x, max_bond=32, dtype=F64)
    ref = OracleNDMPS.from_tensor(x, max_bond=32)
    gpu.compress(cutoff)
    ref.compress(cutoff)
    assert gpu.bond_sizes() == ref.bond_sizes()
    rg, rr = gpu.to_tensor(), ref.to_tensor()
    assert np.linalg.norm(rg - rr) / np.linalg.norm(rr) <= 1e-9
    assert math.isclose(gpu.norm_value, ref.norm_value, rel_tol=1e-10)
    for (lo, hi), (rlo, rhi) in zip(np.abs(np.asarray(gpu.boundary_list)), np.abs(np.asarray(ref.boundary_list))):
        assert math.isclose(max(lo, hi), max(rlo, rhi), rel_tol=1e-7)  # cores agree up to sign


def test_from_tensors_f64_list_and_device_input():
    xs = [synthetic_mri((32, 32, 32), seed=s).astype(np.float64) for s in (1, 2, 3)]
    objs = NDMPS.from_tensors([torch.from_numpy(x).to(DEV) for x in xs], max_bond=16, dtype=F64)
    for x, o in zip(xs, objs):
        one = NDMPS.from_tensor(x, max_bond=16, dtype=F64)
        assert o.bond_sizes() == one.bond_sizes()
        a, b = o.to_tensor(as_torch=True), one.to_tensor(as_torch=True)
        assert a.dtype == F64 and a.is_cuda
        assert float((a - b).norm() / b.norm()) <= 1e-12
    recs = NDMPS.to_tensors(objs)
    assert all(r.dtype == np.float64 for r in recs)


def test_storage_type_is_validated():
    with pytest.raises(ValueError):
        NDMPS.from_tensor(np.zeros((4, 4)), dtype=torch.float16)


# ------------------------------------------------------------------------------------------ kernel checks
@pytest.mark.parametrize("m,n", [(64, 8), (1000, 8), (4096, 64), (777, 33), (5000, 130), (16, 16), (3, 20), (2048, 512)])
def test_gram_f64(m, n):
    lib = _lib.load()
    a = np.random.default_rng(m + n).standard_normal((m, n))
    d_a = torch.from_numpy(a).to(DEV)
    g = torch.empty((n, n), dtype=F64, device=DEV)
    nbytes = lib.ndmps_gram_f64_workspace_bytes(m, n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    _lib.check(lib.ndmps_gram_f64(d_a.data_ptr(), m, n, n, g.data_ptr(), ws.data_ptr(), nbytes, sp()))
    ref = a.T @ a
    out = g.cpu().numpy()
    assert np.abs(out - ref).max() <= 1e-13 * np.abs(ref).max() * math.sqrt(m)
    assert np.array_equal(out, out.T)


@pytest.mark.parametrize("rows,n", [(4, 8), (37, 680), (100, 50), (3, 1)])
def test_dct_f64_last_axis(rows, n):
    from scipy.fftpack import dct, idct

    lib = _lib.load()
    x = np.random.default_rng(rows * n).standard_normal((rows, n))
    d_x = torch.from_numpy(x).to(DEV)
    basis = torch.empty((n, n), dtype=F64, device=DEV)
    y = torch.empty_like(d_x)
    back = torch.empty_like(d_x)
    _lib.check(lib.ndmps_dct_basis_f64(basis.data_ptr(), n, sp()))
    _lib.check(lib.ndmps_dct_last_f64(d_x.data_ptr(), y.data_ptr(), rows, n, basis.data_ptr(), sp()))
    _lib.check(lib.ndmps_idct_last_f64(y.data_ptr(), back.data_ptr(), rows, n, basis.data_ptr(), sp()))
    ref = dct(x, norm="ortho")
    assert np.abs(y.cpu().numpy() - ref).max() <= 1e-13 * max(1.0, np.abs(ref).max()) * math.sqrt(n)
    assert np.abs(back.cpu().numpy() - idct(ref, norm="ortho")).max() <= 1e-13 * math.sqrt(n) * max(1.0, np.abs(x).max())


def test_reductions_scale_and_quantise_f64():
    lib = _lib.load()
    x = np.random.default_rng(9).standard_normal(100_003) * 3.0
    d_x = torch.from_numpy(x).to(DEV)
    ws = torch.empty(lib.ndmps_reduce_workspace_bytes(), dtype=torch.uint8, device=DEV)
    ss = C.c_double()
    _lib.check(lib.ndmps_sumsq_f64(d_x.data_ptr(), x.size, C.byref(ss), ws.data_ptr(), ws.numel(), sp()))
    assert math.isclose(ss.value, float(np.sum(x * x)), rel_tol=1e-13)
    (lo, hi), = hft.minmax_many([d_x])
    assert lo == x.min() and hi == x.max()
    assert hft.minmax(d_x) == (x.min(), x.max())
    y = d_x.clone()
    _lib.check(lib.ndmps_scale_f64(y.data_ptr(), x.size, 1.0 / 3.0, sp()))
    assert np.array_equal(y.cpu().numpy(), x * (1.0 / 3.0))
    for dtype in (np.uint8, np.uint16):
        q = hft.to_numpy_uint(hft.scale_to_dtype(d_x, dtype), dtype)
        assert np.array_equal(q, oft.scale_to_dtype(x, dtype))  # fp64 in, the reference's arithmetic: bit-exact
        back = hft.scale_back(torch.from_numpy(q.view(np.int16) if dtype == np.uint16 else q).to(DEV), x.min(), x.max(),
                              dtype, out_dtype=F64)
        assert np.array_equal(back.cpu().numpy(), oft.scale_back(q, x.min(), x.max(), dtype))


def test_overlap_f64_and_mixed_pair():
    x = synthetic_mri((32, 32, 32), seed=4).astype(np.float64)
    a = NDMPS.from_tensor(x, max_bond=12, dtype=F64)
    b = NDMPS.from_tensor(x, max_bond=20, dtype=F64)
    da, db = a.mps.to_dense().cpu().numpy(), b.mps.to_dense().cpu().numpy()
    assert math.isclose(a.mps @ b.mps, float(da @ db), rel_tol=1e-12)
    c = NDMPS.from_tensor(x, max_bond=20)  # fp32 cores against fp64 cores: contracted in fp64
    dc = np.asarray(omps.mps_to_dense([t.cpu().numpy().astype(np.float64) for t in c.mps.cores])).reshape(-1)
    assert math.isclose(a.mps @ c.mps, float(da @ dc), rel_tol=1e-12)


def test_container_round_trip_keeps_fp64_cores(tmp_path):
    from imgcompressionmps_amd.core import codec

    x = synthetic_mri((32, 32, 32), seed=6).astype(np.float64)
    obj = NDMPS.from_tensor(x, max_bond=16, dtype=F64)
    path = tmp_path / "vol.ndmps"
    codec.save(obj, path, dtype=np.float64)
    back = codec.load(path)
    assert all(c.dtype == F64 for c in back.mps.cores)
    assert np.array_equal(back.to_tensor(), obj.to_tensor())
    q = codec.loads(codec.dumps(obj, dtype=np.uint16))  # the reference's quantisation, from fp64 cores
    assert np.abs(q.to_tensor() - obj.to_tensor()).max() <= 1e-2 * np.abs(x).max()


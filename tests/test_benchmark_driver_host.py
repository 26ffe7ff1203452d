"""Host logic of the benchmark driver with the arithmetic stubbed out, case by case
after the reference's own tests/evaluation/test_benchmark.py (same stubs: a fake NDMPS whose ratios
change on compress, constant SSIM / PSNR / fidelity).  No GPU, no oracle."""
import json
from pathlib import Path

import numpy as np
import pytest

import imgcompressionmps_amd  # noqa: F401
from imgcompressionmps_amd.core import batch as bm
from imgcompressionmps_amd.core import ndmps as ndmps_mod
from imgcompressionmps_amd.utils import loaders
from imgcompressionmps_amd.utils import metrics as metrics_mod


class _FakeNDMPS:
    def __init__(self, data):
        self._data = np.asarray(data)
        self._compressed = False

    @classmethod
    def from_tensor(cls, tensor, norm=False, mode="DCT", max_bond=None, cutoff=1e-10, device=None):
        return cls(tensor)

    def to_tensor(self, as_torch=False):
        return self._data.copy()

    def compress(self, _factors, max_bond=None):
        self._compressed = True

    def compression_ratio(self):
        return 0.5 if self._compressed else 1.0

    def get_storage_space(self, dtype=np.uint16):
        return int(self._data.size * np.dtype(dtype).itemsize / (2 if self._compressed else 1))

    def get_bytesize_on_disk(self, dtype=np.uint16):
        return self.get_storage_space(dtype) // 2

    def compression_ratio_on_disk(self, dtype=np.uint16, replace=True):
        return 0.25 if self._compressed else 1.0

    def bond_sizes(self):
        return [1, 2, 3]


@pytest.fixture(autouse=True)
def _stub_arithmetic(monkeypatch):
    monkeypatch.setattr(ndmps_mod, "NDMPS", _FakeNDMPS)
    monkeypatch.setattr(metrics_mod, "compute_ssim_by_dim", lambda a, b: 1.0)
    monkeypatch.setattr(metrics_mod, "compute_psnr", lambda a, b: 42.0)
    monkeypatch.setattr(metrics_mod, "compute_overlap", lambda mps, ref: 0.999)
    yield


@pytest.fixture
def sample_array():
    return np.arange(24, dtype=np.uint8).reshape(2, 3, 4)


def test_load_tensors_npz(tmp_path, sample_array):
    f = tmp_path / "a.npz"
    np.savez_compressed(f, sequence=sample_array)
    tensors, bits = loaders.load_tensors([str(f)], ".npz")
    assert np.array_equal(tensors[0], sample_array) and bits == [8]
    tensors, _ = loaders.load_tensors([str(f)], ".npz", shape=(1, 2, 2))
    assert tensors[0].shape == (1, 2, 2)


def test_load_tensors_invalid_suffix_and_empty():
    with pytest.raises(ValueError):
        loaders.load_tensors(["dummy.foo"], ".foo")
    assert loaders.load_tensors([], ".npz") == ([], [])


def test_conv_roundtrip_and_empty(sample_array):
    mps_list = bm.conv_to_mps([sample_array])
    assert np.array_equal(bm.conv_to_tensors(mps_list)[0], sample_array)
    assert bm.conv_to_mps([]) == [] and bm.conv_to_tensors([]) == []


def test_compress_list_changes_state(sample_array):
    mps = bm.conv_to_mps([sample_array])[0]
    assert mps.compression_ratio() == 1.0
    bm.compress_list([mps], 0.5)
    assert mps.compression_ratio() == 0.5
    with pytest.raises(Exception):
        bm.compress_list([mps], None)


@pytest.mark.parametrize("metric", ["compression_ratio", "storage", "gzip_bytes", "gzip_ratio", "ssim", "psnr",
                                    "bond_dims", "shape", "fidelity"])
def test_benchmark_metric_all(metric, sample_array):
    mps = bm.conv_to_mps([sample_array])
    assert len(bm.benchmark_metric(mps, [sample_array], metric=metric)) == 1


def test_benchmark_metric_values_and_errors(sample_array):
    mps = bm.conv_to_mps([sample_array])
    ref = [sample_array]
    assert bm.benchmark_metric(mps, ref, metric="ssim") == [1.0]
    assert bm.benchmark_metric(mps, ref, metric="psnr") == [42.0]
    assert bm.benchmark_metric(mps, ref, metric="fidelity") == [0.999]
    assert bm.benchmark_metric(mps, ref, metric="compression_ratio") == [1.0]
    assert bm.benchmark_metric(mps, ref, metric="shape") == [(2, 3, 4)]
    bm.compress_list(mps, 0.5)
    assert bm.benchmark_metric(mps, ref, metric="compression_ratio") == [0.5]
    with pytest.raises(ValueError):
        bm.benchmark_metric(mps, metric="invalid_metric")
    with pytest.raises(IndexError):
        bm.benchmark_metric(mps, [sample_array, sample_array + 1], metric="ssim")


def test_run_benchmark_shapes(sample_array):
    mps_list = bm.conv_to_mps([sample_array, sample_array + 1])
    res = bm.run_benchmark(mps_list, [sample_array, sample_array + 1], np.array([0.8, 0.5]), verbose=False)
    assert set(res).issuperset({"ssim", "compression_ratio", "psnr", "fidelity", "bond_dims"})
    assert res["ssim"].shape == (2, 3)  # 2 files, 3 columns (incl. no compression)
    assert res["compression_ratio"].tolist() == [[1.0, 0.5, 0.5]] * 2
    assert res["bond_dims"] == [[[1, 2, 3]] * 2] * 3  # left as nested lists, step-major


def test_run_benchmark_empty_lists_and_cutoffs(sample_array):
    res = bm.run_benchmark([], [], np.array([0.5]), verbose=False)
    assert all(v == [] for v in res.values())
    res = bm.run_benchmark(bm.conv_to_mps([sample_array]), [sample_array], [], verbose=False)
    for k in ("ssim", "compression_ratio", "psnr", "fidelity", "bond_dims"):
        assert k in res and isinstance(res[k], (list, np.ndarray))


def test_run_benchmark_unsorted_cutoffs(sample_array, capsys):
    res = bm.run_benchmark(bm.conv_to_mps([sample_array]), [sample_array], np.array([0.9, 0.2, 0.5]))
    assert res["ssim"].shape == (1, 4)
    assert "Status: 100.00% - Cutoff: 0.5" in capsys.readouterr().out


def test_run_full_benchmark_invalid_path(tmp_path):
    with pytest.raises(Exception):
        bm.run_full_benchmark(dataset_path=tmp_path / "nonexistent", cutoff_list=np.array([0.5]),
                              result_file="should_fail.json", datatype="MRI", ending=".npz")


def test_run_full_benchmark_relative_and_nested_paths(tmp_path, monkeypatch, sample_array):
    data_dir = tmp_path / "dataset"
    data_dir.mkdir()
    f = data_dir / "scan.npz"
    np.savez_compressed(f, sequence=sample_array)
    monkeypatch.setattr(loaders, "find_specific_files", lambda _, __: [str(f)])
    monkeypatch.chdir(tmp_path)
    bm.run_full_benchmark(dataset_path=data_dir, cutoff_list=np.array([0.9]), result_file="mri.json", datatype="MRI",
                          ending=".npz")
    assert Path("src/evaluation/results/mri.json").exists()  # relative names land under the results folder
    nested = Path("src/evaluation/results/sub/folder/output.json")
    bm.run_full_benchmark(dataset_path=data_dir, cutoff_list=np.array([0.8]), result_file=str(nested),
                          datatype="MRI_Slice", ending=".npz")
    assert nested.exists()


def test_run_full_benchmark_end_to_end(tmp_path, monkeypatch, sample_array):
    data_dir = tmp_path / "dataset"
    data_dir.mkdir()
    f = data_dir / "scan1.npz"
    np.savez_compressed(f, sequence=sample_array)
    monkeypatch.setattr(loaders, "find_specific_files", lambda root, ending: [str(f)])
    monkeypatch.chdir(tmp_path)
    bm.run_full_benchmark(dataset_path=str(data_dir), cutoff_list=np.array([0.7]), result_file="result.json",
                          datatype="MRI_Slice", mode="DCT", ending=".npz")
    data = json.loads(Path("src/evaluation/results/result.json").read_text())
    assert data["datatype"] == "MRI_Slice" and data["mode"] == "DCT" and data["cutoff_list"] == [0.7]
    assert data["files"] == [str(f)] and data["bitsize_list"] == [8, 8, 8]  # three central slices
    assert data["shapes"] == [[3, 4], [2, 4], [2, 3]]
    for key in ("ssim", "compression_ratio", "psnr", "fidelity", "bond_dims", "storage", "gzip_bytes", "gzip_ratio"):
        assert key in data
    assert np.array(data["ssim"]).shape == (3, 2)

"""Host-side checks of the 8f#4 row: volume loaders and the NDMPS container framing (no GPU).

The reference loads ``.nii.gz`` with nibabel (evaluation/benchmark.py:39-42), which is not
installed and ships no NIfTI fixture: the reader is checked against files this test writes field by
field from the NIfTI-1 layout (parity with nibabel itself: unpinned)."""
import gzip
import struct

import numpy as np
import pytest

import imgcompressionmps_amd  # noqa: F401
from imgcompressionmps_amd.core import codec
from imgcompressionmps_amd.utils import loaders


def _write_nifti(path, arr, code, slope=0.0, inter=0.0, endian="<", vox_offset=352.0):
    hdr = bytearray(348)
    struct.pack_into(endian + "i", hdr, 0, 348)
    dims = [arr.ndim] + list(arr.shape) + [1] * (7 - arr.ndim)
    struct.pack_into(endian + "8h", hdr, 40, *dims)
    struct.pack_into(endian + "h", hdr, 70, code)
    struct.pack_into(endian + "h", hdr, 72, arr.dtype.itemsize * 8)
    struct.pack_into(endian + "3f", hdr, 108, vox_offset, slope, inter)
    hdr[344:348] = b"n+1\x00"
    body = arr.astype(arr.dtype.newbyteorder(endian)).tobytes(order="F")
    blob = bytes(hdr) + b"\x00" * (int(vox_offset) - 348) + body
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "wb") as f:
        f.write(blob)


@pytest.mark.parametrize("endian", ["<", ">"])
def test_read_nifti_int16_scaled(tmp_path, endian):
    rng = np.random.default_rng(3)
    arr = rng.integers(-500, 3000, size=(5, 7, 3), dtype=np.int16)
    p = tmp_path / "vol.nii.gz"
    _write_nifti(p, arr, 4, slope=0.5, inter=-10.0, endian=endian)
    data, dtype = loaders.read_nifti(p)
    assert dtype == np.int16 and data.dtype == np.float64 and data.shape == (5, 7, 3)
    assert np.array_equal(data, arr.astype(np.float64) * 0.5 - 10.0)


def test_read_nifti_unscaled_float_and_4d(tmp_path):
    arr = np.random.default_rng(4).random((4, 3, 2, 6)).astype(np.float32)
    p = tmp_path / "f.nii"
    _write_nifti(p, arr, 16, slope=0.0, inter=0.0, vox_offset=400.0)  # slope 0: no scaling; extended offset
    data, dtype = loaders.read_nifti(p)
    assert dtype == np.float32 and np.array_equal(data, arr.astype(np.float64))
    _write_nifti(p, arr, 16, slope=1.0, inter=0.0)
    assert np.array_equal(loaders.read_nifti(p)[0], arr.astype(np.float64))


def test_read_nifti_rejects_garbage(tmp_path):
    p = tmp_path / "x.nii"
    p.write_bytes(b"\x00" * 400)
    with pytest.raises(ValueError):
        loaders.read_nifti(p)
    arr = np.zeros((2, 2), dtype=np.uint8)
    _write_nifti(p, arr, 2)
    blob = bytearray(p.read_bytes())
    blob[344:348] = b"ni1\x00"  # header/image pair: not supported
    p.write_bytes(bytes(blob))
    with pytest.raises(ValueError):
        loaders.read_nifti(p)
    _write_nifti(p, arr, 128)  # RGB24
    with pytest.raises(ValueError):
        loaders.read_nifti(p)
    _write_nifti(p, np.zeros((4, 4), dtype=np.uint8), 2)
    p.write_bytes(p.read_bytes()[:-3])
    with pytest.raises(ValueError):
        loaders.read_nifti(p)


def test_load_tensors_conventions(tmp_path, capsys):
    """benchmark.py:16-55: ending filter, 'sequence' key, (B, H, W) crop, bit sizes of the on-disk dtype."""
    seq = np.arange(4 * 5 * 6, dtype=np.uint8).reshape(4, 5, 6)
    np.savez(tmp_path / "a.npz", sequence=seq)
    vol = np.arange(3 * 4 * 5, dtype=np.int16).reshape(3, 4, 5)
    _write_nifti(tmp_path / "b.nii.gz", vol, 4)
    data, bits = loaders.load_tensors([str(tmp_path / "a.npz")], ".npz", shape=(2, 3, 4))
    assert bits == [8] and np.array_equal(data[0], seq[:2, :3, :4])
    data, bits = loaders.load_tensors([str(tmp_path / "b.nii.gz")], ".gz")
    assert bits == [16] and data[0].dtype == np.float64 and np.array_equal(data[0], vol)
    assert "Loading file 1/1" in capsys.readouterr().out
    with pytest.raises(ValueError):
        loaders.load_tensors([], ".nii")
    assert loaders.get_shapes(data) == [(3, 4, 5)]


def test_mri_to_slices_and_find_files(tmp_path, capsys):
    v = np.arange(4 * 6 * 8).reshape(4, 6, 8)
    slices, bits = loaders.mri_to_slices([v, np.zeros((3, 3))], [12, 8])
    assert "Skipping non-3D volume at index 1" in capsys.readouterr().out
    assert bits == [12, 12, 12]
    assert np.array_equal(slices[0], v[2]) and np.array_equal(slices[1], v[:, 3]) and np.array_equal(slices[2], v[:, :, 4])
    assert loaders.mri_to_slices([v])[1] == [16, 16, 16]
    (tmp_path / "sub").mkdir()
    (tmp_path / "sub" / "x.nii.gz").write_bytes(b"")
    (tmp_path / "y.npz").write_bytes(b"")
    assert sorted(loaders.find_specific_files(tmp_path, ".gz")) == [str(tmp_path / "sub" / "x.nii.gz")]
    assert len(loaders.find_specific_files(tmp_path)) == 2


def test_container_framing_errors():
    with pytest.raises(ValueError):
        codec.loads(b"nope")
    bad_version = codec.MAGIC + struct.pack("<II", 99, 2) + b"{}"
    with pytest.raises(ValueError):
        codec.loads(bad_version)
    truncated = codec.MAGIC + struct.pack("<II", codec.VERSION, 500) + b"{}"
    with pytest.raises(ValueError):
        codec.loads(truncated)

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


def test_load_volume_containers(tmp_path):
    """The two containers of the reference's datasets (benchmark.py:40-48): 'sequence' entry of an .npz,
    NIfTI through read_nifti; bit size of the on-disk dtype."""
    seq = np.arange(4 * 5 * 6, dtype=np.uint8).reshape(4, 5, 6)
    np.savez(tmp_path / "a.npz", sequence=seq)
    vol = np.arange(3 * 4 * 5, dtype=np.int16).reshape(3, 4, 5)
    _write_nifti(tmp_path / "b.nii.gz", vol, 4)
    data, bits = loaders.load_volume(tmp_path / "a.npz")
    assert bits == 8 and np.array_equal(data, seq)
    data, bits = loaders.load_volume(str(tmp_path / "b.nii.gz"))
    assert bits == 16 and data.dtype == np.float64 and np.array_equal(data, vol)
    with pytest.raises(ValueError):
        loaders.load_volume(tmp_path / "c.txt")


def test_container_framing_errors():
    with pytest.raises(ValueError):
        codec.loads(b"nope")
    bad_version = codec.MAGIC + struct.pack("<II", 99, 2) + b"{}"
    with pytest.raises(ValueError):
        codec.loads(bad_version)
    truncated = codec.MAGIC + struct.pack("<II", codec.VERSION, 500) + b"{}"
    with pytest.raises(ValueError):
        codec.loads(truncated)


def _container(header: dict, payload: bytes = b"") -> bytes:
    import json

    hb = json.dumps(header).encode()
    return codec.MAGIC + struct.pack("<II", codec.VERSION, len(hb)) + hb + payload


def test_container_header_is_validated_on_the_host():
    """A crafted header must not reach the device: bonds larger than the rank of their unfolding would make
    the chain contraction write past its N-element buffers (dims [2,2,2] with bonds [1,8,8,1])."""
    good = {"version": 1, "shape": [2, 2, 2], "mode": "Std", "norm": False, "norm_value": 1.0, "dim": 3,
            "qubit_size": [2, 2, 2], "bonds": [1, 2, 2, 1], "dtype": "float32", "bounds": [[0, 0]] * 3,
            "member_bytes": [1, 1, 1]}
    for patch in ({"bonds": [1, 8, 8, 1]}, {"bonds": [2, 2, 2, 1]}, {"bonds": [1, 2, 2]}, {"bonds": [1, 0, 2, 1]},
                  {"bonds": [1, 2.5, 2, 1]}, {"qubit_size": [2, -2, 2]}, {"member_bytes": [1, 1]},
                  {"dtype": "int32"}, {"shape": "abc"}):
        with pytest.raises(ValueError):
            codec.loads(_container({**good, **patch}))
    bad = dict(good)
    del bad["bonds"]
    with pytest.raises(ValueError):
        codec.loads(_container(bad))
    # the fields that are only read after the payload: a missing or mistyped one is a ValueError too, never a
    # KeyError / TypeError from half-way through the load
    for key in ("mode", "norm", "dim"):
        bad = dict(good)
        del bad[key]
        with pytest.raises(ValueError):
            codec.loads(_container(bad))
    for patch in ({"mode": 7}, {"norm": "yes"}, {"dim": 2}, {"dim": True}, {"norm_value": "1.0"}):
        with pytest.raises(ValueError):
            codec.loads(_container({**good, **patch}))
    quantised = {**good, "dtype": "uint16"}
    for patch in ({"bounds": None}, {"bounds": [[0, 1]] * 2}, {"bounds": [[0, 1], [0], [0, 1]]}, {"bounds": [[0, "a"]] * 3}):
        with pytest.raises(ValueError):
            codec.loads(_container({**quantised, **patch}))
    bad = dict(quantised)
    del bad["bounds"]
    with pytest.raises(ValueError):
        codec.loads(_container(bad))


def test_gzip_members_are_inflated_with_a_length_cap():
    bomb = gzip.compress(b"\0" * (1 << 24))  # 16 MiB of zeros in ~16 KB
    with pytest.raises(ValueError):
        codec._gunzip_exactly(bomb, 64, 0)
    with pytest.raises(ValueError):
        codec._gunzip_exactly(gzip.compress(b"abc"), 64, 0)
    with pytest.raises(ValueError):
        codec._gunzip_exactly(b"not gzip", 8, 0)
    assert codec._gunzip_exactly(gzip.compress(b"12345678"), 8, 0) == b"12345678"


def test_chain_contraction_rejects_bonds_no_unfolding_can_have():
    """ndmps_chain_contract_f32 ping-pongs its intermediates through an N-element buffer; the argument check
    runs before any HIP call, so it can be exercised without a GPU."""
    import ctypes as C

    from imgcompressionmps_amd import _lib

    lib = _lib.load()
    dims, fake = _lib.i64_array([2, 2, 2]), (C.c_void_p * 3)(64, 64, 64)
    for bonds in ([1, 8, 8, 1], [1, 2, 8, 1], [1, 0, 2, 1], [2, 2, 2, 1]):
        rc = lib.ndmps_chain_contract_f32(3, dims, _lib.i64_array(bonds), fake, C.c_void_p(64), C.c_void_p(64),
                                          1 << 20, None)
        assert rc == _lib.EINVAL, bonds

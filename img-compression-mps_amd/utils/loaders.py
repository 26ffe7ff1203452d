"""Volume file readers, host side.

The reference reads ``.nii.gz`` through nibabel and ``.npz`` through NumPy
(src/imgcompressionmps/evaluation/benchmark.py:40-48); nibabel is not available here and the
reference's dataset plumbing around it (file discovery, slicing, cropping) is out of scope.
``read_nifti`` is a self-contained NIfTI-1 single-file reader written from the format
definition (348-byte header, column-major voxel data at ``vox_offset``, ``scl_slope`` / ``scl_inter``
scaling) that returns what ``nib.load(p).get_fdata()`` / ``.header.get_data_dtype()`` return: the
scaled float64 array and the on-disk dtype.  Parity with nibabel itself is unpinned (no nibabel, no
NIfTI fixture in the reference); tests check the reader against files written field by field.
"""
import gzip
import struct

import numpy as np

from .filetools import get_num_bits

_NIFTI_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8,
                 512: np.uint16, 768: np.uint32, 1024: np.int64, 1280: np.uint64}


def read_nifti(path):
    """(float64 array scaled like nibabel's get_fdata, on-disk numpy dtype) of a .nii / .nii.gz file."""
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rb") as f:
        raw = f.read()
    if len(raw) < 348:
        raise ValueError(f"{path}: shorter than a NIfTI-1 header")
    for endian in ("<", ">"):
        if struct.unpack(endian + "i", raw[0:4])[0] == 348:
            break
    else:
        raise ValueError(f"{path}: not a NIfTI-1 file (sizeof_hdr != 348)")
    magic = raw[344:348]
    if magic[:3] != b"n+1":
        raise ValueError(f"{path}: only single-file NIfTI-1 ('n+1') is supported, magic {magic!r}")
    dim = struct.unpack(endian + "8h", raw[40:56])
    ndim = dim[0]
    if not 1 <= ndim <= 7:
        raise ValueError(f"{path}: bad dim[0] = {ndim}")
    shape = tuple(int(d) for d in dim[1:1 + ndim])
    datatype = struct.unpack(endian + "h", raw[70:72])[0]
    if datatype not in _NIFTI_DTYPES:
        raise ValueError(f"{path}: unsupported NIfTI datatype code {datatype}")
    dtype = np.dtype(_NIFTI_DTYPES[datatype])
    vox_offset, slope, inter = struct.unpack(endian + "3f", raw[108:120])
    start = int(vox_offset) if vox_offset >= 352 else 352
    count = int(np.prod(shape, dtype=np.int64))
    if start + count * dtype.itemsize > len(raw):
        raise ValueError(f"{path}: truncated voxel data")
    data = np.frombuffer(raw, dtype=dtype.newbyteorder(endian), count=count, offset=start)
    data = data.reshape(shape, order="F").astype(np.float64)
    if np.isfinite(slope) and slope != 0.0 and not (slope == 1.0 and (inter == 0.0 or not np.isfinite(inter))):
        data = data * float(slope) + (float(inter) if np.isfinite(inter) else 0.0)
    return data, dtype


def load_volume(path):
    """(array, bits of the on-disk dtype) of one volume file: ``.nii`` / ``.nii.gz`` through
    ``read_nifti``, ``.npz`` through its ``"sequence"`` entry (the two containers the reference's
    datasets come in, evaluation/benchmark.py:40-48)."""
    name = str(path)
    if name.endswith((".nii", ".nii.gz")):
        data, dtype = read_nifti(path)
    elif name.endswith(".npz"):
        with np.load(path) as archive:  # allow_pickle stays False
            data = archive["sequence"]
        dtype = data.dtype
    else:
        raise ValueError(f"unsupported volume file: {name}")
    return data, get_num_bits(dtype)

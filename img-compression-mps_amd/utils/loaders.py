"""Volume loaders and dataset helpers of the benchmark driver, host side.

Mirrors ``load_tensors`` (src/imgcompressionmps/evaluation/benchmark.py:16-55) and the helpers it
is used with (utils/filetools.py:42-68 ``mri_to_slices``, :83-92 ``find_specific_files``,
:119-123 ``get_shapes``).  The reference reads ``.nii.gz`` through nibabel, which is not available
here; ``read_nifti`` is a self-contained NIfTI-1 single-file reader written from the format
definition (348-byte header, column-major voxel data at ``vox_offset``, ``scl_slope`` / ``scl_inter``
scaling) that returns what ``nib.load(p).get_fdata()`` / ``.header.get_data_dtype()`` return: the
scaled float64 array and the on-disk dtype.  Parity with nibabel itself is unpinned (no nibabel, no
NIfTI fixture in the reference); tests check the reader against files written field by field.
"""
import gzip
import os
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


def load_tensors(files, ending, shape=None):
    """benchmark.py:16-55: list of arrays and of the bit sizes of their on-disk dtypes; ``shape`` =
    (B, H, W) crops the leading three axes."""
    if not (ending.endswith(".gz") or ending.endswith(".npz")):
        raise ValueError(f"Unsupported file extension: {ending}")
    B, H, W = shape if shape else (None, None, None)
    data_list, bitsize_list = [], []
    for i, path in enumerate(files):
        print(f"Loading file {i + 1}/{len(files)}")
        if ending.endswith(".gz"):
            data, dtype = read_nifti(path)
        else:
            with np.load(path) as archive:  # allow_pickle stays False
                data = archive["sequence"]
                dtype = data.dtype
        if shape:
            data = data[:B, :H, :W]
        data_list.append(data)
        bitsize_list.append(get_num_bits(dtype))
    return data_list, bitsize_list


def mri_to_slices(data_list, bitsize_list=None):
    """filetools.py:42-68: the three central 2-D slices of every 3-D volume (non-3-D entries are skipped)."""
    slices, bits = [], []
    for i, volume in enumerate(data_list):
        if volume.ndim != 3:
            print(f"Skipping non-3D volume at index {i} with shape {volume.shape}")
            continue
        slices.extend([volume[volume.shape[0] // 2, :, :], volume[:, volume.shape[1] // 2, :],
                       volume[:, :, volume.shape[2] // 2]])
        bits.extend([bitsize_list[i] if bitsize_list else 16] * 3)
    return slices, bits


def find_specific_files(directory_path, file_extension=None):
    """filetools.py:83-92: recursive listing in os.walk order, optionally filtered by suffix."""
    files = []
    for root, _, filenames in os.walk(directory_path):
        for filename in filenames:
            if file_extension is None or filename.endswith(file_extension):
                files.append(os.path.join(root, filename))
    return files


def get_shapes(data_list):
    return [np.shape(data) for data in data_list]

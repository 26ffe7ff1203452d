"""Serialised form of an NDMPS: a byte string / file that holds everything
``to_tensor`` needs, so a compressed volume can leave the process.

The reference never persists an MPS; it only measures what one *would* take on disk by
gzip-compressing every quantised core in memory (core/ndmps.py:209-234, one gzip member per core,
default level) and by counting ``number_elements * bits`` (:259-277).  The container below stores
exactly those gzip members, in site order, so its payload size IS ``get_bytesize_on_disk(dtype)``:

    magic "NDMPS\\x01\\0\\0" | u32 version | u32 header bytes | header (JSON, utf-8) | members

Header: shape, mode, norm flag, norm_value, dim, qubit_size (site dims), bonds, storage dtype
("uint8" / "uint16" = the reference's ``scale_to_dtype`` min-max quantisation, filetools.py:20-39,
truncating cast; "float32" / "float64" = lossless cores), per-core (min, max) used by ``scale_back``, per-member
byte counts.  The factor lists are a pure function of ``shape`` (utils/core.py:79-126) and are rebuilt on
load.  Quantisation and de-quantisation run on the device (csrc/reduce.hip); gzip runs on the host, as
in the reference.
"""
from __future__ import annotations

import gzip
import io
import json
import struct

import numpy as np

from ..utils import filetools as _ft

MAGIC = b"NDMPS\x01\x00\x00"
VERSION = 1
_DTYPES = {"uint8": np.uint8, "uint16": np.uint16, "float32": np.float32, "float64": np.float64}


def _gzip_member(raw: bytes) -> bytes:
    buf = io.BytesIO()
    with gzip.GzipFile(fileobj=buf, mode="wb", mtime=0) as gz:  # mtime fixed: byte-reproducible files
        gz.write(raw)
    return buf.getvalue()


def dumps(obj, dtype=np.uint16) -> bytes:
    """Serialise ``obj`` (an NDMPS made by ``from_tensor``) with cores stored as ``dtype``
    (np.uint8 / np.uint16: the reference's quantisation; np.float32 / np.float64: exact cores of fp32 / fp64
    storage)."""
    if obj._shape is None:
        raise ValueError("this NDMPS was not created by from_tensor; the tensor shape is unknown")
    name = np.dtype(dtype).name
    if name not in _DTYPES:
        raise ValueError(f"Unsupported dtype {dtype!r}: cores are stored as uint8, uint16, float32 or float64")
    cores = obj.mps.cores
    members, bounds = [], []
    if name in ("float32", "float64"):
        for c in cores:
            host = c.double() if name == "float64" else c.float()
            members.append(_gzip_member(host.cpu().numpy().tobytes()))
            bounds.append([0.0, 0.0])
    else:
        # the (min, max) scale_to_dtype derives from the data is what scale_back must be given
        mm = _ft.minmax_many(cores)
        for c, (lo, hi) in zip(cores, mm):
            q = _ft.scale_to_dtype(c, _DTYPES[name])
            members.append(_gzip_member(_ft.to_numpy_uint(q, _DTYPES[name]).tobytes()))
            bounds.append([lo, hi])
    header = {
        "version": VERSION,
        "shape": [int(s) for s in obj._shape],
        "mode": obj.mode,
        "norm": bool(obj.norm),
        "norm_value": None if obj.norm_value is None else float(obj.norm_value),
        "dim": int(obj.dim),
        "qubit_size": [int(q) for q in obj.qubit_size],
        "bonds": [int(b) for b in obj.mps.bonds],
        "dtype": name,
        "bounds": bounds,
        "member_bytes": [len(m) for m in members],
    }
    hb = json.dumps(header, separators=(",", ":")).encode("utf-8")
    return b"".join([MAGIC, struct.pack("<II", VERSION, len(hb)), hb] + members)


def payload_bytes(data: bytes) -> int:
    """Size of the gzip members of a serialised NDMPS (= ``get_bytesize_on_disk`` for uint dtypes)."""
    return sum(_header(data)[0]["member_bytes"])


def _header(data: bytes):
    if len(data) < 16 or data[:8] != MAGIC:
        raise ValueError("not an NDMPS container (bad magic)")
    version, hlen = struct.unpack("<II", data[8:16])
    if version != VERSION:
        raise ValueError(f"unsupported NDMPS container version {version}")
    if 16 + hlen > len(data):
        raise ValueError("truncated NDMPS container (header)")
    header = json.loads(data[16:16 + hlen].decode("utf-8"))
    return header, 16 + hlen


def _gunzip_exactly(member: bytes, want: int, index: int) -> bytes:
    """Inflate one gzip member that must hold exactly ``want`` bytes; never inflates more than that
    (+1 to notice an oversized member), so a crafted container cannot balloon in memory."""
    import zlib

    inflater = zlib.decompressobj(16 + zlib.MAX_WBITS)
    try:
        raw = inflater.decompress(member, want + 1)
    except zlib.error as exc:
        raise ValueError(f"core {index}: corrupt gzip member ({exc})") from None
    if len(raw) != want or inflater.unconsumed_tail or not inflater.eof:
        raise ValueError(f"core {index}: gzip member does not hold exactly {want} bytes")
    return raw


def _checked_layout(header):
    """(shape, dtype name, bonds, site dims) of a container header, or ValueError: everything the
    device side will size buffers from is validated here, on the host, before any allocation."""
    try:
        shape = tuple(header["shape"])
        name = header["dtype"]
        bonds, dims = list(header["bonds"]), list(header["qubit_size"])
        members = list(header["member_bytes"])
    except (KeyError, TypeError):
        raise ValueError("inconsistent NDMPS container header") from None
    if name not in _DTYPES:
        raise ValueError(f"unsupported core dtype {name!r}")
    if len(bonds) != len(dims) + 1 or len(members) != len(dims) or not dims:
        raise ValueError("inconsistent NDMPS container header")
    ints = [*shape, *bonds, *dims, *members]
    if any(not isinstance(v, int) or isinstance(v, bool) or v < 1 for v in ints):
        raise ValueError("NDMPS container header: shape, site dimensions, bonds and sizes must be positive integers")
    numel = 1
    for d in dims:
        numel *= d
    if bonds[0] != 1 or bonds[-1] != 1:
        raise ValueError("NDMPS container header: open boundary bonds must be 1")
    # the remaining fields are read after the payload has been inflated: validated here as well, so that a crafted
    # container fails with ValueError before anything is allocated
    if not isinstance(header.get("mode"), str):  # any string, as in the reference (ndmps.py:62, :150-153)
        raise ValueError("NDMPS container header: mode must be a string")
    if not isinstance(header.get("norm"), bool):
        raise ValueError("NDMPS container header: norm must be a boolean")
    if not isinstance(header.get("dim"), int) or isinstance(header.get("dim"), bool) or header["dim"] != len(shape):
        raise ValueError("NDMPS container header: dim must equal the number of tensor axes")
    nv = header.get("norm_value", None)
    if nv is not None and (isinstance(nv, bool) or not isinstance(nv, (int, float))):
        raise ValueError("NDMPS container header: norm_value must be a number or null")
    if name not in ("float32", "float64"):
        bounds = header.get("bounds")
        ok = isinstance(bounds, list) and len(bounds) == len(dims) and all(
            isinstance(b, list) and len(b) == 2 and all(isinstance(x, (int, float)) and not isinstance(x, bool) for x in b)
            for b in bounds)
        if not ok:
            raise ValueError("NDMPS container header: quantised cores need one [min, max] pair per core")
    left = 1
    for i, d in enumerate(dims[:-1]):
        left *= d
        if bonds[i + 1] > min(left, numel // left):
            raise ValueError(f"NDMPS container header: bond {i + 1} = {bonds[i + 1]} exceeds the rank of its unfolding")
    return shape, name, bonds, dims


def loads(data: bytes, device=None):
    """Rebuild the NDMPS on ``device`` (default: the current HIP device).  Quantised cores are
    scaled back exactly like ``compress_to_dtype(replace=True)`` leaves them (filetools.py:29-39)."""
    import torch

    from .. import _lib
    from .mps import DeviceMPS
    from .ndmps import NDMPS, _plan_for

    header, off = _header(data)
    shape, name, bonds, dims = _checked_layout(header)
    _lib.require_device()
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    with torch.cuda.device(device):
        plan = _plan_for(shape, device.index or 0)
        if [int(q) for q in plan.qubit_size] != [int(d) for d in dims]:
            raise ValueError("site dimensions in the container do not match the factorisation of its shape")
        cores = []
        for i, nbytes in enumerate(header["member_bytes"]):
            if off + nbytes > len(data):
                raise ValueError("truncated NDMPS container (payload)")
            cshape = (bonds[i], dims[i], bonds[i + 1])
            want = int(np.prod(cshape)) * np.dtype(_DTYPES[name]).itemsize
            raw = _gunzip_exactly(data[off:off + nbytes], want, i)
            off += nbytes
            arr = np.frombuffer(raw, dtype=_DTYPES[name]).reshape(cshape)
            if name in ("float32", "float64"):
                cores.append(torch.from_numpy(arr.copy()).to(device))
            else:
                # torch has no uint16: same bytes as int16 storage (what scale_back reads)
                host = arr.view(np.int16) if name == "uint16" else arr
                q = torch.from_numpy(host.copy()).to(device)
                lo, hi = header["bounds"][i]
                cores.append(_ft.scale_back(q, lo, hi, _DTYPES[name]))
        obj = NDMPS(DeviceMPS(cores), plan.qubit_size.copy(), None, [[0.0, 0.0]] * len(cores), header["norm"],
                    header["norm_value"], header["mode"], header["dim"])
        obj._shape = shape
        obj.update_boundary_list()
    return obj


def save(obj, path, dtype=np.uint16) -> int:
    """Write the container to ``path``; returns the file size in bytes."""
    data = dumps(obj, dtype)
    with open(path, "wb") as f:
        f.write(data)
    return len(data)


def load(path, device=None):
    with open(path, "rb") as f:
        return loads(f.read(), device)

"""Quantise helpers of the storage estimate, device side.

Mirrors src/imgcompressionmps/utils/filetools.py:7-39 (``get_num_bits``,
``scale_to_dtype``, ``scale_back``) for fp32 cores resident in HBM; the arithmetic runs in
csrc/reduce.hip (ndmps_quantize_f32 / ndmps_dequantize_f32 / ndmps_minmax_f32) with the
reference's truncating cast and min-max scaling.
"""
import ctypes as C

import numpy as np

from .. import _lib


def get_num_bits(dtype) -> int:
    dtype = np.dtype(dtype)
    if np.issubdtype(dtype, np.integer):
        return np.iinfo(dtype).bits
    if np.issubdtype(dtype, np.floating):
        return np.finfo(dtype).bits
    raise ValueError(f"Unsupported dtype {dtype!r}")


def _bits(dtype) -> int:
    dtype = np.dtype(dtype)
    if dtype == np.uint8:
        return 8
    if dtype == np.uint16:
        return 16
    raise ValueError(f"Unsupported dtype {dtype!r}: the device path quantises to uint8 or uint16")


def _f32(t):
    """fp32 view of a core for the fp32 reduction / quantisation kernels (bf16 cores are a few MB at most:
    the upcast copy is plumbing, not a data path).  fp64 cores stay fp64: they have kernels of their own."""
    import torch

    return t if t.dtype in (torch.float32, torch.float64) else t.to(torch.float32)


def minmax(t):
    """(min, max) of a device tensor as Python floats."""
    import torch

    lib = _lib.load()
    t = _f32(t)
    if t.dtype == torch.float64:
        return minmax_many([t])[0]
    ws = torch.empty(lib.ndmps_reduce_workspace_bytes(), dtype=torch.uint8, device=t.device)
    lo, hi = C.c_float(), C.c_float()
    t = t.contiguous()
    _lib.check(lib.ndmps_minmax_f32(t.data_ptr(), t.numel(), C.byref(lo), C.byref(hi), ws.data_ptr(),
                                    ws.numel(), _lib.stream_ptr()))
    return float(lo.value), float(hi.value)


def minmax_many(tensors, with_sumsq=False):
    """[(min, max)] of several device fp32 tensors: one launch, one synchronisation.
    ``with_sumsq=True`` also returns the fp64 sums of squares."""
    import torch

    lib = _lib.load()
    tensors = list(tensors)
    if tensors and all(t.dtype == torch.float64 for t in tensors):
        ts = [t if t.is_contiguous() else t.contiguous() for t in tensors]
        count = len(ts)
        nbytes = lib.ndmps_minmax_many_workspace_bytes(count)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=ts[0].device)
        ptrs = (C.c_void_p * count)(*[t.data_ptr() for t in ts])
        lens = _lib.i64_array([t.numel() for t in ts])
        out = (C.c_double * (2 * count))()
        ss = (C.c_double * count)()
        _lib.check(lib.ndmps_minmax_many_f64(count, ptrs, lens, out, ss, ws.data_ptr(), nbytes, _lib.stream_ptr()))
        mm = [(float(out[2 * i]), float(out[2 * i + 1])) for i in range(count)]
        return (mm, [float(v) for v in ss]) if with_sumsq else mm
    ts = []
    for t in tensors:  # fp32 contiguous tensors (the usual case) pass through without a call
        if t.dtype is torch.float32 and t.is_contiguous():
            ts.append(t)
        else:
            u = t.to(torch.float32)  # mixed lists and bf16 cores: fp32 copies
            ts.append(u if u.is_contiguous() else u.contiguous())
    count = len(ts)
    if count == 0:
        return ([], []) if with_sumsq else []
    nbytes = lib.ndmps_minmax_many_workspace_bytes(count)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=ts[0].device)
    ptrs = (C.c_void_p * count)(*[t.data_ptr() for t in ts])
    lens = _lib.i64_array([t.numel() for t in ts])
    out = (C.c_float * (2 * count))()
    ss = (C.c_double * count)()
    _lib.check(lib.ndmps_minmax_many_f32(count, ptrs, lens, out, ss, ws.data_ptr(), nbytes, _lib.stream_ptr()))
    mm = [(float(out[2 * i]), float(out[2 * i + 1])) for i in range(count)]
    return (mm, [float(v) for v in ss]) if with_sumsq else mm


def scale_to_dtype(t, dtype=np.uint8):
    """Device tensor -> unsigned-int device tensor, (x - min) / max(x - min) * iinfo.max, truncated."""
    import torch

    bits = _bits(dtype)
    lib = _lib.load()
    t = _f32(t).contiguous()
    lo, hi = minmax(t)
    # torch has no uint16 arithmetic, but int16 storage has the same bytes
    q = torch.empty(t.shape, dtype=torch.uint8 if bits == 8 else torch.int16, device=t.device)
    fn = lib.ndmps_quantize_f64 if t.dtype == torch.float64 else lib.ndmps_quantize_f32
    _lib.check(fn(t.data_ptr(), t.numel(), lo, hi, bits, q.data_ptr(), _lib.stream_ptr()))
    return q


def scale_back(q, arr_min, arr_max, dtype=np.uint8, out_dtype=None):
    """``out_dtype=torch.float64`` dequantises in the reference's own element type."""
    import torch

    bits = _bits(dtype)
    lib = _lib.load()
    f64 = out_dtype == torch.float64
    out = torch.empty(q.shape, dtype=torch.float64 if f64 else torch.float32, device=q.device)
    fn = lib.ndmps_dequantize_f64 if f64 else lib.ndmps_dequantize_f32
    _lib.check(fn(q.data_ptr(), q.numel(), float(arr_min), float(arr_max), bits, out.data_ptr(), _lib.stream_ptr()))
    return out


def to_numpy_uint(q, dtype):
    """Host copy of a quantised device tensor with the requested unsigned dtype."""
    arr = q.cpu().numpy()
    return arr.view(np.dtype(dtype)) if arr.dtype != np.dtype(dtype) else arr

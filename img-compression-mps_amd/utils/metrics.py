"""Quality metrics on device-resident tensors.

Same function names and argument meaning as the reference's
src/imgcompressionmps/utils/metrics.py (``compute_ssim_2d`` :11-32, ``ssim_3d_axis`` :35-65, ``avg_ssim_3d`` :68-85,
``avg_ssim_4d`` :88-105, ``compute_ssim_by_dim`` :108-129, ``compute_psnr`` :132-146,
``compute_overlap`` :149-160); the arithmetic runs in csrc/metrics.hip (fp64 on fp32 data).
Arguments may be device tensors, NumPy arrays or anything ``torch.as_tensor`` accepts.
"""
import ctypes as C

from .. import _lib


def _dev(t, like=None):
    import torch

    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(t)
    device = like.device if like is not None and like.is_cuda else ("cuda" if not t.is_cuda else t.device)
    return t.to(device=device, dtype=torch.float32).contiguous()


def compute_ssim_by_dim(a, b) -> float:
    import torch

    _lib.require_device()
    lib = _lib.load()
    a = _dev(a, b if isinstance(b, torch.Tensor) else None)
    b = _dev(b, a)
    if a.shape != b.shape:
        raise ValueError("Shape mismatch between arrays.")
    if a.dim() not in (2, 3, 4):
        raise ValueError(f"Unsupported tensor dimension for SSIM: {a.dim()}")
    shape = _lib.i64_array(a.shape)
    nbytes = lib.ndmps_ssim_workspace_bytes(a.dim(), shape)
    if nbytes < 0:
        raise ValueError("win_size exceeds image extent.")
    with torch.cuda.device(a.device):
        ws = torch.empty(nbytes, dtype=torch.uint8, device=a.device)
        out = C.c_double()
        _lib.check(lib.ndmps_ssim_f32(a.data_ptr(), b.data_ptr(), a.dim(), shape, C.byref(out), ws.data_ptr(),
                                      nbytes, _lib.stream_ptr()))
    return float(out.value)


def compute_ssim_2d(original, compressed) -> float:
    return compute_ssim_by_dim(original, compressed)


def ssim_3d_axis(original, compressed, axis: int = 0):
    """Slice-wise SSIM scores along ``axis`` of a 3-D volume (metrics.py:35-65): a list of ``shape[axis]`` floats, the
    second argument clipped at 0, every slice with its own joint data range.  One library call for the whole list."""
    import torch

    _lib.require_device()
    lib = _lib.load()
    a = _dev(original, compressed if isinstance(compressed, torch.Tensor) else None)
    b = _dev(compressed, a)
    if a.shape != b.shape:
        raise ValueError("Shape mismatch between 3D arrays.")
    if a.dim() != 3 or axis >= a.dim() or axis < -a.dim():
        raise ValueError(f"Invalid axis {axis} for 3D SSIM.")
    axis = int(axis) % 3
    shape = _lib.i64_array(a.shape)
    nbytes = lib.ndmps_ssim_slices_workspace_bytes(shape, axis)
    if nbytes < 0:
        raise ValueError("win_size exceeds image extent.")
    with torch.cuda.device(a.device):
        ws = torch.empty(nbytes, dtype=torch.uint8, device=a.device)
        out = (C.c_double * int(a.shape[axis]))()
        _lib.check(lib.ndmps_ssim_slices_f32(a.data_ptr(), b.data_ptr(), shape, axis, out, ws.data_ptr(), nbytes,
                                             _lib.stream_ptr()))
    return [float(x) for x in out]


def avg_ssim_3d(original, compressed) -> float:
    return compute_ssim_by_dim(original, compressed)


def avg_ssim_4d(original, compressed) -> float:
    return compute_ssim_by_dim(original, compressed)


def compute_psnr(original, compressed) -> float:
    import torch

    _lib.require_device()
    lib = _lib.load()
    a = _dev(original, compressed if isinstance(compressed, torch.Tensor) else None)
    b = _dev(compressed, a)
    with torch.cuda.device(a.device):
        nbytes = lib.ndmps_psnr_workspace_bytes()
        ws = torch.empty(nbytes, dtype=torch.uint8, device=a.device)
        out = C.c_double()
        _lib.check(lib.ndmps_psnr_f32(a.data_ptr(), b.data_ptr(), a.numel(), C.byref(out), ws.data_ptr(), nbytes,
                                      _lib.stream_ptr()))
    return float(out.value)


def compute_overlap(mps1, mps2) -> float:
    """Normalized overlap (fidelity) between two NDMPS objects (metrics.py:149-160)."""
    return (mps1.mps @ mps2.mps) / (mps1.norm_value * mps2.norm_value)

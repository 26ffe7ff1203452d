"""Device-resident MPS: what callers of the reference reach through ``NDMPS.mps``.

The reference stores a ``quimb.tensor.MatrixProductState`` there and touches only a
handful of its attributes: ``.arrays`` (mutable, in-place writable),
``@`` (overlap, core/ndmps.py:76,86 and utils/metrics.py:160), ``[i]`` and ``.sites``
(core/ndmps.py:103-105), iteration with ``.size`` (core/ndmps.py:129), ``.bond_sizes()``,
``.show()`` and ``copy.deepcopy`` (evaluation/benchmark.py:154).  ``DeviceMPS`` offers
exactly those on fp32 cores that live in HBM; every operation is a HIP kernel reached
through the C ABI (include/ndmps_hip.h).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _lib


def _torch():
    import torch

    return torch


class DeviceCore:
    """A core (or a view of one) in HBM that behaves like the NumPy arrays in ``mps.arrays``:
    NumPy can read it (``np.asarray``, ``np.min``/``np.max``), and slices of it can be
    assigned or updated in place (``core[:] *= 10``, ``core[:] = other``)."""

    __array_priority__ = 1000

    def __init__(self, tensor):
        self.tensor = tensor

    # -- NumPy-facing protocol ------------------------------------------------------
    @property
    def shape(self):
        return tuple(self.tensor.shape)

    @property
    def ndim(self):
        return self.tensor.dim()

    @property
    def size(self):
        return int(self.tensor.numel())

    @property
    def dtype(self):
        return np.dtype(np.float64 if self.tensor.dtype == _torch().float64 else np.float32)

    def __len__(self):
        return self.tensor.shape[0]

    def __array__(self, dtype=None, copy=None):
        t = self.tensor.detach()
        if t.dtype != _torch().float64:
            t = t.float()  # bf16 cores read as fp32 (NumPy has no bf16)
        arr = t.cpu().numpy()
        return arr.astype(dtype) if dtype is not None else arr

    def numpy(self):
        return self.__array__()

    def min(self, axis=None, out=None, **kwargs):
        from ..utils.filetools import minmax

        if axis is not None or out is not None:
            raise NotImplementedError("DeviceCore.min supports only the full reduction")
        return minmax(self.tensor)[0]

    def max(self, axis=None, out=None, **kwargs):
        from ..utils.filetools import minmax

        if axis is not None or out is not None:
            raise NotImplementedError("DeviceCore.max supports only the full reduction")
        return minmax(self.tensor)[1]

    # -- views and in-place updates -------------------------------------------------
    @staticmethod
    def _unwrap(value, like):
        torch = _torch()
        if isinstance(value, DeviceCore):
            return value.tensor
        if isinstance(value, torch.Tensor):
            return value.to(device=like.device, dtype=like.dtype)
        if isinstance(value, np.ndarray):
            return torch.from_numpy(np.ascontiguousarray(value)).to(device=like.device, dtype=like.dtype)
        return value  # scalar

    def __getitem__(self, idx):
        return DeviceCore(self.tensor[idx])

    def __setitem__(self, idx, value):
        value = self._unwrap(value, self.tensor)
        target = self.tensor[idx]
        if hasattr(value, "data_ptr") and value.data_ptr() == target.data_ptr() and value.shape == target.shape:
            return  # ``core[:] *= k`` writes the view back onto itself
        self.tensor[idx] = value

    def __imul__(self, other):
        self.tensor.mul_(self._unwrap(other, self.tensor))
        return self

    def __itruediv__(self, other):
        self.tensor.div_(self._unwrap(other, self.tensor))
        return self

    def __iadd__(self, other):
        self.tensor.add_(self._unwrap(other, self.tensor))
        return self

    def __isub__(self, other):
        self.tensor.sub_(self._unwrap(other, self.tensor))
        return self

    def __repr__(self):
        return f"DeviceCore(shape={self.shape}, device={self.tensor.device})"


class _Site:
    """``mps[i]`` / iteration item: exposes ``.size``, ``.shape`` and ``.data``."""

    def __init__(self, mps, i):
        self._mps, self._i = mps, i

    @property
    def data(self):
        return self._mps.arrays[self._i]

    @property
    def size(self):
        return int(self._mps.cores[self._i].numel())

    @property
    def shape(self):
        return self._mps.arrays[self._i].shape


class DeviceMPS:
    """Open-boundary MPS with cores ``(chi_i, d_i, chi_{i+1})`` in HBM (fp32; bf16 or fp64 storage on request)."""

    def __init__(self, cores, _trusted=False):
        if _trusted:  # cores made by the sweep: contiguous 3-D views already
            self.cores = list(cores)
            return
        self.cores = [c.contiguous() for c in cores]
        for c in self.cores:
            if c.dim() != 3:
                raise ValueError("cores must be (chi_left, d, chi_right)")

    # -- quimb-like surface -----------------------------------------------------------
    @property
    def L(self):
        return len(self.cores)

    def __len__(self):
        return len(self.cores)

    @property
    def sites(self):
        return tuple(range(len(self.cores)))

    @property
    def dims(self):
        return [int(c.shape[1]) for c in self.cores]

    @property
    def bonds(self):
        return [1] + [int(c.shape[2]) for c in self.cores]

    @property
    def device(self):
        return self.cores[0].device

    @property
    def arrays(self):
        """Edge sites are 2-D like quimb's: (d0, chi) first, (chi, d) last, (d,) if L == 1."""
        L = len(self.cores)
        out = []
        for i, c in enumerate(self.cores):
            if L == 1:
                out.append(DeviceCore(c.view(c.shape[1])))
            elif i == 0:
                out.append(DeviceCore(c.view(c.shape[1], c.shape[2])))
            elif i == L - 1:
                out.append(DeviceCore(c.view(c.shape[0], c.shape[1])))
            else:
                out.append(DeviceCore(c))
        return tuple(out)

    def __getitem__(self, i):
        return _Site(self, i)

    def __iter__(self):
        return (_Site(self, i) for i in range(len(self.cores)))

    def bond_sizes(self):
        return [int(c.shape[2]) for c in self.cores[:-1]]

    def show(self):
        print(" ".join(f"o-{b}-" for b in self.bond_sizes()) + "o")

    def num_elements(self):
        return sum(int(c.numel()) for c in self.cores)

    # -- kernels ----------------------------------------------------------------------
    def _core_ptrs(self):
        return (C.c_void_p * len(self.cores))(*[c.data_ptr() for c in self.cores])

    def _f32_ptrs(self):
        """(keep-alive list, pointer array) of the cores as fp32 (bf16 cores are upcast: a few MB at most)."""
        torch = _torch()
        keep = [c if c.dtype == torch.float32 else c.to(torch.float32) for c in self.cores]
        return keep, (C.c_void_p * len(keep))(*[c.data_ptr() for c in keep])

    def _f64_ptrs(self):
        torch = _torch()
        keep = [c if c.dtype == torch.float64 else c.to(torch.float64) for c in self.cores]
        return keep, (C.c_void_p * len(keep))(*[c.data_ptr() for c in keep])

    @property
    def dtype(self):
        return self.cores[0].dtype

    def __matmul__(self, other):
        """Unconjugated overlap <self|other> (core/ndmps.py:76,86), fp64 transfer matrices."""
        torch = _torch()
        if not isinstance(other, DeviceMPS) or other.dims != self.dims:
            raise ValueError("overlap needs two MPS over the same site dimensions")
        lib = _lib.load()
        L = len(self.cores)
        dims = _lib.i64_array(self.dims)
        ba, bb = _lib.i64_array(self.bonds), _lib.i64_array(other.bonds)
        nbytes = lib.ndmps_overlap_workspace_bytes(L, dims, ba, bb)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        out = C.c_double()
        if self.dtype == torch.float64 or other.dtype == torch.float64:
            keep_a, pa = self._f64_ptrs()  # fp64 cores are contracted as they are (a mixed pair: upcast)
            keep_b, pb = other._f64_ptrs()
            fn = lib.ndmps_overlap_f64
        else:
            keep_a, pa = self._f32_ptrs()
            keep_b, pb = other._f32_ptrs()
            fn = lib.ndmps_overlap_f32
        _lib.check(fn(L, dims, ba, pa, bb, pb, C.byref(out), ws.data_ptr(), nbytes, _lib.stream_ptr()))
        del keep_a, keep_b
        return float(out.value)

    def to_dense(self, out=None):
        """Left->right chain contraction (core/ndmps.py:140); returns N elements in site order, in the
        storage type of the cores (fp32, or bf16 on the bf16 MFMA)."""
        torch = _torch()
        lib = _lib.load()
        L = len(self.cores)
        dims, bonds = _lib.i64_array(self.dims), _lib.i64_array(self.bonds)
        numel = int(np.prod(self.dims, dtype=np.int64))
        if out is None:
            out = torch.empty(numel, dtype=self.dtype, device=self.device)
        f64 = self.dtype == torch.float64
        nbytes = (lib.ndmps_chain_workspace_bytes_f64 if f64 else lib.ndmps_chain_workspace_bytes)(L, dims, bonds)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        fn = (lib.ndmps_chain_contract_f64 if f64 else
              lib.ndmps_chain_contract_bf16 if self.dtype == torch.bfloat16 else lib.ndmps_chain_contract_f32)
        _lib.check(fn(L, dims, bonds, self._core_ptrs(), out.data_ptr(), ws.data_ptr(), nbytes, _lib.stream_ptr()))
        return out

    def to_volume(self, out, n_tail, tables):
        """Chain contraction straight into the C-order volume ``out`` (fp32 cores): the last product scatters
        through the inverse permutation (``tables`` = _Plan.split_tables(n_tail, device))."""
        torch = _torch()
        lib = _lib.load()
        L = len(self.cores)
        dims, bonds = _lib.i64_array(self.dims), _lib.i64_array(self.bonds)
        row_off, col_off, col_perm = tables
        nbytes = lib.ndmps_chain_workspace_bytes(L, dims, bonds)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        _lib.check(lib.ndmps_chain_contract_scatter_f32(L, dims, bonds, self._core_ptrs(), out.data_ptr(),
                                                        row_off.data_ptr(), col_off.data_ptr(), col_perm.data_ptr(),
                                                        int(n_tail), ws.data_ptr(), nbytes, _lib.stream_ptr()))
        return out

    def compress_bond_(self, i, cutoff, max_bond=None):
        """tensor_compress_bond on bond (i-1, i) (core/ndmps.py:104-106); returns the spectrum.  bf16 cores
        are truncated in fp32 and stored back as bf16, fp64 cores in fp64."""
        torch = _torch()
        lib = _lib.load()
        store = self.dtype
        work = torch.float64 if store == torch.float64 else torch.float32
        t1, t2 = self.cores[i - 1].to(work), self.cores[i].to(work)
        chi_l, d1, chi = (int(v) for v in t1.shape)
        _, d2, chi_r = (int(v) for v in t2.shape)
        nbytes = lib.ndmps_compress_bond_workspace_bytes(chi_l, d1, chi, d2, chi_r)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        new1 = torch.empty(chi_l * d1 * chi, dtype=work, device=self.device)
        new2 = torch.empty(chi * d2 * chi_r, dtype=work, device=self.device)
        k = C.c_int64()
        spec = (C.c_double * chi)()
        fn = lib.ndmps_compress_bond_f64 if work == torch.float64 else lib.ndmps_compress_bond_f32
        _lib.check(fn(
            t1.data_ptr(), t2.data_ptr(), chi_l, d1, chi, d2, chi_r, float(cutoff),
            int(max_bond) if max_bond else 0, new1.data_ptr(), new2.data_ptr(), C.byref(k), spec,
            ws.data_ptr(), nbytes, _lib.stream_ptr()))
        k = int(k.value)
        self.cores[i - 1] = new1[: chi_l * d1 * k].view(chi_l, d1, k).to(store, copy=True)
        self.cores[i] = new2[: k * d2 * chi_r].view(k, d2, chi_r).to(store, copy=True)
        return np.array(spec[:], dtype=np.float64)

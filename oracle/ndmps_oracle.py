"""Oracle (test infrastructure): fp64 CPU restatement of the reference ``NDMPS``.

Follows /root/reference/src/imgcompressionmps/core/ndmps.py line by line in
behaviour (not in text):

* ``from_tensor``  <- ndmps.py:36-78  (copy -> map -> norm -> DCT -> scatter -> sweep
                                        -> min/max -> norm)
* ``compress``     <- ndmps.py:94-108 (bonds left->right, rel cutoff, refresh)
* ``continuous_compress`` <- ndmps.py:110-125
* ``to_tensor``    <- ndmps.py:131-153 (contract -> gather -> IDCT)
* bookkeeping      <- ndmps.py:80-92,127-129,155-180
* quantise/gzip    <- ndmps.py:182-277

quimb's part is restated in oracle/mps.py ("parity unpinned", see there).  The
index map is oracle/index_map.py (pinned by the reference's golden vectors).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
``max_bond`` (on ``from_tensor`` and ``compress``) is the keyword BASELINE adds
(SURVEY F3); with ``max_bond=None`` the behaviour is the reference's.
``materialise_map=False`` swaps the (L,*shape) int64 map for the closed-form flat
permutation (bit-identical, verified in tests) for volumes whose map temporaries do
not fit in RAM.
"""
from __future__ import annotations

import gzip
import io

import numpy as np
from scipy.fft import dct as _dct, idct as _idct

from . import index_map as _im
from .filetools import get_num_bits, scale_back, scale_to_dtype
from .mps import OracleMPS, mps_from_dense


class OracleNDMPS:
    def __init__(self, mps=None, qubit_size=None, encoding_map=None, boundary_list=None,
                 norm=True, norm_value=None, mode="Std", dim=None):
        self.qubit_size = qubit_size
        self.encoding_map = encoding_map
        self.mps = mps
        self.dim = dim
        self.norm = norm
        self.norm_value = norm_value
        self.mode = mode
        self.boundary_list = np.array(boundary_list)
        self._flat_dest = None
        self._shape = None

    # ------------------------------------------------------------------ encode
    @classmethod
    def from_tensor(cls, tensor, norm=False, mode="Std", max_bond=None, cutoff=1e-10,
                    materialise_map=True, sweep_from="right"):
        tensor = np.asarray(tensor).astype(np.float64)
        shape = tuple(int(s) for s in tensor.shape)
        if materialise_map:
            qubit_size, enc = _im.gen_encoding_map(shape)
            enc = np.moveaxis(enc, 0, -1)
        else:
            qubit_size, _ = _im.dest_tables(shape)
            enc = None
        if norm:
            tensor /= np.linalg.norm(tensor)
        if mode == "DCT":
            tensor = _dct(tensor, type=2, norm="ortho", axis=-1)

        dense = np.empty(tuple(int(q) for q in qubit_size), dtype=np.float64)
        if enc is not None:
            k = enc.shape[-1]
            flat_idx = enc.reshape(-1, k)
            dense[tuple(flat_idx[:, c] for c in range(k))] = tensor.reshape(-1)
            flat_dest = None
        else:
            flat_dest = _im.flat_destination(shape).reshape(-1)
            dense.reshape(-1)[flat_dest] = tensor.reshape(-1)

        cores, spectra = mps_from_dense(dense, qubit_size, cutoff=cutoff, max_bond=max_bond, sweep_from=sweep_from)
        mps = OracleMPS(cores)
        boundary = [[np.min(a), np.max(a)] for a in mps.arrays]
        norm_value = np.sqrt(mps @ mps)
        obj = cls(mps, qubit_size, enc, boundary, norm, norm_value, mode, tensor.ndim)
        obj._flat_dest = flat_dest
        obj._shape = shape
        obj.sweep_spectra = spectra  # singular values per bond (diagnostic the GPU path also keeps)
        return obj

    # ------------------------------------------------------------- bookkeeping
    def update_boundary_list(self):
        self.boundary_list = np.array([[np.min(t), np.max(t)] for t in self.mps.arrays])

    def update_norm(self):
        self.norm_value = np.sqrt(self.mps @ self.mps)

    def compression_ratio(self):
        return self.number_elements_in_MPS() / np.prod(self.qubit_size)

    def number_elements_in_MPS(self):
        return sum(t.size for t in self.mps)

    def bond_sizes(self):
        return self.mps.bond_sizes()

    def show(self):
        self.mps.show()

    def return_tensors_data(self):
        return [t for t in self.mps.arrays]

    def replace_tensordata(self, tensorlist):
        for i in range(len(self.mps.arrays)):
            assert self.mps.arrays[i].shape == tensorlist[i].shape
            self.mps.arrays[i][:] = tensorlist[i]
        self.update_boundary_list()
        self.update_norm()

    # ---------------------------------------------------------------- truncate
    def compress(self, cutoff, max_bond=None):
        for i in range(1, len(self.mps.sites)):
            self.mps.compress_bond_(i, cutoff, max_bond)
        self.update_boundary_list()
        self.update_norm()

    def continuous_compress(self, cutoff, print_ratio=True):
        for c in np.linspace(0, 1, 20) * cutoff:
            self.compress(c)
            if print_ratio:
                print(f"Compression ratio at {c}: {self.compression_ratio()}")

    # ------------------------------------------------------------- reconstruct
    def to_tensor(self):
        dense = self.mps.to_dense()
        if self.encoding_map is not None:
            k = self.encoding_map.shape[-1]
            rec = dense[tuple(self.encoding_map[..., c] for c in range(k))]
        else:
            rec = dense.reshape(-1)[self._flat_dest].reshape(self._shape)
        if self.mode == "Std":
            return rec
        if self.mode == "DCT":
            return _idct(rec, type=2, norm="ortho", axis=-1)
        return None  # ndmps.py:150-153: unknown modes fall through

    # ------------------------------------------------- quantise / on-disk size
    def compress_to_dtype(self, dtype=np.uint16, replace=False):
        ints = [scale_to_dtype(t, dtype) for t in self.mps.arrays]
        back = [scale_back(t, b[0], b[1], dtype) for t, b in zip(ints, self.boundary_list)]
        if replace:
            self.replace_tensordata(back)
        return ints

    def get_bytesize_on_disk(self, dtype=np.uint16, replace=False):
        total = 0
        for arr in self.compress_to_dtype(dtype, replace):
            buf = io.BytesIO()
            with gzip.GzipFile(fileobj=buf, mode="wb") as gz:
                gz.write(arr.tobytes())
            total += len(buf.getvalue())
        return total

    def compression_ratio_on_disk(self, dtype=np.uint16, replace=False):
        original = np.prod(self.qubit_size) * get_num_bits(dtype) / 8.0
        return self.get_bytesize_on_disk(dtype, replace) / original

    def get_storage_space(self, dtype=np.uint16, verbose=False):
        size_bytes = self.number_elements_in_MPS() * get_num_bits(dtype) / 8
        if verbose:
            print(f"The storage space is approximately: {size_bytes / 1024:.2f} KB")
        return size_bytes

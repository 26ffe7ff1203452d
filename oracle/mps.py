"""Oracle (test infrastructure): fp64 NumPy restatement of the quimb calls made by
the reference hot path.

The arithmetic of the reference lives in the third-party package ``quimb==1.9.0``
(/root/reference/requirements.txt:110), which is absent from /root/reference and
from this image.  This file restates its published algorithm for exactly the five
call sites of /root/reference/src/imgcompressionmps/core/ndmps.py:

* ``MatrixProductState.from_dense``  (ndmps.py:74)   -> :func:`mps_from_dense`
* ``tensor_compress_bond``           (ndmps.py:104-106) -> :func:`compress_bond`
* ``mps ^ ...``                      (ndmps.py:140)  -> :func:`mps_to_dense`
* ``mps @ mps``                      (ndmps.py:76,86; utils/metrics.py:160) -> :func:`mps_overlap`
* ``.arrays/.sites/[i]/.bond_sizes()`` -> :class:`OracleMPS`

PARITY UNPINNED at this boundary: the reference holds no known-answer vector for a
truncated core or reconstruction; the restatement is pinned only by the
reference's own properties (tests/core/test_ndmps.py:35-79: round trip 1e-10,
norm==1, compress shrinks, norm^2 == mps@mps, 0<disk ratio<1, 20 printed lines).

Semantics restated:
  from_dense : sweep i = L-1 .. 1, matricise as (prod_{j<i} d_j) x (d_i chi_{i+1}),
               thin SVD (LAPACK gesdd via np.linalg.svd), keep s_k > cutoff*s_0 with
               cutoff=1e-10 ("rel"), U*S carried left, Vh becomes site i.
  compress_bond : QR(left core), LQ(right core), SVD(R L), keep
               k = max(1, #{s_j > cutoff*s_0}) (only when cutoff > 0), sqrt(s)
               absorbed into both sides, recontract.
  to_dense   : cumulative left -> right contraction.
  overlap    : unconjugated transfer-matrix contraction.

``max_bond`` is the keyword BASELINE.json adds: min(k, max_bond).
"""
from __future__ import annotations

import numpy as np


def _truncate(s, cutoff, max_bond, keep_at_least_one=True):
    """Number of singular values kept under quimb's 'rel' cutoff mode."""
    k = len(s)
    if cutoff > 0.0 and k > 0:
        k = int(np.count_nonzero(s > cutoff * s[0]))
        if keep_at_least_one:
            k = max(k, 1)
    if max_bond is not None:
        k = min(k, int(max_bond))
    return k


def mps_from_dense(dense, dims, cutoff=1e-10, max_bond=None, sweep_from="right"):
    """SVD sweep; returns cores [(chi_i, d_i, chi_{i+1})] and spectra (spectra[i]: bond between sites i-1 and i).

    ``sweep_from="right"`` (default): quimb 1.9.0's ``from_dense`` as SURVEY a4 states it -- sites L-1 .. 1, Vh is the
    site, U S is carried left.  ``sweep_from="left"``: the other convention a from_dense could have (SURVEY a4's
    caveat: quimb's source cannot be read here) -- sites 0 .. L-2, U is the site, S Vh is carried right.  Written out
    on its own (not as a mirrored right sweep) so that it checks the GPU path's mirroring."""
    dims = [int(d) for d in dims]
    L = len(dims)
    cores = [None] * L
    spectra = [None] * L
    if sweep_from == "left":
        work = np.asarray(dense, dtype=np.float64).reshape(1, -1)  # (chi_left, everything to the right)
        chi_l = 1
        for i in range(L - 1):
            cols = work.size // (chi_l * dims[i])
            u, s, vh = np.linalg.svd(work.reshape(chi_l * dims[i], cols), full_matrices=False)
            k = _truncate(s, cutoff, max_bond)
            cores[i] = u[:, :k].reshape(chi_l, dims[i], k)
            spectra[i + 1] = s.copy()
            work = s[:k, None] * vh[:k]
            chi_l = k
        cores[L - 1] = work.reshape(chi_l, dims[L - 1], 1)
        return cores, spectra
    if sweep_from != "right":
        raise ValueError("sweep_from must be 'right' or 'left'")
    work = np.asarray(dense, dtype=np.float64).reshape(-1, 1)  # (rows, chi_right)
    chi_r = 1
    for i in range(L - 1, 0, -1):
        rows = work.size // (dims[i] * chi_r)
        mat = work.reshape(rows, dims[i] * chi_r)
        u, s, vh = np.linalg.svd(mat, full_matrices=False)
        k = _truncate(s, cutoff, max_bond)
        cores[i] = vh[:k].reshape(k, dims[i], chi_r)
        spectra[i] = s.copy()
        work = u[:, :k] * s[:k]
        chi_r = k
    cores[0] = work.reshape(1, dims[0], chi_r)
    return cores, spectra


def compress_bond(t1, t2, cutoff, max_bond=None):
    """quimb ``tensor_compress_bond(reduced=True, absorb='both')`` on two cores.

    t1: (chi_l, d1, chi), t2: (chi, d2, chi_r).  Returns (t1', t2', s).
    """
    chi_l, d1, chi = t1.shape
    chi2, d2, chi_r = t2.shape
    assert chi == chi2
    q1, r = np.linalg.qr(t1.reshape(chi_l * d1, chi))            # (m, r1) (r1, chi)
    q2t, lt = np.linalg.qr(t2.reshape(chi, d2 * chi_r).T)        # LQ via QR of the transpose
    lq_l, lq_q = lt.T, q2t.T                                      # (chi, r2) (r2, n)
    m = r @ lq_l
    u, s, vh = np.linalg.svd(m, full_matrices=False)
    k = _truncate(s, cutoff, max_bond)
    rs = np.sqrt(s[:k])
    new1 = (q1 @ (u[:, :k] * rs)).reshape(chi_l, d1, k)
    new2 = ((rs[:, None] * vh[:k]) @ lq_q).reshape(k, d2, chi_r)
    return new1, new2, s


def mps_to_dense(cores):
    """Cumulative left->right contraction; returns the (d_0..d_{L-1}) tensor."""
    dims = [c.shape[1] for c in cores]
    acc = cores[0].reshape(cores[0].shape[1], cores[0].shape[2])
    for c in cores[1:]:
        chi, d, chi_r = c.shape
        acc = (acc @ c.reshape(chi, d * chi_r)).reshape(-1, chi_r)
    return acc.reshape(dims)


def mps_overlap(a_cores, b_cores):
    """<a|b> without conjugation (real data); equals sum(dense_a * dense_b)."""
    env = np.ones((1, 1), dtype=np.float64)
    for a, b in zip(a_cores, b_cores):
        ca, d, ca2 = a.shape
        cb, _, cb2 = b.shape
        x = env.T @ a.reshape(ca, d * ca2)            # (cb, d*ca2)
        x = x.reshape(cb * d, ca2)
        env = x.T @ b.reshape(cb * d, cb2)            # (ca2, cb2)
    return float(env[0, 0])


class _Site:
    """What ``mps[i]`` / iteration must expose: ``.size`` and ``.data`` (ndmps.py:129)."""

    def __init__(self, owner, i):
        self._owner, self._i = owner, i

    @property
    def data(self):
        return self._owner.arrays[self._i]

    @property
    def size(self):
        return self.data.size

    @property
    def shape(self):
        return self.data.shape


class OracleMPS:
    """Minimal stand-in for ``quimb.tensor.MatrixProductState``."""

    def __init__(self, cores):
        self._cores = [np.ascontiguousarray(c, dtype=np.float64) for c in cores]

    # quimb exposes the edge sites as 2-D arrays: (d0, chi) and (chi, d_last)
    @property
    def arrays(self):
        L = len(self._cores)
        out = []
        for i, c in enumerate(self._cores):
            if L == 1:
                out.append(c.reshape(c.shape[1]))
            elif i == 0:
                out.append(c.reshape(c.shape[1], c.shape[2]))
            elif i == L - 1:
                out.append(c.reshape(c.shape[0], c.shape[1]))
            else:
                out.append(c)
        return tuple(out)

    @property
    def cores(self):
        return self._cores

    @property
    def sites(self):
        return tuple(range(len(self._cores)))

    @property
    def L(self):
        return len(self._cores)

    def __len__(self):
        return len(self._cores)

    def __getitem__(self, i):
        return _Site(self, i)

    def __iter__(self):
        return (_Site(self, i) for i in range(len(self._cores)))

    def __matmul__(self, other):
        return mps_overlap(self._cores, other._cores)

    def bond_sizes(self):
        return [int(c.shape[2]) for c in self._cores[:-1]]

    def show(self):
        print(" ".join(f"o-{b}-" for b in self.bond_sizes()) + "o")

    def compress_bond_(self, i, cutoff, max_bond=None):
        a, b, s = compress_bond(self._cores[i - 1], self._cores[i], cutoff, max_bond)
        self._cores[i - 1], self._cores[i] = np.ascontiguousarray(a), np.ascontiguousarray(b)
        return s

    def to_dense(self):
        return mps_to_dense(self._cores)

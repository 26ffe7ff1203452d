"""Oracle (test infrastructure): CPU restatement of the reference index map.

Restates /root/reference/src/imgcompressionmps/utils/core.py in NumPy int64:

* ``balance_factors``               <- utils/core.py:38-76
* ``get_factorlist``                <- utils/core.py:79-126
* ``hierarchical_block_indexing``   <- utils/core.py:129-168
* ``gen_encoding_map``              <- utils/core.py:6-35

The reference derives per-level digits with a float64 division + floor
(utils/core.py:160-168).  Below 2**53 that equals the integer closed form
``digit[l, j] = (x_j // w[l+1, j]) % f[l, j]`` with ``w`` the suffix products of
the radices; both are implemented here (``faithful=True`` walks the reference's
float route) and are checked against each other and against the reference's
golden vectors (tests/utils/test_core.py:75-88,111-122,151-173,212-237) in
tests/test_oracle_index_map.py.

Pinned by: reference golden vectors + fixtures generated from the reference's own
utils/core.py in the build container (tests/golden/make_golden_index_map.py).
"""
from __future__ import annotations

import numpy as np

_I64MAX = np.iinfo(np.int64).max


def _prime_factors(n: int) -> list[int]:
    """Ascending prime factors with multiplicity (trial division; n >= 2)."""
    out, p = [], 2
    while p * p <= n:
        while n % p == 0:
            out.append(p)
            n //= p
        p += 1 if p == 2 else 2
    if n > 1:
        out.append(n)
    return out


def _check_shape(shape) -> None:
    if len(shape) == 0:
        raise ValueError("Shape cannot be empty.")
    for d in shape:
        if not isinstance(d, int) or d <= 0:
            raise ValueError("All dimensions must be positive integers.")


def balance_factors(factors, target_num):
    """Merge the two smallest factors until ``target_num`` remain (core.py:38-76)."""
    if target_num < 0:
        raise ValueError("target_num must be non-negative.")
    if target_num == 0 and len(factors) > 0:
        raise ValueError("Cannot reduce non-empty factor list to length zero.")
    fs = sorted(factors)
    if len(fs) < target_num:
        raise ValueError("The number of balanced factors cannot be less than the target number.")
    while len(fs) > target_num:
        merged = fs[0] * fs[1]
        fs = sorted([merged] + fs[2:])
    return fs


def get_factorlist(shape):
    """(factor_arr (L, ndim), prod (L+1, ndim)) as in core.py:79-126."""
    _check_shape(shape)
    per_dim = [[1] if d == 1 else _prime_factors(d) for d in shape]
    depth = min(len(f) for f in per_dim)
    per_dim = [balance_factors(f, depth) for f in per_dim]
    # "snake": odd-numbered dimensions run from coarse to fine (core.py:115-117)
    per_dim = [f[::-1] if j % 2 == 1 else f for j, f in enumerate(per_dim)]
    factor_arr = np.array(per_dim, dtype=np.int64).T  # (L, ndim)
    L, nd = factor_arr.shape
    prod = np.ones((L + 1, nd), dtype=np.int64)
    for lvl in range(L - 1, 0, -1):  # suffix products of the later radices
        prod[lvl] = prod[lvl + 1] * factor_arr[lvl]
    prod[0] = _I64MAX
    return factor_arr, prod


def hierarchical_block_indexing(index, prod_block_sizes, faithful=True):
    """Digits (L, ndim, *shape) of every coordinate (core.py:129-168)."""
    index = np.asarray(index)
    prod_block_sizes = np.asarray(prod_block_sizes)
    nd = index.shape[0]
    if (
        prod_block_sizes.ndim != 2
        or prod_block_sizes.shape[1] != nd
        or prod_block_sizes.shape[0] < 2
    ):
        raise ValueError(
            "prod_block_sizes must be of shape (num_levels + 1, ndim) with ndim matching index."
        )
    L = prod_block_sizes.shape[0] - 1
    tail = (1,) * nd
    hi = prod_block_sizes[:-1].reshape((L, nd) + tail)
    lo = prod_block_sizes[1:].reshape((L, nd) + tail)
    rem = np.mod(index[None], hi)
    if faithful:  # the reference's float64 route
        return np.floor(rem / lo).astype(np.int64)
    return (rem // lo).astype(np.int64)


def gen_encoding_map(shape, faithful=True):
    """(qubit_size (L,), enc_map (L, *shape)) as in core.py:6-35."""
    _check_shape(shape)
    factor_arr, prod = get_factorlist(shape)
    digits = hierarchical_block_indexing(np.indices(shape), prod, faithful=faithful)
    L = factor_arr.shape[0]
    enc = np.empty((L,) + tuple(shape), dtype=np.int64)
    for lvl in range(L):
        enc[lvl] = np.ravel_multi_index(tuple(digits[lvl]), tuple(int(f) for f in factor_arr[lvl]))
    return np.prod(factor_arr, axis=1), enc


# ---------------------------------------------------------------------------
# closed form used for volumes whose materialised map does not fit in RAM
# (SURVEY 8d "closed-form-index" CPU variant); checked == gen_encoding_map.
# ---------------------------------------------------------------------------
def dest_tables(shape):
    """Per-dimension tables T_j with flat_dest(x) = sum_j T_j[x_j].

    flat_dest is the C-order offset in the ``qubit_size`` tensor of the voxel
    whose C-order coordinates in ``shape`` are x (core/ndmps.py:66-71 scatter).
    """
    factor_arr, _ = get_factorlist(shape)
    L, nd = factor_arr.shape
    site_dim = np.prod(factor_arr, axis=1)
    site_stride = np.ones(L, dtype=np.int64)
    for lvl in range(L - 2, -1, -1):
        site_stride[lvl] = site_stride[lvl + 1] * site_dim[lvl + 1]
    tables = []
    for j in range(nd):
        x = np.arange(shape[j], dtype=np.int64)
        t = np.zeros(shape[j], dtype=np.int64)
        w = 1
        for lvl in range(L - 1, -1, -1):
            f = int(factor_arr[lvl, j])
            digit = (x // w) % f
            inner = int(np.prod(factor_arr[lvl, j + 1:]))  # row-major ravel inside the site
            t += digit * inner * int(site_stride[lvl])
            w *= f
        tables.append(t)
    return site_dim, tables


def flat_destination(shape):
    """Flat destination offset of every voxel, shape ``shape`` (int64)."""
    _, tables = dest_tables(shape)
    nd = len(shape)
    dest = np.zeros(shape, dtype=np.int64)
    for j, t in enumerate(tables):
        dest += t.reshape((1,) * j + (-1,) + (1,) * (nd - 1 - j))
    return dest

"""Host-side index-map logic of the NDMPS hot path (integer, exact).

Mirrors the public functions of the reference's
src/imgcompressionmps/utils/core.py -- same names, arguments, return values and errors --
so callers and tests read the same:

* ``get_factorlist``   (utils/core.py:79-126)  shape -> (factor_arr, prod_block_sizes)
* ``balance_factors``  (utils/core.py:38-76)
* ``gen_encoding_map`` (utils/core.py:6-35)    materialised map, only for drop-in access
* ``hierarchical_block_indexing`` (utils/core.py:129-168)

The GPU path never materialises the map: ``site_dims`` / ``factor_arr`` go to
``ndmps_plan_create`` (csrc/permute.hip), which turns them into small per-dimension and
per-site offset tables.  The reference computes digits through float64 division; here it is
integer arithmetic (identical below 2**53, see tests/test_host_index_map.py).
"""
from __future__ import annotations

import heapq
from typing import List, Sequence, Tuple

import numpy as np


def _validate(shape) -> None:
    if len(shape) == 0:
        raise ValueError("Shape cannot be empty.")
    if any(not isinstance(d, (int, np.integer)) or isinstance(d, bool) or d <= 0 for d in shape):
        raise ValueError("All dimensions must be positive integers.")


def _factorise(n: int) -> List[int]:
    """Prime factors of n >= 1 in ascending order (1 -> [1], as the reference treats it)."""
    if n == 1:
        return [1]
    out = []
    while n % 2 == 0:
        out.append(2)
        n //= 2
    f = 3
    while f * f <= n:
        while n % f == 0:
            out.append(f)
            n //= f
        f += 2
    if n > 1:
        out.append(n)
    return out


def balance_factors(factors: List[int], target_num: int) -> List[int]:
    if target_num < 0:
        raise ValueError("target_num must be non-negative.")
    if target_num == 0 and len(factors) > 0:
        raise ValueError("Cannot reduce non-empty factor list to length zero.")
    if len(factors) < target_num:
        raise ValueError("The number of balanced factors cannot be less than the target number.")
    heap = list(factors)
    heapq.heapify(heap)
    while len(heap) > target_num:  # fold the two smallest together
        a = heapq.heappop(heap)
        b = heapq.heappop(heap)
        heapq.heappush(heap, a * b)
    return sorted(heap)


def get_factorlist(shape: Sequence[int]) -> Tuple[np.ndarray, np.ndarray]:
    _validate(shape)
    lists = [_factorise(int(d)) for d in shape]
    depth = min(len(f) for f in lists)
    lists = [balance_factors(f, depth) for f in lists]
    for j in range(1, len(lists), 2):  # odd dimensions run coarse -> fine
        lists[j].reverse()
    factor_arr = np.asarray(lists, dtype=np.int64).T.copy()
    L = factor_arr.shape[0]
    prod = np.ones((L + 1, factor_arr.shape[1]), dtype=np.int64)
    running = np.ones(factor_arr.shape[1], dtype=np.int64)
    for lvl in range(L - 1, 0, -1):
        running = running * factor_arr[lvl]
        prod[lvl] = running
    prod[0] = np.iinfo(np.int64).max
    return factor_arr, prod


def site_dims(shape: Sequence[int]) -> np.ndarray:
    """Physical dimension of every MPS site (``qubit_size`` of the reference)."""
    return np.prod(get_factorlist(shape)[0], axis=1)


def hierarchical_block_indexing(index: np.ndarray, prod_block_sizes: np.ndarray) -> np.ndarray:
    index = np.asarray(index)
    prod_block_sizes = np.asarray(prod_block_sizes)
    nd = index.shape[0]
    if prod_block_sizes.ndim != 2 or prod_block_sizes.shape[1] != nd or prod_block_sizes.shape[0] < 2:
        raise ValueError(
            "prod_block_sizes must be of shape (num_levels + 1, ndim) with ndim matching index."
        )
    L = prod_block_sizes.shape[0] - 1
    bshape = (L, nd) + (1,) * nd
    upper = prod_block_sizes[:-1].reshape(bshape)
    lower = prod_block_sizes[1:].reshape(bshape)
    return ((index[None] % upper) // lower).astype(np.int64)


def gen_encoding_map(shape: Sequence[int]) -> Tuple[np.ndarray, np.ndarray]:
    """Materialised (L, *shape) map -- host only, for code that reads ``NDMPS.encoding_map``."""
    _validate(shape)
    shape = tuple(int(s) for s in shape)
    factor_arr, prod = get_factorlist(shape)
    digits = hierarchical_block_indexing(np.indices(shape), prod)
    enc = np.stack([
        np.ravel_multi_index(tuple(digits[lvl]), tuple(int(f) for f in factor_arr[lvl]))
        for lvl in range(factor_arr.shape[0])
    ]).astype(np.int64)
    return np.prod(factor_arr, axis=1), enc

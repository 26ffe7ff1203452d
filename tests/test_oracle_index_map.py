"""Oracle index map vs the reference's golden vectors.

Known-answer arrays below are the ones held by the reference's own tests
(/root/reference/tests/utils/test_core.py:75-88,111-122,151-173,212-237); the .npz /
.json fixtures were generated from the reference's utils/core.py by
tests/golden/make_golden_index_map.py.
"""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import index_map as im

I64MAX = 9223372036854775807


# ---- balance_factors behaviours (test_core.py:14-63) -------------------------
def test_balance_preserves_product():
    r = im.balance_factors([2, 2, 3, 3], 2)
    assert np.prod(r) == 36 and len(r) == 2


def test_balance_identity_sorted_ones_large():
    assert im.balance_factors([2, 3], 2) == [2, 3]
    r = im.balance_factors([7, 3, 2, 2], 3)
    assert r == sorted(r)
    assert im.balance_factors([1, 1, 1, 1], 2) == [1, 1]
    r = im.balance_factors([2] * 10, 2)
    assert len(r) == 2 and np.prod(r) == 1024
    assert im.balance_factors([], 0) == []
    assert im.balance_factors([6], 1) == [6]


def test_balance_errors():
    with pytest.raises(ValueError):
        im.balance_factors([2, 3], -1)
    with pytest.raises(ValueError):
        im.balance_factors([2, 3], 0)
    with pytest.raises(ValueError):
        im.balance_factors([2], 3)


# ---- get_factorlist known answers (test_core.py:75-88,111-122) ---------------
def test_factorlist_256x128():
    f, p = im.get_factorlist((256, 128))
    assert np.array_equal(f, [[2, 2]] * 6 + [[4, 2]])
    assert np.array_equal(
        p, [[I64MAX, I64MAX], [128, 64], [64, 32], [32, 16], [16, 8], [8, 4], [4, 2], [1, 1]]
    )


def test_factorlist_30x40x50():
    f, p = im.get_factorlist((30, 40, 50))
    assert np.array_equal(f, [[2, 5, 2], [3, 4, 5], [5, 2, 5]])
    assert np.array_equal(p, [[I64MAX] * 3, [15, 8, 25], [5, 2, 5], [1, 1, 1]])


def test_factorlist_shapes_snake_errors():
    f, p = im.get_factorlist((30, 24))
    assert f.shape[1] == 2 and p.shape == (f.shape[0] + 1, 2)
    assert np.all(np.prod(f, axis=0) == (30, 24))
    f, p = im.get_factorlist((1, 1))
    assert np.all(f == 1) and np.all(p >= 1)
    f, _ = im.get_factorlist((4, 6))
    assert np.all(np.diff(f[:, 1])[::-1] <= 0)
    f, p = im.get_factorlist((8,))
    assert f.shape[1] == 1 and p.shape[1] == 1
    with pytest.raises(ValueError):
        im.get_factorlist((0, 4))
    with pytest.raises(ValueError):
        im.get_factorlist(())


# ---- gen_encoding_map known answer (test_core.py:151-173) --------------------
def test_encoding_map_8x9():
    q, enc = im.gen_encoding_map((8, 9))
    assert np.array_equal(q, [6, 12])
    lvl0 = np.repeat(np.repeat(np.array([[0, 1, 2], [3, 4, 5]]), 4, axis=0), 3, axis=1)
    lvl1 = np.tile(np.arange(12).reshape(4, 3), (2, 3))
    assert np.array_equal(enc[0], lvl0)
    assert np.array_equal(enc[1], lvl1)


def test_encoding_map_shapes_and_errors():
    q, enc = im.gen_encoding_map((3, 3))
    assert enc.shape[1:] == (3, 3) and len(q) == enc.shape[0]
    assert np.all(im.gen_encoding_map((2, 2))[1] >= 0)
    q, enc = im.gen_encoding_map((1, 4))
    assert enc.shape == (len(q), 1, 4)
    with pytest.raises(ValueError):
        im.gen_encoding_map(())
    with pytest.raises(ValueError):
        im.gen_encoding_map(("a", "b"))


# ---- hierarchical_block_indexing known answer (test_core.py:212-237) ---------
def test_hier_4x6():
    shape = (4, 6)
    _, p = im.get_factorlist(shape)
    r = im.hierarchical_block_indexing(np.indices(shape), p)
    assert r.shape == (2, 2, 4, 6)
    rows = np.arange(4)[:, None] * np.ones((1, 6), dtype=int)
    cols = np.ones((4, 1), dtype=int) * np.arange(6)[None, :]
    assert np.array_equal(r[0, 0], rows // 2)
    assert np.array_equal(r[0, 1], cols // 2)
    assert np.array_equal(r[1, 0], rows % 2)
    assert np.array_equal(r[1, 1], cols % 2)
    with pytest.raises(ValueError):
        im.hierarchical_block_indexing(np.indices((4, 4)), np.array([[1, 1]]))
    z = im.hierarchical_block_indexing(np.indices((1, 1)), im.get_factorlist((1, 1))[1])
    assert np.all(z == 0)


# ---- fixtures generated from the reference's own utils/core.py ----------------
def _small(golden_dir):
    return np.load(os.path.join(golden_dir, "index_map_small.npz"))


def test_against_reference_small_maps(golden_dir):
    g = _small(golden_dir)
    keys = sorted({k.split("/")[0] for k in g.files})
    assert len(keys) >= 12
    for key in keys:
        shape = tuple(int(s) for s in key.split("x"))
        f, p = im.get_factorlist(shape)
        assert np.array_equal(f, g[key + "/factor_arr"]), key
        assert np.array_equal(p, g[key + "/prod"]), key
        for faithful in (True, False):
            q, enc = im.gen_encoding_map(shape, faithful=faithful)
            assert np.array_equal(q, g[key + "/qubit_size"]), key
            assert np.array_equal(enc, g[key + "/enc_map"]), key
        flat = im.flat_destination(shape).reshape(-1)
        assert np.array_equal(flat, g[key + "/flat_dest"]), key
        assert np.array_equal(np.sort(flat), np.arange(flat.size)), "map must be a bijection"


def test_against_reference_large_hashes(golden_dir):
    with open(os.path.join(golden_dir, "index_map_hashes.json")) as fh:
        hashes = json.load(fh)
    for key, rec in hashes.items():
        shape = tuple(int(s) for s in key.split("x"))
        f, _ = im.get_factorlist(shape)
        assert f.tolist() == rec["factor_arr"], key
        q, _ = im.dest_tables(shape)
        assert [int(v) for v in q] == rec["qubit_size"], key
        flat = im.flat_destination(shape).reshape(-1).astype(np.int64)
        assert flat[:16].tolist() == rec["flat_dest_head"], key
        assert hashlib.sha256(flat.tobytes()).hexdigest() == rec["flat_dest_sha256"], key


def test_morton_order_for_power_of_two_cube():
    """for 2^k cubes the map is Z-order with x most significant per triple."""
    flat = im.flat_destination((8, 8, 8))
    for (x, y, z) in [(1, 0, 0), (0, 1, 0), (0, 0, 1), (5, 3, 6), (7, 7, 7)]:
        code = 0
        for b in range(3):
            code |= ((x >> b) & 1) << (3 * b + 2) | ((y >> b) & 1) << (3 * b + 1) | ((z >> b) & 1) << (3 * b)
        assert flat[x, y, z] == code

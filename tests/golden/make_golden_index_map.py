"""Generate golden vectors for the index map from the REFERENCE's own code.

Run in the build container only (the reference does not travel to the GPU box):
    python tests/golden/make_golden_index_map.py
Imports /root/reference/src/imgcompressionmps/utils/core.py and stores
  * factor_arr / prod / qubit_size / full enc_map for small shapes  -> index_map_small.npz
  * sha256 of the flat destination permutation for large shapes      -> index_map_hashes.json
The flat destination of a voxel is ravel_multi_index(enc_map[:, voxel], qubit_size), i.e.
the C-order offset the reference's scatter (core/ndmps.py:66-71) writes the voxel to.
"""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference/src")
from imgcompressionmps.utils.core import gen_encoding_map, get_factorlist  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

SMALL = [(32, 32), (8, 9), (4, 6), (30, 40, 50), (16, 16, 16), (12, 8, 20, 6), (7, 12),
         (1, 4), (64,), (3, 3), (2, 2, 2), (6, 10, 15), (1, 1), (8, 4, 2, 2)]
LARGE = [(512, 680), (8, 512, 680), (128, 128, 128), (64, 64, 64), (96, 80, 112)]


def flat_dest_from_reference(shape):
    qubit_size, enc = gen_encoding_map(shape)
    L = enc.shape[0]
    flat = np.ravel_multi_index(tuple(enc[lvl].reshape(-1) for lvl in range(L)),
                                tuple(int(q) for q in qubit_size))
    return qubit_size, enc, flat.astype(np.int64)


def main():
    small = {}
    for shape in SMALL:
        key = "x".join(map(str, shape))
        f, p = get_factorlist(shape)
        q, enc, flat = flat_dest_from_reference(shape)
        small[key + "/factor_arr"] = f.astype(np.int64)
        small[key + "/prod"] = p.astype(np.int64)
        small[key + "/qubit_size"] = np.asarray(q, dtype=np.int64)
        small[key + "/enc_map"] = enc.astype(np.int32)
        small[key + "/flat_dest"] = flat.astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "index_map_small.npz"), **small)

    hashes = {}
    for shape in LARGE:
        key = "x".join(map(str, shape))
        f, _ = get_factorlist(shape)
        q, _, flat = flat_dest_from_reference(shape)
        hashes[key] = {
            "factor_arr": f.tolist(),
            "qubit_size": [int(v) for v in q],
            "flat_dest_sha256": hashlib.sha256(flat.tobytes()).hexdigest(),
            "flat_dest_head": flat[:16].tolist(),
        }
    with open(os.path.join(HERE, "index_map_hashes.json"), "w") as fh:
        json.dump(hashes, fh, indent=1)
    print("wrote", len(SMALL), "small maps and", len(LARGE), "hashes")


if __name__ == "__main__":
    main()

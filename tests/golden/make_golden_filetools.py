"""Golden vectors for the quantise helpers from the REFERENCE's own filetools.py
(/root/reference/src/imgcompressionmps/utils/filetools.py:7-39).  Build container only:
    python tests/golden/make_golden_filetools.py   -> filetools.npz
"""
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference/src")
from imgcompressionmps.utils.filetools import get_num_bits, scale_back, scale_to_dtype  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    rng = np.random.default_rng(2025)
    out = {}
    for name, shape in [("a", (5, 7)), ("b", (4, 8, 4)), ("c", (33,))]:
        x = rng.standard_normal(shape)
        out[name + "/x"] = x
        for dt in (np.uint8, np.uint16):
            q = scale_to_dtype(x, dt)
            back = scale_back(q, x.min(), x.max(), dt)
            out[f"{name}/{np.dtype(dt).name}/q"] = q
            out[f"{name}/{np.dtype(dt).name}/back"] = back
        # fp32-representable values, as the GPU path stores its cores, through the reference's OWN flow: NDMPS holds
        # float64 arrays (core/ndmps.py:56), so the values are widened first and the arithmetic is float64
        x32 = x.astype(np.float32)
        out[name + "/x32"] = x32
        wide = x32.astype(np.float64)
        for dt in (np.uint8, np.uint16):
            q = scale_to_dtype(wide, dt)
            out[f"{name}/x32/{np.dtype(dt).name}/q"] = q
            out[f"{name}/x32/{np.dtype(dt).name}/back"] = scale_back(q, wide.min(), wide.max(), dt)
    out["bits"] = np.array([get_num_bits(d) for d in (np.uint8, np.uint16, np.int32, np.float32, np.float64)])
    np.savez_compressed(os.path.join(HERE, "filetools.npz"), **out)
    print("wrote filetools.npz")


if __name__ == "__main__":
    main()

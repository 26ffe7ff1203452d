"""Oracle (test infrastructure): restatement of the quantise helpers
/root/reference/src/imgcompressionmps/utils/filetools.py:7-39.

Quirks kept on purpose: ``scale_to_dtype`` truncates (``astype``)
instead of rounding and divides by the post-shift max (NaN for constant input).
Pinned by fixtures generated from the reference's own filetools.py
(tests/golden/make_golden_filetools.py).
"""
import numpy as np


def get_num_bits(dtype):
    dtype = np.dtype(dtype)
    if np.issubdtype(dtype, np.integer):
        return np.iinfo(dtype).bits
    if np.issubdtype(dtype, np.floating):
        return np.finfo(dtype).bits
    raise ValueError(f"Unsupported dtype {dtype!r}")


def scale_to_dtype(array, dtype=np.uint8):
    shifted = array - np.min(array)
    unit = shifted / np.max(shifted)
    return (unit * np.iinfo(dtype).max).astype(dtype)


def scale_back(array, arr_min, arr_max, dtype=np.uint8):
    return (array / np.iinfo(dtype).max) * (arr_max - arr_min) + arr_min

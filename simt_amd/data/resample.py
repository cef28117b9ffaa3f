"""Host-side tables for the device resampler (csrc/input_prep.hip).

The reference's loader resizes with Pillow (dataset/cityscapes_dataset.py:105-106: `image.resize(crop_size, Image.BICUBIC)`,
`label.resize(crop_size, Image.NEAREST)`).  Pillow's 8-bit resampler is integer arithmetic on tables that depend only on the
(source size, target size) pair; they are built here in double precision the way Pillow builds them (Resample.c
precompute_coeffs / normalize_coeffs_8bpc; Geometry.c ImagingScaleAffine) and applied by simt_resample_u8 / simt_label_nearest,
which makes the device output equal to Pillow's byte for byte.  Pure numpy: table construction is O(out * ksize) host work done
once per geometry, not per image.
"""
import functools
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2          # Resample.c: 8-bit pixels, 2 bits of head-room for the negative lobes


def _bicubic(x):
    """Pillow's bicubic kernel, a = -0.5, support 2."""
    a = -0.5
    x = np.abs(x)
    near = ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    far = (((x - 5) * x + 8) * x - 4) * a
    return np.where(x < 1.0, near, np.where(x < 2.0, far, 0.0))


@functools.lru_cache(maxsize=64)
def bicubic_tables(in_size, out_size):
    """-> (ksize, bounds int32 [out, 2] = (first source index, count), kk int32 [out, ksize]) for one axis."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    inv = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        lo = max(int(center - support + 0.5), 0)
        n = min(int(center + support + 0.5), in_size) - lo
        w = _bicubic((np.arange(n, dtype=np.float64) + lo - center + 0.5) * inv)
        total = 0.0
        for v in w:                       # Pillow sums left to right in double: keep the order
            total += float(v)
        if total != 0.0:
            w = w / total
        fixed = w * float(1 << PRECISION_BITS)
        kk[xx, :n] = np.where(w < 0, np.trunc(fixed - 0.5), np.trunc(fixed + 0.5)).astype(np.int64)
        bounds[xx] = (lo, n)
    return ksize, bounds, kk


@functools.lru_cache(maxsize=64)
def nearest_table(in_size, out_size):
    """Source index per target index of Image.resize(..., NEAREST): a double advanced by repeated addition, truncated."""
    a = float(in_size) / out_size
    pos = a * 0.5
    tab = np.empty(out_size, np.int32)
    for x in range(out_size):
        tab[x] = int(pos)
        pos += a
    return np.clip(tab, 0, in_size - 1)

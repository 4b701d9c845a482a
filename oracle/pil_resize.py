"""TEST INFRASTRUCTURE (oracle): numpy restatement of Pillow's 8-bit BILINEAR resampler, the arithmetic behind the
reference's test-time Resize (mega_core/data/transforms/transforms.py:57-59 -> torchvision F.resize ->
PIL.Image.resize(BILINEAR) -> libImaging/Resample.c).  Pillow is a third-party dependency of the reference, not
vendored under /root/reference (INSTALL.md pins none; this image has Pillow 12.2): the algorithm restated here is
Resample.c's precompute_coeffs / normalize_coeffs_8bpc / ImagingResampleHorizontal_8bpc / Vertical_8bpc, and it is
pinned by running Pillow itself beside it (tests/test_feed.py), bit for bit.

Only tests/ may import this module.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def coeffs(in_size, out_size):
    """precompute_coeffs (triangle filter, support 1.0, box = [0, in_size)) + normalize_coeffs_8bpc."""
    scale = in_size / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds, taps = [], []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = []
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) / filterscale)
            w.append(1.0 - a if a < 1.0 else 0.0)
        ww = sum(w)
        w = [v / ww if ww != 0.0 else v for v in w]
        bounds.append((xmin, xmax))
        taps.append([int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS)) for v in w])
    return bounds, taps, ksize


def _pass(x, axis, out_size):
    in_size = x.shape[axis]
    if in_size == out_size:
        return x
    bounds, taps, _ = coeffs(in_size, out_size)
    x = np.moveaxis(x.astype(np.int64), axis, 0)
    out = np.empty((out_size,) + x.shape[1:], dtype=np.int64)
    for o, ((lo, n), k) in enumerate(zip(bounds, taps)):
        acc = np.full(x.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for t in range(n):
            acc += x[lo + t] * k[t]
        out[o] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis).astype(np.uint8)


def resize_bilinear_u8(img_hwc, out_h, out_w):
    """uint8 [H,W,C] -> [out_h,out_w,C]: horizontal pass (to uint8) then vertical, as ImagingResampleInner."""
    return _pass(_pass(img_hwc, 1, out_w), 0, out_h)

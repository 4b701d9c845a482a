"""TEST INFRASTRUCTURE ONLY -- builds oracle/native_oracle.c with gcc and binds it with ctypes.

Imported only by tests/, __graft_entry__.smoke()/build() and bench.py's cpu_baseline leg.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "native_oracle.c")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "liboracle.so")
_lib = None


def build(force=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    os.makedirs(OUT_DIR, exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", SRC, "-o", LIB, "-lm"])
    return LIB


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(LIB)
        _lib.oracle_nms.restype = ctypes.c_int
        _lib.oracle_nms.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int,
                                    ctypes.c_void_p]
        _lib.oracle_roi_align.restype = None
        _lib.oracle_roi_align.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_float] + \
                                         [ctypes.c_int] * 3
    return _lib


def nms(dets, scores, thr, strict_gt=True):
    """numpy in / numpy out: ascending original indices kept (int64)."""
    dets = np.ascontiguousarray(dets, dtype=np.float32).reshape(-1, 4)
    scores = np.ascontiguousarray(scores, dtype=np.float32).reshape(-1)
    n = dets.shape[0]
    keep = np.empty((max(n, 1),), dtype=np.int64)
    cnt = lib().oracle_nms(dets.ctypes.data, scores.ctypes.data, n, float(thr), int(strict_gt), keep.ctypes.data)
    return keep[:cnt].copy()


def roi_align(feat_nchw, rois, spatial_scale, pooled_h, pooled_w, sampling_ratio):
    feat = np.ascontiguousarray(feat_nchw, dtype=np.float32)
    rois = np.ascontiguousarray(rois, dtype=np.float32).reshape(-1, 5)
    B, C, H, W = feat.shape
    K = rois.shape[0]
    out = np.empty((K, C, pooled_h, pooled_w), dtype=np.float32)
    if K:
        lib().oracle_roi_align(feat.ctypes.data, rois.ctypes.data, out.ctypes.data, K, C, H, W, float(spatial_scale),
                               pooled_h, pooled_w, int(sampling_ratio))
    return out

"""Caller-chosen split-K (mega_conv2d_nhwc_ks) on the small-M / long-K layers of config 5's key frame: time per launch for
ksplit in a sweep.  (FlowNetS's coarse levels on 21 pairs, the RPN conv / res5 / fc6 of ONE frame.)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mega.pytorch_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
dt = torch.bfloat16
CASES = [  # name, N, H, W, Cin, Cout, R, stride, pad, dil
    ("flow conv4   ", 21, 38, 63, 256, 512, 3, 2, 1, 1),
    ("flow conv4_1 ", 21, 19, 32, 512, 512, 3, 1, 1, 1),
    ("flow conv5   ", 21, 19, 32, 512, 512, 3, 2, 1, 1),
    ("flow conv5_1 ", 21, 10, 16, 512, 512, 3, 1, 1, 1),
    ("flow conv6   ", 21, 10, 16, 512, 1024, 3, 2, 1, 1),
    ("flow conv6_1 ", 21, 5, 8, 1024, 1024, 3, 1, 1, 1),
    ("rpn conv     ", 1, 38, 63, 1024, 1024, 3, 1, 1, 1),
    ("res5 conv1 a ", 1, 38, 63, 1024, 512, 1, 1, 0, 1),
    ("res5 conv2   ", 1, 38, 63, 512, 512, 3, 1, 2, 2),
    ("res5 conv3   ", 1, 38, 63, 512, 2048, 1, 1, 0, 1),
    ("res5 conv1 b ", 1, 38, 63, 2048, 512, 1, 1, 0, 1),
    ("res5 ds      ", 1, 38, 63, 1024, 2048, 1, 1, 0, 1),
    ("fc6          ", 300, 1, 1, 100352, 1024, 1, 1, 0, 1),
    ("fc7          ", 300, 1, 1, 1024, 1024, 1, 1, 0, 1),
]


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, N, H, W, Cin, Cout, R, st, pad, dil in CASES:
    x = torch.randn((N, H, W, Cin), device=dev).to(dt)
    w = (torch.randn((Cout, R, R, Cin), device=dev) * 0.01).to(dt)
    b = torch.zeros((Cout,), device=dev)
    row = []
    for ks in (None, 1, 2, 3, 4, 6, 8, 12, 16, 24):
        if ks is not None and ks > R * R * Cin // 64:
            continue
        us = timeit(lambda: ops.conv2d_nhwc(x, w, None, b, stride=st, pad=pad, dil=dil, relu=2, ksplit=ks))
        row.append("%s:%6.1f" % ("dflt" if ks is None else "%4d" % ks, us))
    Ho = (H + 2 * pad - dil * (R - 1) - 1) // st + 1
    Wo = (W + 2 * pad - dil * (R - 1) - 1) // st + 1
    print("%s M=%6d N=%5d K=%6d us  %s" % (name, N * Ho * Wo, Cout, R * R * Cin, "  ".join(row)), flush=True)

# the sub-pixel deconvolutions
for name, N, H, W, Cin, C, H2, W2, Cs in (("deconv5", 21, 5, 8, 1024, 512, 10, 16, 512), ("deconv4", 21, 10, 16, 1088, 256, 19, 32, 512),
                                          ("deconv3", 21, 19, 32, 832, 128, 38, 63, 256), ("deconv2", 21, 38, 63, 448, 64, 75, 125, 128)):
    x = torch.randn((N, H, W, Cin), device=dev).to(dt)
    w4 = (torch.randn((4 * C, 2, 2, Cin), device=dev) * 0.01).to(dt)
    b4 = torch.zeros((4 * C,), device=dev)
    out = torch.zeros((N, H2, W2, (Cs + C + 2 + 63) // 64 * 64), device=dev, dtype=dt)
    row = []
    for ks in (1, 2, 3, 4, 6, 8):
        us = timeit(lambda: ops.deconv4x4s2_into(x, w4, b4, out, Cs, relu=2, ksplit=ks))
        row.append("%4d:%6.1f" % (ks, us))
    print("%s M=%6d N=%5d K=%6d us  %s" % (name, N * (H + 1) * (W + 1), 4 * C, 4 * Cin, "  ".join(row)), flush=True)

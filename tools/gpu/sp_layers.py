"""Per-layer time of the split-precision (x3) frame stage on a 40-frame batch: where do the SP GEMMs' 2.28 ms per key frame go?
(shape list of tools/bench_kernels.py; planes in, planes out, 3 K contraction; TF/s = matrix-core work incl. the 3 passes)"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from mega.pytorch_amd import ops  # noqa: E402
import bench_kernels as bk  # noqa: E402

dev = torch.device("cuda:0")
B = 40
tot = 0.0
rows = []
for name, N, H, W, Cin, Cout, R, st, pad, dil, count in bk.conv_shapes(B):
    if Cin % 64 or Cout % 8 or "fc0" in name:
        continue
    g = torch.Generator().manual_seed(1)
    x = ops.split_planes(torch.randn((N, H, W, Cin), generator=g).to(dev).contiguous())
    w = ops.split_conv_weight_x3((torch.randn((Cout, R, R, Cin), generator=g) / math.sqrt(Cin * R * R))).to(dev)
    sc = torch.ones((Cout,), device=dev)
    bi = torch.zeros((Cout,), device=dev)
    Ho, Wo = (H + 2 * pad - dil * (R - 1) - 1) // st + 1, (W + 2 * pad - dil * (R - 1) - 1) // st + 1
    res = ops.split_planes(torch.randn((N, Ho, Wo, Cout), generator=g).to(dev).contiguous()) if "conv3" in name else None
    run = lambda: ops.conv2d_sp(x, w, sc, bi, res, stride=st, pad=pad, dil=dil, relu=True, out_mode="planes")  # noqa: E731
    ms = bk.timeit(run)
    fl = 2.0 * N * Ho * Wo * Cout * R * R * Cin * 3
    by = x.t.numel() * 2 + N * Ho * Wo * Cout * 4 * (2 if res is not None else 1)
    rows.append((name, count, ms, fl / ms / 1e9, by / ms / 1e9))
    tot += ms * count
    del x, w, res
for name, count, ms, tf, tb in rows:
    print("%-30s x%-3d %.4f ms  %6.0f TF/s  %5.2f TB/s   %.3f ms per batch (%.1f %%)" % (name, count, ms, tf, tb, ms * count, 100 * ms * count / tot))
print("sum over the frame stage (without fc0): %.2f ms per 40 frames = %.3f ms per key frame" % (tot, tot / 20))

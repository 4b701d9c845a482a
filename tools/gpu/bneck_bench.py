"""fused layer1 bottleneck (ops.bottleneck64) vs the three launches it replaces, 40 frames of 150 x 250 (the bench's batch)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mega.pytorch_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
N, H, W = int(os.environ.get("NF", 40)), 150, 250
x = torch.randn((N, H, W, 256), generator=g).relu().to(torch.bfloat16).to(dev)
w1 = (torch.randn((64, 1, 1, 256), generator=g) * 0.06).to(torch.bfloat16).to(dev)
w2 = (torch.randn((64, 3, 3, 64), generator=g) * 0.05).to(torch.bfloat16).to(dev)
w3 = (torch.randn((256, 1, 1, 64), generator=g) * 0.1).to(torch.bfloat16).to(dev)
sb = [((torch.rand((n,), generator=g) + 0.5).to(dev), (torch.randn((n,), generator=g) * 0.2).to(dev)) for n in (64, 64, 256)]
def unfused():
    t1 = ops.conv2d_nhwc(x, w1, sb[0][0], sb[0][1], relu=True)
    t2 = ops.conv2d_nhwc(t1, w2, sb[1][0], sb[1][1], pad=1, relu=True)
    return ops.conv2d_nhwc(t2, w3, sb[2][0], sb[2][1], residual=x, relu=True)
def fused():
    return ops.bottleneck64(x, w1, sb[0][0], sb[0][1], w2, sb[1][0], sb[1][1], w3, sb[2][0], sb[2][1])
for name, fn in (("unfused", unfused), ("fused", fused), ("unfused", unfused), ("fused", fused)):
    for _ in range(3):
        y = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y = fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    px = N * H * W
    print("%-8s %.3f ms per block (%d frames): %.0f GB/s of the fused form's algorithmic bytes (1024 B/px), %.0f GB/s of the unfused form's (2083 B/px)"
          % (name, ms, N, px * 1024 / ms / 1e6, px * 2083 / ms / 1e6))
# ---- block 0 (downsample variant): x has 64 channels
x0 = torch.randn((N, H, W, 64), generator=g).relu().to(torch.bfloat16).to(dev)
w10 = (torch.randn((64, 1, 1, 64), generator=g) * 0.12).to(torch.bfloat16).to(dev)
wd = (torch.randn((256, 1, 1, 64), generator=g) * 0.1).to(torch.bfloat16).to(dev)
sbd = ((torch.rand((256,), generator=g) + 0.5).to(dev), (torch.randn((256,), generator=g) * 0.2).to(dev))
def unfused_ds():
    ident = ops.conv2d_nhwc(x0, wd, sbd[0], sbd[1])
    t1 = ops.conv2d_nhwc(x0, w10, sb[0][0], sb[0][1], relu=True)
    t2 = ops.conv2d_nhwc(t1, w2, sb[1][0], sb[1][1], pad=1, relu=True)
    return ops.conv2d_nhwc(t2, w3, sb[2][0], sb[2][1], residual=ident, relu=True)
def fused_ds():
    return ops.bottleneck64_ds(x0, w10, sb[0][0], sb[0][1], w2, sb[1][0], sb[1][1], w3, sb[2][0], sb[2][1], wd, sbd[0], sbd[1])
for name, fn in (("unfused-ds", unfused_ds), ("fused-ds", fused_ds), ("unfused-ds", unfused_ds), ("fused-ds", fused_ds)):
    for _ in range(3):
        y = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y = fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("%-10s %.3f ms per block (%d frames, block 0 with the downsample branch)" % (name, ms, N))
a, b = unfused_ds(), fused_ds()
print("ds bit-equal:", bool(torch.equal(a.view(torch.int16), b.view(torch.int16))), "differing:", int((a.view(torch.int16) != b.view(torch.int16)).sum()))
a, b = unfused(), fused()
print("bit-equal:", bool(torch.equal(a.view(torch.int16), b.view(torch.int16))), "differing:", int((a.view(torch.int16) != b.view(torch.int16)).sum()))

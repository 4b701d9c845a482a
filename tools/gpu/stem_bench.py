"""Time the bf16 stem at the benchmarked shape (40 frames of 600x1000 uint8): fused stem + pool against stem_u8 + max-pool."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mega.pytorch_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
u8 = torch.randint(0, 256, (N, 600, 1000, 3), generator=g, dtype=torch.uint8).to(dev)
w = torch.randn((64, 3, 7, 7), generator=g) * 0.05
sc = (torch.rand((64,), generator=g) + 0.5).to(dev)
bi = (torch.randn((64,), generator=g) * 0.1).to(dev)
mean = (102.9801, 115.9465, 122.7717)
w160 = ops.pack_stem_weight_bf16(w).to(dev)


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


fused = t(lambda: ops.stem_pool(u8, w160, sc, bi, mean, True))
st = t(lambda: ops.stem_u8(u8, w160, sc, bi, mean, True))
y = ops.stem_u8(u8, w160, sc, bi, mean, True)
mp = t(lambda: ops.maxpool3x3s2(y))
ref = ops.maxpool3x3s2(y)
got = ops.stem_pool(u8, w160, sc, bi, mean, True)
print("%d frames: fused stem+pool %.3f ms | stem_u8 %.3f + maxpool %.3f = %.3f ms | bit-equal %s" % (
    N, fused, st, mp, st + mp, torch.equal(ref.view(torch.int16), got.view(torch.int16))))

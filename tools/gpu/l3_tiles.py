"""layer3's HBM-bound 1x1 layers (conv3 256 -> 1024 + residual + ReLU, conv1 1024 -> 256 + ReLU) and layer2's, on every tile
family: is the one-block-per-CU igemm8 streaming class really the best kernel for them?  (MEGA_IGEMM_TILE forces the tile.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mega.pytorch_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for frames in (40, 20):
    for name, H, W, Cin, Cout, res in (("layer3 conv3 256->1024 +res", 38, 63, 256, 1024, True), ("layer3 conv1 1024->256", 38, 63, 1024, 256, False),
                                       ("layer2 conv3 128->512 +res", 75, 125, 128, 512, True), ("layer2 conv1 512->128", 75, 125, 512, 128, False),
                                       ("res5 conv3 512->2048 +res", 38, 63, 512, 2048, True), ("res5 conv1 1024->512", 38, 63, 1024, 512, False)):
        x = torch.randn((frames, H, W, Cin), device=dev).to(torch.bfloat16)
        w = (torch.randn((Cout, 1, 1, Cin), device=dev) * 0.05).to(torch.bfloat16)
        sc = torch.ones((Cout,), device=dev); bi = torch.zeros((Cout,), device=dev)
        r = torch.randn((frames, H, W, Cout), device=dev).to(torch.bfloat16) if res else None
        by = (x.numel() + frames * H * W * Cout * (2 if res else 1)) * 2
        out = "%-28s %2d frames %6.1f MB:" % (name, frames, by / 1e6)
        ref = None
        for tl in ("", "128x128", "128x64", "8:192", "8:256", "256x128"):
            if tl:
                os.environ["MEGA_IGEMM_TILE"] = tl
            else:
                os.environ.pop("MEGA_IGEMM_TILE", None)
            try:
                us = timeit(lambda: ops.conv2d_nhwc(x, w, sc, bi, residual=r, relu=True))
                y = ops.conv2d_nhwc(x, w, sc, bi, residual=r, relu=True)
                if ref is None:
                    ref = y
                same = torch.equal(ref.view(torch.int16), y.view(torch.int16))
                out += "  %s %.1f us (%.2f TB/s)%s" % (tl or "default", us, by / us / 1e6, "" if same else " !bits")
            except Exception as e:  # noqa: BLE001
                out += "  %s ERR" % tl
        os.environ.pop("MEGA_IGEMM_TILE", None)
        print(out, flush=True)

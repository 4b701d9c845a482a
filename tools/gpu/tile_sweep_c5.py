"""Forced tiles (MEGA_IGEMM_TILE) on FlowNetS's mid-size layers at 21 pairs: does the dispatch rule (igemm8 from 128 tiles on)
leave time on the table when a launch has 130-200 igemm8 tiles for 256 CUs?  One process per tile choice (the env is read per call,
but the f32 / bf16 packing caches are per process anyway)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, torch
sys.path.insert(0, %r)
from mega.pytorch_amd import ops
dev = torch.device("cuda:0")
CASES = [("conv3    ", 21, 75, 125, 128, 256, 5, 2, 2), ("conv3_1  ", 21, 38, 63, 256, 256, 3, 1, 1), ("conv4    ", 21, 38, 63, 256, 512, 3, 2, 1),
         ("conv4_1  ", 21, 19, 32, 512, 512, 3, 1, 1), ("conv5    ", 21, 19, 32, 512, 512, 3, 2, 1), ("conv6_1  ", 21, 5, 8, 1024, 1024, 3, 1, 1),
         ("conv2    ", 21, 150, 250, 64, 128, 5, 2, 2)]
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
out = []
for name, N, H, W, Cin, Cout, R, st, pad in CASES:
    x = torch.randn((N, H, W, Cin), device=dev).to(torch.bfloat16)
    w = (torch.randn((Cout, R, R, Cin), device=dev) * 0.01).to(torch.bfloat16)
    b = torch.zeros((Cout,), device=dev)
    try:
        us = timeit(lambda: ops.conv2d_nhwc(x, w, None, b, stride=st, pad=pad, relu=2))
    except Exception as e:
        us = float("nan")
    out.append("%%s%%7.1f" %% (name, us))
print("  ".join(out))
''' % ROOT
for tile in ("", "64x64", "128x64", "128x128", "256x128", "8:256", "8:192"):
    env = dict(os.environ)
    if tile:
        env["MEGA_IGEMM_TILE"] = tile
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print("%-8s %s" % (tile or "default", r.stdout.strip() or r.stderr.strip()[-300:]), flush=True)

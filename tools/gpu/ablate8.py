"""Timing-only ablations of igemm8 on the RPN conv shape (256-row tile): where does the K-loop time go?

The ablation / timeline variants are not in the product library: this script first rebuilds libmega_hip.so with
MEGA_BUILD_EXPERIMENTS=1 (-DMEGA_EXPERIMENTS), runs, and rebuilds the product library afterwards (the GPU box has hipcc).
"""
import os, sys, subprocess
_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_build = "import sys; sys.path.insert(0, %r); from mega.pytorch_amd import build; build.build(force=True)" % _root
subprocess.run([sys.executable, "-c", _build], env=dict(os.environ, MEGA_BUILD_EXPERIMENTS="1"), check=True)
code = r'''
import sys, torch, os
sys.path.insert(0, sys.argv[1])
from mega.pytorch_amd import ops
dev = torch.device("cuda:0")
def t(N,H,W,Cin,Cout,R,pad):
    x = torch.randn((N,H,W,Cin), device=dev).to(torch.bfloat16); w = (torch.randn((Cout,R,R,Cin), device=dev)*0.05).to(torch.bfloat16)
    sc = torch.ones((Cout,), device=dev); bi = torch.zeros((Cout,), device=dev)
    os.environ["MEGA_IGEMM_TILE"] = "8:256"
    for _ in range(3): ops.conv2d_nhwc(x, w, sc, bi, pad=pad, relu=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.conv2d_nhwc(x, w, sc, bi, pad=pad, relu=True)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10
print("ABL=%s  rpn 3x3 K=9216: %.3f ms   r5.conv1 1x1 K=2048 N=512: %.3f ms   l3.conv3 K=256 N=1024: %.3f ms" % (
    os.environ.get("MEGA_IGEMM8_ABLATE", "0"), t(20,38,63,1024,1024,3,1), t(20,38,63,2048,512,1,0), t(20,38,63,256,1024,1,0)))
'''
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for abl in ("0", "1", "2", "3", "4"):
    env = dict(os.environ, MEGA_IGEMM8_ABLATE=abl)
    subprocess.run([sys.executable, "-c", code, root], env=env)
subprocess.run([sys.executable, "-c", _build], env={k: v for k, v in os.environ.items() if k != "MEGA_BUILD_EXPERIMENTS"}, check=True)

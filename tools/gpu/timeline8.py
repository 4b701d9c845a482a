"""Whole-tile s_memtime timeline of igemm8 (ABL = 5 probe): where does a streaming-class tile spend its time?

Every block stamps (waves 0 and 4): 0 entry, 1 K loop starts (prologue issued), 2 K loop done, then per epilogue slab s:
3+2s staged (after the barrier), 4+2s stores issued; 11 all stores acknowledged.  Slot 12 = HW_ID | XCC_ID << 32.
The probe kernels are not in the product library: this script rebuilds libmega_hip.so with MEGA_BUILD_EXPERIMENTS=1,
runs, and rebuilds the product library afterwards (the GPU box has hipcc).
"""
import os, sys, subprocess
_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_build = "import sys; sys.path.insert(0, %r); from mega.pytorch_amd import build; build.build(force=True)" % _root
code = r'''
import sys, os, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from mega.pytorch_amd import ops
dev = torch.device("cuda:0")
out_path = os.environ["MEGA_IGEMM8_TIMELINE_OUT"]
def run(name, N, H, W, Cin, Cout, R, pad, res):
    x = torch.randn((N, H, W, Cin), device=dev).to(torch.bfloat16)
    w = (torch.randn((Cout, R, R, Cin), device=dev) * 0.05).to(torch.bfloat16)
    sc = torch.ones((Cout,), device=dev); bi = torch.zeros((Cout,), device=dev)
    r = torch.randn((N, H, W, Cout), device=dev).to(torch.bfloat16) if res else None
    os.environ["MEGA_IGEMM_TILE"] = "8:256"
    for _ in range(3):
        ops.conv2d_nhwc(x, w, sc, bi, residual=r, pad=pad, relu=True)
    torch.cuda.synchronize()
    t = np.fromfile(out_path, dtype=np.uint64).reshape(-1, 2, 16).astype(np.int64)
    nb = t.shape[0]
    ent = t[:, 0, 0]
    span = t[:, 0, 11].max() - ent[ent > 0].min()
    # shader clock under this kernel: s_memtime ticks (shader cycles) per s_memrealtime tick (100 MHz)
    rt = (t[:, 0, 14] - t[:, 0, 13]).astype(np.float64)
    ghz = ((t[:, 0, 11] - t[:, 0, 0]) / np.maximum(rt, 1.0)) * 0.1
    span_rt = t[:, 0, 14].max() - t[:, 0, 13].min()
    hw = t[:, 0, 12]
    cu = (hw & 0xFFFFFFFF) >> 8 & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; xcc = hw >> 32
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    print("== %s: M=%d N=%d K=%d res=%s: %d blocks on %d distinct CUs, span %d cycles = %.1f us; shader clock median %.2f GHz (p10 %.2f p90 %.2f)" % (
        name, N * H * W, Cout, R * R * Cin, res, nb, len(np.unique(cuid)), span, span_rt / 100.0, np.median(ghz), np.percentile(ghz, 10), np.percentile(ghz, 90)))
    g = t[:, 0, :]          # wave 0
    ph = {"prologue (entry -> K loop)": g[:, 1] - g[:, 0], "K loop": g[:, 2] - g[:, 1]}
    prev = g[:, 2]
    for s in range(4):
        ph["slab %d stage+barrier" % s] = g[:, 3 + 2 * s] - prev
        ph["slab %d read-out (stores issued)" % s] = g[:, 4 + 2 * s] - g[:, 3 + 2 * s]
        prev = g[:, 4 + 2 * s]
    ph["store drain (vmcnt 0)"] = g[:, 11] - g[:, 10]
    tot = g[:, 11] - g[:, 0]
    for k, v in ph.items():
        print("   %-36s median %7d  mean %7d  p90 %7d   (%4.1f %% of a tile)" % (k, np.median(v), v.mean(), np.percentile(v, 90), 100.0 * v.mean() / tot.mean()))
    print("   %-36s median %7d  mean %7d  p90 %7d" % ("tile total", np.median(tot), tot.mean(), np.percentile(tot, 90)))
    # per-CU schedule: gap between a block's end and the next block's entry on the same CU
    gaps, per_cu = [], []
    for c in np.unique(cuid):
        idx = np.where(cuid == c)[0]
        o = idx[np.argsort(g[idx, 0])]
        per_cu.append(len(o))
        for a, b in zip(o[:-1], o[1:]):
            gaps.append(g[b, 0] - g[a, 11])
    gaps = np.array(gaps) if gaps else np.zeros(1)
    print("   blocks per CU: min %d max %d;  end -> next entry gap on a CU: median %d mean %d p90 %d;  busy/span per CU %.2f" % (
        min(per_cu), max(per_cu), np.median(gaps), gaps.mean(), np.percentile(gaps, 90), tot.sum() / (len(per_cu) * float(span))))
    return span
spans = {}
spans["l3.conv3"] = run("l3.conv3 1x1 256->1024 + res", 40, 38, 63, 256, 1024, 1, 0, True)
spans["l3.conv1"] = run("l3.conv1 1x1 1024->256", 40, 38, 63, 1024, 256, 1, 0, False)
spans["l2.conv3"] = run("l2.conv3 1x1 128->512 + res", 40, 75, 125, 128, 512, 1, 0, True)
spans["l1.conv3"] = run("l1.conv3 1x1 64->256 + res", 40, 150, 250, 64, 256, 1, 0, True)
spans["l3.conv2"] = run("l3.conv2 3x3 256->256", 40, 38, 63, 256, 256, 3, 1, False)
spans["rpn.conv"] = run("rpn.conv 3x3 1024->1024", 40, 38, 63, 1024, 1024, 3, 1, False)
'''
timing = r'''
import sys, os, torch
sys.path.insert(0, sys.argv[1])
from mega.pytorch_amd import ops
dev = torch.device("cuda:0")
def t(N, H, W, Cin, Cout, R, pad, res):
    x = torch.randn((N, H, W, Cin), device=dev).to(torch.bfloat16)
    w = (torch.randn((Cout, R, R, Cin), device=dev) * 0.05).to(torch.bfloat16)
    sc = torch.ones((Cout,), device=dev); bi = torch.zeros((Cout,), device=dev)
    r = torch.randn((N, H, W, Cout), device=dev).to(torch.bfloat16) if res else None
    os.environ["MEGA_IGEMM_TILE"] = "8:256"
    for _ in range(3): ops.conv2d_nhwc(x, w, sc, bi, residual=r, pad=pad, relu=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.conv2d_nhwc(x, w, sc, bi, residual=r, pad=pad, relu=True)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3
print("un-probed launch durations (us): l3.conv3 %.1f  l3.conv1 %.1f  l2.conv3 %.1f  l1.conv3 %.1f  l3.conv2 %.1f" % (
    t(40,38,63,256,1024,1,0,True), t(40,38,63,1024,256,1,0,False), t(40,75,125,128,512,1,0,True), t(40,150,250,64,256,1,0,True), t(40,38,63,256,256,3,1,False)))
'''
if __name__ == "__main__":
    out = os.path.join(_root, "gpurun_out", "timeline8.bin")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    prebuilt = "--prebuilt" in sys.argv     # the experiments library was built before the snapshot was taken
    if not prebuilt:
        subprocess.run([sys.executable, "-c", _build], env=dict(os.environ, MEGA_BUILD_EXPERIMENTS="1"), check=True)
    try:
        subprocess.run([sys.executable, "-c", timing, _root], env=dict(os.environ))
        subprocess.run([sys.executable, "-c", code, _root], env=dict(os.environ, MEGA_IGEMM8_ABLATE="5", MEGA_IGEMM8_TIMELINE_OUT=out))
    finally:
        if not prebuilt:
            subprocess.run([sys.executable, "-c", _build], env={k: v for k, v in os.environ.items() if k != "MEGA_BUILD_EXPERIMENTS"}, check=True)
        if os.path.exists(out):
            os.remove(out)

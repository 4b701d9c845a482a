"""GPU micro-benchmark of the kernel shapes that make up one MEGA R-101 steady-state step (run on the GPU box).

  python tools/bench_kernels.py [--frames 16] [--dtype bfloat16] [--what conv,attn,roi,pos]

Prints one line per distinct launch shape: time (median of interleaved rounds), algorithmic TFLOP/s or GB/s.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mega.pytorch_amd import ops  # noqa: E402


def timeit(fn, rounds=7, inner=3):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / inner)
    ts.sort()
    return ts[len(ts) // 2]


def conv_shapes(B):
    """(name, N, H, W, Cin, Cout, R, stride, pad, dil, count per frame-batch) for R-101 C4 + RPN + res5 + fc0."""
    s = []
    s.append(("l1.b0.conv1 1x1 64->64", B, 150, 250, 64, 64, 1, 1, 0, 1, 1))
    s.append(("l1.b0.down 1x1 64->256", B, 150, 250, 64, 256, 1, 1, 0, 1, 1))
    s.append(("l1.conv2 3x3 64->64", B, 150, 250, 64, 64, 3, 1, 1, 1, 3))
    s.append(("l1.conv3 1x1 64->256", B, 150, 250, 64, 256, 1, 1, 0, 1, 3))
    s.append(("l1.conv1 1x1 256->64", B, 150, 250, 256, 64, 1, 1, 0, 1, 2))
    s.append(("l2.b0.conv1 1x1/2 256->128", B, 150, 250, 256, 128, 1, 2, 0, 1, 1))
    s.append(("l2.b0.down 1x1/2 256->512", B, 150, 250, 256, 512, 1, 2, 0, 1, 1))
    s.append(("l2.conv2 3x3 128->128", B, 75, 125, 128, 128, 3, 1, 1, 1, 4))
    s.append(("l2.conv3 1x1 128->512", B, 75, 125, 128, 512, 1, 1, 0, 1, 4))
    s.append(("l2.conv1 1x1 512->128", B, 75, 125, 512, 128, 1, 1, 0, 1, 3))
    s.append(("l3.b0.conv1 1x1/2 512->256", B, 75, 125, 512, 256, 1, 2, 0, 1, 1))
    s.append(("l3.b0.down 1x1/2 512->1024", B, 75, 125, 512, 1024, 1, 2, 0, 1, 1))
    s.append(("l3.conv2 3x3 256->256", B, 38, 63, 256, 256, 3, 1, 1, 1, 23))
    s.append(("l3.conv3 1x1 256->1024", B, 38, 63, 256, 1024, 1, 1, 0, 1, 23))
    s.append(("l3.conv1 1x1 1024->256", B, 38, 63, 1024, 256, 1, 1, 0, 1, 22))
    s.append(("rpn.conv 3x3 1024->1024", B, 38, 63, 1024, 1024, 3, 1, 1, 1, 1))
    s.append(("rpn.pred 1x1 1024->60", B, 38, 63, 1024, 60, 1, 1, 0, 1, 1))
    s.append(("r5.b0.conv1 1x1 1024->512", B, 38, 63, 1024, 512, 1, 1, 0, 1, 1))
    s.append(("r5.b0.down 1x1 1024->2048", B, 38, 63, 1024, 2048, 1, 1, 0, 1, 1))
    s.append(("r5.conv2 3x3d2 512->512", B, 38, 63, 512, 512, 3, 1, 2, 2, 3))
    s.append(("r5.conv3 1x1 512->2048", B, 38, 63, 512, 2048, 1, 1, 0, 1, 3))
    s.append(("r5.conv1 1x1 2048->512", B, 38, 63, 2048, 512, 1, 1, 0, 1, 2))
    nroi = (B // 2) * 300 + (B - B // 2) * 75
    s.append(("fc0 %dx100352->1024" % nroi, nroi, 1, 1, 100352, 1024, 1, 1, 0, 1, 1))
    return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--dtype", default="bfloat16")
    ap.add_argument("--what", default="conv,attn,roi,pos")
    ap.add_argument("--tiles", default="", help="comma list of forced igemm tiles to compare, e.g. 128x128,256x128")
    a = ap.parse_args()
    dt = {"bfloat16": torch.bfloat16, "float32": torch.float32}[a.dtype]
    dev = torch.device("cuda:0")
    what = a.what.split(",")
    tot_ms, tot_fl = 0.0, 0.0
    if "conv" in what:
        for name, N, H, W, Cin, Cout, R, st, pad, dil, cnt in conv_shapes(a.frames):
            x = torch.randn((N, H, W, Cin), device=dev).to(dt)
            w = (torch.randn((Cout, R, R, Cin), device=dev) * 0.05).to(dt)
            sc = torch.ones((Cout,), device=dev); bi = torch.zeros((Cout,), device=dev)
            Ho = (H + 2 * pad - dil * (R - 1) - 1) // st + 1
            Wo = (W + 2 * pad - dil * (R - 1) - 1) // st + 1
            extra = ""
            for tl in [t for t in a.tiles.split(",") if t]:
                os.environ["MEGA_IGEMM_TILE"] = tl
                try:
                    tms = timeit(lambda: ops.conv2d_nhwc(x, w, sc, bi, stride=st, pad=pad, dil=dil, relu=True), rounds=5, inner=2)
                    extra += "  %s %.3f" % (tl, tms)
                except Exception as e:  # noqa: BLE001
                    extra += "  %s ERR" % tl
            os.environ.pop("MEGA_IGEMM_TILE", None)
            ms = timeit(lambda: ops.conv2d_nhwc(x, w, sc, bi, stride=st, pad=pad, dil=dil, relu=True))
            fl = 2.0 * N * Ho * Wo * Cout * R * R * Cin
            by = (x.numel() + w.numel() + N * Ho * Wo * Cout) * x.element_size()
            print("%-30s M=%7d N=%5d K=%6d  %8.3f ms  %7.1f TF/s  %7.1f GB/s  x%d%s" % (
                name, N * Ho * Wo, Cout, R * R * Cin, ms, fl / ms / 1e9, by / ms / 1e6, cnt, extra))
            tot_ms += ms * cnt
            tot_fl += fl * cnt
            del x, w
        print("conv total per %d-frame batch: %.2f ms, %.1f TF/s" % (a.frames, tot_ms, tot_fl / tot_ms / 1e9))
    if "attn" in what:
        for Nq, Nk, pos in [(2175, 750, False), (675, 3750, True), (675, 750, True), (300, 750, True), (300, 750, False)]:
            q = torch.randn((Nq, 1024), device=dev).to(dt); k = torch.randn((Nk, 1024), device=dev).to(dt)
            ldv = (Nk + 31) // 32 * 32
            vt = torch.randn((1024, ldv), device=dev).to(dt)
            p = torch.randn((16, Nq, ldv), device=dev) if pos else None
            ms = timeit(lambda: ops.relation_attention(q, k, vt, Nk, pos=p, resid=q))
            print("attention core Nq=%4d Nk=%4d pos=%d  %8.3f ms  %6.1f TF/s" % (Nq, Nk, pos, ms, 4.0 * Nq * Nk * 1024 / ms / 1e9))
            for M, what_ in ((Nq, "Wq"), (Nk, "Wk")):
                x = torch.randn((M, 1024), device=dev).to(dt); w = torch.randn((1024, 1024), device=dev).to(dt)
                ms = timeit(lambda: ops.linear(x, w))
                print("   proj %s M=%4d  %8.3f ms  %6.1f TF/s" % (what_, M, ms, 2.0 * M * 1024 * 1024 / ms / 1e9))
            x = torch.randn((Nk, 1024), device=dev).to(dt); w = torch.randn((1024, 1024), device=dev).to(dt)
            ms = timeit(lambda: ops.linear_transposed(w, x, ldv))
            print("   proj Vt M=%4d  %8.3f ms  %6.1f TF/s" % (Nk, ms, 2.0 * Nk * 1024 * 1024 / ms / 1e9))
    if "pos" in what:
        for Nq, Nk in [(675, 3750), (675, 750), (300, 750)]:
            bq = torch.rand((Nq, 4), device=dev) * 500; bq[:, 2:] += bq[:, :2]
            bk = torch.rand((Nk, 4), device=dev) * 500; bk[:, 2:] += bk[:, :2]
            wg = torch.randn((64, 16), device=dev) * 0.05; bg = torch.zeros((16,), device=dev)
            dm = torch.full((8,), 1000.0).pow(torch.arange(8) / 8.0).to(dev)
            ms = timeit(lambda: ops.position_logits(bq, bk, wg, bg, dm))
            print("pos_logits Nq=%4d Nk=%4d  %8.3f ms  %7.1f Mpairs/s" % (Nq, Nk, ms, Nq * Nk / ms / 1e3))
    if "roi" in what:
        B = a.frames
        feat = torch.randn((B, 38, 63, 2048), device=dev).to(dt)
        K = (B // 2) * 300 + (B - B // 2) * 75
        g = torch.Generator(device="cpu").manual_seed(0)
        c = torch.rand((K, 2), generator=g) * torch.tensor([900., 500.]); wh = torch.rand((K, 2), generator=g) * 300 + 16
        rois = torch.cat([torch.randint(0, B, (K, 1), generator=g).float(), (c - wh / 2).clamp(min=0), c + wh / 2], dim=1)
        rois[:, 3].clamp_(max=999); rois[:, 4].clamp_(max=599)
        rois = rois.to(dev)
        ms = timeit(lambda: ops.roi_align(feat, rois, 1 / 16., (7, 7), 0))
        print("roi_align K=%d C=2048  %8.3f ms  (out %.1f GB/s)" % (K, ms, K * 49 * 2048 * feat.element_size() / ms / 1e6))


if __name__ == "__main__":
    main()

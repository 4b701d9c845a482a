"""GPU check of igemm8 (LDS-DMA 8-phase kernel) against the register-staged igemm tiles: BIT equality is expected
(same MFMA instruction, same ascending-K order per output element), on conv shapes that exercise padding, stride,
dilation, M / N / K tails, residual, activations, f32 output and split-K; every case is run several times (race
screen: all runs must agree).  Usage (GPU box):  python tools/gpu/igemm8_check.py [--quick]
"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mega.pytorch_amd import ops  # noqa: E402

CASES = [
    # N, H, W, Cin, Cout, R, stride, pad, dil, relu, res, out_f32
    (2, 38, 63, 256, 256, 3, 1, 1, 1, 1, False, False),       # layer3 conv2 (small batch)
    (20, 38, 63, 256, 256, 3, 1, 1, 1, 1, False, False),      # layer3 conv2, bench batch
    (20, 38, 63, 1024, 256, 1, 1, 0, 1, 1, False, False),     # layer3 conv1
    (20, 38, 63, 256, 1024, 1, 1, 0, 1, 1, True, False),      # layer3 conv3 + residual (K = 256: 4 tiles)
    (3, 38, 63, 1024, 1024, 3, 1, 1, 1, 1, False, False),     # rpn conv
    (3, 38, 63, 512, 512, 3, 1, 2, 2, 1, False, False),       # res5 conv2 dilated
    (3, 38, 63, 512, 2048, 1, 1, 0, 1, 1, True, False),       # res5 conv3
    (2, 75, 125, 512, 1024, 1, 2, 0, 1, 0, False, False),     # stride-2 1x1 downsample
    (1, 19, 23, 128, 320, 3, 1, 1, 1, 2, False, False),       # N tail (320 = 256 + 64), K = 1152 (18 tiles), LeakyReLU
    (1, 17, 29, 192, 264, 1, 1, 0, 1, 0, False, True),        # K = 192 (3 tiles: odd), N tail not multiple of 256, f32 out
    (1, 13, 15, 128, 256, 1, 1, 0, 1, 1, True, False),        # K = 128 (2 tiles), M = 195 < one tile
    (700, 1, 1, 1024, 1024, 1, 1, 0, 1, 1, False, False),     # linear
    (375, 1, 1, 100352, 1024, 1, 1, 0, 1, 1, False, False),   # fc0: split-K
    (5, 9, 13, 192, 512, 4, 1, 3, 1, 2, False, False),        # 4x4 conv pad 3 (zero-stuffed deconv style)
]
CASES += [                                                     # K = 64: ONE K-tile (the ring's second tile is all zeros)
    (3, 150, 250, 64, 256, 1, 1, 0, 1, 1, True, False),       # layer1 conv3 + residual
    (3, 150, 250, 64, 256, 1, 1, 0, 1, 0, False, False),      # layer1 downsample
    (1, 13, 15, 64, 320, 1, 1, 0, 1, 1, False, True),         # N tail, f32 out, M < one tile
]


def run(case, force, reps=1):
    N, H, W, Cin, Cout, R, st, pad, dil, relu, use_res, f32o = case
    g = torch.Generator().manual_seed(sum(case[:6]))
    x = torch.randn((N, H, W, Cin), generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn((Cout, R, R, Cin), generator=g) / math.sqrt(Cin * R * R)).to(torch.bfloat16).cuda()
    sc = (torch.rand((Cout,), generator=g) + 0.5).cuda()
    bi = (torch.randn((Cout,), generator=g) * 0.1).cuda()
    Ho = (H + 2 * pad - dil * (R - 1) - 1) // st + 1
    Wo = (W + 2 * pad - dil * (R - 1) - 1) // st + 1
    res = torch.randn((N, Ho, Wo, Cout), generator=g).to(torch.bfloat16).cuda() if use_res else None
    if force:
        os.environ["MEGA_IGEMM_TILE"] = force
    else:
        os.environ.pop("MEGA_IGEMM_TILE", None)
    outs = []
    for _ in range(reps):
        outs.append(ops.conv2d_nhwc(x, w, sc, bi, res, stride=st, pad=pad, dil=dil, relu=relu,
                                    out_dtype=torch.float32 if f32o else None).clone())
    torch.cuda.synchronize()
    os.environ.pop("MEGA_IGEMM_TILE", None)
    return outs


def main():
    quick = "--quick" in sys.argv
    bad = 0
    for case in CASES:
        ref = run(case, "128x128")[0]
        line = "%-52s" % (case,)
        for force in ("8:256", "8:192"):
            outs = run(case, force, reps=2 if quick else 4)
            same_runs = all(torch.equal(outs[0], o) for o in outs[1:])
            eq = torch.equal(outs[0], ref)
            d = (outs[0].float() - ref.float()).abs().max().item()
            nan = not torch.isfinite(outs[0].float()).all().item()
            line += "  %s: %s maxdiff %.3g%s%s" % (force, "BIT-EQUAL" if eq else "DIFF", d, "" if same_runs else " RUN-TO-RUN-DIFF",
                                                  " NAN" if nan else "")
            if not eq or not same_runs:
                bad += 1
                nz = (outs[0].float() - ref.float()).abs().flatten()
                idx = torch.nonzero(nz > 0).flatten()
                if idx.numel():
                    C = outs[0].shape[-1]
                    rows = (idx // C)
                    line += " [bad elems %d, rows %d..%d, cols %d..%d]" % (idx.numel(), rows.min().item(), rows.max().item(),
                                                                        (idx % C).min().item(), (idx % C).max().item())
        print(line, flush=True)
    print("igemm8 check: %s" % ("ALL BIT-EQUAL" if bad == 0 else "%d MISMATCHES" % bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

"""GPU check + timing of stream1x1 (persistent, wave-owned 32 x 256 tiles, weights resident in LDS) against the register-staged
tiles (bit equality expected: same MFMA, same ascending K order, same epilogue arithmetic) and against igemm8 / igemm2 (time):
  python tools/gpu/stream1x1_check.py [--quick] [--time-only] [--f16]"""
import os
import sys

import torch

os.environ["MEGA_IGEMM2"] = "0"
os.environ["MEGA_STREAM1X1"] = "0"     # the natural dispatch of this process = igemm8 (the baseline column); the others are forced per run
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "gpu"))
import igemm8_check as c8  # noqa: E402
import igemm4_check as c4  # noqa: E402

CASES = [
    # N, H, W, Cin, Cout, R, stride, pad, dil, relu, res, out_f32
    (40, 38, 63, 256, 1024, 1, 1, 0, 1, 1, True, False),      # layer3 conv3 at the bench batch (M = 95760: a 16-row tail tile)
    (20, 38, 63, 256, 1024, 1, 1, 0, 1, 1, True, False),      # ... at 20 frames
    (7, 38, 63, 256, 1024, 1, 1, 0, 1, 0, True, False),       # M = 16758, no activation
    (9, 38, 63, 256, 1024, 1, 1, 0, 1, 2, False, False),      # LeakyReLU, no residual
    (40, 75, 125, 128, 512, 1, 1, 0, 1, 1, True, False),      # layer2 conv3 at the bench batch
    (3, 75, 125, 128, 512, 1, 1, 0, 1, 1, False, False),      # K = 128, no residual
    (8, 38, 63, 256, 256, 1, 1, 0, 1, 1, True, False),        # ONE N tile
    (8, 38, 63, 256, 2048, 1, 1, 0, 1, 1, True, False),       # eight N tiles
    (8, 38, 63, 128, 256, 1, 1, 0, 1, 0, False, False),       # K = 128, one N tile, no epilogue extras
]
HOT = [
    ("l3.conv3 1x1 256->1024 + res", 40, 38, 63, 256, 1024, 1, 1, 0, 1, True, False),
    ("l3.conv3 (20 frames)", 20, 38, 63, 256, 1024, 1, 1, 0, 1, True, False),
    ("l3.conv3 (10 frames)", 10, 38, 63, 256, 1024, 1, 1, 0, 1, True, False),
    ("l2.conv3 1x1 128->512 + res", 40, 75, 125, 128, 512, 1, 1, 0, 1, True, False),
    ("l2.conv3 (20 frames)", 20, 75, 125, 128, 512, 1, 1, 0, 1, True, False),
    ("l2.b0.down-like 1x1 256->512", 40, 75, 125, 256, 512, 1, 1, 0, 1, False, False),
]


def main():
    bad = 0
    if "--time-only" not in sys.argv:
        quick = "--quick" in sys.argv
        for case in CASES:
            ref = c8.run(case, "128x128")[0]
            line = "%-52s" % (case,)
            outs = c8.run(case, "s:32", reps=2 if quick else 4)
            same = all(torch.equal(outs[0], o) for o in outs[1:])
            eq = torch.equal(outs[0], ref)
            line += "  s:32: %s maxdiff %.3g%s" % ("BIT-EQUAL" if eq else "DIFF", (outs[0].float() - ref.float()).abs().max().item(),
                                                  "" if same else " RUN-TO-RUN-DIFF")
            if not eq or not same:
                bad += 1
                nz = (outs[0].float() - ref.float()).abs().flatten()
                idx = torch.nonzero(nz > 0).flatten()
                if idx.numel():
                    C = outs[0].shape[-1]
                    rows = idx // C
                    line += " [bad elems %d, rows %d..%d, cols %d..%d]" % (idx.numel(), rows.min().item(), rows.max().item(),
                                                                        (idx % C).min().item(), (idx % C).max().item())
            print(line, flush=True)
        print("stream1x1 check: %s" % ("ALL BIT-EQUAL" if bad == 0 else "%d MISMATCHES" % bad), flush=True)
    print("%-34s %32s %32s %32s" % ("layer (%s)" % str(c4.DT).split(".")[1], "igemm8 natural", "igemm2", "stream1x1"))
    for case in HOT:
        cols = []
        ref = None
        for env in ({"MEGA_IGEMM_TILE": "8:%d" % c4.natural_bm(case)}, {"MEGA_IGEMM_TILE": "2:128"}, {"MEGA_IGEMM_TILE": "s:32"}):
            ms, tf, out = c4.timed(case, env)
            if ref is None:
                ref = out
            ok = torch.equal(out, ref)
            name, N, H, W, Cin, Cout, R, st, pad, dil, use_res, f32o = case
            gb = (N * H * W * Cin * 2 + N * H * W * Cout * (2 + (2 if use_res else 0)) + Cout * Cin * 2) / 1e9
            cols.append("%.4f ms %5.0f TF/s %4.2f TB/s%s" % (ms, tf, gb / ms, "" if ok else " DIFF"))
        print("%-34s %32s %32s %32s" % ((case[0],) + tuple(cols)), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

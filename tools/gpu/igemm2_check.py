"""GPU check + timing of igemm2 (128 x 256 tiles, K-tile 32, 4 waves, two blocks per CU) against igemm8 and the register-staged tiles.
BIT equality is expected on every case of tools/gpu/igemm8_check.py (same MFMA, same ascending K order; shapes igemm2's epilogue
does not serve fall back to igemm8 inside the dispatcher) and on a few more shapes at the bench batch.  Then the hot
layers of a 40-frame batch, igemm8 vs igemm2 (MEGA_IGEMM2_PRE=1: residual rows of slab 0 requested before the K loop):
  python tools/gpu/igemm2_check.py [--quick] [--time-only] [--f16]"""
import os
import sys

import torch

os.environ["MEGA_IGEMM2"] = "0"      # the natural dispatch of this process = igemm8 (the baseline column); igemm2 is forced per run
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "gpu"))
import igemm8_check as c8  # noqa: E402
import igemm4_check as c4  # noqa: E402

EXTRA = [
    # N, H, W, Cin, Cout, R, stride, pad, dil, relu, res, out_f32
    (2, 19, 23, 64, 256, 3, 1, 1, 1, 1, True, False),         # 3x3 with residual, 18 K-tiles of 32
    (1, 17, 29, 64, 512, 1, 1, 0, 1, 0, False, True),         # K = 64, f32 out, two N tiles
    (3, 38, 63, 192, 256, 1, 1, 0, 1, 2, False, False),       # K = 192 (6 K-tiles), LeakyReLU
    (40, 38, 63, 256, 1024, 1, 1, 0, 1, 1, True, False),      # layer3 conv3 at the bench batch
    (40, 75, 125, 128, 512, 1, 1, 0, 1, 1, True, False),      # layer2 conv3 at the bench batch
]
HOT = c4.HOT + [
    ("l2.conv3 1x1 128->512 + res", 40, 75, 125, 128, 512, 1, 1, 0, 1, True, False),
    ("l2.conv1 1x1 512->128", 40, 75, 125, 512, 128, 1, 1, 0, 1, False, False),
    ("l1.conv3 1x1 64->256 + res", 40, 150, 250, 64, 256, 1, 1, 0, 1, True, False),
    ("l3.b0.down 1x1/2 512->1024", 40, 75, 125, 512, 1024, 1, 2, 0, 1, False, False),
    ("r5.b0.down 1x1 1024->2048", 40, 38, 63, 1024, 2048, 1, 1, 0, 1, False, False),
]


def main():
    bad = 0
    if "--time-only" not in sys.argv:
        quick = "--quick" in sys.argv
        for case in c8.CASES + EXTRA:
            ref = c8.run(case, "128x128")[0]
            line = "%-52s" % (case,)
            outs = c8.run(case, "2:128", reps=2 if quick else 4)
            same = all(torch.equal(outs[0], o) for o in outs[1:])
            eq = torch.equal(outs[0], ref)
            line += "  2:128: %s maxdiff %.3g%s" % ("BIT-EQUAL" if eq else "DIFF", (outs[0].float() - ref.float()).abs().max().item(),
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
        print("igemm2 check: %s" % ("ALL BIT-EQUAL" if bad == 0 else "%d MISMATCHES" % bad), flush=True)
    print("%-40s %22s %22s %22s" % ("layer (40-frame batch, %s)" % str(c4.DT).split(".")[1], "igemm8 natural", "igemm2", "igemm2 PRE"))
    for case in HOT:
        cols = []
        ref = None
        for env in ({"MEGA_IGEMM_TILE": "8:%d" % c4.natural_bm(case)}, {"MEGA_IGEMM_TILE": "2:128"}, {"MEGA_IGEMM_TILE": "2:128", "MEGA_IGEMM2_PRE": "1"}):
            if "MEGA_IGEMM2_PRE" in env and not case[10]:
                cols.append("")
                continue
            ms, tf, out = c4.timed(case, env)
            if ref is None:
                ref = out
            ok = torch.equal(out, ref)
            name, N, H, W, Cin, Cout, R, st, pad, dil, use_res, f32o = case
            Ho, Wo = (H + 2 * pad - dil * (R - 1) - 1) // st + 1, (W + 2 * pad - dil * (R - 1) - 1) // st + 1
            gb = (N * H * W * Cin * 2 / (st * st if R == 1 else 1) + N * Ho * Wo * Cout * ((4 if f32o else 2) + (2 if use_res else 0)) + Cout * R * R * Cin * 2) / 1e9
            cols.append("%.4f ms %5.0f TF/s %4.2f TB/s%s" % (ms, tf, gb / ms, "" if ok else " DIFF"))
        print("%-40s %30s %30s %30s" % ((case[0],) + tuple(cols)), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

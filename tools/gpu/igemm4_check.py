"""GPU check + timing of igemm4 (4 waves x 512 registers, 128 x 128 outputs per wave) against igemm8 and the register-staged tiles.
BIT equality is expected on every case of tools/gpu/igemm8_check.py (same MFMA, same ascending K order; shapes igemm4's epilogue
does not serve fall back to igemm8 inside the dispatcher).  Then the hot layers of a 40-frame batch, igemm8 vs igemm4:
  python tools/gpu/igemm4_check.py [--quick] [--time-only] [--f16]"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "gpu"))
from mega.pytorch_amd import ops  # noqa: E402
import igemm8_check as c8  # noqa: E402

DT = torch.float16 if "--f16" in sys.argv else torch.bfloat16

HOT = [
    # name, N, H, W, Cin, Cout, R, stride, pad, dil, res, out_f32
    ("l3.conv1 1x1 1024->256", 40, 38, 63, 1024, 256, 1, 1, 0, 1, False, False),
    ("l3.conv2 3x3 256->256", 40, 38, 63, 256, 256, 3, 1, 1, 1, False, False),
    ("l3.conv3 1x1 256->1024 + res", 40, 38, 63, 256, 1024, 1, 1, 0, 1, True, False),
    ("rpn conv 3x3 1024->1024", 40, 38, 63, 1024, 1024, 3, 1, 1, 1, False, False),
    ("res5.conv1 1x1 1024->512", 40, 38, 63, 1024, 512, 1, 1, 0, 1, False, False),
    ("res5.conv2 3x3 d2 512->512", 40, 38, 63, 512, 512, 3, 1, 2, 2, False, False),
    ("res5.conv3 1x1 512->2048 + res", 40, 38, 63, 512, 2048, 1, 1, 0, 1, True, False),
    ("fc0 7500 x 100352 -> 1024 (f32)", 7500, 1, 1, 100352, 1024, 1, 1, 0, 1, False, True),
    ("agg projection 37500 x 1024 -> 1024", 37500, 1, 1, 1024, 1024, 1, 1, 0, 1, False, False),
]


def timed(case, env, reps=20):
    name, N, H, W, Cin, Cout, R, st, pad, dil, use_res, f32o = case
    g = torch.Generator().manual_seed(1)
    x = torch.randn((N, H, W, Cin), generator=g).to(DT).cuda()
    w = (torch.randn((Cout, R, R, Cin), generator=g) / math.sqrt(Cin * R * R)).to(DT).cuda()
    sc = (torch.rand((Cout,), generator=g) + 0.5).cuda()
    bi = (torch.randn((Cout,), generator=g) * 0.1).cuda()
    Ho, Wo = (H + 2 * pad - dil * (R - 1) - 1) // st + 1, (W + 2 * pad - dil * (R - 1) - 1) // st + 1
    res = torch.randn((N, Ho, Wo, Cout), generator=g).to(DT).cuda() if use_res else None
    for k, v in env.items():
        os.environ[k] = v
    run = lambda: ops.conv2d_nhwc(x, w, sc, bi, res, stride=st, pad=pad, dil=dil, relu=True, out_dtype=torch.float32 if f32o else None)  # noqa: E731
    out = run()
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    for k in env:
        os.environ.pop(k, None)
    ms = e0.elapsed_time(e1) / reps
    return ms, 2.0 * N * Ho * Wo * Cout * R * R * Cin / (ms * 1e9), out


def ablate():
    """timing ablations of igemm4's K loop (experiments build: MEGA_BUILD_EXPERIMENTS=1), 256-row tile"""
    names = {0: "full", 1: "no DMA in the loop", 2: "no barrier", 3: "no vmcnt wait", 4: "no fragment reads", 5: "no MFMAs"}
    for case in (HOT[0], HOT[1], HOT[3]):
        line = "%-28s" % case[0]
        for abl in (0, 1, 2, 3, 4, 5):
            env = {"MEGA_IGEMM_TILE": "4:256"}
            if abl:
                env["MEGA_IGEMM4_ABLATE"] = str(abl)
            ms, tf, _ = timed(case, env)
            line += "  %s %.4f" % (names[abl], ms)
        ms, tf, _ = timed(case, {"MEGA_IGEMM_TILE": "8:256"})
        print(line + "  | igemm8 8:256 %.4f" % ms, flush=True)


def main():
    if "--ablate" in sys.argv:
        return ablate()
    bad = 0
    if "--time-only" not in sys.argv:
        quick = "--quick" in sys.argv
        for case in c8.CASES:
            ref = c8.run(case, "128x128")[0]
            line = "%-52s" % (case,)
            for force in ("4:256", "4:192"):
                outs = c8.run(case, force, reps=2 if quick else 4)
                same = all(torch.equal(outs[0], o) for o in outs[1:])
                eq = torch.equal(outs[0], ref)
                line += "  %s: %s maxdiff %.3g%s" % (force, "BIT-EQUAL" if eq else "DIFF", (outs[0].float() - ref.float()).abs().max().item(),
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
        print("igemm4 check: %s" % ("ALL BIT-EQUAL" if bad == 0 else "%d MISMATCHES" % bad), flush=True)
    print("%-40s %22s %22s %22s %22s" % ("layer (40-frame batch, %s)" % str(DT).split(".")[1], "igemm8 natural", "igemm4 natural", "igemm4 4:256", "igemm4 4:192"))
    for case in HOT:
        cols = []
        ref = None
        for env in ({"MEGA_IGEMM4_OFF_FOR_TIMING": "1"}, {}, {"MEGA_IGEMM_TILE": "4:256"}, {"MEGA_IGEMM_TILE": "4:192"}):
            if "MEGA_IGEMM4_OFF_FOR_TIMING" in env:
                env = {"MEGA_IGEMM_TILE": "8:%d" % natural_bm(case)}
            ms, tf, out = timed(case, env)
            if ref is None:
                ref = out
            ok = torch.equal(out, ref)
            cols.append("%.4f ms %6.0f TF/s%s" % (ms, tf, "" if ok else " DIFF"))
        print("%-40s %22s %22s %22s %22s" % ((case[0],) + tuple(cols)), flush=True)
    return 1 if bad else 0


def natural_bm(case):
    from mega.pytorch_amd import _lib
    name, N, H, W, Cin, Cout, R, st, pad, dil, use_res, f32o = case
    t = _lib.load().mega_conv2d_nhwc_plan_ex(N, H, W, Cin, Cout, R, R, st, pad, dil, Cout, int(use_res), 1, 0 if f32o else 1)
    return (t % 1000000) // 1000


if __name__ == "__main__":
    sys.exit(main())

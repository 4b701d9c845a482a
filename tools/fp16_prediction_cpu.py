#!/usr/bin/env python
"""Prediction BEFORE the kernels exist: what does IEEE half (fp16: 11 significant bits, the bf16 MFMA
rate and bytes) buy over bf16 (8 bits) on this path?  CPU emulation on the oracle-backed twins of the kernels
(tests/cpu_ops.py: f32 arithmetic on operands rounded to the 16-bit type, outputs rounded to the type the HIP kernel writes).

  part A  seeded fixture (tests/golden/oracle_r101_600x1000.npz, what bench.py runs): the table of
          tests/test_e2e_gpu.py::test_r101_bf16_attribution -- proposals matched, logit |err| median / p99, detections
          matched against the f32 oracle -- for   bf16 (all bf16, the headline)
                                                  f16+bf16head (fp16 frame stage, the bf16 head as it is)
                                                  f16 (fp16 frame stage AND fp16 operands in the head)
                                                  2pass-bound (exact activations, fp16-rounded weights: what a hi + lo
                                                  activation-plane form with single-rounded weights could reach at 2x the FLOPs)
  part B  calibrated fixture (score heads with margins): agreement of the 300 kept anchor indices with f32
  range   max |value| of every conv / linear OUTPUT before it is rounded (fp16 overflows at 65 504) and the smallest
          weight magnitudes (fp16 normals end at 6.1e-5; below that precision degrades gradually down to 6e-8)

  python tools/fp16_prediction_cpu.py [--nkey 3] [--variants bf16,f16+bf16head,f16] [--cal-frames 0,5,11] [--threads 8]
"""
import argparse
import os
import sys
import time

os.environ.setdefault("MEGA_STEM_POOL", "0")      # (the fused stem + pool kernel has no CPU twin; same bits as the two kernels)
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import cpu_ops  # noqa: E402
from mega.pytorch_amd import config, engine, modeling, ops, synth  # noqa: E402
from test_e2e_gpu import _bf16_metrics, _fmt, _r101_fixture  # noqa: E402

RANGE = {}          # tag -> [max |pre-rounding output| seen, layer shape that produced it]


def install_twins(tag_ref):
    for name in cpu_ops.ALL:
        setattr(ops, name, getattr(cpu_ops, name))
    conv = cpu_ops.conv2d_nhwc

    def conv_logged(x, w, scale=None, bias=None, residual=None, stride=1, pad=0, dil=1, relu=False, out_dtype=None, out=None):
        y32 = conv(x, w, scale, bias, residual, stride, pad, dil, relu, out_dtype=torch.float32)
        odt = out_dtype or x.dtype
        if x.dtype == torch.float16 or odt == torch.float16:
            r = RANGE.setdefault(tag_ref[0], [0.0, None, 0])
            m = float(y32.abs().max())
            if m > r[0]:
                r[0], r[1] = m, "%s -> %d" % (tuple(x.shape), w.shape[0])
            if odt == torch.float16:
                r[2] += int((y32.abs() > 65504.0).sum())
        y = y32.to(odt)
        if out is not None:
            out.copy_(y)
            return out
        return y
    ops.conv2d_nhwc = conv_logged
    cpu_ops.conv2d_nhwc = conv_logged          # (linear() of the twins goes through it)


def build(dtype, sd, **flags):
    cfg = config.get_cfg("R-101")
    cfg.DTYPE = dtype
    cfg.MODEL.DEVICE = "cpu"
    cfg.NMS_STRICT_GT = True
    rw = flags.pop("_round_weights", None)
    for k, v in flags.items():
        setattr(cfg, k, v)
    m = modeling.build_detection_model(cfg)
    if rw is not None:
        sd = {k: (v.to(rw).to(v.dtype) if (v.dim() >= 2 and v.is_floating_point()) else v) for k, v in sd.items()}
    m.load_state_dict(sd)
    m.eval()
    return m


VARIANTS = {"bf16": ("bfloat16", {}), "f16+bf16head": ("float16", {"HEAD_DTYPE": "bfloat16"}),
            "f16": ("float16", {"HEAD_DTYPE": "float16"}), "f32": ("float32", {}),
            "f16frame+f32head": ("float16", {"_head": "float32"}),      # the fp16 frame stage under the exact-f32 head
            # the BOUND of VERDICT r05 item 1c's two-pass form (activation planes hi + lo against single-rounded fp16 weights,
            # K' = 2K): exact activations, every weight matrix rounded to fp16 once
            "2pass-bound": ("float32", {"_round_weights": torch.float16})}


def weight_floor(sd):
    tiny = tot = 0
    for k, v in sd.items():
        if v.dim() >= 2:
            a = v.abs().float()
            tiny += int(((a < 6.1e-5) & (a > 0)).sum())
            tot += a.numel()
    return tiny, tot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nkey", type=int, default=3)
    ap.add_argument("--variants", default="bf16,f16+bf16head,f16")
    ap.add_argument("--cal-frames", default="0,5,11")
    ap.add_argument("--threads", type=int, default=8)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    tag = ["-"]
    install_twins(tag)

    # ---------------- part A
    d, gen = _r101_fixture()
    sd, clip, gfor = gen.inputs()
    nkey, T = min(args.nkey, int(d["cfg_nkey"])), int(d["cfg_T"])
    keep = [int(k) for k in d["keep"] if int(k) < nkey]
    tiny, tot = weight_floor(sd)
    print("seeded weights: %d of %d matrix elements (%.3f %%) lie below fp16's smallest normal 6.1e-5" % (tiny, tot, 100.0 * tiny / tot))
    for variant in [v for v in args.variants.split(",") if v]:
        dt, flags = VARIANTS[variant]
        tag[0] = "seeded/" + variant
        flags = dict(flags)
        head = flags.pop("_head", None)
        m = build(dt, sd, **flags)
        fm = None
        if head is not None:
            fm, m = m, build(head, sd)
        eng = engine.ClipEngine(m, steps_per_batch=4, keep_logits=True, frame_model=fm)
        t0 = time.time()
        with torch.no_grad():
            dets = eng.run(clip, T, gfor, first=0, last=nkey)
        print("== part A, variant %s (%.0fs)" % (variant, time.time() - t0), flush=True)
        for idx in keep:
            mm = _bf16_metrics(d, idx, eng.key_boxes_log[idx].numpy(), eng.logits_log[idx].numpy(), dets[idx])
            print(_fmt(variant, idx, mm), flush=True)
        if tag[0] in RANGE:
            r = RANGE[tag[0]]
            print("   fp16 range: max |output before rounding| %.4g at %s; %d values above 65504" % (r[0], r[1], r[2]), flush=True)
        del eng, m

    # ---------------- part B
    frames_b = [int(x) for x in args.cal_frames.split(",") if x != ""]
    if frames_b:
        import make_oracle_r101_calibrated as cal
        sdc, clipc, _ = cal.inputs()
        tiny, tot = weight_floor(sdc)
        print("calibrated weights: %d of %d matrix elements (%.3f %%) below 6.1e-5" % (tiny, tot, 100.0 * tiny / tot))
        fr = synth.preprocess_cpu(clipc)
        W, H = fr.shape[3], fr.shape[2]
        models = {"f32": build("float32", sdc), "bf16": build("bfloat16", sdc), "f16": build("float16", sdc)}

        def select(model, c4):
            rpn = model.rpn
            out = rpn.head.run(c4)
            B, Hh, Ww, _ = c4.shape
            cell = next(iter(rpn.anchor_generator.cell_anchors)).float().contiguous()
            r = ops.rpn_select(out, cell, Hh, Ww, rpn.anchor_generator.strides[0], rpn.pre_nms_top_n["key"], rpn.post_nms_top_n["key"],
                               rpn.nms_thresh, rpn.min_size, W, H, rpn.strict_gt, want_index=True)
            return r[3][0, :int(r[2][0])].tolist()
        res = {}
        for f in frames_b:
            img = fr[f:f + 1]
            c4, idx = {}, {}
            with torch.no_grad():
                for k, m in models.items():
                    tag[0] = "calibrated/" + k
                    c4[k] = m.backbone.body.forward(img)[0].permute(0, 2, 3, 1).contiguous()
                    idx[k] = select(m, c4[k])
            line = "part B frame %2d:" % f
            for k in ("bf16", "f16"):
                rel = ((c4[k].float() - c4["f32"]).abs().mean() / c4["f32"].abs().mean()).item()
                agree = len(set(idx[k]) & set(idx["f32"])) / max(len(idx["f32"]), 1)
                same = float(np.mean([a == b for a, b in zip(idx[k], idx["f32"])]))
                a75 = len(set(idx[k][:75]) & set(idx["f32"][:75])) / 75.0
                res.setdefault(k, []).append((agree, same, a75, rel))
                line += "  %s: C4 rel err %.2e, kept indices %.1f%% (same position %.1f%%, first 75: %.1f%%)" % (k, rel, 100 * agree, 100 * same, 100 * a75)
            print(line, flush=True)
        for k, v in res.items():
            a = np.array(v)
            print("part B mean, %s: kept indices %.1f%%, same position %.1f%%, first 75 %.1f%%, C4 rel err %.2e" % (
                k, 100 * a[:, 0].mean(), 100 * a[:, 1].mean(), 100 * a[:, 2].mean(), a[:, 3].mean()))
        if "calibrated/f16" in RANGE:
            r = RANGE["calibrated/f16"]
            print("   fp16 range (calibrated): max |output before rounding| %.4g at %s; %d values above 65504" % (r[0], r[1], r[2]))


if __name__ == "__main__":
    main()

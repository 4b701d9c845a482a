#!/usr/bin/env python
"""Where do the bf16 frame stage's proposal flips come from?  CPU emulation on the oracle-backed twins (tests/cpu_ops.py: f32
arithmetic on bf16-rounded operands, outputs rounded to the dtype the HIP kernel writes) with the CALIBRATED score heads
(tests/golden/oracle_r101_calibrated_600x1000.npz): kept-proposal anchor indices of a few frames for
  f32            the exact path (reference: the oracle's own selection)
  bf16           everything bf16 (what bench.py runs)
  bb16+rpn32     bf16 backbone, RPN head (3x3 conv + logits / deltas) in f32 on the bf16 C4 map
  bb32+rpn16     f32 backbone, C4 rounded to bf16 once, RPN head in bf16
  bb16+t32       bf16 backbone and bf16 RPN conv operands, but the conv's OUTPUT t kept f32 into the 1x1 logits (f32 weights)
  res11 / res15  everything bf16 EXCEPT the backbone's residual stream, carried with 11 / 15 mantissa bits (bf16 has 7): the
                 conv inputs are its bf16 rounding, the residual add reads / writes the wide value (a 4-bit / 8-bit remainder plane)
-> agreement of the 300 kept anchor indices with f32 (as sets, and of the first 75 = the reference-role rows).

  MEGA_STEM_POOL=0 python tools/frame_precision_cpu.py [--frames 0,5,11] [--threads 16]
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
from mega.pytorch_amd import config, modeling, ops, synth  # noqa: E402
import make_oracle_r101_calibrated as cal  # noqa: E402


def install_twins():
    for name in cpu_ops.ALL:
        setattr(ops, name, getattr(cpu_ops, name))
    for name in getattr(cpu_ops, "EXTRA", []):
        setattr(ops, name, getattr(cpu_ops, name))


def build(dtype, sd):
    cfg = config.get_cfg("R-101")
    cfg.DTYPE = dtype
    cfg.MODEL.DEVICE = "cpu"
    cfg.NMS_STRICT_GT = True
    m = modeling.build_detection_model(cfg)
    m.load_state_dict(sd)
    m.eval()
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", default="0,5,11")
    ap.add_argument("--threads", type=int, default=16)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    install_twins()
    sd, clip, _ = cal.inputs()
    frames = synth.preprocess_cpu(clip)
    m32, m16 = build("float32", sd), build("bfloat16", sd)
    W, H = frames.shape[3], frames.shape[2]

    def select(model, c4_nhwc, head_dtype=None, t32=False):
        rpn = model.rpn
        if t32:
            # bf16 conv operands, f32 output t, f32 logits on it
            pk16 = rpn.head._packed(torch.bfloat16, c4_nhwc.device)
            pk32 = m32.rpn.head._packed(torch.float32, c4_nhwc.device)
            t = ops.conv2d_nhwc(c4_nhwc, pk16["w1"], None, pk16["b1"], pad=1, relu=True, out_dtype=torch.float32)
            out = ops.conv2d_nhwc(t, pk32["w2"], None, pk32["b2"], out_dtype=torch.float32)
        else:
            out = rpn.head.run(c4_nhwc)
        B, Hh, Ww, _ = c4_nhwc.shape
        cell = next(iter(rpn.anchor_generator.cell_anchors)).float().contiguous()
        r = ops.rpn_select(out, cell, Hh, Ww, rpn.anchor_generator.strides[0], rpn.pre_nms_top_n["key"], rpn.post_nms_top_n["key"],
                           rpn.nms_thresh, rpn.min_size, W, H, rpn.strict_gt, want_index=True)
        return r[3][0, :int(r[2][0])].tolist(), out

    def round_bits(v, bits):
        """f32 -> nearest value with `bits` mantissa bits (round to nearest even on the dropped bits)"""
        i = v.contiguous().view(torch.int32)
        drop = 23 - bits
        r = ((i >> drop) & 1) + ((1 << (drop - 1)) - 1)
        return (((i + r) >> drop) << drop).view(torch.float32)

    def backbone_wide(img, bits):
        """the bf16 backbone with the RESIDUAL stream carried at `bits` mantissa bits (bf16 = 7): conv inputs are the bf16
        rounding of the stream, the residual add reads and writes the wide value (a bf16 plane + a remainder plane)"""
        body = m16.backbone.body
        y = body.stem.run(img.float().contiguous(), torch.bfloat16).float()     # (the stem's output is bf16 either way)
        for name in body.stages:
            for blk in getattr(body, name):
                pk = blk._packed(torch.bfloat16, y.device)
                x16 = y.to(torch.bfloat16)
                ident = y
                if blk.downsample is not None:
                    ident = round_bits(ops.conv2d_nhwc(x16, pk["wd"], pk["sd"], pk["bd"], stride=blk.down_stride,
                                                       out_dtype=torch.float32), bits)
                t = ops.conv2d_nhwc(x16, pk["w1"], pk["s1"], pk["b1"], stride=blk.stride, relu=True)
                t = ops.conv2d_nhwc(t, pk["w2"], pk["s2"], pk["b2"], pad=blk.dilation, dil=blk.dilation, relu=True)
                o = ops.conv2d_nhwc(t, pk["w3"], pk["s3"], pk["b3"], out_dtype=torch.float32)
                y = round_bits(torch.relu(o + ident), bits)
        return y                                                               # NHWC f32 (wide)

    res = {}
    for f in [int(x) for x in a.frames.split(",")]:
        t0 = time.time()
        img = frames[f:f + 1]
        with torch.no_grad():
            c32 = m32.backbone.body.forward(img)[0]      # logical NCHW view of NHWC f32
            c16 = m16.backbone.body.forward(img)[0]
            n32 = c32.permute(0, 2, 3, 1).contiguous()
            n16 = c16.permute(0, 2, 3, 1).contiguous()
            ref, lo32 = select(m32, n32)
            rows = {"bf16": select(m16, n16)[0],
                    "bb16+rpn32": select(m32, n16.float())[0],
                    "bb32+rpn16": select(m16, n32.to(torch.bfloat16))[0],
                    "bb16+t32": select(m16, n16, t32=True)[0]}
            for bits in (11, 15):
                nw = backbone_wide(img, bits)
                rows["res%d" % bits] = select(m16, nw.to(torch.bfloat16))[0]
                if bits == 15:
                    relw = ((nw - n32).abs().mean() / n32.abs().mean()).item()
        rel = ((n16.float() - n32).abs().mean() / n32.abs().mean()).item()
        line = "frame %2d (%.0fs): C4 mean |err| / mean |x|: bf16 %.2e, 15-bit residual stream %.2e;" % (f, time.time() - t0, rel, relw)
        for k, v in rows.items():
            agree = len(set(v) & set(ref)) / max(len(ref), 1)
            a75 = len(set(v[:75]) & set(ref[:75])) / 75.0
            res.setdefault(k, []).append(agree)
            line += "  %s %.1f%% (first 75: %.1f%%)" % (k, 100 * agree, 100 * a75)
        print(line, flush=True)
    print("mean agreement of the kept anchor indices with f32:", {k: "%.1f%%" % (100 * np.mean(v)) for k, v in res.items()})


if __name__ == "__main__":
    main()

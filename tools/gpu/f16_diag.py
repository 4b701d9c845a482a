"""Per-stage error of the bf16 / f16 frame stage against the exact-f32 one on ONE frame of the R-101 600x1000 fixture clip, the
same 300 proposals (the f32 model's) through every model's ROIAlign + fc0: where does a 16-bit mode lose its bits?
CPU-twin reference (tools/fp16_prediction_cpu.py's models, /tmp diag of round 6): f16 stem+pool 1.8e-4, layer1 4.4e-4, layer2
5.0e-4, layer3 6.8e-4, res5 6.9e-4, pooled 3.6e-4, fc0 4.8e-4, rpn_out 9.4e-4 (bf16: 8x those)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mega.pytorch_amd import config, modeling, ops  # noqa: E402
from test_e2e_gpu import _r101_fixture  # noqa: E402

dev = torch.device("cuda:0")
d, gen = _r101_fixture()
sd, clip, gfor = gen.inputs()


def build(dt):
    cfg = config.get_cfg("R-101")
    cfg.DTYPE = dt
    cfg.MODEL.DEVICE = "cuda:0"
    cfg.NMS_STRICT_GT = True
    m = modeling.build_detection_model(cfg)
    m.load_state_dict(sd)
    return cfg, m.to(dev)


frame = int(sys.argv[1]) if len(sys.argv) > 1 else 0
u8 = clip[frame:frame + 1].to(dev)
outs = {}
for dt in ("float32", "bfloat16", "float16"):
    cfg, m = build(dt)
    mean, bgr = tuple(cfg.INPUT.PIXEL_MEAN), bool(cfg.INPUT.TO_BGR255)
    img = ops.preprocess_frames(u8, mean, bgr)
    body = m.backbone.body
    tr = []
    with torch.no_grad():
        for src in (("u8", None), ("f32img", img)):
            if dt == "float32" and src[0] == "u8":
                continue
            if src[0] == "u8":
                y = body.stem.run_u8(u8, mean, bgr, body.dtype)
            else:
                y = body.stem.run(img, body.dtype)
            tr.append(("stem+pool(%s)" % src[0], y.float()))
        for name in body.stages:
            for blk in getattr(body, name):
                y = blk.run(y)
            tr.append((name, y.float()))
        fe = m.roi_heads.box.feature_extractor
        x5 = fe.res5_features(y)
        tr.append(("res5", x5.float()))
        if dt == "float32":
            a = m.frame_stage_a1(y, u8.shape[2], u8.shape[1])
            props = a["props"][0].clone()
        rois5 = torch.cat([torch.zeros((props.shape[0], 1), device=dev), props], 1)
        pooled = ops.roi_align(x5, rois5, fe.scale, (7, 7), fe.sampling_ratio)
        tr.append(("pooled", pooled.float()))
        tr.append(("fc0", fe.pooled_fc(x5, rois5).float()))
        tr.append(("rpn_out", m.rpn.head.run(y).float()))
    outs[dt] = dict(tr)
    del m
    torch.cuda.empty_cache()
ref = outs["float32"]
for name in outs["float16"]:
    r = ref[name if name in ref else "stem+pool(f32img)"]
    line = "%-18s max %.3g |" % (name, float(r.abs().max()))
    for k in ("bfloat16", "float16"):
        t = outs[k][name]
        line += "  %s mean rel %.2e" % (k, float((t - r).abs().mean() / r.abs().mean()))
    print(line)

"""Seeded synthetic inputs for parity tests and the benchmark (no datasets / checkpoints offline):
a calibrated random ``state_dict`` with the reference's key names and shapes and a
synthetic ImageNet-VID-style clip.

The reference's default init + FrozenBN identity stats makes activations explode over 33 residual
blocks and saturates every logit, so parity tests would never exercise top-k / NMS / softmax paths.
Calibration (scale only, architecture untouched): damped bn3 gains, random BN statistics, scaled stem
and attention projections.  The same dict loads into the reference model via load_state_dict.
"""
import math

import numpy as np
import torch

FE = "roi_heads.box.feature_extractor."


def _kaiming_uniform(gen, shape, a=1.0):
    fan_in = int(np.prod(shape[1:]))
    gain = math.sqrt(2.0 / (1 + a * a))
    bound = gain * math.sqrt(3.0 / fan_in)
    return (torch.rand(shape, generator=gen) * 2 - 1) * bound


def _normal(gen, shape, std):
    return torch.randn(shape, generator=gen) * std


def _bn(sd, p, n, gen, gain=(0.5, 1.0)):
    sd[p + "weight"] = torch.rand(n, generator=gen) * (gain[1] - gain[0]) + gain[0]
    sd[p + "bias"] = _normal(gen, (n,), 0.1)
    sd[p + "running_mean"] = _normal(gen, (n,), 0.1)
    sd[p + "running_var"] = torch.rand(n, generator=gen) * 1.5 + 0.5


def _bottleneck(sd, p, cin, cmid, cout, gen):
    if cin != cout:
        sd[p + "downsample.0.weight"] = _kaiming_uniform(gen, (cout, cin, 1, 1))
        _bn(sd, p + "downsample.1.", cout, gen, (0.6, 0.9))
    sd[p + "conv1.weight"] = _kaiming_uniform(gen, (cmid, cin, 1, 1))
    _bn(sd, p + "bn1.", cmid, gen, (0.9, 1.4))
    sd[p + "conv2.weight"] = _kaiming_uniform(gen, (cmid, cmid, 3, 3))
    _bn(sd, p + "bn2.", cmid, gen, (0.9, 1.4))
    sd[p + "conv3.weight"] = _kaiming_uniform(gen, (cout, cmid, 1, 1))
    _bn(sd, p + "bn3.", cout, gen, (0.2, 0.4))


def make_state_dict(blocks=(3, 4, 23), reduce_channel=False, stage=3, global_res_stage=1, num_classes=31,
                    pooler_resolution=7, seed=0, anchor_sizes=(64, 128, 256, 512), aspect_ratios=(0.5, 1.0, 2.0),
                    anchor_stride=16):
    """Calibrated random weights with the reference's MEGA state_dict layout (R-101: blocks (3,4,23),
    R-50: (3,4,6) + reduce_channel, global_res_stage 0)."""
    gen = torch.Generator().manual_seed(seed)
    sd = {}
    p = "backbone.body."
    sd[p + "stem.conv1.weight"] = _kaiming_uniform(gen, (64, 3, 7, 7)) * 0.02
    _bn(sd, p + "stem.bn1.", 64, gen, (0.8, 1.2))
    cin = 64
    for li, nb in enumerate(blocks):
        cmid, cout = 64 * 2 ** li, 256 * 2 ** li
        for b in range(nb):
            _bottleneck(sd, "%slayer%d.%d." % (p, li + 1, b), cin, cmid, cout, gen)
            cin = cout
    # RPN (rpn/rpn.py:73-106): 3x3 conv + two 1x1 convs, A = len(sizes)*len(ratios)
    A = len(anchor_sizes) * len(aspect_ratios)
    sd["rpn.anchor_generator.cell_anchors.0"] = _cell_anchors(anchor_stride, anchor_sizes, aspect_ratios)
    sd["rpn.head.conv.weight"] = _normal(gen, (1024, 1024, 3, 3), 0.01)
    sd["rpn.head.conv.bias"] = _normal(gen, (1024,), 0.01)
    sd["rpn.head.cls_logits.weight"] = _normal(gen, (A, 1024, 1, 1), 0.02)
    sd["rpn.head.cls_logits.bias"] = _normal(gen, (A,), 0.01)
    sd["rpn.head.bbox_pred.weight"] = _normal(gen, (4 * A, 1024, 1, 1), 0.01)
    sd["rpn.head.bbox_pred.bias"] = _normal(gen, (4 * A,), 0.01)
    # res5 head
    cin = 1024
    for b in range(3):
        _bottleneck(sd, "%shead.layer4.%d." % (FE, b), cin, 512, 2048, gen)
        cin = 2048
    pooled_c = 2048
    if reduce_channel:
        sd[FE + "conv.weight"] = _kaiming_uniform(gen, (256, 2048, 1, 1))
        sd[FE + "conv.bias"] = _normal(gen, (256,), 0.01)
        pooled_c = 256
    fc_in = pooled_c * pooler_resolution ** 2
    for i in range(stage):
        sd["%sl_fcs.%d.weight" % (FE, i)] = _kaiming_uniform(gen, (1024, fc_in if i == 0 else 1024))
        sd["%sl_fcs.%d.bias" % (FE, i)] = _normal(gen, (1024,), 0.01)
        sd["%sl_Wgs.%d.weight" % (FE, i)] = _normal(gen, (16, 64, 1, 1), 0.05)
        sd["%sl_Wgs.%d.bias" % (FE, i)] = _normal(gen, (16,), 0.02)
        _attn(sd, "l_", i, gen)
    for i in range(global_res_stage + 1):
        _attn(sd, "g_", i, gen)
    sd["roi_heads.box.predictor.cls_score.weight"] = _normal(gen, (num_classes, 1024), 0.05)
    sd["roi_heads.box.predictor.cls_score.bias"] = _normal(gen, (num_classes,), 0.1)
    sd["roi_heads.box.predictor.bbox_pred.weight"] = _normal(gen, (num_classes * 4, 1024), 0.01)
    sd["roi_heads.box.predictor.bbox_pred.bias"] = _normal(gen, (num_classes * 4,), 0.01)
    return sd


def _attn(sd, kind, i, gen):
    sd["%s%sWqs.%d.weight" % (FE, kind, i)] = _kaiming_uniform(gen, (1024, 1024)) * 0.3
    sd["%s%sWqs.%d.bias" % (FE, kind, i)] = _normal(gen, (1024,), 0.01)
    sd["%s%sWks.%d.weight" % (FE, kind, i)] = _kaiming_uniform(gen, (1024, 1024)) * 0.3
    sd["%s%sWks.%d.bias" % (FE, kind, i)] = _normal(gen, (1024,), 0.01)
    sd["%s%sWvs.%d.weight" % (FE, kind, i)] = _normal(gen, (1024, 1024, 1, 1), 0.02)
    sd["%s%sWvs.%d.bias" % (FE, kind, i)] = _normal(gen, (1024,), 0.01)
    sd["%s%sus.%d" % (FE, kind, i)] = _normal(gen, (16, 1, 64), 0.05)


def _cell_anchors(stride, sizes, aspect_ratios):
    """mega_core/modeling/rpn/anchor_generator.py:220-289 generate_anchors (Detectron rounding)."""
    def whctrs(a):
        w = a[2] - a[0] + 1; h = a[3] - a[1] + 1
        return w, h, a[0] + 0.5 * (w - 1), a[1] + 0.5 * (h - 1)

    def mk(ws, hs, xc, yc):
        ws = ws[:, None]; hs = hs[:, None]
        return np.hstack((xc - 0.5 * (ws - 1), yc - 0.5 * (hs - 1), xc + 0.5 * (ws - 1), yc + 0.5 * (hs - 1)))

    scales = np.array(sizes, dtype=np.float64) / stride
    ratios = np.array(aspect_ratios, dtype=np.float64)
    w, h, xc, yc = whctrs(np.array([1, 1, stride, stride], dtype=np.float64) - 1)
    ws = np.round(np.sqrt((w * h) / ratios)); hs = np.round(ws * ratios)
    ra = mk(ws, hs, xc, yc)
    out = []
    for i in range(ra.shape[0]):
        w, h, xc, yc = whctrs(ra[i])
        out.append(mk(w * scales, h * scales, xc, yc))
    return torch.from_numpy(np.vstack(out)).float()


def make_clip(T, H=600, W=1000, seed=0):
    """uint8 [T,H,W,3] RGB: low-frequency noise background + a few moving textured rectangles
    (SURVEY.md 8d 'Synthetic inputs')."""
    rng = np.random.RandomState(seed)
    gh, gw = max(H // 32, 2), max(W // 32, 2)
    base = torch.from_numpy(rng.rand(1, 3, gh, gw).astype(np.float32))
    bg = torch.nn.functional.interpolate(base, size=(H, W), mode="bilinear", align_corners=False)[0]
    nobj = 5
    pos = rng.rand(nobj, 2) * np.array([W * 0.7, H * 0.7])
    vel = (rng.rand(nobj, 2) - 0.5) * np.array([W, H]) * 0.02
    size = (rng.rand(nobj, 2) * 0.25 + 0.08) * np.array([W, H])
    col = rng.rand(nobj, 3).astype(np.float32)
    frames = torch.empty((T, H, W, 3), dtype=torch.uint8)
    for t in range(T):
        img = bg.clone()
        drift = torch.from_numpy(rng.rand(1, 3, gh, gw).astype(np.float32))
        img = 0.85 * img + 0.15 * torch.nn.functional.interpolate(drift, size=(H, W), mode="bilinear",
                                                                 align_corners=False)[0]
        for o in range(nobj):
            x0 = int(np.clip(pos[o, 0] + vel[o, 0] * t, 0, W - 2)); y0 = int(np.clip(pos[o, 1] + vel[o, 1] * t, 0, H - 2))
            x1 = int(min(W, x0 + size[o, 0])); y1 = int(min(H, y0 + size[o, 1]))
            stripes = ((torch.arange(x0, x1) // 6) % 2).float().view(1, 1, -1) * 0.25
            img[:, y0:y1, x0:x1] = torch.from_numpy(col[o]).view(3, 1, 1) * 0.75 + stripes
        frames[t] = (img.clamp(0, 1) * 255).round().to(torch.uint8).permute(1, 2, 0)
    return frames


PIXEL_MEAN = (102.9801, 115.9465, 122.7717)  # config/defaults.py:51 (BGR)


def preprocess_cpu(frames_u8):
    """CPU form of the reference test transform for frames already at target size
    (data/transforms/transforms.py:83-129: ToTensor, BGR*255, minus mean): uint8 [T,H,W,3] -> f32 [T,3,H,W]."""
    x = frames_u8.permute(0, 3, 1, 2).float() / 255.0
    x = x[:, [2, 1, 0]] * 255.0
    return x - torch.tensor(PIXEL_MEAN).view(1, 3, 1, 1)


def make_dff_state_dict(blocks=(3, 4, 6), reduce_channel=True, num_classes=31, seed=0):
    """FGFA layout minus embednet.*, plus flownet.Convolution5_scale.weight [1024,194,1,1] (flownet.py:36-38)."""
    sd = {k: v for k, v in make_fgfa_state_dict(blocks, reduce_channel, num_classes, seed).items()
          if not k.startswith("embednet.")}
    gen = torch.Generator().manual_seed(seed + 3000)
    sd["flownet.Convolution5_scale.weight"] = _normal(gen, (1024, 194, 1, 1), 0.02)
    return sd


def make_fgfa_state_dict(blocks=(3, 4, 6), reduce_channel=True, num_classes=31, seed=0):
    """Calibrated random weights with the reference's FGFA state_dict layout
    (backbone.*, flownet.*, embednet.*, rpn.*, roi_heads.box.feature_extractor.{head,conv,fc6,fc7}, predictor)."""
    base = make_state_dict(blocks=blocks, reduce_channel=reduce_channel, stage=1, global_res_stage=0,
                           num_classes=num_classes, seed=seed)
    sd = {k: v for k, v in base.items() if not any(t in k for t in (".l_", ".g_"))}
    gen = torch.Generator().manual_seed(seed + 1000)

    def conv(name, co, ci, k, gain=1.0, transposed=False):
        shape = (ci, co, k, k) if transposed else (co, ci, k, k)
        fan_in = ci * k * k
        sd[name + ".weight"] = (torch.rand(shape, generator=gen) * 2 - 1) * math.sqrt(3.0 / fan_in) * gain
        sd[name + ".bias"] = _normal(gen, (co,), 0.02)
    for name, ci, co, k in [("flow_conv1", 6, 64, 7), ("conv2", 64, 128, 5), ("conv3", 128, 256, 5), ("conv3_1", 256, 256, 3),
                            ("conv4", 256, 512, 3), ("conv4_1", 512, 512, 3), ("conv5", 512, 512, 3),
                            ("conv5_1", 512, 512, 3), ("conv6", 512, 1024, 3), ("conv6_1", 1024, 1024, 3)]:
        conv("flownet." + name, co, ci, k, gain=1.6)
    for name, ci in [("Convolution1", 1024), ("Convolution2", 1026), ("Convolution3", 770), ("Convolution4", 386),
                     ("Convolution5", 194)]:
        conv("flownet." + name, 2, ci, 3, gain=2.0)
    for name, ci, co in [("deconv5", 1024, 512), ("deconv4", 1026, 256), ("deconv3", 770, 128), ("deconv2", 386, 64)]:
        conv("flownet." + name, co, ci, 4, gain=1.6, transposed=True)
    for name in ("upsample_flow6to5", "upsample_flow5to4", "upsample_flow4to3", "upsample_flow3to2"):
        conv("flownet." + name, 2, 2, 4, gain=1.0, transposed=True)
    conv("embednet.embed_conv1", 512, 1024, 1)
    conv("embednet.embed_conv2", 512, 512, 3)
    conv("embednet.embed_conv3", 2048, 512, 1)
    pooled_c = 256 if reduce_channel else 2048
    sd[FE + "fc6.weight"] = _kaiming_uniform(gen, (1024, pooled_c * 49))
    sd[FE + "fc6.bias"] = _normal(gen, (1024,), 0.01)
    sd[FE + "fc7.weight"] = _kaiming_uniform(gen, (1024, 1024))
    sd[FE + "fc7.bias"] = _normal(gen, (1024,), 0.01)
    return sd


def make_rdn_state_dict(blocks=(3, 4, 6), reduce_channel=True, base_stage=2, advanced_stage=0, num_classes=31, seed=0):
    """Calibrated random weights with the reference's RDN state_dict layout (RDNFeatureExtractor:
    fcs.i / Wgs.i / Wqs.i / Wks.i / Wvs.i, roi_box_feature_extractors.py:311-333)."""
    base = make_state_dict(blocks=blocks, reduce_channel=reduce_channel, stage=1, global_res_stage=0,
                           num_classes=num_classes, seed=seed)
    sd = {k: v for k, v in base.items() if not any(t in k for t in (".l_", ".g_"))}
    gen = torch.Generator().manual_seed(seed + 2000)
    pooled_c = 256 if reduce_channel else 2048
    n_fc = base_stage + advanced_stage
    n_att = base_stage + advanced_stage + 1 if advanced_stage > 0 else base_stage
    for i in range(n_fc):
        sd["%sfcs.%d.weight" % (FE, i)] = _kaiming_uniform(gen, (1024, pooled_c * 49 if i == 0 else 1024))
        sd["%sfcs.%d.bias" % (FE, i)] = _normal(gen, (1024,), 0.01)
    for i in range(n_att):
        sd["%sWgs.%d.weight" % (FE, i)] = _normal(gen, (16, 64, 1, 1), 0.05)
        sd["%sWgs.%d.bias" % (FE, i)] = _normal(gen, (16,), 0.02)
        sd["%sWqs.%d.weight" % (FE, i)] = _kaiming_uniform(gen, (1024, 1024)) * 0.3
        sd["%sWqs.%d.bias" % (FE, i)] = _normal(gen, (1024,), 0.01)
        sd["%sWks.%d.weight" % (FE, i)] = _kaiming_uniform(gen, (1024, 1024)) * 0.3
        sd["%sWks.%d.bias" % (FE, i)] = _normal(gen, (1024,), 0.01)
        sd["%sWvs.%d.weight" % (FE, i)] = _normal(gen, (1024, 1024, 1, 1), 0.02)
        sd["%sWvs.%d.bias" % (FE, i)] = _normal(gen, (1024,), 0.01)
    return sd

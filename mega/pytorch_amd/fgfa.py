"""FGFA detector (BASELINE configs[4]) on the HIP kernels: host mirror of

  mega_core/modeling/detector/generalized_rcnn_fgfa.py:21-219   GeneralizedRCNNFGFA (test path)
  mega_core/modeling/backbone/flownet.py:16-118                 FlowNetS
  mega_core/modeling/backbone/embednet.py:8-24                  EmbedNet
  mega_core/modeling/roi_heads/box_head/roi_box_feature_extractors.py:55-118  ResNetConv52MLPFeatureExtractor

with the reference's module / parameter names (state_dict compatible: flownet.*, embednet.*,
roi_heads.box.feature_extractor.{head,conv,fc6,fc7}.*).

Kernel mapping: every Conv2d / ConvTranspose2d / Linear is the implicit-GEMM MFMA kernel (igemm.hip).  A
ConvTranspose2d(Cin, C, 4, stride=2) is ONE sub-pixel GEMM (its four output phases = a 2 x 2 / pad 1 conv with 4 C output
columns, ops.deconv4x4s2_into) that writes its cropped result straight into the level's concatenation buffer; the 2-channel
flow up-sampling, the skip connection and the zero padding of that buffer are one more kernel (ops.flow_level_assemble).
`FlowNetS.subpixel = False` keeps the literal form (stride-1 4x4 conv with the flipped kernel over the zero-stuffed input,
crop, cat, pad: 4x the matrix work) -- tests compare the two.  Channel counts that are not a multiple of the GEMM K-vector
(6, 2, 1026, 770, 386, 194) are zero-padded (weights too), which changes no result.  The flow-guided warp + cosine weights + softmax + sum is ONE fused kernel
(fgfa.hip).  torch is used for memory plumbing only (concat / crop / zero-stuffing copies).
"""
from collections import deque

import torch
from torch import nn

from . import ops
from .modeling import (DETECTION_META_ARCHITECTURES, ROI_BOX_FEATURE_EXTRACTORS, ROI_BOX_PREDICTOR, CombinedROIHeads,
                       PostProcessor, ResNetHead, _Packed, _nhwc, _nchw_view, _pack_conv, build_backbone, build_rpn,
                       compute_dtype, convert_to_roi_format)
from .structures import BoxList, to_image_list


def _mult(dtype):
    return 64 if dtype == torch.bfloat16 else 32


def _padc(x, mult):
    """zero-pad the channel (last) dimension of an NHWC tensor to a multiple of `mult`."""
    c = x.shape[-1]
    cp = (c + mult - 1) // mult * mult
    if cp == c:
        return x.contiguous()
    z = torch.zeros(x.shape[:-1] + (cp - c,), dtype=x.dtype, device=x.device)
    return torch.cat([x, z], dim=-1).contiguous()


def _pack_w(w_oihw, dtype, mult, scale=1.0):
    """OIHW -> OHWI with Cin zero-padded to a multiple of mult."""
    w = (w_oihw.detach().float() * scale).permute(0, 2, 3, 1)
    return _padc(w, mult).to(dtype).contiguous()


def _zero_stuff(x):
    """[N,H,W,C] -> [N,2H-1,2W-1,C] with the input on the even positions (ConvTranspose2d stride 2 as a conv)."""
    N, H, W, C = x.shape
    z = torch.zeros((N, 2 * H - 1, 2 * W - 1, C), dtype=x.dtype, device=x.device)
    z[:, ::2, ::2] = x
    return z


def _crop_like(x, target):
    """flownet.py:9-13 crop_like on NHWC."""
    if x.shape[1:3] == target.shape[1:3]:
        return x
    return x[:, 1:target.shape[1] + 1, 1:target.shape[2] + 1].contiguous()


class FlowNetS(_Packed):
    """backbone/flownet.py:16-118 (method "fgfa": returns Convolution5 * 2.5)."""

    CONVS = [("flow_conv1", 6, 64, 7, 2, 3), ("conv2", 64, 128, 5, 2, 2), ("conv3", 128, 256, 5, 2, 2),
             ("conv3_1", 256, 256, 3, 1, 1), ("conv4", 256, 512, 3, 2, 1), ("conv4_1", 512, 512, 3, 1, 1),
             ("conv5", 512, 512, 3, 2, 1), ("conv5_1", 512, 512, 3, 1, 1), ("conv6", 512, 1024, 3, 2, 1),
             ("conv6_1", 1024, 1024, 3, 1, 1)]
    PREDS = [("Convolution1", 1024), ("Convolution2", 1026), ("Convolution3", 770), ("Convolution4", 386),
             ("Convolution5", 194)]
    DECONVS = [("deconv5", 1024, 512), ("deconv4", 1026, 256), ("deconv3", 770, 128), ("deconv2", 386, 64)]
    UPS = ["upsample_flow6to5", "upsample_flow5to4", "upsample_flow4to3", "upsample_flow3to2"]
    subpixel = True        # the refinement levels' transposed convs as sub-pixel GEMMs (False: zero-stuffed, the literal form)

    def __init__(self, cfg):
        super().__init__()
        self.method = cfg.MODEL.VID.METHOD
        if self.method == "dff":           # flownet.py:36-38: 1x1 conv 194 -> 1024 without bias, "+ 1" in forward
            self.Convolution5_scale = nn.Conv2d(194, 1024, 1, bias=False)
            nn.init.zeros_(self.Convolution5_scale.weight)
        for name, ci, co, k, s, p in self.CONVS:
            setattr(self, name, nn.Conv2d(ci, co, k, stride=s, padding=p))
        for name, ci in self.PREDS:
            setattr(self, name, nn.Conv2d(ci, 2, 3, stride=1, padding=1))
        for name, ci, co in self.DECONVS:
            setattr(self, name, nn.ConvTranspose2d(ci, co, 4, stride=2))
        for name in self.UPS:
            setattr(self, name, nn.ConvTranspose2d(2, 2, 4, stride=2))

    def _pack(self, dtype, device):
        m = _mult(dtype)
        pk = {}
        for name, ci, co, k, s, p in self.CONVS:
            mod = getattr(self, name)
            # the image pair is fed un-normalised: the reference's "/ 255" (generalized_rcnn_fgfa.py:198) is linear
            # and is folded into the first conv's weights
            pk[name] = (_pack_w(mod.weight, dtype, m, 1.0 / 255 if name == "flow_conv1" else 1.0).to(device),
                        mod.bias.detach().float().to(device).contiguous(), s, p)
        for name, _ in self.PREDS:
            mod = getattr(self, name)
            pk[name] = (_pack_w(mod.weight, dtype, m).to(device), mod.bias.detach().float().to(device).contiguous())
        for name in [d[0] for d in self.DECONVS] + self.UPS:
            mod = getattr(self, name)
            w = mod.weight.detach().permute(1, 0, 2, 3).flip(2, 3)      # [in,out,kh,kw] -> conv kernel [out,in,kh,kw]
            pk[name] = (_pack_w(w, dtype, m).to(device), mod.bias.detach().float().to(device).contiguous())
        pk["Convolution5.b2.5"] = self.Convolution5.bias.detach().float().to(device).contiguous() * 2.5
        for name, _ in self.PREDS:           # Conv2d(Cin, 2, 3, padding=1) as a 1 x 1 conv with 18 (tap, channel) columns (of 64)
            mod = getattr(self, name)
            w = mod.weight.detach().float()                                  # [2, Cin, 3, 3]
            w18 = torch.zeros((64, 1, 1, w.shape[1]), dtype=torch.float32)
            w18[:18, 0, 0] = w.permute(2, 3, 0, 1).reshape(18, w.shape[1])   # row (r*3 + s)*2 + c
            pk[name + ".18"] = (_padc(w18, m).to(dtype).to(device).contiguous(), mod.bias.detach().float().to(device).contiguous())
        for name, _, co in self.DECONVS:     # the sub-pixel form: [4 C, 2, 2, Cin] + the bias once per phase
            mod = getattr(self, name)
            pk[name + ".sp"] = (ops.pack_deconv4x4s2(mod.weight, dtype, m).to(device),
                                mod.bias.detach().float().repeat(4).to(device).contiguous(), co)
        for name in self.UPS:                # f32 ConvTranspose2d weight [in, out, kh, kw] as it is
            mod = getattr(self, name)
            pk[name + ".sp"] = (mod.weight.detach().float().to(device).contiguous(), mod.bias.detach().float().to(device).contiguous())
        if self.method == "dff":
            pk["scale_w"] = _pack_w(self.Convolution5_scale.weight, dtype, m).to(device)
            pk["ones"] = torch.ones((1024,), dtype=torch.float32, device=device)
        pk["x2.5"] = torch.full((2,), 2.5, dtype=torch.float32, device=device)
        pk["m"] = m
        if dtype in (torch.bfloat16, torch.float16):
            # flow_conv1 over ops.fgfa_pair_taps' operand: a 7 x 1 conv whose 64 input channels are the seven horizontal taps
            # x (6 image channels + 2 zeros) + 8 zeros: w[o][r][0][s * 8 + c] = W[o][c][r][s] / 255  (K = 448, not 49 x 64)
            w1 = self.flow_conv1.weight.detach().float() / 255.0                     # [64,6,7,7] (o, c, r, s)
            wt = torch.zeros((64, 7, 1, 64), dtype=torch.float32)
            for s_ in range(7):
                wt[:, :, 0, s_ * 8:s_ * 8 + 6] = w1[:, :, :, s_].permute(0, 2, 1)   # [o, r, c]
            pk["flow_conv1_taps"] = wt.to(dtype).to(device).contiguous()
            # the per-FRAME halves (conv1_parts): against the operand of the pair (frame, frame), rows 0..63 see the key-frame
            # channels (c < 3) only, rows 64..127 the reference-frame channels (c = 3..5) only
            wab = torch.zeros((128, 7, 1, 64), dtype=torch.float32)
            for s_ in range(7):
                wab[:64, :, 0, s_ * 8:s_ * 8 + 3] = wt[:, :, 0, s_ * 8:s_ * 8 + 3]
                wab[64:, :, 0, s_ * 8 + 3:s_ * 8 + 6] = wt[:, :, 0, s_ * 8 + 3:s_ * 8 + 6]
            pk["flow_conv1_ab"] = wab.to(dtype).to(device).contiguous()
        return pk

    def _conv(self, pk, name, x, act=2, ksplit=None):
        w, b, s, p = pk[name]
        return ops.conv2d_nhwc(_padc(x, pk["m"]), w, None, b, stride=s, pad=p, relu=act, ksplit=ksplit)

    def _pred(self, pk, name, x, scale=1.0, out_dtype=None):
        """flow prediction * scale: [N,H,W,2]"""
        if self.subpixel:      # one pass over x for the 18 (tap, channel) partial maps in f32, then the shifted sum
            w18, b = pk[name + ".18"]
            z = ops.conv2d_nhwc(x, w18, None, None, out_dtype=torch.float32)
            return ops.flow_pred_finish(z, pk[name + ".b2.5"] if scale == 2.5 else (b * scale if scale != 1.0 else b), scale,
                                        out_dtype or x.dtype)
        w, b = pk[name]
        if scale != 1.0:
            return ops.conv2d_nhwc(x, w, pk["x2.5"] * (scale / 2.5), b * scale, pad=1, relu=0, out_dtype=out_dtype)
        return ops.conv2d_nhwc(x, w, None, b, pad=1, relu=0, out_dtype=out_dtype)

    def _deconv(self, pk, name, x, act):
        w, b = pk[name]
        xs = _zero_stuff(_padc(x, pk["m"]))
        return ops.conv2d_nhwc(xs, w, None, b, pad=3, relu=act)

    def run(self, pair_nhwc):
        """pair_nhwc [T,H,W,6 (+pad)] un-normalised image pairs (cur | ref) -> flow [T,2,Hf,Wf] f32."""
        dt = pair_nhwc.dtype
        pk = self._packed(dt, pair_nhwc.device)
        m = _mult(dt)
        # pool at 8 channels (one 16-byte vector per pixel), THEN zero-pad to the GEMM's K vector: padding the full-resolution
        # pair to 64 channels first wrote and re-read 1.6 GB per key frame for the same values
        x = _padc(ops.avgpool2x2_ceil(_padc(pair_nhwc, 8 if dt == torch.bfloat16 else 4)), m)
        r1 = self._conv(pk, "flow_conv1", x)
        return self._trunk(pk, r1, m)

    def run_pairs(self, refs, cur=None, order=None, dtype=torch.bfloat16):
        """The same network fed from the f32 NCHW frames (16-bit compute dtypes): refs [T,3,H,W], the key frame `cur`
        ([1,3,H,W] or one per pair) or ring slot order[0] -- the pair assembly, cast, pool and flow_conv1's horizontal taps are
        ONE kernel (ops.fgfa_pair_taps) and flow_conv1 a 7 x 1 conv over K = 448 (round 6: the concatenation, permute, two
        channel pads and a conv over K = 3136 it replaces were a quarter of config 5's key frame)."""
        pk = self._packed(dtype, refs.device)
        x = ops.fgfa_pair_taps(refs, cur, order, dtype)
        w, b, s, p = pk["flow_conv1"]
        r1 = ops.conv2d_nhwc(x, pk["flow_conv1_taps"], None, b, stride=2, pad=0, relu=2)
        return self._trunk(pk, r1, _mult(dtype))

    def conv1_parts(self, frames, dtype):
        """flow_conv1 is linear before its LeakyReLU and its input is cat([key, frame]): conv(pair) = A(key) + B(frame) with
        A / B the convs of ONE frame against the key / reference half of the weights.  frames f32 [n,3,H,W] -> f32
        [n,h1,w1,128] = [A | B] (no bias): computed once per frame, when it enters the window (16-bit compute dtypes)."""
        pk = self._packed(dtype, frames.device)
        x = ops.fgfa_pair_taps(frames, frames, None, dtype)            # the operand of the pair (frame, frame)
        return ops.conv2d_nhwc(x, pk["flow_conv1_ab"], None, None, stride=2, pad=0, relu=0, out_dtype=torch.float32)

    def run_parts(self, ab, dtype, key=None, order=None, T=None):
        """flow of the pairs (key frame, frame t), t = 0 .. T-1, from the frames' conv1 halves ab [S,h1,w1,128] (conv1_parts);
        the key frame is slot `key` or order[0].  flow_conv1 = leaky(A[key] + B[t] + bias): one element-wise kernel per key frame
        instead of the pair assembly + a K = 448 conv over 21 pairs (0.30 -> 0.06 ms of config 5's key frame)."""
        pk = self._packed(dtype, ab.device)
        r1 = ops.flow_conv1_combine(ab, pk["flow_conv1"][1], dtype, key=key, order=order, T=T)
        return self._trunk(pk, r1, _mult(dtype))

    def run_parts_multi(self, ab, dtype, orders):
        """run_parts for several key frames in ONE trunk pass: orders i32 [G, 1 + T] on the device, row g = [slot of key frame g,
        slot of window position 0 .. T-1] -> flow [G * T, 2, h, w], rows g * T + t = the pair (key frame g, window position
        t): exactly the key frames' pairs, in window order.  The trunk's kernels are batch-invariant: the bits of G separate
        run_parts calls on the windows' frames."""
        pk = self._packed(dtype, ab.device)
        r1 = ops.flow_conv1_combine(ab, pk["flow_conv1"][1], dtype, order=orders, nwin=orders.shape[1] - 1)
        return self._trunk(pk, r1, _mult(dtype))

    def pairs(self, refs, cur, dtype, order=None):
        """flow of the pairs (cur | refs[t]): the one-kernel input stage for the 16-bit dtypes, the generic path otherwise"""
        if dtype in (torch.bfloat16, torch.float16):
            return self.run_pairs(refs, cur, order, dtype)
        T = refs.shape[0]
        if cur is None:
            cur = refs.index_select(0, order[0:1].long())
        pair = torch.cat([cur.expand(T, -1, -1, -1) if cur.shape[0] == 1 else cur, refs], dim=1)
        return self.run(pair.permute(0, 2, 3, 1).contiguous().to(dtype))

    def _trunk(self, pk, r1, m):
        r2 = self._conv(pk, "conv2", r1)
        r3 = self._conv(pk, "conv3", r2)
        r4 = self._conv(pk, "conv3_1", r3)
        r5 = self._conv(pk, "conv4", r4)
        r6 = self._conv(pk, "conv4_1", r5)
        r7 = self._conv(pk, "conv5", r6)
        r8 = self._conv(pk, "conv5_1", r7)
        r9 = self._conv(pk, "conv6", r8)
        r10 = self._conv(pk, "conv6_1", r9, ksplit=4 if self.subpixel else None)    # 840 rows x K = 9216: 53 -> 39 us

        def level(feat_in, skip, pred_name, up_name, deconv_name):
            flow = self._pred(pk, pred_name, feat_in)                       # [.,.,.,2]
            if self.subpixel:
                w4, b4, C = pk[deconv_name + ".sp"]
                Cs = skip.shape[-1]
                cc = torch.empty(skip.shape[:3] + ((Cs + C + 2 + m - 1) // m * m,), dtype=skip.dtype, device=skip.device)
                ops.deconv4x4s2_into(feat_in, w4, b4, cc, Cs, relu=2)       # cc[..., Cs:Cs+C] = crop(leaky(deconv(feat_in)))
                wu, bu = pk[up_name + ".sp"]
                return ops.flow_level_assemble(skip, flow, wu, bu, cc, C)  # skip | . | crop(up(flow)) | zeros
            up = _crop_like(self._deconv(pk, up_name, flow, 0), skip)
            dec = _crop_like(self._deconv(pk, deconv_name, feat_in, 2), skip)
            return _padc(torch.cat([skip, dec, up], dim=-1), m)             # concatN (channel-padded)
        c2 = level(r10, r8, "Convolution1", "upsample_flow6to5", "deconv5")
        c3 = level(c2, r6, "Convolution2", "upsample_flow5to4", "deconv4")
        c4 = level(c3, r4, "Convolution3", "upsample_flow4to3", "deconv3")
        c5 = level(c4, r2, "Convolution4", "upsample_flow3to2", "deconv2")
        c5 = ops.avgpool2x2_ceil(c5)
        flow = self._pred(pk, "Convolution5", c5, 2.5, torch.float32)       # Convolution5 * 2.5 (flownet.py:118)
        flow = flow.permute(0, 3, 1, 2).contiguous()
        if self.method == "dff":          # Convolution5_scale + 1 (flownet.py:112-116), NHWC [T,h,w,1024]
            return flow, ops.conv2d_nhwc(c5, pk["scale_w"], None, pk["ones"], relu=0)
        return flow

    def forward(self, x):
        """reference signature: x [T,6,H,W] = cat([cur/255, ref/255]) -> flow [T,2,h,w]."""
        dt = self._dtype
        out = self.run((x * 255.0).permute(0, 2, 3, 1).contiguous().to(dt))
        if self.method == "dff":
            return out[0], _nchw_view(out[1])
        return out


class EmbedNet(_Packed):
    """backbone/embednet.py:8-24."""

    def __init__(self, cfg):
        super().__init__()
        self.embed_conv1 = nn.Conv2d(1024, 512, 1)
        self.embed_conv2 = nn.Conv2d(512, 512, 3, padding=1)
        self.embed_conv3 = nn.Conv2d(512, 2048, 1)

    def _pack(self, dtype, device):
        return {n: (_pack_conv(getattr(self, n), dtype).to(device), getattr(self, n).bias.detach().float().to(device).contiguous())
                for n in ("embed_conv1", "embed_conv2", "embed_conv3")}

    def run(self, x):
        pk = self._packed(x.dtype, x.device)
        x = ops.conv2d_nhwc(x, pk["embed_conv1"][0], None, pk["embed_conv1"][1], relu=True)
        x = ops.conv2d_nhwc(x, pk["embed_conv2"][0], None, pk["embed_conv2"][1], pad=1, relu=True)
        return ops.conv2d_nhwc(x, pk["embed_conv3"][0], None, pk["embed_conv3"][1])

    def forward(self, x):
        return _nchw_view(self.run(_nhwc(x)))


@ROI_BOX_FEATURE_EXTRACTORS.register("ResNetConv52MLPFeatureExtractor")
class ResNetConv52MLPFeatureExtractor(_Packed):
    """roi_box_feature_extractors.py:55-118: res5 on the full map (+1x1 reduce) -> ROIAlign -> fc6 -> fc7."""

    def __init__(self, cfg, in_channels):
        super().__init__()
        rb = cfg.MODEL.ROI_BOX_HEAD
        self.head = ResNetHead(cfg.MODEL.RESNETS.RES5_DILATION)
        self.conv = nn.Conv2d(2048, 256, 1) if cfg.MODEL.VID.ROI_BOX_HEAD.REDUCE_CHANNEL else None
        self.pooled_c = 256 if self.conv is not None else 2048
        self.resolution, self.scale, self.sampling_ratio = rb.POOLER_RESOLUTION, rb.POOLER_SCALES[0], rb.POOLER_SAMPLING_RATIO
        rep = rb.MLP_HEAD_DIM
        self.fc6 = nn.Linear(self.pooled_c * self.resolution ** 2, rep)
        self.fc7 = nn.Linear(rep, rep)
        self.out_channels = rep

    def _pack(self, dtype, device):
        r2 = self.resolution ** 2
        w6 = self.fc6.weight.detach()
        w6 = w6.view(w6.shape[0], self.pooled_c, r2).permute(0, 2, 1).reshape(w6.shape[0], -1)   # (c,ph,pw)->(ph,pw,c)
        pk = {"w6": w6.contiguous().to(dtype).to(device), "b6": self.fc6.bias.detach().float().to(device).contiguous(),
              "w7": self.fc7.weight.detach().to(dtype).to(device).contiguous(),
              "b7": self.fc7.bias.detach().float().to(device).contiguous()}
        if self.conv is not None:
            pk["rc_w"] = _pack_conv(self.conv, dtype).to(device)
            pk["rc_b"] = self.conv.bias.detach().float().to(device).contiguous()
        return pk

    FC6_KSPLIT = 24      # fc6 on <= 300 rows, K = 100352: 211 us with the library's 3 ranges, 120 us with 24 (tools/gpu/ksplit_sweep.py)

    def full_map(self, x):
        """res5 (+ the 1x1 reduce conv) on the whole map: needs no proposals (the engine runs it beside the RPN selection)"""
        feat = _nhwc(x[0] if isinstance(x, (list, tuple)) else x)
        pk = self._packed(feat.dtype, feat.device)
        y = self.head.run(feat)
        if self.conv is not None:
            y = ops.conv2d_nhwc(y, pk["rc_w"], None, pk["rc_b"], relu=True)
        return y

    def pooled_fc(self, y, proposals):
        pk = self._packed(y.dtype, y.device)
        pooled = ops.roi_align(y, convert_to_roi_format(proposals), self.scale, (self.resolution, self.resolution),
                               self.sampling_ratio)
        # split-K chosen by K alone (the rows of a batch keep their bits whatever the batch): long-K fc6 only
        ks = self.FC6_KSPLIT if pk["w6"].shape[1] >= 32768 else None
        h = ops.linear(pooled.view(pooled.shape[0], -1), pk["w6"], pk["b6"], relu=True, ksplit=ks)
        return ops.linear(h, pk["w7"], pk["b7"], relu=True)

    def forward(self, x, proposals):
        return self.pooled_fc(self.full_map(x), proposals)


class ROIBoxHead(nn.Module):
    """roi_heads/box_head/box_head.py:11-62 (test path)."""

    def __init__(self, cfg, in_channels):
        super().__init__()
        self.feature_extractor = ROI_BOX_FEATURE_EXTRACTORS[cfg.MODEL.ROI_BOX_HEAD.FEATURE_EXTRACTOR](cfg, in_channels)
        self.predictor = ROI_BOX_PREDICTOR[cfg.MODEL.ROI_BOX_HEAD.PREDICTOR](cfg, self.feature_extractor.out_channels)
        self.post_processor = PostProcessor(cfg)

    def forward(self, features, proposals, targets=None):
        if self.training:
            raise NotImplementedError("inference path only")
        x = self.feature_extractor(features, proposals)
        class_logits, box_regression = self.predictor(x)
        return x, self.post_processor((class_logits, box_regression), proposals), {}


class GeneralizedRCNNFGFA(nn.Module):
    """detector/generalized_rcnn_fgfa.py:21-219, inference: per-video deques of images and [features | embeddings]
    (maxlen ALL_FRAME_INTERVAL, key = slot KEY_FRAME_LOCATION), FlowNetS on the (key, frame) pairs, flow-guided
    aggregation, then the plain RPN + box head on the aggregated map."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.device = torch.device(cfg.MODEL.DEVICE)
        self.dtype = compute_dtype(cfg)
        self.backbone = build_backbone(cfg)
        self.flownet = FlowNetS(cfg)
        self.flownet._dtype = self.dtype
        self.embednet = EmbedNet(cfg)
        self.rpn = build_rpn(cfg, self.backbone.out_channels)
        self.rpn.head.conv_ksplit = 4        # one frame per call: see RPNHead.conv_ksplit
        self.roi_heads = CombinedROIHeads(cfg, [("box", ROIBoxHead(cfg, self.backbone.out_channels))])
        self.all_frame_interval = cfg.MODEL.VID.FGFA.ALL_FRAME_INTERVAL
        self.key_frame_location = cfg.MODEL.VID.FGFA.KEY_FRAME_LOCATION
        self.eval()

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        for m in self.modules():
            if isinstance(m, _Packed):
                m._pk = None
        return out

    @torch.no_grad()
    def _update_feature(self, img=None, feats=None):
        """:144-150.  img [1,3,H,W] f32; the stored feature is NHWC [1,h,w,1024+2048] (features | embeddings)."""
        if feats is None:
            f = _nhwc(self.backbone(img)[0])
            feats = torch.cat([f, self.embednet.run(f)], dim=-1)
            if self.dtype in (torch.bfloat16, torch.float16):      # FlowNetS's first conv, per frame (FlowNetS.conv1_parts)
                feats = (feats, self.flownet.conv1_parts(img, self.dtype))
        if isinstance(feats, tuple):
            self.flow_ab.append(feats[1])
        self.images.append(img)
        self.features.append(feats[0] if isinstance(feats, tuple) else feats)
        return feats

    @torch.no_grad()
    def forward(self, images, targets=None):
        """reference call convention (:78-105): images = {"cur", "ref": [frame t+MAX_OFFSET], "frame_category",
        "seg_len", "pattern", "img_dir", "transforms"}; extension "ref_init": preprocessed frames 1..9 for
        frame_category 0 (replaces the PIL read inside forward, :166-176)."""
        if targets is not None:
            raise ValueError("In testing mode, targets should be None")
        cur = to_image_list(images["cur"]).tensors.to(self.device).float()
        H, W = cur.shape[-2:]
        if images["frame_category"] == 0:
            self.seg_len = images["seg_len"]
            self.end_id = 0
            self.images = deque(maxlen=self.all_frame_interval)
            self.features = deque(maxlen=self.all_frame_interval)
            self.flow_ab = deque(maxlen=self.all_frame_interval)
            f0 = self._update_feature(cur)
            while len(self.images) < self.key_frame_location + 1:
                self._update_feature(cur, f0)
            init = images.get("ref_init")
            while len(self.images) < self.all_frame_interval:
                self.end_id = min(self.end_id + 1, self.seg_len - 1)
                if self.end_id == 0:
                    t = cur
                elif init is not None:
                    t = to_image_list(init[self.end_id - 1]).tensors
                else:
                    from PIL import Image
                    im = Image.open(images["img_dir"] % (images["pattern"] % self.end_id)).convert("RGB")
                    t = images["transforms"](im)
                    t = t[0] if isinstance(t, tuple) else t
                    t = t.view(1, *t.shape)
                self._update_feature(t.to(self.device).float())
        elif images["frame_category"] == 1:
            self.end_id = min(self.end_id + 1, self.seg_len - 1)
            self._update_feature(to_image_list(images["ref"][0]).tensors.to(self.device).float())
        all_images = torch.cat(list(self.images), dim=0)                       # [T,3,H,W]
        all_features = torch.cat(list(self.features), dim=0)                   # [T,h,w,3072] NHWC
        cur_image = self.images[self.key_frame_location]
        T = all_images.shape[0]
        if len(self.flow_ab) == T:                                              # :196-198 (the /255 lives in conv1)
            flow = self.flownet.run_parts(torch.cat(list(self.flow_ab), dim=0), self.dtype, key=self.key_frame_location)
        else:
            flow = self.flownet.pairs(all_images, cur_image, self.dtype)
        nfeat = self.backbone.out_channels
        agg = ops.fgfa_warp_aggregate(all_features.contiguous(), flow, nfeat, self.key_frame_location)   # [h,w,1024]
        feats = (_nchw_view(agg.unsqueeze(0)),)
        il = to_image_list(cur)
        proposals, _ = self.rpn(il, feats, None)
        _, result, _ = self.roi_heads(feats, proposals, None)
        return result


DETECTION_META_ARCHITECTURES.register("GeneralizedRCNNFGFA", GeneralizedRCNNFGFA)


class GeneralizedRCNN(nn.Module):
    """detector/generalized_rcnn.py:16-65, inference (BASELINE config 1, configs/vid_R_50_C4_1x.yaml): single frame,
    backbone -> RPN -> box head (ResNetConv52MLPFeatureExtractor); no cross-frame state."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.device = torch.device(cfg.MODEL.DEVICE)
        self.dtype = compute_dtype(cfg)
        self.backbone = build_backbone(cfg)
        self.rpn = build_rpn(cfg, self.backbone.out_channels)
        self.rpn.head.conv_ksplit = 4        # one frame per call: see RPNHead.conv_ksplit
        self.roi_heads = CombinedROIHeads(cfg, [("box", ROIBoxHead(cfg, self.backbone.out_channels))])
        self.eval()

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        for m in self.modules():
            if isinstance(m, _Packed):
                m._pk = None
        return out

    @torch.no_grad()
    def forward(self, images, targets=None):
        if targets is not None:
            raise ValueError("In testing mode, targets should be None")
        il = to_image_list(images)
        il = type(il)(il.tensors.to(self.device).float(), il.image_sizes)
        features = self.backbone(il.tensors)
        proposals, _ = self.rpn(il, features, None)
        _, result, _ = self.roi_heads(features, proposals, None)
        return result


DETECTION_META_ARCHITECTURES.register("GeneralizedRCNN", GeneralizedRCNN)


class GeneralizedRCNNDFF(nn.Module):
    """detector/generalized_rcnn_dff.py:19-138, inference: the backbone runs on key frames only
    (images["is_key_frame"], every KEY_FRAME_INTERVAL-th frame in data/datasets/vid_dff.py); every frame's features
    are the key frame's C4 map warped by FlowNetS(frame, key) and multiplied by the predicted scale map."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.device = torch.device(cfg.MODEL.DEVICE)
        self.dtype = compute_dtype(cfg)
        self.backbone = build_backbone(cfg)
        self.flownet = FlowNetS(cfg)
        self.flownet._dtype = self.dtype
        self.rpn = build_rpn(cfg, self.backbone.out_channels)
        self.rpn.head.conv_ksplit = 4        # one frame per call: see RPNHead.conv_ksplit
        self.roi_heads = CombinedROIHeads(cfg, [("box", ROIBoxHead(cfg, self.backbone.out_channels))])
        self.key_images = None
        self.key_feats = None
        self.eval()

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        for m in self.modules():
            if isinstance(m, _Packed):
                m._pk = None
        return out

    @torch.no_grad()
    def forward(self, images, targets=None):
        """images = {"cur", "is_key_frame", ...} (vid_dff.py test feed)."""
        if targets is not None:
            raise ValueError("In testing mode, targets should be None")
        cur = to_image_list(images["cur"]).tensors.to(self.device).float()
        if images["is_key_frame"]:
            self.key_images = cur
            self.key_feats = _nhwc(self.backbone(cur)[0]).contiguous()
        if self.key_feats is None:
            raise RuntimeError("the first frame of a video must be a key frame")
        flow, scale = self.flownet.pairs(self.key_images, cur, self.dtype)       # :132 cat([cur, key]) (the /255 lives in conv1)
        agg = ops.dff_warp_scale(self.key_feats[0], flow[0].contiguous(), scale[0].contiguous())
        feats = (_nchw_view(agg.unsqueeze(0)),)
        proposals, _ = self.rpn(to_image_list(cur), feats, None)
        _, result, _ = self.roi_heads(feats, proposals, None)
        return result


DETECTION_META_ARCHITECTURES.register("GeneralizedRCNNDFF", GeneralizedRCNNDFF)


class FgfaClipEngine(object):
    """Clip-level driver for GeneralizedRCNNFGFA (BASELINE configs[4]): the MI355X-first way to run the reference's
    per-key-frame loop (generalized_rcnn_fgfa.py:144-219; feed: data/datasets/vid_fgfa.py test mode).

    The reference (and `model(images)` here) runs ~200 launches per key frame at batch 1, re-concatenates the window's
    21 images and 21 x 3072-channel maps every step, and reads the detection count back before the next frame: on
    MI355X that is host-bound and its single-frame launches fill a fraction of the chip.  Here
      * backbone + EmbedNet (+ the per-frame halves of FlowNetS's first conv) run for `lookahead` upcoming frames in ONE
        batch (the kernels are batch-invariant: same bits), once per frame of the video;
      * a frame's maps live in ring slot `frame id mod R`, R = T + group - 1: the window of key frame k is the frames
        clamp(k - key + t, 0, L - 1), t = 0 .. T-1 (what the reference's deque holds: frame 0 replicated at the start, the
        last frame at the end), i.e. a row of a device index table (`order[b]` = [slot of the key frame, slot of window
        position t ...]); FlowNetS takes a key frame's pairs in slot order over the whole ring and the warp kernel visits the
        window in window order through the table (mega_fgfa_warp_aggregate_ring) -- the bits of the contiguous call;
      * `group` consecutive key frames (default 10) share ONE FlowNetS pass over exactly their group x T pairs (its coarse
        levels have 840-12 768 GEMM rows at 21 pairs and leave half the chip idle: 1.05 ms per key frame at 21 pairs, 0.84 at
        42, 0.77 at 84) and ONE batched box-head pass; the flow fields come out in window order per key frame
        (mega_fgfa_warp_aggregate_ring_pos).  Same box, batched head: group 1 / 2 / 4 / 5 / 10 / 20 = 518 / 595 / 630 / 638 /
        665 / 669 FPS (with the box head replayed per key frame the optimum was 2: profiles/r06_c5_group_ab.txt);
      * the key frame is TWO hipGraphs on two streams: A = FlowNetS + warp of a group, B = RPN selection, res5 + ROIAlign +
        fc6 / fc7, predictor, post-processing (fixed 300 proposal rows per frame, the device-side proposal counts go to the
        post-processor) of the group's key frames as ONE batched launch chain (batch_head; or one replay per key frame);
        B of one group runs beside A of the next (its one-block-per-frame selection / NMS kernels and 300-row GEMMs leave
        most of the chip idle); detection counts are read a batch of steps later.
    Detections are identical to `model(images)` frame by frame (tests/test_e2e_gpu.py::test_fgfa_engine_equals_model)."""

    DEPTH = 3        # groups the first stream may run ahead of the second (staging buffers of aggregated maps)
    lanes = 1        # graph-B lanes (streams): 1 hides the box head beside FlowNetS here; DffClipEngine uses more
    fork_select = True   # graph B forks the one-block proposal selection to a side stream beside res5
    batch_head = False   # graph B on all maps of a group in ONE batched replay (_body_bb) instead of one replay per map

    def __init__(self, model, lookahead=20, graphs=True, pipeline=True, group=10, lanes=1, batch_head=True):
        self.m = model
        self.batch_head = bool(batch_head)
        self.lanes = max(1, int(lanes))
        self.fork_select = self.lanes == 1   # (see DffClipEngine: forked graphs on several lanes crash the HIP runtime)
        self.pipeline = pipeline             # graphs A and B on two streams (see _step); False: one graph on one stream
        self.parts = model.dtype in (torch.bfloat16, torch.float16)    # FlowNetS's first conv per frame, kept in a ring
        self.group = max(1, int(group))      # key frames per FlowNetS pass
        self._sb = None
        self.T = model.all_frame_interval
        self.key = model.key_frame_location
        self.R = self.T + self.group - 1
        self.lookahead = lookahead
        self.use_graphs = graphs
        self.graph = None
        self.fgraphs = {}
        self.replays = 0
        self.feat_ring = None
        self.keep_intermediates = False      # tests / diagnostics: self._dbg = per key frame (flow, aggregated map, proposals, ...)

    # ---- features of a batch of frames (backbone + EmbedNet), replayed from a hipGraph per batch size
    def _features(self, imgs):
        m = self.m

        def body(x):
            f = _nhwc(m.backbone(x)[0])
            out = torch.cat([f, m.embednet.run(f)], dim=-1)
            if self.parts:               # + FlowNetS's first conv per frame (FlowNetS.conv1_parts)
                return out, m.flownet.conv1_parts(x, m.dtype)
            return out, None
        if not (self.use_graphs and imgs.is_cuda):
            return body(imgs)
        ent = self.fgraphs.setdefault(tuple(imgs.shape), {})
        if "seen" not in ent:            # first use of a shape: eager (packs weights, warms the allocator)
            ent["seen"] = True
            return body(imgs)
        if "graph" not in ent:
            ent["in"] = imgs.clone()
            torch.cuda.current_stream().synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                ent["out"] = body(ent["in"])
            ent["graph"] = g
        ent["in"].copy_(imgs)
        ent["graph"].replay()
        return tuple(None if t is None else t.clone() for t in ent["out"])

    # ---- a group of key frames on the ring state, in two halves (these bodies are what the graphs capture)
    def _body_a(self):
        """FlowNetS on the pairs of the group's key frames + flow-guided aggregation -> aggregated C4 maps [group,h,w,1024]"""
        m = self.m
        G, T = self.group, self.T
        nfeat = m.backbone.out_channels
        if self.parts:                  # key frame b = ring slot order[b][0]; ONE trunk pass over exactly the G x T pairs
            flow = m.flownet.run_parts_multi(self.ab_ring, m.dtype, self.order)
            flows = [flow[b * T:(b + 1) * T] for b in range(G)]
            if not self.keep_intermediates:     # the group's warps in ONE launch (a key frame alone is 1.17 rounds of blocks)
                return ops.fgfa_warp_aggregate_group(self.feat_ring, flow, nfeat, self.order, self.key)
            aggs = [ops.fgfa_warp_aggregate(self.feat_ring, flows[b], nfeat, 0, order=self.order[b], flow_pos=self.key)
                    for b in range(G)]
        else:                           # (exact-f32 mode: the generic pair path, a key frame's pairs in slot order over the ring)
            flows = [m.flownet.pairs(self.img_ring, None, m.dtype, order=self.order[b]) for b in range(G)]
            aggs = [ops.fgfa_warp_aggregate(self.feat_ring, flows[b], nfeat, 0, order=self.order[b]) for b in range(G)]
        if self.keep_intermediates:
            self._dbg_a = [(flows[b], aggs[b]) for b in range(G)]
        return torch.stack(aggs, dim=0)

    def _body_b(self, agg, size, lane=0):
        """RPN + conv5 box head + post-processing on ONE aggregated map [h,w,1024]"""
        m = self.m
        W, H = size
        feats = (_nchw_view(agg.unsqueeze(0)),)
        box = m.roi_heads.box
        fe = box.feature_extractor
        if agg.is_cuda and not ops.profiling() and self.fork_select:
            # the proposal selection is ONE block (top-k, decode, NMS of one frame: ~0.38 ms on 1 of 256 CUs) and res5 on the
            # whole map does not need its result: fork it to a side stream (one per lane), join before ROIAlign
            sides = self.__dict__.setdefault("_sides", {})
            if lane not in sides:
                sides[lane] = torch.cuda.Stream(device=agg.device)
            side = sides[lane]
            hold = []
            props, _, cnt = m.rpn.propose(_nhwc(feats[0]), W, H, "key", select_stream=side, hold=hold)
            y = fe.full_map(feats)
            torch.cuda.current_stream(agg.device).wait_stream(side)
            del hold
            x = fe.pooled_fc(y, [props[0]])
        else:
            props, _, cnt = m.rpn.propose(_nhwc(feats[0]), W, H, "key")
            x = fe(feats, [props[0]])
        logits, deltas = box.predictor(x)
        pp = box.post_processor
        if self.keep_intermediates:
            self._dbg_b = (props, logits, deltas, x, cnt)
        return ops.postprocess(logits.float().contiguous(), deltas.float().contiguous(), props[0].contiguous(), cnt,
                               pp.weights, W, H, pp.score_thresh, pp.nms, pp.detections_per_img, pp.strict_gt)

    def _body_bb(self, aggs, size):
        """_body_b for ALL maps of a group at once ([G,h,w,1024]): the RPN head, selection (one block per frame), res5, ROIAlign,
        fc6 / fc7, predictor and post-processing as ONE batched launch chain (every kernel is batch-invariant and the batched
        post-processor gives each image the bits of its own call) -> (boxes [G,cap,4], scores [G,cap], labels [G,cap], counts [G])"""
        m = self.m
        W, H = size
        G = aggs.shape[0]
        feats = (_nchw_view(aggs),)
        box = m.roi_heads.box
        fe = box.feature_extractor
        if aggs.is_cuda and not ops.profiling():
            sides = self.__dict__.setdefault("_sides", {})
            if 0 not in sides:
                sides[0] = torch.cuda.Stream(device=aggs.device)
            hold = []
            props, _, cnt = m.rpn.propose(_nhwc(feats[0]), W, H, "key", select_stream=sides[0], hold=hold)
            y = fe.full_map(feats)
            torch.cuda.current_stream(aggs.device).wait_stream(sides[0])
            del hold
        else:
            props, _, cnt = m.rpn.propose(_nhwc(feats[0]), W, H, "key")
            y = fe.full_map(feats)
        x = fe.pooled_fc(y, [props[b] for b in range(G)])
        logits, deltas = box.predictor(x)
        pp = box.post_processor
        return ops.postprocess_batched(logits.float().contiguous(), deltas.float().contiguous(), props.reshape(-1, 4).contiguous(), G,
                                       pp.weights, W, H, pp.score_thresh, pp.nms, pp.detections_per_img, pp.strict_gt, nprop=cnt)

    def _eager(self, size, n):
        aggs = self._body_a()
        if self.batch_head and not self.keep_intermediates:
            ob, os_, ol, oc = self._body_bb(aggs, size)
            return [(ob[b], os_[b], ol[b], oc[b:b + 1]) for b in range(n)]
        outs = []
        for b in range(n):
            outs.append(self._body_b(aggs[b], size))
            if self.keep_intermediates:
                self._dbg = self._dbg_a[b] + self._dbg_b
        return outs

    def _step(self, size, n):
        """the group on the ring state (order holds `group` rows; the first n are real key frames) -> n output tuples"""
        if not (self.use_graphs and self.feat_ring.is_cuda):
            return self._eager(size, n)
        if self.graph is None:
            self.graph = "armed"
            return self._eager(size, n)
        cur = torch.cuda.current_stream()
        if self.graph == "armed":
            # Graph A (FlowNetS + warp of the group: whole-chip GEMMs) replays on the current stream, graph B (RPN, box head,
            # post-processing of ONE key frame: ~0.8 ms of one-block kernels and GEMMs on 2394 / 300 rows) on a second stream
            # (pipeline=True), so that B of a group runs BESIDE A of the following groups.  A's maps are copied (on A's stream,
            # 10 MB) into one of DEPTH staging buffers; B takes them from there, so A runs up to DEPTH groups ahead of B: with a
            # lead of one group A stalled 0.9 ms per key frame behind B's stretched (contended) replays (tools/gpu/trace_c5.sh).
            # The graphs replay concurrently: each has its own memory pool (torch's default for separately captured graphs).
            cur.synchronize()
            nl = 1 if self.batch_head else (self.lanes if self.pipeline else 1)
            if self._sb is None and self.pipeline:
                self._sb = [torch.cuda.Stream(device=self.feat_ring.device) for _ in range(nl)]
            ga = torch.cuda.CUDAGraph()
            with torch.cuda.graph(ga, capture_error_mode="thread_local"):
                self._agg_out = self._body_a()
            self._agg_stage = [self._agg_out.clone() for _ in range(self.DEPTH)]
            # graph B once per LANE (its own input, outputs, pool and side stream): the key frames of a group are dealt to the
            # lanes round-robin, each lane replaying on its own stream -- B is a latency chain of small launches (0.8-0.9 ms per
            # key frame), and when graph A is short (DffClipEngine: 10 pairs per 10 frames) ONE lane of B is the bottleneck
            self._agg_in, self._out, gbs = [], [], []
            for l in range(nl):
                self._agg_in.append(self._agg_out.clone() if self.batch_head else self._agg_out[0].clone())
                cur.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    self._out.append(self._body_bb(self._agg_in[l], size) if self.batch_head else
                                     self._body_b(self._agg_in[l], size, lane=l))
                gbs.append(g)
            self.graph = (ga, gbs)
            self._es = [torch.cuda.Event() for _ in range(self.DEPTH)]      # staging buffer j holds a group's maps
            self._eb = [[torch.cuda.Event() for _ in range(nl)] for _ in range(self.DEPTH)]   # lane l has consumed buffer j
            for ev in self._eb:
                for e in ev:
                    e.record(cur)
            self._turn = 0
        ga, gbs = self.graph
        j = self._turn
        self._turn = (j + 1) % self.DEPTH
        for e in self._eb[j]:
            cur.wait_event(e)                 # B is done with the maps this group's copy overwrites (DEPTH groups ago)
        ga.replay()
        self._agg_stage[j].copy_(self._agg_out)
        self._es[j].record(cur)
        outs = [None] * n
        for l, g in enumerate(gbs):
            sb = self._sb[l] if self.pipeline else cur
            sb.wait_event(self._es[j])
            with torch.cuda.stream(sb):
                if self.batch_head:               # the whole group in one replay
                    self._agg_in[l].copy_(self._agg_stage[j])
                    g.replay()
                    ob, os_, ol, oc = (t.clone() for t in self._out[l])
                    for b in range(n):
                        outs[b] = (ob[b], os_[b], ol[b], oc[b:b + 1])
                else:
                    for b in range(l, n, len(gbs)):
                        self._agg_in[l].copy_(self._agg_stage[j][b])
                        g.replay()
                        outs[b] = tuple(t.clone() for t in self._out[l])
                self._eb[j][l].record(sb)
        for o in outs:
            for t in o:
                t.record_stream(cur)          # read on the current stream after the join in run()'s flush
        self.replays += n
        return outs

    def _reset(self, frames):
        """rings for a video of this size (kept, with the captured graphs that read them, when the size repeats)"""
        m = self.m
        dev = frames.device
        H, W = frames.shape[-2:]
        sig = (H, W, str(dev))
        if self.feat_ring is None or self._sig != sig:
            f, ab = self._features(frames[0:1].float())
            self.feat_ring = f.new_zeros((self.R,) + tuple(f.shape[1:]))
            self.ab_ring = None if ab is None else ab.new_zeros((self.R,) + tuple(ab.shape[1:]))
            self.img_ring = None if self.parts else frames.new_zeros((self.R, 3, H, W), dtype=torch.float32)
            self.order = torch.zeros((self.group, self.T + 1), dtype=torch.int32, device=dev)
            self._sig = sig
            self.graph = None
        self.slot_fid = [-1] * self.R
        self.cache = {}                                  # frame id -> ([h,w,3072], conv1 halves) computed ahead of need

    def _ensure(self, frames, fids):
        """the frames `fids` resident in their ring slots (slot = id mod R); features of the next `lookahead` new frames in one
        batch (one hipGraph shape)"""
        L = frames.shape[0]
        dev = frames.device
        for fid in fids:
            s = fid % self.R
            if self.slot_fid[s] == fid:
                continue
            if fid not in self.cache:
                ids = sorted(set(min(fid + j, L - 1) for j in range(self.lookahead)))
                if len(ids) == self.lookahead:                            # a contiguous run: a slice (no index upload: a
                    batch = frames[ids[0]:ids[0] + self.lookahead]        # pageable H2D copy would drain the stream)
                else:
                    ids = ids + [ids[-1]] * (self.lookahead - len(ids))  # keep the batch shape
                    batch = frames[torch.tensor(ids, device=dev)]
                fb, ab = self._features(batch.float())
                self.cache = {i: (fb[j], None if ab is None else ab[j]) for j, i in enumerate(ids)}
            f, ab = self.cache[fid]
            self.feat_ring[s].copy_(f)
            if ab is not None:
                self.ab_ring[s].copy_(ab)
            if self.img_ring is not None:
                self.img_ring[s].copy_(frames[fid])
            self.slot_fid[s] = fid

    @torch.no_grad()
    def run(self, frames, first=0, last=None, sync_every=16):
        """frames: preprocessed f32 [L,3,H,W] on the device (the whole video: inference.resident_video builds it from a
        feed.FrameSource).  Key frames first..last-1 (first = 0 starts a new video; first > 0 continues the previous call's video
        on the ring state).  -> list[BoxList]."""
        L = frames.shape[0]
        last = L if last is None else last
        H, W = frames.shape[-2:]
        T, key, G, R = self.T, self.key, self.group, self.R
        out, pending = [], []

        def flush():
            if not pending:
                return
            for s_ in (self._sb or []):
                torch.cuda.current_stream().wait_stream(s_)           # (the second halves of the pending key frames)
            counts = torch.cat([p[3] for p in pending]).tolist()
            for (ob, os_, ol, _), n in zip(pending, counts):
                out.append(PostProcessor.materialize((ob, os_, ol, None), int(n), (W, H)))
            del pending[:]

        if first == 0 or self.feat_ring is None:
            self._reset(frames)
        # the index tables of every group of this call, uploaded ONCE (a pinned upload per group cost the host a pinned
        # allocation and the stream a host round trip per group)
        plan, rows = [], []
        idx = first
        while idx < last:
            n = min(G, last - idx)
            keys = [idx + b for b in range(n)] + [idx + n - 1] * (G - n)        # (a short last group repeats its last key frame)
            # generalized_rcnn_fgfa.py:163-176,:178-181: the deque of key frame k holds frames clamp(k - key + t, 0, L - 1)
            wins = [[min(max(k - key + t, 0), L - 1) for t in range(T)] for k in keys]
            plan.append((n, sorted(set(f for w in wins for f in w))))
            rows.append([[k % R] + [f % R for f in w] for k, w in zip(keys, wins)])
            idx += n
        if not plan:
            return out
        od = torch.tensor(rows, dtype=torch.int32)                               # [groups, G, 1 + T]
        od = od.pin_memory().to(self.order.device, non_blocking=True) if self.order.is_cuda else od
        for gi, (n, fids) in enumerate(plan):
            self._ensure(frames, fids)
            self.order.copy_(od[gi])
            pending.extend(self._step((W, H), n))
            if len(pending) >= sync_every:
                flush()
        flush()
        return out


class DffClipEngine(FgfaClipEngine):
    """Clip-level driver for GeneralizedRCNNDFF (generalized_rcnn_dff.py:19-138, feed: vid_dff.py test mode --
    a key frame every `interval` = 10 frames): the two-graph / two-stream form of FgfaClipEngine for deep feature flow.

    `model(images)` runs, per frame and at batch 1, FlowNetS on ONE image pair (its coarse levels have 40-600 GEMM rows), a warp,
    the RPN and the conv5 box head, and reads the detection count back: host- and latency-bound (600 FPS on R-101).  Here
      * the backbone runs on `lookahead` upcoming KEY frames in one batch;
      * graph A = FlowNetS on the `interval` pairs (frame, key frame) of a key-frame interval in ONE pass + their warp x scale
        (mega_dff_warp_scale) -> `interval` feature maps;
      * graph B = RPN selection, res5 + ROIAlign + fc6 / fc7, predictor, post-processing of one frame, replayed per frame on a
        second stream beside graph A of the following intervals (FgfaClipEngine._step).
    Detections are identical to `model(images)` frame by frame (tests/test_e2e_gpu.py::test_dff_engine_equals_model)."""

    def __init__(self, model, interval=10, lookahead=8, graphs=True, pipeline=True, lanes=2, batch_head=True):
        self.m = model
        self.batch_head = bool(batch_head)   # the box head of the interval's frames as ONE batched graph (else per frame, on lanes)
        self.pipeline = pipeline
        self.lanes = max(1, int(lanes))      # graph-B lanes: two frames' box heads in flight (FgfaClipEngine._step)
        # With several lanes graph B does NOT fork its proposal selection to a side stream: two lanes of forked graphs replayed
        # concurrently segfault inside hipGraphLaunch on this runtime (ROCm 7.0.2; reproducibly at 2 lanes, not at 3 or 4:
        # profiles/r06_dff_engine_lanes.txt) -- and two plain lanes are as fast as three forked ones (1260 vs 1270 FPS)
        self.fork_select = self.lanes == 1
        self.group = int(interval)           # frames per graph A = the key-frame interval (vid_dff.py:62: frame_id % 10 == 0)
        self._sb = None
        self.lookahead = lookahead
        self.use_graphs = graphs
        self.graph = None
        self.fgraphs = {}
        self.replays = 0
        self.feat_ring = None
        self.keep_intermediates = False

    def _features(self, imgs):
        """backbone of a batch of key frames -> NHWC [n,h,w,1024] (replayed from a hipGraph per batch size)"""
        m = self.m

        def body(x):
            return _nhwc(m.backbone(x)[0]).contiguous()
        if not (self.use_graphs and imgs.is_cuda):
            return body(imgs)
        ent = self.fgraphs.setdefault(tuple(imgs.shape), {})
        if "seen" not in ent:
            ent["seen"] = True
            return body(imgs)
        if "graph" not in ent:
            ent["in"] = imgs.clone()
            torch.cuda.current_stream().synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                ent["out"] = body(ent["in"])
            ent["graph"] = g
        ent["in"].copy_(imgs)
        ent["graph"].replay()
        return ent["out"].clone()

    def _body_a(self):
        """FlowNetS on the interval's pairs (frame, key frame) + warp x scale -> [interval,h,w,1024]"""
        m = self.m
        n = self.group
        flow, scale = m.flownet.pairs(self.key_img.expand(n, -1, -1, -1), self.cur_imgs, m.dtype)      # :132 cat([cur, key])
        aggs = [ops.dff_warp_scale(self.key_feat, flow[i].contiguous(), scale[i].contiguous()) for i in range(n)]
        if self.keep_intermediates:
            self._dbg_a = [(flow[i], aggs[i]) for i in range(n)]
        return torch.stack(aggs, dim=0)

    @torch.no_grad()
    def run(self, frames, first=0, last=None, sync_every=20):
        """frames: preprocessed f32 [L,3,H,W] on the device.  Frames first..last-1 (first a multiple of the interval)
        -> list[BoxList]."""
        L = frames.shape[0]
        last = L if last is None else last
        H, W = frames.shape[-2:]
        I = self.group
        if first % I:
            raise ValueError("DffClipEngine.run: first = %d is not a key frame (multiple of %d)" % (first, I))
        out, pending = [], []

        def flush():
            if not pending:
                return
            for s_ in (self._sb or []):
                torch.cuda.current_stream().wait_stream(s_)
            counts = torch.cat([p[3] for p in pending]).tolist()
            for (ob, os_, ol, _), n in zip(pending, counts):
                out.append(PostProcessor.materialize((ob, os_, ol, None), int(n), (W, H)))
            del pending[:]

        sig = (H, W, str(frames.device))
        if self.feat_ring is None or self._sig != sig:
            f = self._features(frames[0:1].float())
            self.key_feat = torch.zeros_like(f[0])
            self.feat_ring = self.key_feat                        # (what FgfaClipEngine._step asks for the device)
            self.key_img = frames.new_zeros((1, 3, H, W), dtype=torch.float32)
            self.cur_imgs = frames.new_zeros((I, 3, H, W), dtype=torch.float32)
            self._sig = sig
            self.graph = None
        cache = {}
        k0 = first
        while k0 < last:
            if k0 not in cache:                                   # the backbone of the next `lookahead` key frames in one batch
                ids = [k for k in range(k0, L, I)][:self.lookahead]
                ids = ids + [ids[-1]] * (self.lookahead - len(ids))
                fb = self._features(frames[torch.tensor(ids, device=frames.device)].float())
                cache = {k: fb[j] for j, k in enumerate(ids)}
            n = min(I, last - k0, L - k0)
            self.key_feat.copy_(cache[k0])
            self.key_img.copy_(frames[k0:k0 + 1])
            self.cur_imgs[:n].copy_(frames[k0:k0 + n])
            if n < I:                                             # a short last interval: the graph's shape stays, the tail repeats
                self.cur_imgs[n:].copy_(frames[k0 + n - 1:k0 + n].expand(I - n, -1, -1, -1))
            pending.extend(self._step((W, H), n))
            k0 += I
            if len(pending) >= sync_every:
                flush()
        flush()
        return out


class BaseClipEngine(FgfaClipEngine):
    """Clip-level driver for the single-frame GeneralizedRCNN (BASELINE configs[0], detector/generalized_rcnn.py:16-65): no
    cross-frame state, so graph A is simply the backbone on `group` consecutive frames (one batched launch chain instead of
    `group` single-frame ones) and graph B -- RPN selection, res5 + ROIAlign + fc6 / fc7, predictor, post-processing of one
    frame -- replays per frame on two lanes / streams beside it (FgfaClipEngine._step).  Detections are identical to
    `model(image)` frame by frame (tests/test_e2e_gpu.py::test_base_engine_equals_model)."""

    def __init__(self, model, group=20, graphs=True, pipeline=True, lanes=2, batch_head=True):
        self.m = model
        self.batch_head = bool(batch_head)   # the box head of the group's frames as ONE batched graph (else per frame, on lanes)
        self.pipeline = pipeline
        self.group = int(group)
        self.lanes = max(1, int(lanes))
        self.fork_select = self.lanes == 1
        self._sb = None
        self.use_graphs = graphs
        self.graph = None
        self.replays = 0
        self.feat_ring = None
        self.keep_intermediates = False

    def _body_a(self):
        maps = _nhwc(self.m.backbone(self.cur_imgs)[0]).contiguous()            # [group,h,w,1024]
        if self.keep_intermediates:
            self._dbg_a = [(None, maps[i]) for i in range(self.group)]
        return maps

    @torch.no_grad()
    def run(self, frames, first=0, last=None, sync_every=40):
        """frames: preprocessed f32 [L,3,H,W] on the device.  Frames first..last-1 -> list[BoxList]."""
        L = frames.shape[0]
        last = L if last is None else last
        H, W = frames.shape[-2:]
        G = self.group
        out, pending = [], []

        def flush():
            if not pending:
                return
            for s_ in (self._sb or []):
                torch.cuda.current_stream().wait_stream(s_)
            counts = torch.cat([p[3] for p in pending]).tolist()
            for (ob, os_, ol, _), n in zip(pending, counts):
                out.append(PostProcessor.materialize((ob, os_, ol, None), int(n), (W, H)))
            del pending[:]

        sig = (H, W, str(frames.device))
        if self.feat_ring is None or self._sig != sig:
            self.cur_imgs = frames.new_zeros((G, 3, H, W), dtype=torch.float32)
            self.feat_ring = self.cur_imgs                        # (what FgfaClipEngine._step asks for the device)
            self._sig = sig
            self.graph = None
        k0 = first
        while k0 < last:
            n = min(G, last - k0)
            self.cur_imgs[:n].copy_(frames[k0:k0 + n])
            if n < G:
                self.cur_imgs[n:].copy_(frames[k0 + n - 1:k0 + n].expand(G - n, -1, -1, -1))
            pending.extend(self._step((W, H), n))
            k0 += n
            if len(pending) >= sync_every:
                flush()
        flush()
        return out

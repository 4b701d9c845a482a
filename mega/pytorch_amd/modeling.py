"""Host-side mirror of the reference's mega_core.modeling modules for the MEGA inference path, running on
the HIP kernels of libmega_hip.so.  Same class names, registry names, constructor signatures
``(cfg, in_channels)``, call signatures at the seams and ``state_dict`` keys as the reference
(SURVEY.md 8b), so a reference checkpoint loads with ``load_state_dict`` and the detector drops into
``tools/test_net.py`` / ``demo/predictor.py``.

Reference files mirrored:
  detector/generalized_rcnn_mega.py:21-225     GeneralizedRCNNMEGA
  backbone/resnet.py:81-366, backbone.py:12-20 ResNet C4 body / stem / bottleneck, build_backbone
  rpn/rpn.py:73-125,:200-260, rpn/inference.py RPNHead, RPNWithRefModule, RPNPostProcessor
  roi_heads/box_head/roi_box_feature_extractors.py:457-933   MEGAFeatureExtractor (test path)
  roi_heads/box_head/{box_head.py:65-124, roi_box_predictors.py:35-57, inference.py:12-149}, roi_heads.py:9-76

Layout contract at the seams: feature maps are logical NCHW tensors in channels-last memory
(``x.permute(0,2,3,1)`` is contiguous), dtype = the compute dtype (cfg.DTYPE: float32 | bfloat16).
There is no CPU path: modules raise if the kernels are unavailable.
"""
import os
from collections import OrderedDict, deque

import torch
from torch import nn

from . import ops
from .relation import (RelationWeights, cat_rows, cat_rows_many, position_logits_for, relation_attend, relation_attend_batched,
                       relation_attention_forward, relation_project_batched, op_dtype, project_v)
from .structures import BoxList, cat_boxlist, to_image_list
from .synth import _cell_anchors

_DTYPES = {"float32": torch.float32, "bfloat16": torch.bfloat16, "float": torch.float32, "bf16": torch.bfloat16,
           "float16": torch.float16, "fp16": torch.float16, "half": torch.float16}
_HALF = (torch.bfloat16, torch.float16)      # the 16-bit matrix-core operand types (same MFMA rate, 8 / 11 significant bits)


def compute_dtype(cfg):
    return _DTYPES[str(getattr(cfg, "DTYPE", "float32"))]


def stream_dtype(cfg):
    """dtype of the aggregation head's ACTIVATION STREAM -- the fc0 output, x + attention, the stage FC outputs, the
    predictor input (roi_box_feature_extractors.py:806-829,:898-933).  f32 compute: f32.  bf16 compute: cfg.HEAD_STREAM,
    default float32: the Wq / Wk / Wv projections, Q K^T and P V run on the bf16 matrix cores as before (their errors
    average out over the keys), but nothing on the direct path proposal features -> class logits is rounded to bf16: the
    stage FCs and the predictor run in exact-f32 MFMA on the f32 stream.  "bfloat16" restores the round-3 hand-offs (8
    bf16 roundings of the stream: logit error median 1.7e-3 against the f32 oracle instead of ~1e-4)."""
    if compute_dtype(cfg) == torch.float32:
        return torch.float32
    st = _DTYPES[str(getattr(cfg, "HEAD_STREAM", "float32"))]
    if compute_dtype(cfg) == torch.float16 and st != torch.float32:
        raise ValueError("DTYPE float16 keeps the head's activation stream in float32 (HEAD_STREAM)")
    return st


def head_dtype(cfg):
    """operand dtype of the aggregation head's matrix-core GEMMs / attention (Wq, Wk, Wv projections, Q K^T, P V, the position
    term).  The compute dtype, except in float16 mode: there cfg.HEAD_DTYPE chooses -- "bfloat16" (default): the FRAME STAGE
    runs in fp16 and the head is the bf16 head (f32 activation stream, bf16 rounded copies into the projections / Q K^T / P V;
    its own error, 1.0-1.8e-4 of the logits, is not what limits the fp16 frame stage: 5.5e-4 -> 5.4e-4 with the head in fp16);
    "float16" (round 6): the head's kernels instantiated for IEEE-half operands (relation.hip: v_mfma_f32_32x32x16_f16 /
    16x16x32_f16, fp16 tile-ordered position logits) -- the bf16 head's rate and bytes at an eighth of its rounding noise.
    Its use is BEHIND ANOTHER MODE'S FRAME STAGE (engine.ClipEngine(frame_model=...)): split-precision frame stage + fp16 head
    holds the parity mode's bounds (logit error median 1.3-2.5e-5, p99 <= 1.5e-4, 100 % of the proposals and detections:
    tests/test_e2e_gpu.py::test_r101_600x1000_f16_head_vs_oracle) at the bf16 head's speed."""
    d = compute_dtype(cfg)
    if d == torch.float16:
        return _DTYPES[str(getattr(cfg, "HEAD_DTYPE", "bfloat16"))]
    return d


def conv_mode(cfg):
    """How the frame stage's convolutions / first FC run:
      "f32"   cfg.DTYPE float32: exact-f32 MFMA (the parity mode, 157 TF/s roof)
      "x3"    cfg.DTYPE float32 + cfg.F32_CONV "bf16x3": split-precision -- activations live as [hi | lo] bf16 planes
              (ops.Planes), every conv contracts [hi | lo | hi] against [Wh | Wh | Wl] on the bf16 matrix cores with f32
              accumulation (x.W to ~2^-16); the stem, the narrow RPN outputs and ROIAlign stay exact f32.  The parity mode
              at matrix-core rates.
      "bf16"  cfg.DTYPE bfloat16 (what bench.py times)
      "h2"    cfg.DTYPE float16 + cfg.F16_CONV "x2" (round 6): the two-pass form -- activations as float16 [hi | lo] planes
              (x to ~2^-22), every conv / fc0 ONE fp16 GEMM over K x 2 against weights [W | W] rounded to fp16 once:
              exact activations x single-rounded weights at twice the fp16 mode's matrix-core work; stem, the narrow RPN
              outputs and ROIAlign exact f32 as in "x3"; the head is the bf16 head
      "f16"   cfg.DTYPE float16 (round 6): the same kernels instantiated for IEEE half operands -- identical MFMA rate and
              bytes, 11 significant bits instead of 8 (1/8 of the rounding noise); activations are bounded by 65 504
      "wide"  cfg.DTYPE bfloat16 + cfg.RESIDUAL_STREAM "planes": bf16 convolutions, but the residual trunk of layer1-3 /
              res5 (resnet.py:324-344 `out += identity`, 33 + 3 times in R-101-C4) is carried as planes: a conv reads the
              hi plane (= the bf16 tensor it always read), conv3 adds hi + lo in f32 and writes both planes, so the trunk is
              never rounded to 8 bits."""
    if compute_dtype(cfg) == torch.float32:
        return "x3" if str(getattr(cfg, "F32_CONV", "exact")) == "bf16x3" else "f32"
    if compute_dtype(cfg) == torch.float16:
        return "h2" if str(getattr(cfg, "F16_CONV", "single")) == "x2" else "f16"
    return "wide" if str(getattr(cfg, "RESIDUAL_STREAM", "bfloat16")) == "planes" else "bf16"


def _nhwc(x):
    """logical NCHW / channels-last memory -> contiguous NHWC view (no copy when the contract holds)."""
    if isinstance(x, ops.Planes):
        return x
    y = x.permute(0, 2, 3, 1)
    return y if y.is_contiguous() else y.contiguous()


def _nchw_view(x_nhwc):
    return x_nhwc.permute(0, 3, 1, 2)


class Registry(dict):
    """mega_core/utils/registry.py:9-45."""

    def register(self, name, module=None):
        if module is not None:
            assert name not in self
            self[name] = module
            return module

        def deco(fn):
            assert name not in self
            self[name] = fn
            return fn
        return deco


BACKBONES = Registry()
RPN_HEADS = Registry()
ROI_BOX_FEATURE_EXTRACTORS = Registry()
ROI_BOX_PREDICTOR = Registry()
DETECTION_META_ARCHITECTURES = Registry()


# ================================================================================================= backbone
class FrozenBatchNorm2d(nn.Module):
    """layers/batch_norm.py:7-31 (buffers only; applied as the conv kernel's scale/bias epilogue)."""

    def __init__(self, n):
        super().__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))

    def folded(self):
        scale = self.weight.float() * self.running_var.float().rsqrt()
        bias = self.bias.float() - self.running_mean.float() * scale
        return scale.contiguous(), bias.contiguous()


def _pack_conv(conv, dtype):
    """OIHW parameter -> OHWI kernel operand in the compute dtype."""
    return conv.weight.detach().permute(0, 2, 3, 1).contiguous().to(dtype)


_PLANES3 = ("x3", "h2")          # conv modes whose EVERY conv reads / writes planes (three per bottleneck)


def _spk(mode):
    """conv_mode -> the x3 argument of ops.conv2d_sp: True ([hi | lo | hi] . [Wh | Wh | Wl], bf16), "h2" ([hi | lo] . [W | W],
    float16), False (the hi plane against plain bf16 weights: mode "wide")"""
    return {"x3": True, "h2": "h2"}.get(mode, False)


def _plane_dtype(mode):
    return torch.float16 if mode == "h2" else torch.bfloat16


def _pack_conv_h2(conv, dtype=None):
    """OIHW parameter -> the two-pass fp16 operand [Cout,R,S,2C] = [W | W] per tap (ops.conv2d_sp(x3="h2"))."""
    return ops.split_conv_weight_h2(conv.weight.detach().float().permute(0, 2, 3, 1).contiguous())


def _pack_conv_x3(conv, dtype=None):
    """OIHW parameter -> split-precision operand [Cout,R,S,3C] = [Wh | Wh | Wl] per tap (ops.conv2d_sp(x3=True))."""
    return ops.split_conv_weight_x3(conv.weight.detach().float().permute(0, 2, 3, 1).contiguous())


class _Packed(nn.Module):
    """Modules that cache kernel-ready operands; invalidated on load_state_dict / dtype change."""

    def __init__(self):
        super().__init__()
        self._pk = None
        self._pk_key = None

    def _packed(self, dtype, device):
        key = (dtype, str(device))
        if self._pk is None or self._pk_key != key:
            with torch.no_grad():
                self._pk = self._pack(dtype, device)
            self._pk_key = key
        return self._pk

    def invalidate(self):
        self._pk = None
        for m in self.children():
            if hasattr(m, "invalidate"):
                m.invalidate()

    def _load_from_state_dict(self, *a, **k):
        self._pk = None
        return super()._load_from_state_dict(*a, **k)


class Bottleneck(_Packed):
    """backbone/resnet.py:239-344 with FrozenBN, stride_in_1x1=True: three fused conv+BN(+ReLU) launches,
    the residual add + final ReLU live in conv3's epilogue."""

    def __init__(self, in_channels, bottleneck_channels, out_channels, stride, dilation=1):
        super().__init__()
        self.downsample = None
        if in_channels != out_channels:
            self.down_stride = stride if dilation == 1 else 1
            self.downsample = nn.Sequential(nn.Conv2d(in_channels, out_channels, 1, bias=False),
                                            FrozenBatchNorm2d(out_channels))
        if dilation > 1:
            stride = 1
        self.stride, self.dilation = stride, dilation
        self.conv1 = nn.Conv2d(in_channels, bottleneck_channels, 1, bias=False)
        self.bn1 = FrozenBatchNorm2d(bottleneck_channels)
        self.conv2 = nn.Conv2d(bottleneck_channels, bottleneck_channels, 3, bias=False)
        self.bn2 = FrozenBatchNorm2d(bottleneck_channels)
        self.conv3 = nn.Conv2d(bottleneck_channels, out_channels, 1, bias=False)
        self.bn3 = FrozenBatchNorm2d(out_channels)

    def _pack(self, dtype, device):
        pk = {}
        pack = {"x3": _pack_conv_x3, "h2": _pack_conv_h2}.get(dtype, _pack_conv)
        for i in (1, 2, 3):
            s, b = getattr(self, "bn%d" % i).folded()
            pk["w%d" % i] = pack(getattr(self, "conv%d" % i), dtype).to(device)
            pk["s%d" % i], pk["b%d" % i] = s.to(device), b.to(device)
        if self.downsample is not None:
            s, b = self.downsample[1].folded()
            pk["wd"] = pack(self.downsample[0], dtype).to(device)
            pk["sd"], pk["bd"] = s.to(device), b.to(device)
        return pk

    sp_mode = None     # "x3" / "wide" (conv_mode): set on every block by the module that owns the stage

    def run_sp(self, x, out_mode="planes"):
        """The block on split-precision planes (conv_mode "x3" / "wide").  x: ops.Planes, or -- wide mode, the stem's
        output -- a plain bf16 tensor.  -> Planes, or out_mode "f32" / "bf16": a plain tensor (the last block of res5)."""
        x3 = _spk(self.sp_mode)
        pk = self._packed(self.sp_mode if x3 else torch.bfloat16, x.device)
        identity = x
        if self.downsample is not None:
            identity = ops.conv2d_sp(x, pk["wd"], pk["sd"], pk["bd"], stride=self.down_stride, out_mode="planes", x3=x3)
        elif not isinstance(x, ops.Planes):
            raise ValueError("an identity block needs its input as planes")
        if x3:
            t = ops.conv2d_sp(x, pk["w1"], pk["s1"], pk["b1"], stride=self.stride, relu=True, x3=x3)
            t = ops.conv2d_sp(t, pk["w2"], pk["s2"], pk["b2"], pad=self.dilation, dil=self.dilation, relu=True, x3=x3)
        else:       # bf16 inside the block: conv1 reads the hi plane, conv2 is the plain launch
            t = ops.conv2d_sp(x, pk["w1"], pk["s1"], pk["b1"], stride=self.stride, relu=True, out_mode="bf16", x3=False)
            t = ops.conv2d_nhwc(t, pk["w2"], pk["s2"], pk["b2"], pad=self.dilation, dil=self.dilation, relu=True)
        return ops.conv2d_sp(t, pk["w3"], pk["s3"], pk["b3"], residual=identity, relu=True, out_mode=out_mode, x3=x3)

    fuse = True        # layer1's identity blocks as ONE kernel (ops.bottleneck64); False / MEGA_FUSE_BOTTLENECK=0: three launches

    def _fusable(self, x):
        return (self.fuse and x.dtype in _HALF and x.is_cuda and self.downsample is None and self.stride == 1
                and self.dilation == 1 and self.conv1.in_channels == 256 and self.conv1.out_channels == 64
                and self.conv3.out_channels == 256 and x.shape[0] * (-(-x.shape[1] // 8)) * (-(-x.shape[2] // 16)) >= 128
                and os.environ.get("MEGA_FUSE_BOTTLENECK", "1") != "0")

    def _fusable_ds(self, x):
        return (self.fuse and x.dtype in _HALF and x.is_cuda and self.downsample is not None and self.stride == 1
                and self.down_stride == 1 and self.dilation == 1 and self.conv1.in_channels == 64
                and self.conv1.out_channels == 64 and self.conv3.out_channels == 256
                and x.shape[0] * (-(-x.shape[1] // 8)) * (-(-x.shape[2] // 16)) >= 128
                and os.environ.get("MEGA_FUSE_BOTTLENECK", "1") not in ("0", "id"))

    def run(self, x, out=None):
        """out: optional destination of the block's output (a contiguous [N,Ho,Wo,Cout] view, e.g. a batch slice)"""
        if isinstance(x, ops.Planes) or self.sp_mode is not None:
            assert out is None
            return self.run_sp(x)
        pk = self._packed(x.dtype, x.device)
        if out is not None:
            assert not self._fusable(x) and not self._fusable_ds(x)
        if self._fusable(x):
            return ops.bottleneck64(x, pk["w1"], pk["s1"], pk["b1"], pk["w2"], pk["s2"], pk["b2"], pk["w3"], pk["s3"], pk["b3"])
        if self._fusable_ds(x):
            return ops.bottleneck64_ds(x, pk["w1"], pk["s1"], pk["b1"], pk["w2"], pk["s2"], pk["b2"], pk["w3"], pk["s3"],
                                       pk["b3"], pk["wd"], pk["sd"], pk["bd"])
        identity = x
        if self.downsample is not None:
            identity = ops.conv2d_nhwc(x, pk["wd"], pk["sd"], pk["bd"], stride=self.down_stride)
        t = ops.conv2d_nhwc(x, pk["w1"], pk["s1"], pk["b1"], stride=self.stride, relu=True)
        t = ops.conv2d_nhwc(t, pk["w2"], pk["s2"], pk["b2"], pad=self.dilation, dil=self.dilation, relu=True)
        return ops.conv2d_nhwc(t, pk["w3"], pk["s3"], pk["b3"], residual=identity, relu=True, out=out)


def _make_stage(in_channels, bottleneck_channels, out_channels, block_count, first_stride, dilation=1):
    blocks, stride = [], first_stride
    for _ in range(block_count):
        blocks.append(Bottleneck(in_channels, bottleneck_channels, out_channels, stride, dilation))
        stride, in_channels = 1, out_channels
    return nn.Sequential(*blocks)


# MEGA_L3_SPLIT=1 (opt-in, same bits): layer3 runs over the two halves of a frame batch one after the other (see
# ResNet.forward).  Measured +0.3 % on one box, -0.05 % on another: neutral, so the whole batch per layer stays the default.
_L3_SPLIT = os.environ.get("MEGA_L3_SPLIT", "0") == "1"
# MEGA_STEM_POOL=0: the stem and its max-pool as two kernels (A/B leg; same bits)
_FUSE_STEM_POOL = os.environ.get("MEGA_STEM_POOL", "1") != "0"


class BaseStem(_Packed):
    """backbone/resnet.py:347-366: 7x7/2 conv + FrozenBN + ReLU (one direct-conv kernel), 3x3/2 max-pool."""

    def __init__(self, out_channels=64):
        super().__init__()
        self.conv1 = nn.Conv2d(3, out_channels, 7, stride=2, padding=3, bias=False)
        self.bn1 = FrozenBatchNorm2d(out_channels)

    def _pack(self, dtype, device):
        s, b = self.bn1.folded()
        w = self.conv1.weight.detach().float().permute(1, 2, 3, 0).reshape(147, 64).contiguous()
        pk = {"w": w.to(device), "s": s.to(device), "b": b.to(device), "w160": None}
        if dtype in _HALF:
            pk["w160"] = ops.pack_stem_weight_bf16(self.conv1.weight, dtype).to(device)
        return pk

    def run(self, img_nchw_f32, dtype):
        pk = self._packed(dtype, img_nchw_f32.device)
        # (float16 on the device: the fused kernel only -- stem_mfma_kernel / stem_conv_kernel have no fp16 instantiation)
        if pk["w160"] is not None and (_FUSE_STEM_POOL or (dtype == torch.float16 and img_nchw_f32.is_cuda)):
            return ops.stem_pool(img_nchw_f32, pk["w160"], pk["s"], pk["b"])
        y = ops.stem(img_nchw_f32, pk["w"], pk["s"], pk["b"], dtype, w_n160=pk["w160"])
        return ops.maxpool3x3s2(y)

    def run_u8(self, frames_u8, mean, to_bgr, dtype=torch.bfloat16):
        """bf16 / f16 mode: the stem reads the uint8 frames [N,H,W,3] themselves (preprocessing on the patch load)"""
        pk = self._packed(dtype, frames_u8.device)
        if _FUSE_STEM_POOL or dtype == torch.float16:
            return ops.stem_pool(frames_u8, pk["w160"], pk["s"], pk["b"], mean, to_bgr)
        return ops.maxpool3x3s2(ops.stem_u8(frames_u8, pk["w160"], pk["s"], pk["b"], mean, to_bgr))


_STAGE_BLOCKS = {"R-50-C4": (3, 4, 6), "R-101-C4": (3, 4, 23)}


class ResNet(nn.Module):
    """backbone/resnet.py:81-152 for the *-C4 bodies."""

    def __init__(self, cfg):
        super().__init__()
        self.dtype = compute_dtype(cfg)
        self.mode = conv_mode(cfg)
        blocks = _STAGE_BLOCKS[cfg.MODEL.BACKBONE.CONV_BODY]
        self.stem = BaseStem(cfg.MODEL.RESNETS.STEM_OUT_CHANNELS)
        in_ch = cfg.MODEL.RESNETS.STEM_OUT_CHANNELS
        self.stages = []
        for i, n in enumerate(blocks):
            mid, out = 64 * 2 ** i, cfg.MODEL.RESNETS.RES2_OUT_CHANNELS * 2 ** i
            name = "layer%d" % (i + 1)
            self.add_module(name, _make_stage(in_ch, mid, out, n, first_stride=int(i > 0) + 1))
            self.stages.append(name)
            in_ch = out
        if self.mode in ("x3", "wide", "h2"):
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    m.sp_mode = self.mode

    def forward(self, x, u8_norm=None):
        """x [N,3,H,W] f32 image batch -> [C4] (logical NCHW, channels-last memory, compute dtype).
        u8_norm = (mean, to_bgr) with x the uint8 frames [N,H,W,3] (bf16 mode): preprocessing fused into the stem.
        conv_mode "x3" / "wide": C4 as an f32 tensor (hi + lo of the planes run_nhwc returns)."""
        y = self.run_nhwc(x, u8_norm)
        if isinstance(y, ops.Planes):      # (the seam's tensor: f32 = hi + lo in mode "x3", the bf16 hi plane in mode "wide")
            y = y.float() if self.mode in _PLANES3 else y.hi().contiguous()
        return [_nchw_view(y)]

    def run_nhwc(self, x, u8_norm=None):
        """forward() without the NCHW view: -> C4 as a contiguous NHWC tensor, or as ops.Planes (conv_mode "x3" / "wide")"""
        if u8_norm is not None and self.mode == "h2":       # (the two-pass mode's stem is the exact-f32 one: preprocess first)
            x, u8_norm = ops.preprocess_frames(x.contiguous(), u8_norm[0], u8_norm[1]), None
        if u8_norm is not None:
            assert self.dtype in _HALF and x.dtype == torch.uint8
            y = self.stem.run_u8(x.contiguous(), u8_norm[0], u8_norm[1], self.dtype)
        else:
            y = self.stem.run(x.float().contiguous(), torch.float32 if self.mode == "h2" else self.dtype)
        if self.mode in _PLANES3:
            y = ops.split_planes(y, _plane_dtype(self.mode))   # the stem (exact f32 direct conv + max-pool) hands over its f32 map as planes
        for name in self.stages:
            blocks = list(getattr(self, name))
            n = y.shape[0]
            if name == "layer3" and _L3_SPLIT and not isinstance(y, ops.Planes) and y.is_cuda and y.dtype in _HALF and n >= 32 and n % 2 == 0:
                # layer3 in two halves of the batch, each through all its blocks: at 20 frames of 600x1000 the working set of a
                # conv3 + residual layer (221 MB) stays in the 256 MB Infinity Cache -- 4.6 TB/s of algorithmic bytes against
                # 3.5-3.8 at 40 frames (tools/gpu/l3_tiles.py); every conv is batch-invariant, so the bits do not change
                out = None
                for lo in (0, n // 2):
                    z = y[lo:lo + n // 2]
                    for bi, blk in enumerate(blocks):
                        if bi + 1 < len(blocks):
                            z = blk.run(z)
                        else:
                            if out is None:
                                out = torch.empty((n,) + tuple(z.shape[1:3]) + (blk.conv3.out_channels,), dtype=z.dtype,
                                                  device=z.device)
                            blk.run(z, out=out[lo:lo + n // 2])
                y = out
                continue
            for blk in blocks:
                y = blk.run(y)
        return y


@BACKBONES.register("R-50-C4")
@BACKBONES.register("R-101-C4")
def build_resnet_backbone(cfg):
    """backbone/backbone.py:12-20."""
    model = nn.Sequential(OrderedDict([("body", ResNet(cfg))]))
    model.out_channels = cfg.MODEL.RESNETS.BACKBONE_OUT_CHANNELS
    return model


def build_backbone(cfg):
    return BACKBONES[cfg.MODEL.BACKBONE.CONV_BODY](cfg)


class ResNetHead(nn.Module):
    """backbone/resnet.py:155-204 as built by roi_box_feature_extractors.py:462-472: res5, stride_init=1,
    dilation=RES5_DILATION, run on the whole C4 map."""

    def __init__(self, dilation=2):
        super().__init__()
        self.layer4 = _make_stage(1024, 512, 2048, 3, first_stride=1, dilation=dilation)
        self.out_channels = 2048

    def run(self, x, out_mode=None):
        """out_mode (conv_mode "x3" / "wide": x is ops.Planes): what the LAST block writes -- "f32" / "bf16" (a plain map for
        ROIAlign) or "planes" (a 1x1 reduce conv follows)"""
        blocks = list(self.layer4)
        for bi, blk in enumerate(blocks):
            if isinstance(x, ops.Planes):
                x = blk.run_sp(x, out_mode if bi + 1 == len(blocks) and out_mode else "planes")
            else:
                x = blk.run(x)
        return x


# ================================================================================================= RPN
class BufferList(nn.Module):
    """rpn/anchor_generator.py:11-31."""

    def __init__(self, buffers=None):
        super().__init__()
        for i, b in enumerate(buffers or []):
            self.register_buffer(str(i), b)

    def __len__(self):
        return len(self._buffers)

    def __iter__(self):
        return iter(self._buffers.values())


class AnchorGenerator(nn.Module):
    """rpn/anchor_generator.py:34-125.  Only the cell anchors are state; the [H*W*A,4] grid is generated on the
    fly inside the proposal kernel from (cell anchor, x*stride, y*stride)."""

    def __init__(self, sizes, aspect_ratios, anchor_strides, straddle_thresh=0):
        super().__init__()
        assert len(anchor_strides) == 1, "C4 path: single feature level"
        self.strides = anchor_strides
        self.cell_anchors = BufferList([_cell_anchors(anchor_strides[0], sizes, aspect_ratios)])
        self.straddle_thresh = straddle_thresh

    def num_anchors_per_location(self):
        return [len(c) for c in self.cell_anchors]


@RPN_HEADS.register("SingleConvRPNHead")
class RPNHead(_Packed):
    """rpn/rpn.py:73-106.  The two 1x1 convs are merged into one 5A-wide GEMM with f32 output."""

    def __init__(self, cfg, in_channels, num_anchors):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, in_channels, 3, padding=1)
        self.cls_logits = nn.Conv2d(in_channels, num_anchors, 1)
        self.bbox_pred = nn.Conv2d(in_channels, num_anchors * 4, 1)
        for l in (self.conv, self.cls_logits, self.bbox_pred):
            nn.init.normal_(l.weight, std=0.01)
            nn.init.constant_(l.bias, 0)

    def _pack(self, dtype, device):
        w2 = torch.cat([self.cls_logits.weight.detach(), self.bbox_pred.weight.detach()], dim=0)
        b2 = torch.cat([self.cls_logits.bias.detach(), self.bbox_pred.bias.detach()], dim=0)
        if dtype in _PLANES3:   # split-precision 3x3 conv -> f32; the narrow 1x1 outputs stay exact f32
            return {"w1": (_pack_conv_x3 if dtype == "x3" else _pack_conv_h2)(self.conv).to(device), "b1": self.conv.bias.detach().float().to(device).contiguous(),
                    "w2": w2.permute(0, 2, 3, 1).contiguous().float().to(device), "b2": b2.float().to(device).contiguous()}
        return {"w1": _pack_conv(self.conv, dtype).to(device), "b1": self.conv.bias.detach().float().to(device).contiguous(),
                "w2": w2.permute(0, 2, 3, 1).contiguous().to(dtype).to(device), "b2": b2.float().to(device).contiguous()}

    sp_mode = None     # "x3" / "wide" when the backbone hands over C4 as ops.Planes (set by RPNWithRefModule)
    conv_ksplit = None  # the 3x3 conv's caller-chosen split-K (ops.conv2d_nhwc): set by the ONE-frame detectors (FGFA / DFF /
                        # base: 2394 rows x K = 9216 on 64 x 64 tiles, 94 -> 63 us with 4 K ranges); MEGA's batched frame
                        # stage never sets it (a frame's bits must not depend on the batch it is computed in)

    def run(self, feat_nhwc):
        """-> [B, H*W, 5A] f32 (channel a = objectness of anchor a, A + 4a + j = delta j)."""
        if isinstance(feat_nhwc, ops.Planes):
            x3 = _spk(self.sp_mode)
            pk = self._packed(self.sp_mode if x3 else torch.bfloat16, feat_nhwc.device)
            t = ops.conv2d_sp(feat_nhwc, pk["w1"], None, pk["b1"], pad=1, relu=True, out_mode="f32" if x3 else "bf16", x3=x3)
            o = ops.conv2d_nhwc(t, pk["w2"], None, pk["b2"], out_dtype=torch.float32)
            B, H, W, C = o.shape
            return o.view(B, H * W, C)
        pk = self._packed(feat_nhwc.dtype, feat_nhwc.device)
        t = ops.conv2d_nhwc(feat_nhwc, pk["w1"], None, pk["b1"], pad=1, relu=True, ksplit=self.conv_ksplit)
        o = ops.conv2d_nhwc(t, pk["w2"], None, pk["b2"], out_dtype=torch.float32)
        B, H, W, C = o.shape
        return o.view(B, H * W, C)

    def forward(self, x):
        """reference signature: list of features -> (logits list [N,A,H,W], bbox_reg list [N,4A,H,W])."""
        logits, reg = [], []
        for f in x:
            o = self.run(_nhwc(f))
            B, _, H, W = f.shape
            A = self.cls_logits.weight.shape[0]
            o = o.view(B, H, W, 5 * A).permute(0, 3, 1, 2)
            logits.append(o[:, :A])
            reg.append(o[:, A:])
        return logits, reg


class RPNWithRefModule(nn.Module):
    """rpn/rpn.py:200-243 (+ RPNModule :110-197, RPNPostProcessor rpn/inference.py:13-149), test path only."""

    def __init__(self, cfg, in_channels):
        super().__init__()
        c = cfg.MODEL.RPN
        self.anchor_generator = AnchorGenerator(c.ANCHOR_SIZES, c.ASPECT_RATIOS, c.ANCHOR_STRIDE, c.STRADDLE_THRESH)
        self.head = RPN_HEADS[c.RPN_HEAD](cfg, in_channels, self.anchor_generator.num_anchors_per_location()[0])
        self.pre_nms_top_n = {"key": c.PRE_NMS_TOP_N_TEST, "ref": cfg.MODEL.VID.RPN.REF_PRE_NMS_TOP_N}
        self.post_nms_top_n = {"key": c.POST_NMS_TOP_N_TEST, "ref": cfg.MODEL.VID.RPN.REF_POST_NMS_TOP_N}
        self.nms_thresh, self.min_size = c.NMS_THRESH, c.MIN_SIZE
        self.strict_gt = bool(getattr(cfg, "NMS_STRICT_GT", True))
        if conv_mode(cfg) in ("x3", "wide", "h2"):
            self.head.sp_mode = conv_mode(cfg)
        self.keep_index = False       # tests: frame records also carry the kept proposals' flat anchor indices

    def propose(self, feat_nhwc, im_w, im_h, version="key", want_index=False, select_stream=None, hold=None):
        """Batched, sync-free: -> (proposals [B,post,4], objectness [B,post], counts [B] i32[, anchor index [B,post] i32])
        on device.  select_stream: the selection kernels (one block per frame: top-k, decode, NMS) are launched on that
        stream, forked from the current one after the head's convs; the CALLER joins (current.wait_stream(select_stream))
        before it reads the results and keeps `hold` (a list that receives the workspace) alive until then."""
        rpn_out = self.head.run(feat_nhwc)
        B, H, W, _ = feat_nhwc.shape
        cell = next(iter(self.anchor_generator.cell_anchors)).to(feat_nhwc.device).float().contiguous()
        if select_stream is not None:
            select_stream.wait_stream(torch.cuda.current_stream(feat_nhwc.device))
            if hold is not None:
                hold.append(rpn_out)
        with ops.launch_on(select_stream):
            return ops.rpn_select(rpn_out, cell, H, W, self.anchor_generator.strides[0], self.pre_nms_top_n[version],
                                  self.post_nms_top_n[version], self.nms_thresh, self.min_size, im_w, im_h, self.strict_gt,
                                  want_index=want_index, hold=hold)

    def forward(self, images, features, targets=None, version="key"):
        if self.training:
            raise NotImplementedError("inference path only (training is out of scope, SURVEY.md section 2)")
        images = to_image_list(images)
        im_h, im_w = images.image_sizes[0]
        props, scores, cnt = self.propose(_nhwc(features[0]), im_w, im_h, version)
        boxes = []
        for b, n in enumerate(cnt.tolist()):
            bl = BoxList(props[b, :n], (im_w, im_h), "xyxy")
            bl.add_field("objectness", scores[b, :n])
            boxes.append(bl)
        return (boxes, {}) if version == "key" else boxes


def build_rpn(cfg, in_channels):
    """rpn/rpn.py:246-262: METHOD 'mega' -> RPNWithRefModule."""
    # "mega" -> RPNWithRefModule; "fgfa" / "base" use the plain RPNModule in the reference, which is the "key" path
    # of the same module (identical parameters and proposals)
    assert cfg.MODEL.VID.METHOD in ("mega", "rdn", "fgfa", "dff", "base")
    return RPNWithRefModule(cfg, in_channels)


# ================================================================================================= box head
def convert_to_roi_format(boxes):
    """modeling/poolers.py:78-89."""
    rows = []
    for i, b in enumerate(boxes):
        bb = b.bbox if isinstance(b, BoxList) else b
        ids = torch.full((bb.shape[0], 1), float(i), dtype=torch.float32, device=bb.device)
        rows.append(torch.cat([ids, bb.float()], dim=1))
    return torch.cat(rows, dim=0).contiguous()


def _linear(mod):
    return nn.Linear(mod[0], mod[1])


@ROI_BOX_FEATURE_EXTRACTORS.register("MEGAFeatureExtractor")
class MEGAFeatureExtractor(_Packed):
    """roi_box_feature_extractors.py:457-933, test-time path (_forward_ref :885, _forward_test :898,
    generate_feats_test :754, update_memory :678, update_global :674, update_lm :690)."""

    def __init__(self, cfg, in_channels):
        super().__init__()
        rb = cfg.MODEL.ROI_BOX_HEAD
        vid = cfg.MODEL.VID
        self.head = ResNetHead(cfg.MODEL.RESNETS.RES5_DILATION)
        pooled_c = 2048
        self.conv = None
        if vid.ROI_BOX_HEAD.REDUCE_CHANNEL:
            self.conv = nn.Conv2d(2048, 256, 1)
            pooled_c = 256
        self.resolution = rb.POOLER_RESOLUTION
        self.scale = rb.POOLER_SCALES[0]
        self.sampling_ratio = rb.POOLER_SAMPLING_RATIO
        rep = rb.MLP_HEAD_DIM
        self.all_frame_interval = vid.MEGA.ALL_FRAME_INTERVAL
        att = vid.ROI_BOX_HEAD.ATTENTION
        assert att.ENABLE and att.GROUP == 16 and att.EMBED_DIM == 64 and rep == 1024, "kernels are built for 16x64 heads"
        self.embed_dim, self.groups, self.feat_dim, self.stage = att.EMBED_DIM, att.GROUP, rep, att.STAGE
        self.base_num = vid.RPN.REF_POST_NMS_TOP_N
        self.advanced_num = int(self.base_num * vid.MEGA.RATIO)
        in0 = pooled_c * self.resolution ** 2
        self.pooled_c = pooled_c
        self.l_fcs = nn.ModuleList([nn.Linear(in0 if i == 0 else rep, rep) for i in range(self.stage)])
        self.l_Wgs = nn.ModuleList([nn.Conv2d(self.embed_dim, self.groups, 1) for _ in range(self.stage)])
        self.l_Wqs = nn.ModuleList([nn.Linear(rep, rep) for _ in range(self.stage)])
        self.l_Wks = nn.ModuleList([nn.Linear(rep, rep) for _ in range(self.stage)])
        self.l_Wvs = nn.ModuleList([nn.Conv2d(rep * self.groups, rep, 1, groups=self.groups) for _ in range(self.stage)])
        self.l_us = nn.ParameterList([nn.Parameter(torch.randn(self.groups, 1, self.embed_dim) * 0.01)
                                      for _ in range(self.stage)])
        self.memory_enable = vid.MEGA.MEMORY.ENABLE
        self.global_enable = vid.MEGA.GLOBAL.ENABLE
        if self.global_enable:
            self.global_size = vid.MEGA.GLOBAL.SIZE
            self.global_res_stage = vid.MEGA.GLOBAL.RES_STAGE
            n = self.global_res_stage + 1
            self.g_Wqs = nn.ModuleList([nn.Linear(rep, rep) for _ in range(n)])
            self.g_Wks = nn.ModuleList([nn.Linear(rep, rep) for _ in range(n)])
            self.g_Wvs = nn.ModuleList([nn.Conv2d(rep * self.groups, rep, 1, groups=self.groups) for _ in range(n)])
            self.g_us = nn.ParameterList([nn.Parameter(torch.randn(self.groups, 1, self.embed_dim) * 0.01)
                                          for _ in range(n)])
        else:
            self.global_res_stage = 0
        self.out_channels = rep
        self.dtype = compute_dtype(cfg)      # frame stage: res5, ROIAlign, fc0
        self.hdtype = head_dtype(cfg)        # the head's matrix-core operands (= self.dtype except in float16 mode, see head_dtype)
        self.stream = stream_dtype(cfg)
        self.mode = conv_mode(cfg)
        # cfg.F32_HEAD_LINEAR: "auto" (with F32_CONV "bf16x3": the head's Wq / Wk / Wv projections and stage FCs run in
        # split precision like the frame stage; the attention core, position logits and predictor stay exact f32) | "exact"
        self.head_x3 = self.mode == "x3" and str(getattr(cfg, "F32_HEAD_LINEAR", "auto")) != "exact"
        if self.mode in ("x3", "wide", "h2"):
            for m_ in self.head.modules():
                if isinstance(m_, Bottleneck):
                    m_.sp_mode = self.mode
        self.mem = None
        self.global_cache = None
        self.cache_memory_kv = True      # keep the Wk / Wv projections of memory rows (False: re-project every step)
        self.static_pools = None         # set by engine.StaticAggregation while it owns the pools

    # ---- kernel operands
    def _pack(self, dtype, device):
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        hd = self.hdtype if dtype == self.dtype else dtype
        sv = self.stream != hd         # f32 head stream over 16-bit operands: split Wv (relation.project_v)
        hx3 = self.head_x3             # conv_mode "x3": the head's projections / stage FCs in split precision too
        pk = {"local": [RelationWeights(sd, "", "l_", i, hd, device, with_pos=True, split_v=sv, x3=hx3)
                        for i in range(self.stage)],
              "global": [RelationWeights(sd, "", "g_", i, hd, device, with_pos=False, split_v=sv, x3=hx3)
                         for i in range(self.global_res_stage + 1)] if self.global_enable else []}
        # fc0 consumes the bin-major [K, 49, C] ROIAlign output: permute its columns from (c, ph, pw) to (ph, pw, c)
        w0 = self.l_fcs[0].weight.detach()
        r2 = self.resolution ** 2
        w0 = w0.view(w0.shape[0], self.pooled_c, r2).permute(0, 2, 1).reshape(w0.shape[0], -1)
        # (conv_mode "x3" reads fc0 as pk["fc0_x3"] only: no resident f32 copy of the 411 MB matrix beside it -- ADVICE r05)
        pk["fc_w"] = [None if self.mode in _PLANES3 else w0.contiguous().to(dtype).to(device)] + [
            self.l_fcs[i].weight.detach().to(dtype).to(device).contiguous() for i in range(1, self.stage)]
        pk["fc_b"] = [self.l_fcs[i].bias.detach().float().to(device).contiguous() for i in range(self.stage)]
        if self.stream != dtype:      # f32 activation stream in bf16 mode: the stage FCs read and write it at ~2^-16 --
            for i in range(1, self.stage):     # split-precision operands on the bf16 matrix cores (ops.split_bf16x3)
                pk["fc_w"][i] = ops.split_weight_bf16x3(self.l_fcs[i].weight.detach().float()).to(device)
        if self.conv is not None:
            pk["rc_w"] = _pack_conv(self.conv, dtype).to(device)
            pk["rc_b"] = self.conv.bias.detach().float().to(device).contiguous()
        if hx3:
            for i in range(1, self.stage):
                pk["fc_w"][i] = ops.X3Weight(self.l_fcs[i].weight, device)
        if self.mode in _PLANES3:  # split-precision fc0 (and reduce conv) operands
            splitw = ops.split_conv_weight_x3 if self.mode == "x3" else ops.split_conv_weight_h2
            pk["fc0_x3"] = splitw(w0.float().contiguous().view(w0.shape[0], 1, 1, -1)).view(w0.shape[0], -1).to(device)
            if self.conv is not None:
                pk["rc_w_x3"] = (_pack_conv_x3 if self.mode == "x3" else _pack_conv_h2)(self.conv).to(device)
        return pk

    # ---- per-frame stage (independent per frame; the multi-GPU sharding unit)
    def box_features(self, feat_nhwc, rois5):
        """res5 (+1x1 reduce) on the full C4 maps -> ROIAlign -> fc0 + ReLU.  feat [B,H,W,1024], rois5 [K,5]
        -> [K,1024].  (:885-896 and :898-907)"""
        return self.pooled_fc(self.res5_features(feat_nhwc), rois5)

    def res5_features(self, feat_nhwc):
        """the proposal-independent half of box_features: res5 (+1x1 reduce) on the full C4 maps"""
        pk = self._packed(self.dtype, feat_nhwc.device)
        if self.mode in ("x3", "wide", "h2") and not isinstance(feat_nhwc, ops.Planes):
            # the reference call signature (forward(x, proposals): x = the backbone's C4 TENSOR, f32 = hi + lo in mode "x3",
            # the bf16 hi plane in mode "wide"): back to planes, so that res5's blocks -- which run on planes in these modes --
            # end in a plain map for ROIAlign (the tensor branch handed a Planes object to the reduce conv)
            feat_nhwc = (ops.split_planes(feat_nhwc.float().contiguous(), _plane_dtype(self.mode)) if self.mode in _PLANES3
                         else feat_nhwc.contiguous())
        if isinstance(feat_nhwc, ops.Planes) or (self.mode == "wide" and self.head.layer4[0].sp_mode is not None):
            plain = "f32" if self.mode in _PLANES3 else "bf16"
            if self.conv is None:
                return self.head.run(feat_nhwc, out_mode=plain)
            x = self.head.run(feat_nhwc, out_mode="planes")
            return ops.conv2d_sp(x, pk["rc_w_x3"] if self.mode in _PLANES3 else pk["rc_w"], None, pk["rc_b"], relu=True,
                                 out_mode=plain, x3=_spk(self.mode))
        x = self.head.run(feat_nhwc)
        if self.conv is not None:
            x = ops.conv2d_nhwc(x, pk["rc_w"], None, pk["rc_b"], relu=True)
        return x

    def pooled_fc(self, x5, rois5):
        """ROIAlign on the res5 maps -> fc0 + ReLU"""
        pk = self._packed(self.dtype, x5.device)
        if self.mode in _PLANES3:
            # f32 ROIAlign (exact term order) writing planes -> split-precision fc0, in row chunks: a chunk's planes tensor
            # ([rows, 2 x 49 C] bf16) stays below the kernels' 2 GiB operand limit.  Rows are independent (split-K depends
            # on K alone), so the chunking does not change a row's bits.
            K = rois5.shape[0]
            per = max(1, min(K, (0x7FF00000 // (4 * self.pooled_c * self.resolution ** 2)) // 64 * 64))
            out = torch.empty((K, self.feat_dim), dtype=torch.float32, device=x5.device)
            for o in range(0, K, per):
                pooled = ops.roi_align_planes(x5, rois5[o:o + per].contiguous(), self.scale, (self.resolution, self.resolution),
                                              self.sampling_ratio, _plane_dtype(self.mode))
                out[o:o + per] = ops.linear_sp(pooled, pk["fc0_x3"], pk["fc_b"][0], relu=True)
            return out
        pooled = ops.roi_align(x5, rois5, self.scale, (self.resolution, self.resolution), self.sampling_ratio)
        return ops.linear(pooled.view(pooled.shape[0], -1), pk["fc_w"][0], pk["fc_b"][0], relu=True,
                          out_dtype=self.stream)

    # ---- test-time state (:657-688)
    def init_memory(self):
        n = self.all_frame_interval
        self.mem_queue_list = [{"rois": deque(maxlen=n), "feats": deque(maxlen=n),
                                "k": deque(maxlen=n), "vt": deque(maxlen=n)} for _ in range(self.stage)]   # k / vt:
        self.mem = [dict() for _ in range(self.stage)]      # the rows' Wk / Wv projections, kept with the rows

    def init_global(self):
        self.global_queue_list = [{"feats": deque(maxlen=self.global_size)}]
        self.global_cache = [dict()]

    def update_global(self, feats):
        self.global_queue_list[0]["feats"].append(feats)
        self.global_cache[0]["feats"] = torch.cat(list(self.global_queue_list[0]["feats"]), dim=0)

    def update_memory(self, i, cache):
        n = self.base_num if i == 0 else self.advanced_num
        self.mem_queue_list[i]["rois"].append(cache["rois_ref"][:n])
        self.mem[i] = {"rois": torch.cat(list(self.mem_queue_list[i]["rois"]), dim=0)}
        if not self.cache_memory_kv:     # raw features are only needed when their projections are not kept
            self.mem_queue_list[i]["feats"].append(cache["feats_ref"][:n])
            self.mem[i]["feats"] = torch.cat(list(self.mem_queue_list[i]["feats"]), dim=0)

    def _remember_kv(self, i, k, vt):
        """The rows update_memory(i, .) pushed this step are the first rows of this step's `ref`: keep their key /
        value projections (k [Nr,1024], vt [1024,ld] of the whole ref) with them -- next steps read them as memory."""
        n = self.mem_queue_list[i]["rois"][-1].shape[0]
        q = self.mem_queue_list[i]
        q["k"].append(k[:n])
        q["vt"].append(vt[:, :n].contiguous())      # contiguous pieces: the concatenation below is ONE batched copy
        self.mem[i]["k"] = torch.cat(list(q["k"]), dim=0)
        self.mem[i]["vt"] = torch.cat(list(q["vt"]), dim=1)

    def _push_memory(self, i, rois_n, k_n, vt_n):
        """update_memory + _remember_kv for rows that are already cut to this stage's entry size."""
        q = self.mem_queue_list[i]
        q["rois"].append(rois_n)
        q["k"].append(k_n)
        q["vt"].append(vt_n.contiguous())
        self.mem[i] = {"rois": torch.cat(list(q["rois"]), dim=0), "k": torch.cat(list(q["k"]), dim=0),
                       "vt": torch.cat(list(q["vt"]), dim=1)}

    def update_lm(self, feats, i=0):
        pk = self._packed(self.dtype, feats.device)
        return relation_attention_forward(pk["global"][i], feats, self.global_cache[-1]["feats"], residual=True)

    def _stage_fc(self, pk, i, x):
        """relu(l_fcs[i](x)) (:826-827) on the activation stream: in the stream's dtype; an f32 stream over bf16 compute
        goes through the bf16 matrix cores with split operands ([hi | lo | hi] . [Wh | Wh | Wl]: x . W to ~2^-16)."""
        if self.stream != self.dtype:
            return ops.linear(ops.split_bf16x3(x.contiguous()), pk["fc_w"][i], pk["fc_b"][i], relu=True,
                              out_dtype=torch.float32)
        return ops.linear(x, pk["fc_w"][i], pk["fc_b"][i], relu=True)

    # ---- aggregation for one key frame (:898-933 after the fc0 line)
    def aggregate(self, x, rois_key, rois, rois_dis, x_ref, dis_index=None, x_ref_dis=None):
        """x [nk,1024] key-frame fc0 features, rois_key [nk,4]; rois [Nl,4] / x_ref [Nl,1024] the local window
        (oldest frame first); rois_dis [Nd,4] with either dis_index [Nd] (rows of x_ref forming the 'dis' set;
        update_lm is row-wise, so update_lm(x_ref_dis) == update_lm(x_ref)[dis_index]) or, in the reference's
        call convention, the explicit x_ref_dis [Nd,1024] tensor."""
        pk = self._packed(self.dtype, x.device)
        nkey = x.shape[0]
        nl = x_ref.shape[0]
        if self.global_enable and self.global_cache and "feats" in self.global_cache[-1]:
            # :757-760 -- ONE launch chain for all query sets (rows are independent)
            parts = [x, x_ref] + ([x_ref_dis] if x_ref_dis is not None else [])
            z = self.update_lm(torch.cat(parts, dim=0))
            x, x_ref = z[:nkey], z[nkey:nkey + nl]
            if x_ref_dis is not None:
                x_ref_dis = z[nkey + nl:]
        if x_ref_dis is None:
            x_ref_dis = x_ref.index_select(0, dis_index)
        rois_cur01 = torch.cat([rois_key, rois_dis], dim=0)
        cache = [{"rois_cur": rois_cur01, "rois_ref": rois, "feats_cur": torch.cat([x, x_ref_dis], dim=0),
                  "feats_ref": x_ref}]
        for _ in range(self.stage - 2):
            cache.append({"rois_cur": rois_cur01, "rois_ref": rois_dis})
        cache.append({"rois_cur": rois_key, "rois_ref": rois_dis})
        sp = self.static_pools       # engine.StaticAggregation: pools as fixed-address tensors (hipGraph-able step)
        for i in range(self.stage):
            if sp is not None:
                memory = sp.read_memory(i)
            else:
                memory = self.mem[i] if self.mem[i] else None             # read BEFORE the push (:914-917)
                if self.memory_enable:
                    self.update_memory(i, cache[i])
            rois_cur, rois_ref = cache[i]["rois_cur"], cache[i]["rois_ref"]
            feats_cur, feats_ref = cache[i]["feats_cur"], cache[i]["feats_ref"]
            mem_kv = None
            if memory is not None:
                rois_ref = torch.cat([rois_ref, memory["rois"]], dim=0)
                if "k" in memory:      # projections made when these rows were in the local window (same values)
                    mem_kv = (memory["k"], memory["vt"])
                else:
                    feats_ref = torch.cat([feats_ref, memory["feats"]], dim=0)
            feats_cur, k_loc, vt_loc = relation_attention_forward(
                pk["local"][i], feats_cur.contiguous(), feats_ref.contiguous(), rois_cur.contiguous(),
                rois_ref.contiguous(), residual=True, mem_kv=mem_kv, return_kv=True)
            if sp is not None:          # the push comes after the read and after every use of the old rows
                n = self.base_num if i == 0 else self.advanced_num
                sp.push_memory(i, cache[i]["rois_ref"][:n], k_loc[:n], vt_loc[:, :n])
            elif self.memory_enable and self.cache_memory_kv:
                self._remember_kv(i, k_loc, vt_loc)
            if i != self.stage - 1:
                feats_cur = self._stage_fc(pk, i + 1, feats_cur)
            if i == self.stage - 1:
                x = feats_cur
            elif i == self.stage - 2:
                cache[i + 1]["feats_cur"] = feats_cur[:nkey]
                cache[i + 1]["feats_ref"] = feats_cur[nkey:]
            else:
                cache[i + 1]["feats_cur"] = feats_cur
                cache[i + 1]["feats_ref"] = feats_cur[nkey:]
        for i in range(self.global_res_stage):
            x = self.update_lm(x.contiguous(), i + 1)
        return x

    # ---- aggregation of SEVERAL consecutive key frames at once (engine batches)
    batched_attention = True  # the attention core / position logits of all key frames of a stage in ONE launch each
                              # (False: one relation_attend call per key frame -- the A/B switch of that change)

    def _update_lm_batched(self, xs, globs, i=0, also_cat=()):
        """update_lm (:690-699) for several key frames: xs[t] (a tensor or a tuple of row blocks) attends to globs[t]
        (that step's global pool).  Returns consecutive row blocks of one buffer (and, with also_cat, the extra
        concatenations that rode along in the same copy launch)."""
        pk = self._packed(self.dtype, globs[0].device)
        w = pk["global"][i]
        res = relation_project_batched(w, xs, globs, want_x=True, also_cat=also_cat, pad_refs=self.batched_attention)
        qs, ks, vts, xc = res[:4]
        if self.batched_attention:
            # the key sets are the projections' own (32-aligned, zero-padded) blocks: nothing to assemble per frame
            z = relation_attend_batched(w, [{"x": xc[t], "q": qs[t], "k_all": ks[t], "vt_all": vts[t],
                                             "Nk": ks[t].shape[0]} for t in range(len(xs))])
        else:
            z = [relation_attend(w, xc[t], qs[t], ks[t], vts[t]) for t in range(len(xs))]
        return (z, res[4]) if also_cat else z

    # aggregate_batch: the attention kernel reads a stage's key set [local window ; memory snapshot] as two segments where
    # they lie (MEGA_ATTN_SEGMENTS=0: copied into one K / V^T buffer per key frame first -- the A/B switch; same bits)
    attn_segments = os.environ.get("MEGA_ATTN_SEGMENTS", "1") != "0"
    # aggregate_batch, MEGA_EARLY_POS=1 (opt-in; same bits): the position logits of EVERY stage are computed at its start, on
    # a side stream, instead of inside each stage in stream order.  Measured round 4: -0.3 % with one step-batch per block,
    # -4 % with two (the VALU-bound position kernel beside the chain's GEMMs / attention slows those by as much as it hides,
    # and the forked hipGraph branch adds cross-queue dependencies): off by default.
    early_pos = os.environ.get("MEGA_EARLY_POS", "0") == "1"

    def _early_position_logits(self, pk, frames, own, rois_cur01):
        """The position logits of all stages of a step-batch, ahead of time.  They depend on boxes only -- the queries'
        boxes and, per stage, the key set's [local window ; memory snapshot] boxes, all known when the batch starts -- while
        the serial chain of a stage is projections -> attention -> FC.  The position kernel is VALU-bound (64 sin / cos per
        pair; 1.1 of the aggregation's 5.3 ms per 20 key frames), the chain's other kernels are matrix-core- or
        memory-bound: launched on a side stream at the start of the batch (forked from / joined to the current stream with
        events, so a hipGraph capture records them as a parallel branch) they run beside update_lm, the projections and
        the global attention instead of between them.
        -> dict(tape=[boxes tape of memory pool i, as _push_memory_batch lays it out], pos=[per stage: list over own],
                ev=[per stage: event to wait for before the attention], keep=...)."""
        S = len(frames)
        dev = frames[0]["x"].device
        rois_ref, new_rois, tgroups, tidx = [], [], [], []
        for i in range(self.stage):
            n_push = self.base_num if i == 0 else self.advanced_num
            rr = [f["rois"] if i == 0 else f["rois_dis"] for f in frames]
            rois_ref.append(rr)
            new_rois.append([r[:min(n_push, r.shape[0])] for r in rr])
            if self.memory_enable:
                have_old = len(self.mem_queue_list[i]["rois"]) > 0
                tidx.append(len(tgroups))
                tgroups.append(((([self.mem[i]["rois"]] if have_old else []) + new_rois[i]), 0))
            else:
                tidx.append(None)
        tapes = ops.multi_cat(tgroups) if tgroups else []
        tape = [None if j is None else tapes[j] for j in tidx]
        rgroups, nk = [], []
        for i in range(self.stage):
            snaps = {}
            if tape[i] is not None:       # the snapshot frame t reads: the live entries before its own push (:914-917)
                q = self.mem_queue_list[i]["rois"]
                cap, old = q.maxlen, [r.shape[0] for r in q]
                off = [0]
                for n in old + [r.shape[0] for r in new_rois[i]]:
                    off.append(off[-1] + n)
                for t in own:
                    hi = len(old) + t
                    if hi > 0:
                        snaps[t] = tape[i][off[max(0, hi - cap)]:off[hi]]
            pieces, n_i = [], {}
            for t in own:
                pieces.append(rois_ref[i][t])
                n_i[t] = rois_ref[i][t].shape[0]
                if t in snaps:
                    pieces.append(snaps[t])
                    n_i[t] += snaps[t].shape[0]
            rgroups.append((pieces, 0))
            nk.append(n_i)
        r_flat = ops.multi_cat(rgroups)
        side = None
        if dev.type == "cuda" and not ops.profiling():
            side = getattr(self, "_pos_stream", None)
            if side is None or side.device != dev:
                side = self._pos_stream = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
        pos, evs = [], []
        for i in range(self.stage):
            w = pk["local"][i]
            if not w.with_pos:
                pos.append(None)
                evs.append(None)
                continue
            last = i == self.stage - 1
            rq = [frames[t]["rois_key"] if last else rois_cur01[t] for t in own]
            rk, o = [], 0
            for t in own:
                rk.append(r_flat[i][o:o + nk[i][t]])
                o += nk[i][t]
            with ops.launch_on(side):          # (buffers allocated under the current stream, kernels on the side stream)
                pos.append(position_logits_for(w, rq, rk))
            ev = None
            if side is not None:
                ev = torch.cuda.Event()
                ev.record(side)
            evs.append(ev)
        return {"tape": tape, "pos": pos, "ev": evs, "nk": nk, "keep": (tapes, r_flat)}

    def _zero_cols(self, like, n):
        """[rows of like, n] zeros (n < 32), cut from one cached block: the pad columns between V^T blocks"""
        z = getattr(self, "_zpad", None)
        if z is None or z.dtype != like.dtype or z.device != like.device or z.shape[0] != like.shape[0]:
            z = like.new_zeros((like.shape[0], 32))
            self._zpad = z
        return z[:, :n]

    def aggregate_batch(self, frames, shard=None):
        """aggregate() for a list of consecutive key frames (oldest first), each a dict with the arguments of aggregate():
        x, rois_key, rois, rois_dis, x_ref, dis_index, and "glob" = the global pool [Ng,1024] of THAT step (or None).
        Returns the list of box-head outputs x [nk,1024] (entries of frames another rank owns are None).

        Why this is legal: the per-video state looks sequential (memory pools), but stage i of key frame t only
        depends on stage i-1 of key frames <= t -- what a step pushes into memory[i] are its stage-(i-1) OUTPUT rows
        of the oldest window frame (update_memory :678-688 is called before the attention, :914-917), never its own
        stage-i output.  The dependency depth is the number of stages, not the number of frames.  So the batch is
        processed stage by stage: the Wq / Wk / Wv projections and the stage FCs of all its frames run as ONE GEMM each
        (M = S x 675 ... S x 1875 rows instead of 64x64-tile launches at 4 % of the MFMA peak), the position logits and
        the attention core as one launch each.  All kernels are batch-invariant, so every frame's result has the same
        bits as aggregate()'s (tests: engine == reference call convention).

        Data movement: the memory pool a frame reads is a SLIDING window over the entries in push order, so per stage
        the old pool and the batch's new entries are laid out once as one "tape" (3 concatenations: boxes, K, V^T) and
        frame t's read-before-push snapshot is a row / column range of it; the key sets [local window ; memory] of all
        frames are then assembled by 3 more concatenations into flat buffers whose row (K, boxes) / 32-aligned column
        (V^T) blocks are handed to the kernels as views.  Per-frame torch.cat launches (7 per frame and stage before)
        were the largest kernel family of the aggregation; the operands' values and order are unchanged.

        shard (engine.KeyFrameShard, multi-GPU): the key frames of the batch are dealt round-robin to the ranks; a
        rank runs the stages only for its own frames.  What other frames need from a frame are its memory entries
        (75 / 15 / 15 rows): they are all-gathered once per stage (3 small collectives per BATCH), their Wk / Wv
        projections recomputed by every rank (one tiny GEMM pair per stage), and every rank replays all pushes in
        frame order, so the memory pools stay replicated.  No key frame is aggregated twice: the step no longer has a
        serial (replicated) part."""
        assert self.cache_memory_kv and self.static_pools is None
        pk = self._packed(self.dtype, frames[0]["x"].device)
        S = len(frames)
        own = [t for t in range(S) if shard is None or shard.owner(t) == shard.rank]
        nkey = [f["x"].shape[0] for f in frames]
        nl = [f["x_ref"].shape[0] for f in frames]
        ndis = [f["rois_dis"].shape[0] for f in frames]
        xs, x_refs = {}, {}
        use_glob = self.global_enable and frames[0].get("glob") is not None
        z = None
        rc_pieces = [r for t in own for r in (frames[t]["rois_key"], frames[t]["rois_dis"])]   # stage queries' boxes
        rc_all = None
        if use_glob and own:                                                     # :757-760
            z, extra = self._update_lm_batched([(frames[t]["x"], frames[t]["x_ref"]) for t in own],
                                               [frames[t]["glob"] for t in own], also_cat=(rc_pieces,))
            rc_all = extra[0]
            for j, t in enumerate(own):
                xs[t], x_refs[t] = z[j][:nkey[t]], z[j][nkey[t]:nkey[t] + nl[t]]
        elif own:
            for t in own:
                xs[t], x_refs[t] = frames[t]["x"], frames[t]["x_ref"]
        # stage-0 queries [key rows ; 'dis' rows of the window] of every own frame: ONE gather from the update_lm output
        feats_cur = {}
        sig = tuple((nkey[t], nl[t], frames[t].get("dis_key")) for t in own)
        if z is not None and own and all(sg[2] is not None for sg in sig):
            z_all = cat_rows(z)
            cache = getattr(self, "_cur_index", None)
            if cache is None or cache[0] != sig or cache[1].device != z_all.device:
                # built on the host from the window's proposal counts (dis_key; the same rule as _dis_index) and
                # uploaded once: the counts of an untrained RPN change from batch to batch, and the device-side
                # arange / add / cat form was 40 tiny launches per batch
                idx, o = [], 0
                an = self.advanced_num
                for t in own:
                    idx.append(torch.arange(o, o + nkey[t]))
                    off = o + nkey[t]
                    for n in frames[t]["dis_key"]:        # rows per window record: its first advanced_num are 'dis' rows
                        idx.append(torch.arange(off, off + min(an, n)))
                        off += n
                    assert off == o + nkey[t] + nl[t]
                    o += nkey[t] + nl[t]
                cache = (sig, torch.cat(idx).to(z_all.device, non_blocking=True))
                self._cur_index = cache
            cur_all = z_all.index_select(0, cache[1])
            o = 0
            for t in own:
                feats_cur[t] = cur_all[o:o + nkey[t] + ndis[t]]
                o += nkey[t] + ndis[t]
        else:
            for t in own:
                feats_cur[t] = torch.cat([xs[t], x_refs[t].index_select(0, frames[t]["dis_index"])], dim=0)
        feats_ref = x_refs
        rois_cur01 = {}
        if own:
            if rc_all is None:
                rc_all = cat_rows_many([rc_pieces])[0]
            o = 0
            for t in own:
                rois_cur01[t] = rc_all[o:o + nkey[t] + ndis[t]]
                o += nkey[t] + ndis[t]
        early = None
        if self.early_pos and self.batched_attention and own:
            early = self._early_position_logits(pk, frames, own, rois_cur01)
        for i in range(self.stage):
            last = i == self.stage - 1
            w = pk["local"][i]
            n_push = self.base_num if i == 0 else self.advanced_num
            rois_ref = [f["rois"] if i == 0 else f["rois_dis"] for f in frames]
            n_ent = [min(n_push, r.shape[0]) for r in rois_ref]           # rows of each frame's memory entry
            qs, ks, vts = {}, {}, {}
            # will every own frame read a memory snapshot (then the attention takes its key set as two segments)?  Frame t's
            # snapshot is the pool before its own push: empty only for the first frame of a video's first batch
            seg_ok = (self.attn_segments and self.batched_attention and self.memory_enable and bool(own)
                      and (len(self.mem_queue_list[i]["rois"]) > 0 or 0 not in own))
            if own:
                # (segments: every frame's key rows are followed by zero rows up to a multiple of 32, so its V^T block starts
                #  at a 32-aligned column of the GEMM's output -- the first segment's loads stay 16-byte aligned)
                q_, k_, v_ = relation_project_batched(w, [feats_cur[t] for t in own], [feats_ref[t] for t in own],
                                                      pad_refs=seg_ok)
                for j, t in enumerate(own):
                    qs[t], ks[t], vts[t] = q_[j], k_[j], v_[j][:, :k_[j].shape[0]]
            snaps = {}
            if self.memory_enable:
                if shard is not None:
                    # memory entries of ALL frames: gather the entry rows, project them here (same bits as the owner's)
                    ent = shard.gather_rows({t: feats_ref[t][:n_ent[t]] for t in own}, n_ent, frames[0]["x"])
                    e_all = op_dtype(w, torch.cat(ent, dim=0))
                    ek_all = ops.linear(e_all, w.wk, w.bk)
                    evt_all = project_v(w, e_all, (e_all.shape[0] + 31) // 32 * 32)
                    o, new_k, new_vt = 0, [], []
                    for t in range(S):
                        new_k.append(ek_all[o:o + n_ent[t]])
                        new_vt.append(evt_all[:, o:o + n_ent[t]])
                        o += n_ent[t]
                else:
                    new_k = [ks[t][:n_ent[t]] for t in range(S)]
                    new_vt = [vts[t][:, :n_ent[t]] for t in range(S)]
                snaps = self._push_memory_batch(i, [rois_ref[t][:n_ent[t]] for t in range(S)], new_k, new_vt,
                                                tr=None if early is None else early["tape"][i])
            jobs = []
            if own:
                # key sets [local window ; memory snapshot] of all own frames
                kp, vp, rp, Nk, ldv = [], [], [], {}, {}
                segments = seg_ok and all(t in snaps for t in own)
                assert segments == seg_ok
                for t in own:
                    m = snaps.get(t)
                    kp.append(ks[t]); vp.append(vts[t]); rp.append(rois_ref[t])
                    Nk[t] = ks[t].shape[0]
                    if m is not None:
                        kp.append(m["k"]); vp.append(m["vt"]); rp.append(m["rois"])
                        Nk[t] += m["k"].shape[0]
                    ldv[t] = (Nk[t] + 31) // 32 * 32
                    if ldv[t] > Nk[t]:
                        vp.append(self._zero_cols(vts[t], ldv[t] - Nk[t]))
                r_flat = None
                if early is None:
                    r_flat = ops.multi_cat([(rp, 0)])[0]
                else:                    # (the boxes of the key sets were laid out, and used, before the stages)
                    assert all(early["nk"][i][t] == Nk[t] for t in own)
                if segments:
                    # the attention kernel reads both parts where they lie -- the projections' output (ks / vts) and the
                    # memory tape (the snapshot's row / column range) -- as two key segments: no K / V^T copies at all
                    # (they were 0.3 ms per 20 key frames: 155 MB each way at stage 0)
                    ok = 0
                    for t in own:
                        rc = frames[t]["rois_key"] if last else rois_cur01[t]
                        m = snaps[t]
                        jobs.append({"x": feats_cur[t], "q": qs[t], "k": ks[t], "vt": vts[t], "N1": ks[t].shape[0],
                                     "k2": m["k"], "vt2": m["vt"], "Nk": Nk[t], "rois_q": rc,
                                     "rois_k": None if r_flat is None else r_flat[ok:ok + Nk[t]]})
                        ok += Nk[t]
                else:
                    # assembled by concatenation
                    k_flat = ops.multi_cat([(kp, 0)])[0]
                    vt_flat = ops.multi_cat([(vp, 1)])[0]
                    ok = oc = 0
                    for t in own:
                        rc = frames[t]["rois_key"] if last else rois_cur01[t]
                        jobs.append({"x": feats_cur[t], "q": qs[t], "k_all": k_flat[ok:ok + Nk[t]],
                                     "vt_all": vt_flat[:, oc:oc + ldv[t]], "Nk": Nk[t], "rois_q": rc,
                                     "rois_k": None if r_flat is None else r_flat[ok:ok + Nk[t]]})
                        ok += Nk[t]
                        oc += ldv[t]
            if self.batched_attention:
                pos_i = None
                if early is not None and early["pos"][i] is not None:
                    pos_i = early["pos"][i]
                    if early["ev"][i] is not None:          # join: the side stream's launch of this stage's logits
                        torch.cuda.current_stream().wait_event(early["ev"][i])
                outs = dict(zip(own, relation_attend_batched(w, jobs, pos=pos_i)))
                if early is not None:
                    early["pos"][i] = None                  # (frees the stage's logits as before: after its attention)
            else:
                outs = dict(zip(own, [relation_attend_batched(w, [j])[0] for j in jobs]))
            if last:
                xs = outs
                break
            feats_cur, feats_ref = {}, {}
            if own:
                ncur = [outs[t].shape[0] for t in own]
                fc = self._stage_fc(pk, i + 1, cat_rows([outs[t] for t in own]))
                o = 0
                for j, t in enumerate(own):
                    nx = fc[o:o + ncur[j]]
                    o += ncur[j]
                    feats_cur[t] = nx[:nkey[t]] if i == self.stage - 2 else nx
                    feats_ref[t] = nx[nkey[t]:]
        for i in range(self.global_res_stage):                                   # :930-931
            if own:
                z = self._update_lm_batched([xs[t] for t in own], [frames[t]["glob"] for t in own], i + 1)
                for j, t in enumerate(own):
                    xs[t] = z[j]
        return [xs.get(t) for t in range(S)]

    def _push_memory_batch(self, i, new_rois, new_k, new_vt, tr=None):
        """The pushes of S consecutive key frames into memory[i] (update_memory + _remember_kv, in frame order) as one
        tape: [live entries ; the S new entries] laid out by three concatenations.  Returns {t: snapshot} where
        snapshot = dict(rois, k, vt) views of the pool frame t READS (the live entries before its own push, :914-917;
        absent while the pool is empty).  The deques / self.mem[i] are left as S single pushes would leave them
        (their tensors are views of the tape).  tr: the boxes tape if it was laid out earlier."""
        q = self.mem_queue_list[i]
        cap = q["rois"].maxlen
        old = [r.shape[0] for r in q["rois"]]
        E0, S = len(old), len(new_rois)
        have_old = E0 > 0
        # (one launch for all three tapes: the copy kernel takes any alignment, so the 150-byte V^T column blocks ride along)
        groups = [(([self.mem[i]["k"]] if have_old else []) + list(new_k), 0),
                  (([self.mem[i]["vt"]] if have_old else []) + list(new_vt), 1)]
        if tr is None:
            groups.append(((([self.mem[i]["rois"]] if have_old else []) + list(new_rois)), 0))
            tk, tv, tr = ops.multi_cat(groups)
        else:      # the boxes tape already exists (_early_position_logits laid it out from the same pieces)
            tk, tv = ops.multi_cat(groups)
        off = [0]
        for n in old + [r.shape[0] for r in new_rois]:
            off.append(off[-1] + n)

        def view(lo, hi):      # entries lo .. hi-1 of the tape
            a, b = off[lo], off[hi]
            return {"rois": tr[a:b], "k": tk[a:b], "vt": tv[:, a:b]}
        snaps = {}
        for t in range(S):
            hi = E0 + t
            if hi > 0:
                snaps[t] = view(max(0, hi - cap), hi)
        for t in range(S):
            e = view(E0 + t, E0 + t + 1)
            q["rois"].append(e["rois"])
            q["k"].append(e["k"])
            q["vt"].append(e["vt"])
        self.mem[i] = view(max(0, E0 + S - cap), E0 + S)
        return snaps

    # ---- reference call signatures
    def forward(self, x, proposals, pre_calculate=False, key_features=None):
        if self.training:
            raise NotImplementedError("inference path only")
        if pre_calculate:                                                  # _forward_ref (:885-896)
            return self.box_features(_nhwc(x), convert_to_roi_format(proposals))
        props, proposals_ref, proposals_ref_dis, x_ref, x_ref_dis = proposals  # _forward_test (:898-933)
        xk = key_features if key_features is not None else self.box_features(_nhwc(x), convert_to_roi_format(props))
        return self.aggregate(xk, props[0].bbox, proposals_ref.bbox, proposals_ref_dis.bbox, x_ref,
                              x_ref_dis=x_ref_dis)


def make_roi_box_feature_extractor(cfg, in_channels):
    return ROI_BOX_FEATURE_EXTRACTORS[cfg.MODEL.ROI_BOX_HEAD.FEATURE_EXTRACTOR](cfg, in_channels)


@ROI_BOX_PREDICTOR.register("FPNPredictor")
class FPNPredictor(_Packed):
    """roi_box_predictors.py:35-57; both Linears as one 5*NC-wide GEMM with f32 output."""

    def __init__(self, cfg, in_channels):
        super().__init__()
        nc = cfg.MODEL.ROI_BOX_HEAD.NUM_CLASSES
        self.num_classes = nc
        self.cls_score = nn.Linear(in_channels, nc)
        self.bbox_pred = nn.Linear(in_channels, nc * 4)
        nn.init.normal_(self.cls_score.weight, std=0.01)
        nn.init.normal_(self.bbox_pred.weight, std=0.001)
        for l in (self.cls_score, self.bbox_pred):
            nn.init.constant_(l.bias, 0)

    def _pack(self, dtype, device):
        w = torch.cat([self.cls_score.weight.detach(), self.bbox_pred.weight.detach()], dim=0)
        b = torch.cat([self.cls_score.bias.detach(), self.bbox_pred.bias.detach()], dim=0)
        return {"w": w.to(dtype).to(device).contiguous(), "b": b.float().to(device).contiguous()}

    def forward(self, x):
        pk = self._packed(x.dtype, x.device)
        o = ops.linear(x.contiguous(), pk["w"], pk["b"], out_dtype=torch.float32)
        return o[:, :self.num_classes].contiguous(), o[:, self.num_classes:].contiguous()


class PostProcessor(nn.Module):
    """roi_heads/box_head/inference.py:12-149 on device: no per-class host loop, no CPU kthvalue."""

    def __init__(self, cfg):
        super().__init__()
        rh = cfg.MODEL.ROI_HEADS
        self.score_thresh, self.nms, self.detections_per_img = rh.SCORE_THRESH, rh.NMS, rh.DETECTIONS_PER_IMG
        self.weights = tuple(rh.BBOX_REG_WEIGHTS)
        self.strict_gt = bool(getattr(cfg, "NMS_STRICT_GT", True))

    def run(self, x, bl):
        """Sync-free form: padded outputs + device-side count (ob [cap,4], os [cap], ol [cap] i64, oc [1] i32)."""
        class_logits, box_regression = x
        im_w, im_h = bl.size
        return ops.postprocess(class_logits.float().contiguous(), box_regression.float().contiguous(),
                               bl.bbox.float().contiguous(), None, self.weights, im_w, im_h,
                               self.score_thresh, self.nms, self.detections_per_img, self.strict_gt)

    def run_batch(self, x, boxes, size):
        """run() for several images with the same number of rows (x = the concatenated logits / deltas, boxes = the
        list of their [R,4] proposal boxes): one launch chain; returns one run()-style tuple per image."""
        class_logits, box_regression = x
        B = len(boxes)
        ob, os_, ol, oc = ops.postprocess_batched(class_logits.float().contiguous(), box_regression.float().contiguous(),
                                                  torch.cat([b.float() for b in boxes], dim=0), B, self.weights,
                                                  size[0], size[1], self.score_thresh, self.nms,
                                                  self.detections_per_img, self.strict_gt)
        return [(ob[b], os_[b], ol[b], oc[b:b + 1]) for b in range(B)]

    @staticmethod
    def materialize(padded, n, size):
        ob, os_, ol, _ = padded
        res = BoxList(ob[:n], size, "xyxy")
        res.add_field("scores", os_[:n])
        res.add_field("labels", ol[:n])
        return res

    def forward(self, x, boxes):
        assert len(boxes) == 1, "MEGA test path: one key frame per call (data/collate_batch.py:22)"
        padded = self.run(x, boxes[0])
        return [self.materialize(padded, int(padded[3].item()), boxes[0].size)]   # the one host sync


class ROIAttentionBoxHead(nn.Module):
    """roi_heads/box_head/box_head.py:65-124."""

    def __init__(self, cfg, in_channels):
        super().__init__()
        self.feature_extractor = make_roi_box_feature_extractor(cfg, in_channels)
        self.predictor = ROI_BOX_PREDICTOR[cfg.MODEL.ROI_BOX_HEAD.PREDICTOR](cfg, self.feature_extractor.out_channels)
        self.post_processor = PostProcessor(cfg)

    def forward(self, features, proposals, targets=None, key_features=None):
        if self.training:
            raise NotImplementedError("inference path only")
        x = self.feature_extractor(features, proposals, key_features=key_features)
        class_logits, box_regression = self.predictor(x)
        result = self.post_processor((class_logits, box_regression), proposals[0])
        return x, result, {}


class CombinedROIHeads(nn.ModuleDict):
    """roi_heads/roi_heads.py:9-76 (box head only on this path)."""

    def __init__(self, cfg, heads):
        super().__init__(heads)
        self.cfg = cfg.clone() if hasattr(cfg, "clone") else cfg

    def forward(self, features, proposals, targets=None, **kw):
        x, detections, loss_box = self.box(features, proposals, targets, **kw)
        return x, detections, dict(loss_box)


def build_roi_heads(cfg, in_channels):
    return CombinedROIHeads(cfg, [("box", ROIAttentionBoxHead(cfg, in_channels))])


# ================================================================================================= detector
class GeneralizedRCNNMEGA(nn.Module):
    """detector/generalized_rcnn_mega.py:21-225, inference.

    Same per-video state machine (deques of maxlen ALL_FRAME_INTERVAL, key = slot KEY_FRAME_LOCATION,
    frame-0 replication, end-of-video clamping), with two work-saving re-arrangements that do not change
    results: a frame's RPN + res5 + ROIAlign + fc0 are computed ONCE when it enters the window, for the
    300 'key' proposals (the 75 'ref' proposals are their first 75 rows: same NMS keep list cut earlier,
    rpn/inference.py:116-121, boxlist_ops.py:27-29), instead of being recomputed when the frame becomes
    the key frame (generalized_rcnn_mega.py:211, roi_box_feature_extractors.py:901-907).
    """
    _ref_key = "ref_l"          # images[...] key of the frame entering the local window
    stem_reads_u8 = True        # frame_stage_a0(frames_u8, u8_norm=(mean, to_bgr)): preprocessing fused into the bf16 stem

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.device = torch.device(cfg.MODEL.DEVICE)
        self.backbone = build_backbone(cfg)
        self.rpn = build_rpn(cfg, self.backbone.out_channels)
        self.roi_heads = build_roi_heads(cfg, self.backbone.out_channels)
        mega = cfg.MODEL.VID.MEGA
        self.memory_enable = mega.MEMORY.ENABLE
        self.global_enable = mega.GLOBAL.ENABLE
        self.base_num = cfg.MODEL.VID.RPN.REF_POST_NMS_TOP_N
        self.advanced_num = int(self.base_num * mega.RATIO)
        self.all_frame_interval = mega.ALL_FRAME_INTERVAL
        self.key_frame_location = mega.KEY_FRAME_LOCATION
        self.key_num = cfg.MODEL.RPN.POST_NMS_TOP_N_TEST
        assert cfg.MODEL.VID.RPN.REF_PRE_NMS_TOP_N == cfg.MODEL.RPN.PRE_NMS_TOP_N_TEST and self.base_num <= self.key_num, \
            "ref proposals must be a prefix of the key proposals for the single-pass frame stage"
        self.eval()

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        for m in self.modules():
            if isinstance(m, _Packed):
                m._pk = None
        return out

    # ------------------------------------------------------------------ frame stage (batched, frame-independent)
    @torch.no_grad()
    def frame_stage_a(self, imgs):
        """First half of the frame stage: backbone + RPN head + proposal selection for imgs [B,3,H,W] (f32, preprocessed),
        no host sync.  -> {"c4", "props" [B,K,4], "scores" [B,K], "cnt" [B] (device)}.  The proposal COUNTS exist at the end
        of this half: the engine copies them to the host here, so that it can lay out the aggregation while the second half
        (res5 + ROIAlign + fc0, ~40 % of the stage) is still running."""
        return self.frame_stage_a1(self.frame_stage_a0(imgs), imgs.shape[3], imgs.shape[2])

    # The four quarters of the frame stage, for engines that run the two branches below C4 side by side (engine.py:
    # the RPN branch's proposal selection is a chain of one-block-per-frame kernels -- 40 blocks on 256 CUs for ~0.4 ms
    # per 40-frame batch -- that hides completely under res5's convolutions):
    #   a0: backbone -> C4        a1: RPN head + proposal selection (needs C4)
    #   b1: res5 on the full maps (needs C4 only)        b2: ROIAlign + fc0 (needs a1's proposals and b1's maps)
    @torch.no_grad()
    def frame_stage_a0(self, imgs, u8_norm=None):
        """imgs: preprocessed f32 [B,3,H,W]; or, with u8_norm = (mean, to_bgr) in bf16 mode, the uint8 frames [B,H,W,3]"""
        body = self.backbone.body
        if hasattr(body, "run_nhwc"):      # (C4 stays NHWC -- or ops.Planes in conv_mode "x3" / "wide")
            return body.run_nhwc(imgs, u8_norm)
        if u8_norm is not None:
            return _nhwc(body(imgs, u8_norm=u8_norm)[0])
        return _nhwc(self.backbone(imgs)[0])

    @torch.no_grad()
    def frame_stage_a1(self, c4, W, H):
        res = self.rpn.propose(c4, W, H, "key", want_index=self.rpn.keep_index)   # [B,K,4], [B,K], [B] (device)
        a = {"c4": c4, "props": res[0], "scores": res[1], "cnt": res[2]}
        if self.rpn.keep_index:
            a["index"] = res[3]
        return a

    @torch.no_grad()
    def frame_stage_b1(self, c4):
        return self.roi_heads.box.feature_extractor.res5_features(c4)

    @torch.no_grad()
    def frame_stage_b(self, a, want):
        """Second half: res5 + ROIAlign + fc0 on the proposals of frame_stage_a.  want[b] = proposal rows computed for
        frame b (key_num for local frames, base_num for global-pool frames).  Shapes are static: frame b always gets
        want[b] ROI rows; rows past its (device-side) proposal count are all-zero boxes whose features are simply never
        used.  Returns the handle frame_stage_resolve() takes."""
        return self.frame_stage_b2(self.frame_stage_b1(a["c4"]), a, want)

    @torch.no_grad()
    def frame_stage_b2(self, x5, a, want):
        """ROIAlign + fc0 of frame_stage_b on res5 maps x5 = frame_stage_b1(a["c4"])."""
        fe = self.roi_heads.box.feature_extractor
        c4, props = a["c4"], a["props"]
        want = tuple(int(w) for w in want)
        key = (want, str(c4.device))
        if not hasattr(self, "_roi_index_cache"):
            self._roi_index_cache = {}
        cache = self._roi_index_cache.get(key)
        if cache is None:
            # entries are NEVER evicted: a captured hipGraph keeps reading these index tensors on every replay
            K = props.shape[1]
            flat = torch.cat([b * K + torch.arange(w) for b, w in enumerate(want)])
            ids = torch.cat([torch.full((w,), float(b)) for b, w in enumerate(want)]).view(-1, 1)
            cache = (key, flat.to(c4.device), ids.to(c4.device))
            self._roi_index_cache[key] = cache
        boxes = props.view(-1, 4).index_select(0, cache[1])
        rois5 = torch.cat([cache[2], boxes], dim=1)
        feats = fe.pooled_fc(x5, rois5)
        st = {"props": props, "scores": a["scores"], "cnt": a["cnt"], "feats": feats, "want": want}
        if "index" in a:
            st["index"] = a["index"]
        return st

    @torch.no_grad()
    def frame_stage_async(self, imgs, want):
        """Enqueue the whole frame stage for imgs [B,3,H,W] (f32, preprocessed) WITHOUT any host sync:
        frame_stage_b(frame_stage_a(imgs), want).  Returns a handle for frame_stage_resolve()."""
        return self.frame_stage_b(self.frame_stage_a(imgs), want)

    @staticmethod
    def frame_stage_resolve(st, counts=None):
        """Handle -> list of records {"boxes": [n,4], "scores": [n], "feats": [n,1024]}, n = min(count, want).
        counts: host list of the per-frame proposal counts (default: read st["cnt"], one host sync)."""
        if counts is None:
            counts = st["cnt"].tolist()
        out, o = [], 0
        for b, w in enumerate(st["want"]):
            n = min(int(counts[b]), w)
            out.append({"boxes": st["props"][b, :n], "scores": st["scores"][b, :n], "feats": st["feats"][o:o + n]})
            if "index" in st:
                out[-1]["index"] = st["index"][b, :n]
            o += w
        return out

    @torch.no_grad()
    def frame_stage(self, imgs, want):
        """frame_stage_async + frame_stage_resolve (one host sync: the proposal counts)."""
        return self.frame_stage_resolve(self.frame_stage_async(imgs, want))

    # ------------------------------------------------------------------ state machine
    def _reset(self, seg_len):
        n = self.all_frame_interval
        self.seg_len = seg_len
        self.end_id = 0
        self.records = deque(maxlen=n)          # one record per window slot (key_num proposals each)
        fe = self.roi_heads.box.feature_extractor
        fe.init_memory()
        if self.global_enable:
            fe.init_global()

    def _dis_index(self, ns, device):
        """rows of the concatenated window (ns[r] rows per record) that form the 'dis' set: the first advanced_num of
        each record.  Cached per row-count signature (one signature in steady state)."""
        cache = getattr(self, "_dis_cache", None)
        if cache is None or not isinstance(cache, dict):
            cache = self._dis_cache = {}
        hit = cache.get(ns)
        if hit is None or hit.device != device:
            an, rows, off = self.advanced_num, [], 0
            for n in ns:
                rows.append(torch.arange(off, off + min(an, n)))
                off += n
            if len(cache) > 64:
                cache.clear()
            hit = cache[ns] = torch.cat(rows).to(device)
        return hit

    def _window(self):
        """Concatenated local window (oldest first) in the reference's terms (:213-216)."""
        bn, an = self.base_num, self.advanced_num
        ns = tuple(min(bn, r["boxes"].shape[0]) for r in self.records)
        rois = torch.cat([r["boxes"][:n] for r, n in zip(self.records, ns)], 0)
        feats = torch.cat([r["feats"][:n] for r, n in zip(self.records, ns)], 0)
        rois_dis = torch.cat([r["boxes"][:min(an, n)] for r, n in zip(self.records, ns)], 0)
        return rois, rois_dis, feats, self._dis_index(ns, feats.device), ns

    @torch.no_grad()
    def step(self, new_local=None, new_globals=(), im_size=None, defer=False):
        """Advance by one key frame given already-computed frame records: new_local enters the window (None at
        the end-clamped tail is NOT allowed: the reference re-processes the last frame, so pass its record
        again), new_globals go to the global pool.  Returns a BoxList on the device; with defer=True returns the
        padded outputs + device-side count instead (no host sync; see PostProcessor.materialize)."""
        fe = self.roi_heads.box.feature_extractor
        if new_local is not None:
            self.records.append(new_local)
        for g in new_globals:
            fe.update_global(g["feats"][:self.base_num])
        key = self.records[self.key_frame_location]
        rois, rois_dis, x_ref, dis_index, _ = self._window()
        x = fe.aggregate(key["feats"], key["boxes"], rois, rois_dis, x_ref, dis_index)
        logits, deltas = self.roi_heads.box.predictor(x)
        self.last_logits = logits
        kb = BoxList(key["boxes"], im_size, "xyxy")
        kb.add_field("objectness", key["scores"])
        pp = self.roi_heads.box.post_processor
        if defer:
            return pp.run((logits, deltas), kb)
        return pp((logits, deltas), [kb])[0]

    # ------------------------------------------------------------------ batched form of step() (ClipEngine)
    def prepare_step(self, new_local=None, new_globals=()):
        """First half of step(): advance the window / global pool by one key frame and snapshot the inputs of its
        aggregation (no attention yet).  The snapshots of several consecutive key frames go to step_batch()."""
        fe = self.roi_heads.box.feature_extractor
        if new_local is not None:
            self.records.append(new_local)
        for g in new_globals:
            fe.update_global(g["feats"][:self.base_num])
        key = self.records[self.key_frame_location]
        rois, rois_dis, x_ref, dis_index, ns = self._window()
        glob = fe.global_cache[-1].get("feats") if (self.global_enable and fe.global_cache) else None
        return {"x": key["feats"], "rois_key": key["boxes"], "scores": key["scores"], "rois": rois,
                "rois_dis": rois_dis, "x_ref": x_ref, "dis_index": dis_index, "dis_key": ns, "glob": glob}

    def prepare_batch(self, steps):
        """prepare_step() for S consecutive key frames, steps = [(new_local or None, new_globals), ...]: the same
        snapshots, but the local windows -- 25 records sliding by one per key frame -- are row ranges of ONE tape of the
        S + 24 records involved (3 concatenations per batch instead of 3 per key frame), and so are the global pools."""
        fe = self.roi_heads.box.feature_extractor
        bn, an, cap = self.base_num, self.advanced_num, self.records.maxlen
        L, ends = list(self.records), []
        for new_local, _ in steps:
            if new_local is not None:
                self.records.append(new_local)
                L.append(new_local)
            ends.append(len(L))                           # frame j's window = the last <= cap records of L[:ends[j]]
        lo0 = max(0, ends[0] - cap)
        used = L[lo0:]
        ns = [min(bn, r["boxes"].shape[0]) for r in used]
        nd = [min(an, n) for n in ns]
        groups = [([r["boxes"][:n] for r, n in zip(used, ns)], 0), ([r["feats"][:n] for r, n in zip(used, ns)], 0),
                  ([r["boxes"][:n] for r, n in zip(used, nd)], 0)]
        gplan = None
        if self.global_enable:        # the global-pool tape rides in the same copy launch
            gq = fe.global_queue_list[0]["feats"]
            G, gends = list(gq), []
            for _, new_globals in steps:
                for g in new_globals:
                    e = g["feats"][:bn]
                    gq.append(e)
                    G.append(e)
                gends.append(len(G))
            if G:
                glo0 = max(0, gends[0] - gq.maxlen)
                gused = G[glo0:]
                gplan = (gq, gends, glo0, gused)
                if len(gused) > 1:
                    groups.append((gused, 0))
        cat = ops.multi_cat(groups)
        tape_r, tape_f, tape_d = cat[0], cat[1], cat[2]
        offc, offd = [0], [0]
        for n, d in zip(ns, nd):
            offc.append(offc[-1] + n)
            offd.append(offd[-1] + d)
        # global pools: a sliding window over the pushed entries as well
        globs = [None] * len(steps)
        if gplan is not None:
            gq, gends, glo0, gused = gplan
            tape_g = cat[3] if len(gused) > 1 else gused[0]
            goff = [0]
            for e in gused:
                goff.append(goff[-1] + e.shape[0])
            for j, ge in enumerate(gends):
                if ge > 0:
                    globs[j] = tape_g[goff[max(0, ge - gq.maxlen) - glo0]:goff[ge - glo0]]
            if globs[-1] is not None:
                fe.global_cache[0]["feats"] = globs[-1]
        frames = []
        for j, e in enumerate(ends):
            a, b = max(0, e - cap) - lo0, e - lo0
            key = used[a + self.key_frame_location]
            nsj = tuple(ns[a:b])
            frames.append({"x": key["feats"], "rois_key": key["boxes"], "scores": key["scores"], "index_key": key.get("index"),
                           "rois": tape_r[offc[a]:offc[b]], "rois_dis": tape_d[offd[a]:offd[b]],
                           "x_ref": tape_f[offc[a]:offc[b]], "dis_index": self._dis_index(nsj, tape_f.device),
                           "dis_key": nsj, "glob": globs[j]})
        return frames

    batched_postprocess = True      # post-processing of a step-batch's key frames as one launch chain

    @torch.no_grad()
    def step_batch(self, frames, im_size, shard=None):
        """Aggregation + predictor + post-processing of the key frames prepared by prepare_step(), stage by stage over
        the whole list (MEGAFeatureExtractor.aggregate_batch).  Returns one padded post-processing output per frame
        (PostProcessor.run form: no host sync).  Same bits as calling step() once per frame.
        shard: see aggregate_batch; the padded outputs of the frames other ranks own are all-gathered, so every rank
        returns the outputs of ALL frames."""
        fe = self.roi_heads.box.feature_extractor
        xs = fe.aggregate_batch(frames, shard)
        own = [t for t, x in enumerate(xs) if x is not None]
        pp = self.roi_heads.box.post_processor
        outs, self.last_logits_batch = [None] * len(frames), [None] * len(frames)
        if own:
            n = [xs[t].shape[0] for t in own]
            logits, deltas = self.roi_heads.box.predictor(cat_rows([xs[t] for t in own]).contiguous())
            o = 0
            same = len(own) > 1 and len(set(n)) == 1 and self.batched_postprocess
            if same:       # the usual case (every key frame has all its proposals): one launch chain for all
                res = pp.run_batch((logits, deltas), [frames[t]["rois_key"] for t in own], im_size)
            for j, t in enumerate(own):
                lg, dl = logits[o:o + n[j]], deltas[o:o + n[j]]
                o += n[j]
                self.last_logits_batch[t] = lg
                outs[t] = res[j] if same else pp.run((lg, dl), BoxList(frames[t]["rois_key"], im_size, "xyxy"))
            self.last_logits = self.last_logits_batch[own[-1]]
        if shard is not None:
            outs = shard.gather_detections(outs, len(frames), pp.detections_per_img, frames[0]["rois_key"].device)
        return outs

    def forward(self, images, targets=None):
        """Reference call convention (generalized_rcnn_mega.py:48-78, data/datasets/vid_mega.py:95-142):
        images = {"cur", "ref_l": [frame t+MAX_OFFSET], "ref_g": [...], "frame_category", "seg_len", "pattern",
        "img_dir", "transforms"}.  Extension: "ref_l_init" may hold the preprocessed frames 1..12 needed at
        frame_category 0, which replaces the PIL read inside forward (:185-191)."""
        if targets is not None:
            raise ValueError("In testing mode, targets should be None")
        if self.training:
            raise NotImplementedError("inference path only (training is out of scope)")
        cur = to_image_list(images["cur"]).tensors.to(self.device)
        H, W = cur.shape[-2:]
        ref_g = [to_image_list(g).tensors.to(self.device) for g in images.get("ref_g", [])] if self.global_enable else []
        new_local = None
        locals_batch = []
        if images["frame_category"] == 0:
            self._reset(images["seg_len"])
            locals_batch.append(cur)
            init = images.get(self._ref_key + "_init")   # init[j] = preprocessed frame j + 1
            for _ in range(self.all_frame_interval - self.key_frame_location - 1):
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
                locals_batch.append(t.to(self.device))
        elif images["frame_category"] == 1:
            self.end_id = min(self.end_id + 1, self.seg_len - 1)
            locals_batch.append(to_image_list(images[self._ref_key][0]).tensors.to(self.device))
        batch = torch.cat(locals_batch + ref_g, dim=0)
        want = [self.key_num] * len(locals_batch) + [self.base_num] * len(ref_g)
        recs = self.frame_stage(batch, want)
        loc, glob = recs[:len(locals_batch)], recs[len(locals_batch):]
        if images["frame_category"] == 0:
            for _ in range(self.key_frame_location + 1):
                self.records.append(loc[0])
            for r in loc[1:]:
                self.records.append(r)
        else:
            new_local = loc[0]
        return [self.step(new_local, glob, (W, H))]


DETECTION_META_ARCHITECTURES.register("GeneralizedRCNNMEGA", GeneralizedRCNNMEGA)


def build_detection_model(cfg):
    """detector/detectors.py:9-18."""
    return DETECTION_META_ARCHITECTURES[cfg.MODEL.META_ARCHITECTURE](cfg)

"""GPU: the IEEE-half (fp16) instantiation of the frame-stage kernels (round 6; cfg.DTYPE "float16").  The same kernels as the
bf16 path -- igemm / igemm8 / conv64 / bneck64 / stem_pool / ROIAlign / the cast kernels -- compiled for _Float16 operands
(v_mfma_f32_32x32x16_f16, f32 accumulation, round-to-nearest-even conversions): each is checked against torch-CPU fp32
arithmetic on the SAME fp16-rounded operands, and the fused / LDS-DMA forms against the plain tiles bit for bit, exactly as
the bf16 instantiations are in tests/test_kernels_gpu.py.  Reference layers: mega_core/modeling/backbone/resnet.py:324-366,
rpn/rpn.py:99-106, roi_box_feature_extractors.py:894-907, csrc/cuda/ROIAlign_cuda.cu:64-122."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_kernels_gpu import CONV_CASES, _random_rois, _relerr

pytestmark = pytest.mark.gpu
H16 = torch.float16


def _ops():
    from mega.pytorch_amd import ops
    return ops


@pytest.mark.parametrize("case", CONV_CASES)
def test_f16_conv2d_nhwc(dev, case):
    """every conv class of tests/test_kernels_gpu.py::test_conv2d_nhwc in fp16: 11 significant bits -> the bound is 1/8 of
    the bf16 one (the operands are rounded BEFORE the reference, so what is measured is accumulation order + the output's
    own rounding)"""
    ops = _ops()
    N, H, W, Cin, Cout, R, stride, pad, dil, relu, use_res = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn((N, Cin, H, W), generator=g)
    w = torch.randn((Cout, Cin, R, R), generator=g) / math.sqrt(Cin * R * R)
    scale = torch.rand((Cout,), generator=g) + 0.5
    bias = torch.randn((Cout,), generator=g) * 0.1
    ref = F.conv2d(x.to(H16).float(), w.to(H16).float(), stride=stride, padding=pad, dilation=dil)
    ref = ref * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)
    res = None
    if use_res:
        res = torch.randn(ref.shape, generator=g).to(H16)
        ref = ref + res.float()
    if relu:
        ref = F.leaky_relu(ref, 0.1) if relu == 2 else F.relu(ref)
    out = ops.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().to(H16).to(dev), w.permute(0, 2, 3, 1).contiguous().to(H16).to(dev),
                          scale.to(dev), bias.to(dev), None if res is None else res.permute(0, 2, 3, 1).contiguous().to(dev),
                          stride=stride, pad=pad, dil=dil, relu=relu)
    assert out.dtype == H16
    err = _relerr(out.float().cpu().permute(0, 3, 1, 2), ref)
    assert err < 2.5e-3, "conv %s fp16 relerr %.3g" % (case, err)


def test_f16_subnormal_operands_are_not_flushed(dev):
    """fp16 normals end at 6.1e-5: weights / activations below that are subnormal operands of v_mfma_f32_32x32x16_f16.  They
    must enter the products with their value (gradual underflow), not as zeros: a GEMM whose weights ALL lie in the
    subnormal range against f32 arithmetic on the same fp16 values."""
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    x = torch.randn((300, 1024), generator=g).to(H16)
    w = (torch.randn((256, 1024), generator=g) * 1e-5).to(H16)          # |w| ~ 1e-5: subnormal, 7-8 significant bits left
    assert float(w.float().abs().max()) < 6.1e-5 and float(w.float().abs().median()) > 1e-6
    ref = F.linear(x.float(), w.float())
    out = ops.linear(x.to(dev), w.to(dev), out_dtype=torch.float32)
    err = _relerr(out.cpu(), ref)
    assert err < 1e-4, "subnormal fp16 operands: relerr %.3g (flushed to zero?)" % err


def test_f16_in_f32_out_and_first_fc(dev):
    """fp16 operands with an f32 output (the RPN head's logits, fc0's f32 stream): narrow output through the generic tile, and
    fc0 at its real size (K = 100 352: one split-K igemm8 launch + finalize), incl. batch invariance"""
    ops = _ops()
    torch.set_num_threads(16)
    g = torch.Generator().manual_seed(5)
    x = torch.randn((700, 1024), generator=g).to(H16)
    w = (torch.randn((60, 1024), generator=g) / 32).to(H16)
    b = torch.randn((60,), generator=g)
    out = ops.linear(x.to(dev), w.to(dev), b.to(dev), out_dtype=torch.float32)
    assert out.dtype == torch.float32 and _relerr(out.cpu(), F.linear(x.float(), w.float(), b)) < 1e-4
    M, K, N = 375, 100352, 1024
    x = torch.randn((M, K), generator=g).to(H16)
    w = (torch.randn((N, K), generator=g) / math.sqrt(K)).to(H16)
    b = torch.randn((N,), generator=g) * 0.1
    ref = F.relu(F.linear(x.float(), w.float(), b))
    xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
    out = ops.linear(xd, wd, bd, relu=True, out_dtype=torch.float32)
    assert _relerr(out.cpu(), ref) < 1e-4
    out16 = ops.linear(xd, wd, bd, relu=True)
    assert out16.dtype == H16 and _relerr(out16.float().cpu(), ref) < 2.5e-3
    out2 = ops.linear(torch.cat([xd, xd.flip(0)], dim=0), wd, bd, relu=True, out_dtype=torch.float32)
    assert torch.equal(out2[:M], out)


HOT = [
    # N, H, W, Cin, Cout, R, stride, pad, dil, use_res     (the frame stage's layer classes on a 2-frame batch)
    (2, 38, 63, 1024, 256, 1, 1, 0, 1, False),     # layer3 conv1
    (2, 38, 63, 256, 256, 3, 1, 1, 1, False),      # layer3 conv2
    (2, 38, 63, 256, 1024, 1, 1, 0, 1, True),      # layer3 conv3 + residual (streaming class)
    (1, 38, 63, 1024, 1024, 3, 1, 1, 1, False),    # RPN conv
    (1, 38, 63, 512, 512, 3, 1, 2, 2, False),      # res5 conv2 (dilated)
    (1, 75, 125, 512, 1024, 1, 2, 0, 1, False),    # layer3 block 0 downsample (stride-2 1x1)
    (3, 75, 125, 64, 256, 1, 1, 0, 1, True),       # one K-tile + residual
    (1, 17, 13, 128, 520, 1, 1, 0, 1, True),       # Cout % 256 != 0, M tail: the general epilogue
]


@pytest.mark.parametrize("case", HOT)
def test_f16_igemm8_bit_equal_to_register_staged_tiles(dev, case, monkeypatch):
    """igemm8<..., f16_t> (LDS-DMA, 256 / 192 x 256 tiles) gives the bits of the register-staged 128 x 128 tile, as in bf16:
    same MFMA (v_mfma_f32_32x32x16_f16), same ascending K order -> tile choice never changes a result (batch invariance).
    bf16-typed and f32 outputs; three runs each (race screen)."""
    ops = _ops()
    N, H, W, Cin, Cout, R, stride, pad, dil, use_res = case
    g = torch.Generator().manual_seed(Cin + Cout + R)
    x = torch.randn((N, H, W, Cin), generator=g).to(H16).to(dev)
    w = (torch.randn((Cout, R, R, Cin), generator=g) / math.sqrt(Cin * R * R)).to(H16).to(dev)
    sc = (torch.rand((Cout,), generator=g) + 0.5).to(dev)
    bi = (torch.randn((Cout,), generator=g) * 0.1).to(dev)
    Ho, Wo = (H + 2 * pad - dil * (R - 1) - 1) // stride + 1, (W + 2 * pad - dil * (R - 1) - 1) // stride + 1
    res = torch.randn((N, Ho, Wo, Cout), generator=g).to(H16).to(dev) if use_res else None
    for odt in (H16, torch.float32):
        if odt == torch.float32 and use_res:
            continue
        monkeypatch.setenv("MEGA_IGEMM_TILE", "128x128")
        ref = ops.conv2d_nhwc(x, w, sc, bi, res, stride=stride, pad=pad, dil=dil, relu=True, out_dtype=odt)
        for force in ("8:256", "8:192"):
            monkeypatch.setenv("MEGA_IGEMM_TILE", force)
            outs = [ops.conv2d_nhwc(x, w, sc, bi, res, stride=stride, pad=pad, dil=dil, relu=True, out_dtype=odt) for _ in range(3)]
            torch.cuda.synchronize()
            for o in outs:
                assert torch.equal(o, ref), "%s %s %s: %d elements differ (max |d| %.3g)" % (
                    case, force, odt, (o != ref).sum().item(), (o.float() - ref.float()).abs().max().item())
        monkeypatch.delenv("MEGA_IGEMM_TILE")
    # and the natural dispatch against f32 arithmetic
    out = ops.conv2d_nhwc(x, w, sc, bi, res, stride=stride, pad=pad, dil=dil, relu=True)
    r32 = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), w.float().cpu().permute(0, 3, 1, 2), stride=stride, padding=pad, dilation=dil)
    r32 = r32 * sc.cpu().view(1, -1, 1, 1) + bi.cpu().view(1, -1, 1, 1)
    if res is not None:
        r32 = r32 + res.float().cpu().permute(0, 3, 1, 2)
    assert _relerr(out.float().cpu().permute(0, 3, 1, 2), F.relu(r32)) < 2.5e-3


@pytest.mark.parametrize("shape", [(2, 150, 250), (3, 151, 249), (40, 33, 47)])
def test_f16_conv64_bit_equal_to_generic_tiles(dev, shape, monkeypatch):
    ops = _ops()
    N, H, W = shape
    g = torch.Generator().manual_seed(17)
    x = torch.randn((N, H, W, 64), generator=g).to(H16).to(dev)
    w = (torch.randn((64, 3, 3, 64), generator=g) * 0.05).to(H16).to(dev)
    sc = (torch.rand((64,), generator=g) + 0.5).to(dev)
    bi = (torch.randn((64,), generator=g) * 0.1).to(dev)
    from mega.pytorch_amd import _lib
    assert _lib.load().mega_conv2d_nhwc_plan_ex(N, H, W, 64, 64, 3, 3, 1, 1, 1, 64, 0, 2, 2) // 1000000 == 6
    for relu in (True, False, 2):
        y = ops.conv2d_nhwc(x, w, sc, bi, pad=1, relu=relu)
        monkeypatch.setenv("MEGA_IGEMM_TILE", "128x64")
        y_ref = ops.conv2d_nhwc(x, w, sc, bi, pad=1, relu=relu)
        monkeypatch.delenv("MEGA_IGEMM_TILE")
        torch.cuda.synchronize()
        assert torch.equal(y.view(torch.int16), y_ref.view(torch.int16)), "relu=%s: bit patterns differ" % (relu,)
    ref = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), w.float().cpu().permute(0, 3, 1, 2), padding=1)
    ref = F.relu(ref * sc.cpu().view(1, -1, 1, 1) + bi.cpu().view(1, -1, 1, 1)).permute(0, 2, 3, 1)
    assert _relerr(ops.conv2d_nhwc(x, w, sc, bi, pad=1, relu=True).float().cpu(), ref) < 2.5e-3


@pytest.mark.parametrize("shape", [(2, 150, 250), (3, 37, 53), (1, 8, 16), (40, 9, 17)])
def test_f16_fused_bottlenecks_bit_equal_to_unfused(dev, shape):
    """bneck64_kernel<f16_t> / bneck64_ds_kernel<f16_t> against the conv2d_nhwc launches they replace, fp16: the same bits
    (same MFMA, K order, fp16 roundings of t1 / t2 / the identity, same epilogue arithmetic)"""
    ops = _ops()
    N, H, W = shape
    g = torch.Generator().manual_seed(H * 7 + W)
    x = torch.randn((N, H, W, 256), generator=g).relu().to(H16).to(dev)
    w1 = (torch.randn((64, 1, 1, 256), generator=g) * 0.06).to(H16).to(dev)
    w2 = (torch.randn((64, 3, 3, 64), generator=g) * 0.05).to(H16).to(dev)
    w3 = (torch.randn((256, 1, 1, 64), generator=g) * 0.1).to(H16).to(dev)
    sb = [((torch.rand((n,), generator=g) + 0.5).to(dev), (torch.randn((n,), generator=g) * 0.2).to(dev)) for n in (64, 64, 256, 256)]
    t1 = ops.conv2d_nhwc(x, w1, sb[0][0], sb[0][1], relu=True)
    t2 = ops.conv2d_nhwc(t1, w2, sb[1][0], sb[1][1], pad=1, relu=True)
    ref = ops.conv2d_nhwc(t2, w3, sb[2][0], sb[2][1], residual=x, relu=True)
    got = ops.bottleneck64(x, w1, sb[0][0], sb[0][1], w2, sb[1][0], sb[1][1], w3, sb[2][0], sb[2][1])
    torch.cuda.synchronize()
    nd = (got.view(torch.int16) != ref.view(torch.int16)).sum().item()
    assert got.dtype == H16 and nd == 0, "identity block: %d of %d elements differ" % (nd, got.numel())
    # f32 reference of the whole block (fp16 intermediates: the bound of three chained roundings)
    xf = x.float().cpu().permute(0, 3, 1, 2)

    def bn(y, i):
        return y * sb[i][0].cpu().view(1, -1, 1, 1) + sb[i][1].cpu().view(1, -1, 1, 1)
    y = F.relu(bn(F.conv2d(xf, w1.float().cpu().permute(0, 3, 1, 2)), 0))
    y = F.relu(bn(F.conv2d(y, w2.float().cpu().permute(0, 3, 1, 2), padding=1), 1))
    y = F.relu(bn(F.conv2d(y, w3.float().cpu().permute(0, 3, 1, 2)), 2) + xf).permute(0, 2, 3, 1)
    assert _relerr(got.float().cpu(), y) < 3e-3
    # the downsample block (64 -> 256)
    x6 = torch.randn((N, H, W, 64), generator=g).relu().to(H16).to(dev)
    w1d = (torch.randn((64, 1, 1, 64), generator=g) * 0.12).to(H16).to(dev)
    wd = (torch.randn((256, 1, 1, 64), generator=g) * 0.1).to(H16).to(dev)
    ident = ops.conv2d_nhwc(x6, wd, sb[3][0], sb[3][1])
    t1 = ops.conv2d_nhwc(x6, w1d, sb[0][0], sb[0][1], relu=True)
    t2 = ops.conv2d_nhwc(t1, w2, sb[1][0], sb[1][1], pad=1, relu=True)
    ref = ops.conv2d_nhwc(t2, w3, sb[2][0], sb[2][1], residual=ident, relu=True)
    got = ops.bottleneck64_ds(x6, w1d, sb[0][0], sb[0][1], w2, sb[1][0], sb[1][1], w3, sb[2][0], sb[2][1], wd, sb[3][0], sb[3][1])
    torch.cuda.synchronize()
    nd = (got.view(torch.int16) != ref.view(torch.int16)).sum().item()
    assert nd == 0, "downsample block: %d of %d elements differ" % (nd, got.numel())


@pytest.mark.parametrize("shape", [(2, 600, 1000), (2, 75, 131), (1, 9, 7), (1, 17, 130)])
def test_f16_stem_pool(dev, shape):
    """stem_pool_kernel<., f16_t>: conv 7x7/2 + FrozenBN + ReLU + max_pool2d(3, 2, 1) (backbone/resnet.py:355-366) from the
    uint8 frames and from the preprocessed f32 image.  Reference: f32 arithmetic on the fp16-rounded pixels and weights, the
    stem map rounded to fp16 before the max (what the kernel stages in LDS); summation order differs, so single-ulp flips
    are allowed: bound 2 ulp of fp16 relative to the map's scale."""
    ops = _ops()
    N, H, W = shape
    g = torch.Generator().manual_seed(H * 3 + W)
    u8 = torch.randint(0, 256, (N, H, W, 3), generator=g, dtype=torch.uint8)
    w = torch.randn((64, 3, 7, 7), generator=g) * 0.05
    sc = torch.rand((64,), generator=g) + 0.5
    bi = torch.randn((64,), generator=g) * 0.1
    mean = (102.9801, 115.9465, 122.7717)
    w160 = ops.pack_stem_weight_bf16(w, H16).to(dev)
    img = ops.preprocess_frames(u8.to(dev), mean, True)
    conv = F.conv2d(img.cpu().to(H16).float(), w.to(H16).float(), stride=2, padding=3)
    stem = F.relu(conv * sc.view(1, -1, 1, 1) + bi.view(1, -1, 1, 1)).to(H16).float()
    ref = F.max_pool2d(stem, 3, 2, 1).permute(0, 2, 3, 1)
    for got in (ops.stem_pool(u8.to(dev), w160, sc.to(dev), bi.to(dev), mean, True), ops.stem_pool(img, w160, sc.to(dev), bi.to(dev))):
        assert got.dtype == H16 and got.shape == ref.shape
        assert _relerr(got.float().cpu(), ref) < 2e-3
        assert (got.view(torch.int16) < 0).sum().item() == 0            # +0, never -0 (the pool orders bit patterns)
    a, b = ops.stem_pool(u8.to(dev), w160, sc.to(dev), bi.to(dev), mean, True), ops.stem_pool(img, w160, sc.to(dev), bi.to(dev))
    assert torch.equal(a.view(torch.int16), b.view(torch.int16))         # the u8 patch load == preprocess + f32 patch load


def test_f16_roi_align_hot_shape(dev):
    """ROIAlign on fp16 NHWC features at the hot shape (C = 2048, the XCD-sliced separable kernel) against native_oracle.c on
    the same fp16 values, and the exact-term-order kernel (sampling_ratio = 2)"""
    ops = _ops()
    from oracle import native
    g = torch.Generator().manual_seed(3)
    B, Hh, Ww, C = 2, 38, 63, 2048
    feat = torch.randn((B, Hh, Ww, C), generator=g).to(H16)
    rois = _random_rois(g, 200, B, 1000.0, 600.0)
    for sr in (0, 2):
        ref = torch.from_numpy(native.roi_align(feat.float().permute(0, 3, 1, 2).contiguous().numpy(), rois.numpy(), 1.0 / 16, 7, 7, sr))
        ref = ref.permute(0, 2, 3, 1).reshape(rois.shape[0], 49, C)
        got = ops.roi_align(feat.to(dev), rois.to(dev), 1.0 / 16, (7, 7), sr)
        assert got.dtype == H16 and _relerr(got.float().cpu(), ref) < 2e-3, "sampling_ratio %d" % sr


@pytest.mark.parametrize("n", [8, 1024 * 675, 1024 * 3 + 5, 7])
def test_f16_cast_and_cat_cast(dev, n):
    """ops.cast_half(x, float16) / ops.cat_rows_cast_bf16(pieces, float16) == torch's float -> half (round to nearest even,
    overflow to inf, gradual underflow), bit for bit"""
    ops = _ops()
    g = torch.Generator().manual_seed(n)
    x = torch.randn((n,), generator=g) * 3
    if n >= 8:
        x[:8] = torch.tensor([0.0, -0.0, 1e-40, 65504.0, 65520.0, 1e-7, -6.0e-5, 2.0 ** -25])
    got = ops.cast_half(x.to(dev), H16)
    assert got.dtype == H16 and torch.equal(got.cpu().view(torch.int16), x.to(H16).view(torch.int16))
    if n >= 1024:
        big = (torch.randn((700, 2048), generator=g) * 3).to(dev)
        pieces = [big[:300, :1024], big[300:301, 1024:], big[301:700, 1024:].contiguous()] + [big[i:i + 2, :1024] for i in range(0, 140, 2)]
        got = ops.cat_rows_cast_bf16(pieces, H16)
        want = torch.cat(pieces, dim=0).to(H16)
        assert torch.equal(got.view(torch.int16), want.view(torch.int16))


def test_f16_model_frame_stage_is_batch_invariant_and_close_to_f32(dev):
    """cfg.DTYPE float16 through the model: backbone -> RPN -> res5 -> ROIAlign -> fc0 on two 192 x 320 frames; the frame
    records of a batch of 2 equal those of two batches of 1 bit for bit, and fc0's f32 output stays within fp16 noise of the
    float32 model's on the SAME proposals (float32 model's boxes fed to both ROIAlign passes)."""
    from mega.pytorch_amd import config, modeling, synth
    sd = synth.make_state_dict(blocks=(3, 4, 6), reduce_channel=True, global_res_stage=0, seed=3)
    ms = {}
    for dt in ("float16", "float32"):
        cfg = config.get_cfg("R-50")
        cfg.DTYPE = dt
        cfg.MODEL.DEVICE = str(dev)
        m = modeling.build_detection_model(cfg)
        m.load_state_dict(sd)
        ms[dt] = m.to(dev)
    g = torch.Generator().manual_seed(9)
    u8 = torch.randint(0, 256, (2, 192, 320, 3), generator=g, dtype=torch.uint8).to(dev)
    from mega.pytorch_amd import ops
    img = ops.preprocess_frames(u8, tuple(cfg.INPUT.PIXEL_MEAN), True)
    m16, m32 = ms["float16"], ms["float32"]
    assert m16.backbone.body.dtype == H16 and m16.roi_heads.box.feature_extractor.hdtype == torch.bfloat16
    r2 = m16.frame_stage(img, [300, 300])
    r1 = [m16.frame_stage(img[i:i + 1].contiguous(), [300])[0] for i in range(2)]
    for a, b in zip(r2, r1):
        assert torch.equal(a["boxes"], b["boxes"]) and torch.equal(a["feats"], b["feats"])
        assert a["feats"].dtype == torch.float32
    # same ROIs through both models' res5 + ROIAlign + fc0
    c16, c32 = m16.frame_stage_a0(img), m32.frame_stage_a0(img)
    assert c16.dtype == H16
    rel = float((c16.float() - c32).abs().mean() / c32.abs().mean())
    assert rel < 1.5e-3, "C4 mean relative error %.3g" % rel            # (bf16: ~5e-3; fp16 predicted 4-7e-4)
    a32 = m32.frame_stage_a1(c32, 320, 192)
    a16 = dict(a32)
    a16["c4"] = c16
    f32 = m32.frame_stage_b(a32, [300, 300])["feats"]
    f16 = m16.frame_stage_b(a16, [300, 300])["feats"]
    rel = float((f16 - f32).abs().mean() / f32.abs().mean())
    assert rel < 1.5e-3, "fc0 mean relative error %.3g" % rel


# ------------------------------------------------------------------------------------------------ the two-pass form (conv_mode "h2")
from test_kernels_gpu import SP_CASES, _sp_inputs  # noqa: E402


@pytest.mark.parametrize("case", SP_CASES)
def test_h2_conv2d_sp_vs_f64(dev, case):
    """The two-pass fp16 conv (float16 [hi | lo] planes against [W | W], W rounded to fp16 ONCE, f32 accumulation) against the
    f64 convolution of the same f32 activations with the SAME fp16-rounded weights: what is left is the planes' representation
    (~2^-22 of a value, the lo plane's subnormal floor 6e-8 absolute) and f32 accumulation -- the exact-f32 kernels' level."""
    ops = _ops()
    N, H, W, C, Cout, R, stride, pad, dil, relu, use_res, out_mode = case
    x, w, scale, bias, res = _sp_inputs(case)
    w16 = w.to(H16)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w16.permute(0, 3, 1, 2).double(), stride=stride, padding=pad, dilation=dil)
    ref = ref * scale.double().view(1, -1, 1, 1) + bias.double().view(1, -1, 1, 1)
    if use_res:
        ref = ref + res.permute(0, 3, 1, 2).double()
    if relu:
        ref = ref.clamp(min=0)
    ref = ref.permute(0, 2, 3, 1)
    xp = ops.split_planes(x.to(dev).contiguous(), H16)
    assert xp.t.dtype == H16 and (xp.float().cpu() - x).abs().max() <= 2.0 ** -20 * x.abs().max() + 1e-7
    rp = ops.split_planes(res.to(dev).contiguous(), H16) if use_res else None
    wh2 = ops.split_conv_weight_h2(w).to(dev)
    assert wh2.dtype == H16 and wh2.shape[-1] == 2 * C
    y = ops.conv2d_sp(xp, wh2, scale.to(dev), bias.to(dev), residual=rp, stride=stride, pad=pad, dil=dil, relu=relu,
                      out_mode=out_mode, x3="h2")
    got = (y.float() if out_mode == "planes" else y).cpu().double()
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    print("conv2d_sp h2 %s: max err / scale = %.3g" % (case, err))
    assert got.shape == ref.shape and err < 2e-5, err


def test_h2_linear_sp_split_k_and_roi_align_planes(dev):
    """fc0's shape class (K >= 32768: three K ranges + finalize) through the fp16 SP kernels; the f32 ROIAlign writing fp16
    [hi | lo] planes == split_planes(., float16) of its f32 output, bit for bit"""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    M, K, Nout = 300, 49 * 1024, 1024
    x = torch.randn((M, K), generator=g)
    w = torch.randn((Nout, K), generator=g) / math.sqrt(K)
    b = torch.randn((Nout,), generator=g) * 0.1
    ref = (x.double() @ w.to(H16).double().t() + b.double()).clamp(min=0)
    y = ops.linear_sp(ops.split_planes(x.to(dev), H16), ops.split_conv_weight_h2(w.view(Nout, 1, 1, K)).to(dev).view(Nout, 2 * K),
                      b.to(dev), relu=True).cpu().double()
    err = (y - ref).abs().max().item() / ref.abs().max().item()
    print("linear_sp h2 split-K: max err / scale = %.3g" % err)
    assert err < 2e-5
    for (B, Hh, Ww, C, Kr) in ((2, 38, 63, 2048, 300), (1, 12, 17, 32, 9)):
        g = torch.Generator().manual_seed(C + Kr)
        feat = torch.randn((B, Hh, Ww, C), generator=g).to(dev)
        rois = _random_rois(g, Kr, B, Ww * 16, Hh * 16).to(dev)
        f32 = ops.roi_align(feat, rois, 1.0 / 16, (7, 7), 0)
        want = ops.split_planes(f32.view(Kr, -1).contiguous(), H16)
        got = ops.roi_align_planes(feat, rois, 1.0 / 16, (7, 7), 0, H16)
        assert got.C == want.C and got.t.dtype == H16
        if C == 2048:     # the XCD-sliced hot shape runs the separable form (round 6): f32 round-off, far below the planes' own
            n = got.C     # rounding (fp16 [hi | lo]: ~2^-22 relative; the small lo values leave fp16's normal range: 2^-24 absolute)
            a = got.t[:, :n].float() + got.t[:, n:].float()
            scale = f32.abs().max().item()
            assert (a - f32.view(Kr, -1)).abs().max().item() <= 4e-6 * scale
        else:             # (small shapes keep the exact-term-order kernel: bit for bit)
            assert torch.equal(got.t.view(torch.int16), want.t.view(torch.int16))

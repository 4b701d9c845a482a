"""-m gpu: every HIP kernel, called through the C ABI (mega/pytorch_amd/ops.py -> libmega_hip.so),
against the CPU oracle (oracle/) or a plain torch-CPU fp32 reference of the same op, on seeded inputs.

Tolerances: integer / index outputs bit-exact; f32 kernels 1e-4 relative to the tensor scale (MFMA f32
is an exact fmaf chain, only the summation order differs from MKL); bf16 kernels 2e-2 of tensor scale.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ops():
    from mega.pytorch_amd import ops
    return ops


def _relerr(a, b):
    a = a.double(); b = b.double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


CONV_CASES = [
    # N, H, W, Cin, Cout, R, stride, pad, dil, relu, use_res
    (1, 19, 23, 64, 64, 1, 1, 0, 1, True, False),
    (2, 19, 23, 64, 256, 1, 1, 0, 1, False, True),
    (1, 21, 17, 256, 128, 1, 2, 0, 1, True, False),      # stride-2 1x1 (STRIDE_IN_1X1)
    (1, 19, 23, 64, 64, 3, 1, 1, 1, True, False),
    (1, 13, 15, 128, 192, 3, 1, 2, 2, True, False),      # dilated res5-style
    (1, 38, 63, 256, 60, 1, 1, 0, 1, False, False),      # RPN-style narrow output, N tail
    (300, 1, 1, 1024, 155, 1, 1, 0, 1, False, False),    # linear, M and N tails
    (1, 38, 63, 128, 512, 3, 1, 1, 1, True, True),
    (2, 150, 250, 64, 64, 1, 1, 0, 1, True, False),      # one K-tile (bf16), many blocks: LDS re-use races show up
    (2, 150, 250, 64, 64, 3, 1, 1, 1, True, False),      # odd K-tile count (9)
    (2, 75, 125, 64, 256, 1, 1, 0, 1, False, True),
    (3, 32, 48, 64, 64, 7, 2, 3, 1, 2, False),           # FlowNetS conv1 (Cin zero-padded 6->64), LeakyReLU(0.1) epilogue
    (2, 9, 13, 192, 128, 4, 1, 3, 1, 2, False),          # zero-stuffed deconv as a 4x4 conv with pad 3
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_nhwc(dev, case, dtype):
    ops = _ops()
    N, H, W, Cin, Cout, R, stride, pad, dil, relu, use_res = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn((N, Cin, H, W), generator=g)
    w = torch.randn((Cout, Cin, R, R), generator=g) / math.sqrt(Cin * R * R)
    scale = torch.rand((Cout,), generator=g) + 0.5
    bias = torch.randn((Cout,), generator=g) * 0.1
    xq, wq = x.to(dtype).float(), w.to(dtype).float()
    ref = F.conv2d(xq, wq, stride=stride, padding=pad, dilation=dil)
    ref = ref * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)
    res = None
    if use_res:
        res = torch.randn(ref.shape, generator=g).to(dtype)
        ref = ref + res.float()
    if relu:
        ref = F.leaky_relu(ref, 0.1) if relu == 2 else F.relu(ref)
    out = ops.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().to(dtype).to(dev),
                          w.permute(0, 2, 3, 1).contiguous().to(dtype).to(dev), scale.to(dev), bias.to(dev),
                          None if res is None else res.permute(0, 2, 3, 1).contiguous().to(dev),
                          stride=stride, pad=pad, dil=dil, relu=relu)
    got = out.float().cpu().permute(0, 3, 1, 2)
    assert got.shape == ref.shape
    err = _relerr(got, ref)
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    assert err < tol, "conv %s %s relerr %.3g" % (case, dtype, err)


def test_conv_bf16_in_f32_out(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    x = torch.randn((700, 1024), generator=g).to(torch.bfloat16)
    w = (torch.randn((60, 1024), generator=g) / 32).to(torch.bfloat16)
    b = torch.randn((60,), generator=g)
    ref = F.linear(x.float(), w.float(), b)
    out = ops.linear(x.to(dev), w.to(dev), b.to(dev), out_dtype=torch.float32)
    assert out.dtype == torch.float32
    assert _relerr(out.cpu(), ref) < 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_stem_and_maxpool(dev, dtype):
    ops = _ops()
    from oracle import mega_oracle as mo
    from mega.pytorch_amd import synth
    sd = synth.make_state_dict(blocks=(1, 1, 1), seed=2)
    g = torch.Generator().manual_seed(1)
    x = torch.randn((2, 3, 75, 101), generator=g) * 60
    p = "backbone.body.stem."
    conv = F.conv2d(x, sd[p + "conv1.weight"], stride=2, padding=3)
    ref_c = F.relu(mo.frozen_bn(conv, sd, p + "bn1."))
    ref_p = F.max_pool2d(ref_c, 3, 2, 1)
    scale = sd[p + "bn1.weight"] * sd[p + "bn1.running_var"].rsqrt()
    bias = sd[p + "bn1.bias"] - sd[p + "bn1.running_mean"] * scale
    w_tap = sd[p + "conv1.weight"].permute(1, 2, 3, 0).reshape(147, 64).contiguous()
    y = ops.stem(x.to(dev), w_tap.to(dev), scale.to(dev), bias.to(dev), dtype)
    tol = 1e-4 if dtype == torch.float32 else 1e-2
    assert _relerr(y.float().cpu().permute(0, 3, 1, 2), ref_c) < tol
    z = ops.maxpool3x3s2(y)
    ref_p2 = F.max_pool2d(y.float().cpu().permute(0, 3, 1, 2), 3, 2, 1)
    assert torch.equal(z.float().cpu().permute(0, 3, 1, 2), ref_p2)
    assert _relerr(z.float().cpu().permute(0, 3, 1, 2), ref_p) < tol


def _random_rois(g, K, B, W, H):
    x1 = torch.rand((K,), generator=g) * W * 0.8
    y1 = torch.rand((K,), generator=g) * H * 0.8
    w = torch.rand((K,), generator=g) * W * 0.6 + 1
    h = torch.rand((K,), generator=g) * H * 0.6 + 1
    b = torch.randint(0, B, (K,), generator=g).float()
    rois = torch.stack([b, x1, y1, (x1 + w).clamp(max=W - 1), (y1 + h).clamp(max=H - 1)], dim=1)
    rois[0] = torch.tensor([0, 0, 0, W - 1, H - 1])          # whole image (largest sampling grid)
    rois[1] = torch.tensor([0, 5.3, 7.1, 5.9, 7.4])          # tiny: forced to 1x1
    return rois


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("layouts", [(True, True), (False, False), (True, False)])
def test_roi_align(dev, dtype, layouts):
    ops = _ops()
    from oracle import native
    in_nhwc, out_nhwc = layouts
    g = torch.Generator().manual_seed(7)
    B, C, H, W = 2, 96, 38, 63
    feat = torch.randn((B, C, H, W), generator=g).to(dtype)
    rois = _random_rois(g, 40, B, W * 16, H * 16)
    ref = torch.from_numpy(native.roi_align(feat.float().numpy(), rois.numpy(), 1 / 16., 7, 7, 0))
    f_in = feat.permute(0, 2, 3, 1).contiguous() if in_nhwc else feat
    out = ops.roi_align(f_in.to(dev), rois.to(dev), 1 / 16., (7, 7), 0, in_nhwc=in_nhwc, out_nhwc=out_nhwc).float().cpu()
    if out_nhwc:
        out = out.view(-1, 7, 7, C).permute(0, 3, 1, 2)
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    assert _relerr(out, ref) < tol


def test_roi_align_bf16_hot_shape(dev):
    """The frame stage's ROIAlign as it runs in the benchmark: res5 maps of 2048 channels at 38 x 63, bf16, the
    per-ROI separable kernel with XCD channel slicing -- against native_oracle.c (pinned to the compiled reference
    csrc/cpu ROIAlign) on 640 ROIs over 4 images; and at the full launch size (6000 ROIs = 20 frames x 300) every ROI's
    rows must equal the rows the same ROI gets in the small launch (block-independence: bit equality)."""
    ops = _ops()
    from oracle import native
    g = torch.Generator().manual_seed(21)
    B, C, H, W = 4, 2048, 38, 63
    feat = torch.randn((B, H, W, C), generator=g).to(torch.bfloat16)
    rois = _random_rois(g, 640, B, W * 16, H * 16)
    rois[2] = torch.tensor([1, 0.0, 0.0, 15.9, 15.9])               # one cell
    rois[3] = torch.tensor([2, 500.0, 0.0, 999.0, 599.0])           # right half, full height: grid 5 x 6
    ref = torch.from_numpy(native.roi_align(feat.float().permute(0, 3, 1, 2).contiguous().numpy(), rois.numpy(), 1 / 16., 7, 7, 0))
    out = ops.roi_align(feat.to(dev), rois.to(dev), 1 / 16., (7, 7), 0).float().cpu()           # [K,49,C]
    got = out.view(-1, 7, 7, C).permute(0, 3, 1, 2)
    err = (got - ref).abs()
    print("roi_align bf16 C=2048: max |err| %.3g, rel %.3g" % (err.max(), _relerr(got, ref)))
    assert _relerr(got, ref) < 1e-2 and err.max() < 0.08          # bf16 inputs, f32 sums, one bf16 rounding at the end
    big = rois[torch.arange(6000) % 640].contiguous()
    out6 = ops.roi_align(feat.to(dev), big.to(dev), 1 / 16., (7, 7), 0)
    assert torch.equal(out6[:640].cpu().float(), out) and torch.equal(out6[5120:5760].cpu().float(), out)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_roi_align_unclipped_rois_and_fixed_sampling_ratio(dev, dtype):
    """the public op with ROIs the MEGA path never produces.  (a) boxes larger than the map /
    partly outside it with the adaptive grid: grid = ceil(roi / 7) exceeds the separable kernel's 10-pixel patch
    tables (it must fall back, not read past them); (b) sampling_ratio = 2 on a 200-wide map: a bin's two samples lie
    bin_size / 2 apart, i.e. a sparse patch.  Both against the oracle."""
    ops = _ops()
    from oracle import native
    g = torch.Generator().manual_seed(5)
    B, C, H, W = 2, 64, 40, 200
    feat = torch.randn((B, C, H, W), generator=g).to(dtype)
    rois = torch.tensor([[0, -400.0, -300.0, 4000.0, 900.0],          # far larger than the 3200 x 640 image
                         [1, 0.0, 0.0, 3199.0, 639.0],                # the whole map: grid 29 x 6
                         [0, 2500.0, -200.0, 3600.0, 300.0],          # sticking out at the top right
                         [1, -50.0, 100.0, 30.0, 500.0],
                         [0, 100.0, 100.0, 400.0, 300.0]])
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    f_in = feat.permute(0, 2, 3, 1).contiguous().to(dev)
    for ratio in (0, 2):
        ref = torch.from_numpy(native.roi_align(feat.float().numpy(), rois.numpy(), 1 / 16., 7, 7, ratio))
        out = ops.roi_align(f_in, rois.to(dev), 1 / 16., (7, 7), ratio).float().cpu().view(-1, 7, 7, C).permute(0, 3, 1, 2)
        assert torch.isfinite(out).all()
        assert _relerr(out, ref) < tol, (ratio, _relerr(out, ref))


def test_nms_golden_and_random(dev):
    ops = _ops()
    from oracle import native
    inputs = np.array([10, 10, 50, 60, 0.5, 11, 12, 48, 60, 0.7, 8, 9, 40, 50, 0.6, 100, 100, 150, 140, 0.9,
                       99, 110, 155, 139, 0.8], dtype=np.float32).reshape(-1, 5)
    gt = [[1, 3], [1, 3], [1, 3], [1, 2, 3, 4], [0, 1, 2, 3, 4]]
    b = torch.from_numpy(inputs[:, :4].copy()).to(dev)
    s = torch.from_numpy(inputs[:, 4].copy()).to(dev)
    for thr, want in zip([0.1, 0.3, 0.5, 0.8, 0.9], gt):     # reference tests/test_nms.py:16-58
        for strict in (False, True):
            keep = ops.nms(b, s, thr, strict_gt=strict).cpu().tolist()
            assert keep == want, (thr, strict, keep)
    rng = np.random.RandomState(0)
    for n, thr in [(1, 0.5), (63, 0.5), (64, 0.3), (65, 0.7), (300, 0.5), (1000, 0.7), (1024, 0.5), (2500, 0.3),
                   (6000, 0.7), (8192, 0.6)]:   # >= 1024: the lazy one-block-per-problem kernel
        ctr = rng.rand(n, 2) * 500
        wh = rng.rand(n, 2) * 120 + 4
        boxes = np.concatenate([ctr - wh / 2, ctr + wh / 2], axis=1).astype(np.float32)
        scores = rng.rand(n).astype(np.float32)
        if n >= 300:
            scores[::7] = scores[3]                                  # ties
        for strict in (True, False):
            want = native.nms(boxes, scores, thr, strict)
            got = ops.nms(torch.from_numpy(boxes).to(dev), torch.from_numpy(scores).to(dev), thr, strict).cpu().numpy()
            assert np.array_equal(got, want), "n=%d thr=%g strict=%s: %d vs %d kept" % (n, thr, strict, len(got), len(want))
    assert ops.nms(torch.zeros((0, 4), device=dev), torch.zeros((0,), device=dev), 0.5).numel() == 0


def _rpn_inputs(g, B, A, Hf, Wf):
    obj = torch.randn((B, A, Hf, Wf), generator=g) * 2
    reg = torch.randn((B, 4 * A, Hf, Wf), generator=g) * 0.5
    obj[:, :, 0, :5] = obj[0, 0, 1, 1]                              # exact ties in the logits
    return obj, reg


@pytest.mark.parametrize("shape", [(2, 38, 63, 6000, 300), (1, 10, 16, 6000, 75), (3, 20, 30, 1000, 300)])
def test_rpn_select(dev, shape):
    ops = _ops()
    from oracle import mega_oracle as mo
    B, Hf, Wf, pre, post = shape
    g = torch.Generator().manual_seed(11)
    cell = mo.generate_anchors(16)
    A = cell.shape[0]
    obj, reg = _rpn_inputs(g, B, A, Hf, Wf)
    im_w, im_h = Wf * 16 - 8, Hf * 16 - 8
    anchors = mo.grid_anchors(cell, Hf, Wf, 16)
    rpn_out = torch.cat([obj, reg], dim=1).permute(0, 2, 3, 1).reshape(B, Hf * Wf, 5 * A).contiguous()
    props, scores, cnt, index = ops.rpn_select(rpn_out.to(dev), cell.to(dev), Hf, Wf, 16, pre, post, 0.7, 0, im_w, im_h, True,
                                               want_index=True)
    props, scores, cnt, index = props.cpu(), scores.cpu(), cnt.cpu(), index.cpu()
    p2, s2, c2 = ops.rpn_select(rpn_out.to(dev), cell.to(dev), Hf, Wf, 16, pre, post, 0.7, 0, im_w, im_h, True)
    assert torch.equal(p2.cpu(), props) and torch.equal(s2.cpu(), scores) and torch.equal(c2.cpu(), cnt)
    for b in range(B):
        wb, ws, wi = mo.rpn_select(obj[b], reg[b], anchors, im_w, im_h, pre, post, 0.7, 0, True, want_index=True)
        n = int(cnt[b])
        assert n == wb.shape[0], "frame %d: kept %d vs oracle %d" % (b, n, wb.shape[0])
        # the proposal INDICES after NMS, bit for bit (north_star), incl. the tie order (lower anchor index first)
        assert torch.equal(index[b, :n].long(), wi), "frame %d: kept anchor indices differ" % b
        assert (index[b, n:] == -1).all()
        assert (props[b, :n] - wb).abs().max() < 1e-3, (props[b, :n] - wb).abs().max()
        assert (scores[b, :n] - ws).abs().max() < 1e-6
        assert props[b, n:].abs().max() == 0 if n < post else True


@pytest.mark.parametrize("R", [300, 77])
def test_postprocess(dev, R):
    ops = _ops()
    from oracle import mega_oracle as mo
    g = torch.Generator().manual_seed(R)
    NC = 31
    logits = torch.randn((R, NC), generator=g) * 1.5
    deltas = torch.randn((R, NC * 4), generator=g) * 0.5
    ctr = torch.rand((R, 2), generator=g) * torch.tensor([900., 500.])
    wh = torch.rand((R, 2), generator=g) * 200 + 10
    props = torch.cat([ctr - wh / 2, ctr + wh / 2], dim=1).clamp(min=0)
    props[:, 2].clamp_(max=999); props[:, 3].clamp_(max=599)
    cfg = mo.OracleCfg()
    wb, ws, wl = mo.postprocess(logits, deltas, props, 1000, 600, cfg)
    ob, os_, ol, oc = ops.postprocess(logits.to(dev), deltas.to(dev), props.to(dev), None, cfg.bbox_reg_weights, 1000,
                                      600, cfg.score_thresh, cfg.nms, cfg.detections_per_img, True)
    n = int(oc.item())
    assert n == wb.shape[0], "dets %d vs oracle %d" % (n, wb.shape[0])
    assert torch.equal(ol[:n].cpu(), wl)
    assert (ob[:n].cpu() - wb).abs().max() < 1e-3
    assert (os_[:n].cpu() - ws).abs().max() < 1e-6


@pytest.mark.parametrize("B,R", [(10, 300), (3, 77), (2, 1024)])
def test_postprocess_batched_equals_per_image(dev, B, R):
    """the image is a grid dimension of the same kernels: every image has the bits of its own mega_postprocess call"""
    ops = _ops()
    from oracle import mega_oracle as mo
    g = torch.Generator().manual_seed(B * 1000 + R)
    NC = 31
    logits = (torch.randn((B * R, NC), generator=g) * 1.5).to(dev)
    deltas = (torch.randn((B * R, NC * 4), generator=g) * 0.5).to(dev)
    ctr = torch.rand((B * R, 2), generator=g) * torch.tensor([900., 500.])
    wh = torch.rand((B * R, 2), generator=g) * 200 + 10
    props = torch.cat([ctr - wh / 2, ctr + wh / 2], dim=1).clamp(min=0).to(dev)
    cfg = mo.OracleCfg()
    args = (cfg.bbox_reg_weights, 1000, 600, cfg.score_thresh, cfg.nms, cfg.detections_per_img, True)
    ob, os_, ol, oc = ops.postprocess_batched(logits, deltas, props, B, *args)
    for b in range(B):
        sl = slice(b * R, (b + 1) * R)
        wb, ws, wl, wc = ops.postprocess(logits[sl].contiguous(), deltas[sl].contiguous(), props[sl].contiguous(), None, *args)
        n = int(wc.item())
        assert int(oc[b].item()) == n and n > 0
        assert torch.equal(ob[b, :n], wb[:n]) and torch.equal(os_[b, :n], ws[:n]) and torch.equal(ol[b, :n], wl[:n])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(300, 750, True), (675, 1111, True), (70, 33, False), (129, 64, True),
                                   (40, 1500, True)])     # (40, 1500): 16 blocks -> the key range is split 3 ways + combine
def test_relation_attention(dev, dtype, shape):
    """position logits + attention core + the projections (ops.linear) vs the literal reference formula
    (oracle.attention_module_multi_head = roi_box_feature_extractors.py:567-646)."""
    ops = _ops()
    from oracle import mega_oracle as mo
    from mega.pytorch_amd import synth
    from mega.pytorch_amd.relation import RelationWeights, relation_attention_forward
    Nq, Nk, use_pos = shape
    sd = synth.make_state_dict(blocks=(1, 1, 1), seed=4)
    g = torch.Generator().manual_seed(Nq)
    x = torch.randn((Nq, 1024), generator=g).to(dtype)
    r = torch.randn((Nk, 1024), generator=g).to(dtype)

    def boxes(n):
        c = torch.rand((n, 2), generator=g) * torch.tensor([900., 500.])
        wh = torch.rand((n, 2), generator=g) * 250 + 2
        return torch.cat([c - wh / 2, c + wh / 2], dim=1)
    bq, bk = boxes(Nq), boxes(Nk)
    ver = "local" if use_pos else "global"
    pe = mo.cal_position_embedding(bq, bk) if use_pos else None
    ref = x.float() + mo.attention_module_multi_head(sd, mo.FE, ver, 0, x.float(), r.float(), pe)
    wts = RelationWeights(sd, mo.FE, "l_" if use_pos else "g_", 0, dtype, dev, with_pos=use_pos)
    out = relation_attention_forward(wts, x.to(dev), r.to(dev), bq.to(dev) if use_pos else None,
                                     bk.to(dev) if use_pos else None, residual=True)
    tol = 2e-4 if dtype == torch.float32 else 3e-2
    err = _relerr(out.float().cpu(), ref)
    assert err < tol, "attention %s %s relerr %.3g" % (shape, dtype, err)


def test_position_logits(dev):
    ops = _ops()
    from oracle import mega_oracle as mo
    from mega.pytorch_amd import synth
    sd = synth.make_state_dict(blocks=(1, 1, 1), seed=4)
    g = torch.Generator().manual_seed(3)
    c = torch.rand((90, 2), generator=g) * torch.tensor([900., 500.])
    wh = torch.rand((90, 2), generator=g) * 250 + 2
    b = torch.cat([c - wh / 2, c + wh / 2], dim=1)
    bq, bk = b[:37], b[20:]
    pe = mo.cal_position_embedding(bq, bk)
    w, bias = sd[mo.FE + "l_Wgs.0.weight"], sd[mo.FE + "l_Wgs.0.bias"]
    ref = (F.relu(F.conv2d(pe, w, bias)) + 1e-6).log()[0]           # [16, Nq, Nk]
    got = ops.position_logits(bq.to(dev), bk.to(dev), w.view(16, 64).t().contiguous().to(dev), bias.to(dev),
                              mo.dim_mat_values().to(dev)).cpu()[:, :, :bk.shape[0]]
    # log() near relu's zero amplifies 1-ulp differences of the pre-activation: compare exp (the softmax weight)
    assert (got.exp() - ref.exp()).abs().max() < 2e-5
    big = ref > -8
    assert (got[big] - ref[big]).abs().max() < 5e-2
    # fast mode (bf16 path): Cody-Waite reduction + hardware sin/cos, embedding and Wg rounded to bf16 for the
    # matrix-core contraction (64 products of ~2^-8 relative error each): bf16-class tolerance on the softmax weight
    fast = ops.position_logits(bq.to(dev), bk.to(dev), w.view(16, 64).t().contiguous().to(dev), bias.to(dev),
                               mo.dim_mat_values().to(dev), precise=False).cpu()[:, :, :bk.shape[0]]
    err = (fast.exp() - ref.exp()).abs()
    assert err.max() < 6e-3 and err.mean() < 1e-3, (err.max(), err.mean())
    assert torch.isfinite(fast).all()


def test_preprocess(dev):
    ops = _ops()
    from mega.pytorch_amd import synth
    fr = synth.make_clip(2, 40, 64, seed=1)
    ref = synth.preprocess_cpu(fr)
    got = ops.preprocess_frames(fr.to(dev), synth.PIXEL_MEAN).cpu()
    assert torch.equal(got, ref)          # ToTensor (/255), *255, -mean: three separately rounded f32 ops, no FMA
    allv = torch.arange(256, dtype=torch.uint8).repeat(3, 1).t().reshape(1, 16, 16, 3).contiguous()
    assert torch.equal(ops.preprocess_frames(allv.to(dev), synth.PIXEL_MEAN).cpu(), synth.preprocess_cpu(allv))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(5, 9, 13, 16, 24), (19, 38, 63, 1024, 2048), (21, 12, 17, 64, 128)])
def test_fgfa_warp_aggregate(dev, dtype, shape):
    """Fused FGFA warp + cosine weights + softmax + sum vs the oracle restatement of
    generalized_rcnn_fgfa.py:45-76,:201-211 (config 5: T=19 reference default / 21 BASELINE, 38x63, 1024+2048)."""
    ops = _ops()
    from oracle import mega_oracle as mo
    T, H, W, Cf, Ce = shape
    g = torch.Generator().manual_seed(T * H)
    feats = torch.randn((T, Cf + Ce, H, W), generator=g).to(dtype)
    flow = torch.randn((T, 2, H, W), generator=g) * 3
    flow[T // 2] *= 0.05
    flow[0, :, 0, :] = 50.0                                   # far outside: border padding
    key = T // 2
    want, want_w = mo.fgfa_aggregate(feats.float(), flow, key, nfeat=Cf)
    out, w = ops.fgfa_warp_aggregate(feats.permute(0, 2, 3, 1).contiguous().to(dev), flow.to(dev), Cf, key,
                                     want_weights=True)
    assert (w.cpu() - want_w[:, 0]).abs().max() < (2e-5 if dtype == torch.float32 else 2e-3)
    err = _relerr(out.float().cpu().permute(2, 0, 1)[None], want)
    assert err < (1e-5 if dtype == torch.float32 else 1e-2), err


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 7, 9, 8), (3, 38, 63, 32), (1, 1, 5, 8)])
def test_avgpool2x2_ceil(dev, dtype, shape):
    """F.avg_pool2d(x, 2, stride=2, ceil_mode=True) (flownet.py:55,117): edge windows divide by the in-bounds count."""
    ops = _ops()
    x = torch.randn(shape, generator=torch.Generator().manual_seed(7)).to(dtype)
    ref = F.avg_pool2d(x.float().permute(0, 3, 1, 2), 2, stride=2, ceil_mode=True).permute(0, 2, 3, 1)
    got = ops.avgpool2x2_ceil(x.to(dev)).float().cpu()
    assert got.shape == ref.shape
    assert (got - ref).abs().max() < (1e-6 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(3, 37, 51), (21, 60, 100), (2, 8, 1002)])
def test_fgfa_pair_taps(dev, dtype, shape):
    """mega_fgfa_pair_taps (round 6): cat([cur, ref]) -> 16-bit -> AvgPool2d(2, 2, ceil_mode) -> the seven horizontal taps of
    flow_conv1 as one 64-channel row, three zero rows above / below -- BIT-equal to the arithmetic of the steps it replaces
    (tests/cpu_ops.py's twin: each f32 pixel rounded, the in-bounds taps summed in f32 in (dy, dx) order, divided by their count,
    rounded once), for the three ways the key frame is given: one for all pairs, one per pair, a ring slot (order[0]); odd
    sizes (ceil-mode edges) and a row longer than one 250-pixel segment."""
    import cpu_ops
    ops = _ops()
    T, H, W = shape
    g = torch.Generator().manual_seed(T + H + W)
    refs = (torch.rand((T, 3, H, W), generator=g) * 255.0 - 110.0)
    cur1 = (torch.rand((1, 3, H, W), generator=g) * 255.0 - 110.0)
    curT = (torch.rand((T, 3, H, W), generator=g) * 255.0 - 110.0)
    order = torch.tensor([T - 1] + list(range(T)), dtype=torch.int32)
    for cur, od in ((cur1, None), (curT, None), (None, order)):
        if cur is not None and cur.shape[0] == T:        # one key frame per pair: the twin pair by pair
            want = torch.cat([cpu_ops.fgfa_pair_taps(refs[t:t + 1], cur[t:t + 1], None, dtype) for t in range(T)], 0)
        else:
            want = cpu_ops.fgfa_pair_taps(refs, cur, od, dtype)
        got = ops.fgfa_pair_taps(refs.to(dev), None if cur is None else cur.to(dev), None if od is None else od.to(dev), dtype).cpu()
        assert got.shape == want.shape == (T, (H + 1) // 2 + 6, (W + 1) // 2, 64)
        assert torch.equal(got.view(torch.int16), want.view(torch.int16)), "pair taps differ from the twin (max |d| %.3g)" % (
            (got.float() - want.float()).abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dff_warp_scale(dev, dtype):
    """warp(key feats, flow) * scale vs F.grid_sample(bilinear, border) (generalized_rcnn_dff.py:41-60,:134-135)."""
    import cpu_ops
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    H, W, C = 19, 31, 64
    feats = torch.randn((H, W, C), generator=g).to(dtype)
    scale = (1 + 0.2 * torch.randn((H, W, C), generator=g)).to(dtype)
    flow = torch.randn((2, H, W), generator=g) * 3
    flow[:, 0, :] -= 8          # exercise the border clamp
    ref = cpu_ops.dff_warp_scale(feats.float(), flow, scale.float())
    got = ops.dff_warp_scale(feats.to(dev), flow.to(dev), scale.to(dev)).float().cpu()
    assert (got - ref).abs().max() < (1e-5 if dtype == torch.float32 else 3e-2) * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("tile", ["64x64", "128x64", "128x128", "256x128", "256x256"])
def test_conv_every_tile_shape(dev, tile, monkeypatch):
    """every instantiated igemm tile (normally picked by the cost model) on shapes with M / N / K tails, odd K-tile
    counts, residual + ReLU and f32 output: forced through MEGA_IGEMM_TILE."""
    ops = _ops()
    monkeypatch.setenv("MEGA_IGEMM_TILE", tile)
    g = torch.Generator().manual_seed(11)
    for (N, H, W, Cin, Cout, R, pad, res, odt) in [(2, 37, 41, 192, 320, 1, 0, True, torch.bfloat16),
                                                   (1, 29, 31, 64, 200, 3, 1, False, torch.float32),
                                                   (700, 1, 1, 1024, 155, 1, 0, False, torch.float32)]:
        x = torch.randn((N, H, W, Cin), generator=g).to(torch.bfloat16)
        w = (torch.randn((Cout, R, R, Cin), generator=g) / math.sqrt(Cin * R * R)).to(torch.bfloat16)
        sc = torch.rand((Cout,), generator=g) + 0.5
        bi = torch.randn((Cout,), generator=g) * 0.1
        ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), padding=pad)
        ref = ref * sc.view(1, -1, 1, 1) + bi.view(1, -1, 1, 1)
        r = None
        if res:
            r = torch.randn(ref.permute(0, 2, 3, 1).shape, generator=g).to(torch.bfloat16)
            ref = ref + r.float().permute(0, 3, 1, 2)
        ref = F.relu(ref)
        out = ops.conv2d_nhwc(x.to(dev), w.to(dev), sc.to(dev), bi.to(dev), None if r is None else r.contiguous().to(dev),
                              pad=pad, relu=True, out_dtype=odt)
        got = out.float().cpu().permute(0, 3, 1, 2)
        err = _relerr(got, ref)
        assert err < (2e-2 if odt == torch.bfloat16 else 1e-4), "tile %s shape %s relerr %.3g" % (tile, (N, H, W, Cin, Cout, R), err)


def test_stem_mfma_bf16(dev):
    """matrix-core stem (bf16 pixels x bf16 weights, f32 accumulate) vs the f32 conv of the SAME bf16-rounded operands
    (tight), vs the direct f32-math kernel (bf16-class), and on a size with partial tiles on both edges."""
    ops = _ops()
    from oracle import mega_oracle as mo
    from mega.pytorch_amd import synth
    sd = synth.make_state_dict(blocks=(1, 1, 1), seed=2)
    g = torch.Generator().manual_seed(5)
    p = "backbone.body.stem."
    w = sd[p + "conv1.weight"]
    scale = sd[p + "bn1.weight"] * sd[p + "bn1.running_var"].rsqrt()
    bias = sd[p + "bn1.bias"] - sd[p + "bn1.running_mean"] * scale
    w_tap = w.permute(1, 2, 3, 0).reshape(147, 64).contiguous()
    w160 = ops.pack_stem_weight_bf16(w)
    assert w160.shape == (64, 176) and not w160.view(64, 22, 8)[:, :, 7].any() and not w160[:, 168:].any()
    for (N, H, W) in [(2, 75, 101), (1, 600, 1000), (3, 33, 70)]:
        x = torch.randn((N, 3, H, W), generator=g) * 60
        # leave NaN patterns behind in the CUs' LDS (block-wide sorts stage their keys there): the kernel's zero-weight
        # taps read patch padding, which must have been written (0 x NaN is NaN)
        torch.full((1 << 22,), float("nan"), device=dev).sort()
        got = ops.stem(x.to(dev), w_tap.to(dev), scale.to(dev), bias.to(dev), torch.bfloat16, w_n160=w160.to(dev))
        got = got.float().cpu().permute(0, 3, 1, 2)
        conv = F.conv2d(x.bfloat16().float(), w.bfloat16().float(), stride=2, padding=3)
        ref = F.relu(conv * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1))
        assert got.shape == ref.shape
        # output is stored as bf16: half an ulp of the result is the floor
        assert (got - ref).abs().max() <= 2 ** -8 * ref.abs().max() + 1e-6, (N, H, W, (got - ref).abs().max())
        direct = ops.stem(x.to(dev), w_tap.to(dev), scale.to(dev), bias.to(dev), torch.bfloat16).float().cpu().permute(0, 3, 1, 2)
        assert _relerr(got, direct) < 2e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_linear_split_k(dev, dtype):
    """K >= 32768 (the box head's first FC) runs split-K through the workspace entry point: value parity, fused
    bias / ReLU in the finalize kernel, and batch invariance (a row gives the same bits in a batch of 37 or 300)."""
    ops = _ops()
    g = torch.Generator().manual_seed(13)
    M, N, K = 300, 200, 32768 + 1024
    x = (torch.randn((M, K), generator=g) * 0.5).to(dtype)
    w = (torch.randn((N, K), generator=g) / math.sqrt(K)).to(dtype)
    b = torch.randn((N,), generator=g) * 0.1
    ref = F.relu(x.float() @ w.float().t() + b)
    got = ops.linear(x.to(dev), w.to(dev), b.to(dev), relu=True)
    assert _relerr(got.float().cpu(), ref) < (2e-5 if dtype == torch.float32 else 2e-2)
    part = ops.linear(x[:37].contiguous().to(dev), w.to(dev), b.to(dev), relu=True)
    assert torch.equal(part, got[:37])
    from mega.pytorch_amd import _lib
    lib = _lib.load()
    assert lib.mega_conv2d_nhwc_workspace_bytes(M, N, K) == 3 * M * N * 4 and lib.mega_conv2d_nhwc_workspace_bytes(M, N, 4096) == 0   # 3 K ranges


def _untile_pos(t, Nk):
    """bf16 [16, KT, Nq, 32] in (h2, rq, e) tile order -> f32 [16, Nq, Nk]."""
    G, KT, Nq, _ = t.shape
    x = t.float().view(G, KT, Nq, 2, 4, 4).permute(0, 2, 1, 4, 3, 5)      # [G, Nq, KT, rq, h2, e]: key = 8rq + 4h2 + e
    return x.reshape(G, Nq, KT * 32)[:, :, :Nk]


def _tile_pos(p, Nk):
    """f32 [16, Nq, >=Nk] -> bf16 [16, KT, Nq, 32] tile order (pad keys = 0)."""
    G, Nq = p.shape[0], p.shape[1]
    KT = (Nk + 31) // 32
    full = torch.zeros((G, Nq, KT * 32))
    full[:, :, :Nk] = p[:, :, :Nk]
    x = full.view(G, Nq, KT, 4, 2, 4).permute(0, 2, 1, 4, 3, 5)           # [G, KT, Nq, h2, rq, e]
    return x.reshape(G, KT, Nq, 32).to(torch.bfloat16).contiguous()


@pytest.mark.parametrize("shape", [(37, 70), (300, 750), (5, 33), (64, 16)])
def test_position_logits_tiled_bf16(dev, shape):
    """bf16 tile-ordered logits (what the bf16-mode attention consumes) == the f32-row fast kernel, bf16-rounded."""
    ops = _ops()
    from oracle import mega_oracle as mo
    from mega.pytorch_amd import synth
    Nq, Nk = shape
    sd = synth.make_state_dict(blocks=(1, 1, 1), seed=4)
    g = torch.Generator().manual_seed(Nq + Nk)
    c = torch.rand((Nq + Nk, 2), generator=g) * torch.tensor([900., 500.])
    wh = torch.rand((Nq + Nk, 2), generator=g) * 250 + 2
    b = torch.cat([c - wh / 2, c + wh / 2], dim=1)
    bq, bk = b[:Nq], b[Nq:]
    w, bias = sd[mo.FE + "l_Wgs.0.weight"], sd[mo.FE + "l_Wgs.0.bias"]
    args = (bq.to(dev), bk.to(dev), w.view(16, 64).t().contiguous().to(dev), bias.to(dev), mo.dim_mat_values().to(dev))
    rows = ops.position_logits(*args, precise=False).cpu()[:, :, :Nk]
    tiled = ops.position_logits(*args, precise=False, tiled=True).cpu()
    assert tiled.dtype == torch.bfloat16 and tuple(tiled.shape) == (16, (Nk + 31) // 32, Nq, 32)
    got = _untile_pos(tiled, Nk)
    assert torch.equal(got, rows.to(torch.bfloat16).float())


@pytest.mark.parametrize("shape", [(300, 750), (675, 1111), (33, 70), (129, 64), (40, 1500)])   # last: split keys + combine
def test_relation_attention_tiled_pos(dev, shape):
    """bf16 attention with tile-ordered bf16 logits == the same kernel fed the same (bf16-rounded) logits as f32 rows,
    bit for bit: only the fetch path differs."""
    ops = _ops()
    Nq, Nk = shape
    g = torch.Generator().manual_seed(Nq * 3 + Nk)
    q = (torch.randn((Nq, 1024), generator=g) * 0.3).to(torch.bfloat16)
    k = (torch.randn((Nk, 1024), generator=g) * 0.3).to(torch.bfloat16)
    ld = (Nk + 31) // 32 * 32
    vt = torch.zeros((1024, ld), dtype=torch.bfloat16)
    vt[:, :Nk] = torch.randn((1024, Nk), generator=g).to(torch.bfloat16)
    pos = (torch.randn((16, Nq, ld), generator=g) * 2 - 3).to(torch.bfloat16).float()
    resid = torch.randn((Nq, 1024), generator=g).to(torch.bfloat16)
    bv = torch.randn((1024,), generator=g) * 0.1
    a = ops.relation_attention(q.to(dev), k.to(dev), vt.to(dev), Nk, pos=pos.to(dev), resid=resid.to(dev), bias_v=bv.to(dev))
    b = ops.relation_attention(q.to(dev), k.to(dev), vt.to(dev), Nk, pos=_tile_pos(pos, Nk).to(dev), resid=resid.to(dev),
                               bias_v=bv.to(dev))
    assert torch.equal(a, b)


# ------------------------------------------------------------------------------------------------ hot shapes
# The exact shapes of the benchmarked configuration (MEGA R-101, 600x1000, 20-frame frame-stage batches):
# SURVEY.md Appendix A attention call list, the first FC of the box head, the RPN conv through its natural dispatch.

@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(675, 3750, True), (1875, 750, False), (300, 750, True)])
def test_relation_attention_hot_shapes(dev, dtype, shape):
    """local stage 0 (675 x 3750 with the position term: 1875 window rows + 1875 memory rows), the global stage over
    the whole window (1875 x 750) and the last local stage (300 x 750) against the literal reference formula."""
    from oracle import mega_oracle as mo
    from mega.pytorch_amd import synth
    from mega.pytorch_amd.relation import RelationWeights, relation_attention_forward
    Nq, Nk, use_pos = shape
    torch.set_num_threads(16)
    sd = synth.make_state_dict(blocks=(1, 1, 1), seed=4)
    g = torch.Generator().manual_seed(Nq + Nk)
    x = torch.randn((Nq, 1024), generator=g).to(dtype)
    r = torch.randn((Nk, 1024), generator=g).to(dtype)

    def boxes(n):
        c = torch.rand((n, 2), generator=g) * torch.tensor([900., 500.])
        wh = torch.rand((n, 2), generator=g) * 250 + 2
        return torch.cat([c - wh / 2, c + wh / 2], dim=1)
    bq, bk = boxes(Nq), boxes(Nk)
    ver = "local" if use_pos else "global"
    with torch.no_grad():
        pe = mo.cal_position_embedding(bq, bk) if use_pos else None
        ref = x.float() + mo.attention_module_multi_head(sd, mo.FE, ver, 0, x.float(), r.float(), pe)
    del pe
    wts = RelationWeights(sd, mo.FE, "l_" if use_pos else "g_", 0, dtype, dev, with_pos=use_pos)
    out = relation_attention_forward(wts, x.to(dev), r.to(dev), bq.to(dev) if use_pos else None,
                                     bk.to(dev) if use_pos else None, residual=True)
    err = _relerr(out.float().cpu(), ref)
    # (float16, round 6: the same kernels on IEEE-half operands -- 11 significant bits: an eighth of the bf16 bound)
    bound = {torch.float32: 2e-4, torch.bfloat16: 3e-2, torch.float16: 4e-3}[dtype]
    print("attention %s %s relerr %.3g" % (shape, dtype, err))
    assert err < bound, "attention %s %s relerr %.3g" % (shape, dtype, err)
    # the engine's form of the same call: the second half of the keys arrives as cached memory projections
    if use_pos and Nk == 3750:
        h = Nk // 2
        _, k1, vt1 = relation_attention_forward(wts, x[:8].to(dev), r[h:].to(dev), bq[:8].to(dev), bk[h:].to(dev),
                                                residual=True, return_kv=True)
        out2 = relation_attention_forward(wts, x.to(dev), r[:h].to(dev), bq.to(dev), bk.to(dev), residual=True,
                                          mem_kv=(k1, vt1[:, :Nk - h].contiguous()))
        err2 = _relerr(out2.float().cpu(), ref)
        assert err2 < bound, "attention with cached memory K/V: relerr %.3g" % err2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_first_fc_hot_shape(dev, dtype):
    """l_fcs[0] at its real size: K = 100352 (2048 x 7 x 7), 1024 outputs, M = 375 rows (one 300-row local frame + one
    75-row global frame), through the split-K path its natural dispatch takes."""
    ops = _ops()
    torch.set_num_threads(16)
    g = torch.Generator().manual_seed(21)
    M, K, N = 375, 100352, 1024
    x = torch.randn((M, K), generator=g).to(dtype)
    w = (torch.randn((N, K), generator=g) / math.sqrt(K)).to(dtype)
    b = torch.randn((N,), generator=g) * 0.1
    ref = F.relu(F.linear(x.float(), w.float(), b))
    out = ops.linear(x.to(dev), w.to(dev), b.to(dev), relu=True)
    err = _relerr(out.float().cpu(), ref)
    assert err < (1e-4 if dtype == torch.float32 else 2e-2), "fc0 %s relerr %.3g" % (dtype, err)
    # batch invariance: the same rows inside a bigger batch give the same bits
    x2 = torch.cat([x, x.flip(0)], dim=0)
    out2 = ops.linear(x2.to(dev), w.to(dev), b.to(dev), relu=True)
    assert torch.equal(out2[:M], out)


def test_rpn_conv_natural_dispatch_hot_shape(dev):
    """The RPN 3x3 conv (1024 -> 1024, K = 9216) on a 20-frame batch of 38x63 maps (M = 47880), through the tile its
    natural dispatch picks, bf16: (i) two whole frames against torch-CPU fp32 on the same bf16-rounded operands,
    (ii) all 20 frames bit-identical to the same frames computed in batches of 1 (batch invariance across tiles)."""
    ops = _ops()
    lib = __import__("mega.pytorch_amd._lib", fromlist=["load"]).load()
    torch.set_num_threads(16)
    g = torch.Generator().manual_seed(33)
    Nf, H, W, C = 20, 38, 63, 1024
    x = (torch.randn((Nf, H, W, C), generator=g)).to(torch.bfloat16)
    w = (torch.randn((C, 3, 3, C), generator=g) / math.sqrt(9 * C)).to(torch.bfloat16)
    b = torch.randn((C,), generator=g) * 0.1
    tile = lib.mega_conv2d_nhwc_tile(Nf * H * W, C, 9 * C)
    print("RPN conv natural tile: %dx%d" % (tile // 1000, tile % 1000))
    xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
    out = ops.conv2d_nhwc(xd, wd, None, bd, pad=1, relu=True)
    for f in (0, Nf - 1):
        ref = F.relu(F.conv2d(x[f:f + 1].float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b, padding=1))
        err = _relerr(out[f:f + 1].float().cpu().permute(0, 3, 1, 2), ref)
        assert err < 2e-2, "frame %d relerr %.3g" % (f, err)
    for f in (0, 7, Nf - 1):
        one = ops.conv2d_nhwc(xd[f:f + 1].contiguous(), wd, None, bd, pad=1, relu=True)
        assert torch.equal(one[0], out[f]), "frame %d differs between batch-of-1 and batch-of-%d dispatch" % (f, Nf)


def test_igemm8_bit_equal_to_register_staged_tiles(dev):
    """igemm8 (LDS-DMA staging, 8 waves, 256/192 x 256 tiles, the kernel the big frame-stage layers are dispatched to)
    must give the SAME BITS as the register-staged igemm tiles on every shape class: same MFMA instruction, same
    ascending-K order per output element -> tile / kernel choice never changes a result (batch invariance).  Each case
    runs several times (race screen)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "igemm8_check", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "gpu",
                                     "igemm8_check.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    for case in chk.CASES:
        ref = chk.run(case, "128x128")[0]
        for force in ("8:256", "8:192"):
            outs = chk.run(case, force, reps=3)
            assert all(torch.equal(outs[0], o) for o in outs[1:]), "%s %s: run-to-run difference" % (case, force)
            assert torch.equal(outs[0], ref), "%s %s: differs from the 128x128 tile (max |d| %.3g)" % (
                case, force, (outs[0].float() - ref.float()).abs().max().item())


def test_experimental_gemm_kernels_bit_equal_to_register_staged_tiles(dev):
    """Round 6's opt-in GEMM kernels -- igemm4 (4 waves x 512 registers, 128 x 128 outputs per wave), igemm2 (128 x 256 tiles, K-tile
    32, two blocks per CU) and stream1x1 (persistent, wave-owned 32 x 256 tiles, weights resident in LDS) -- are not dispatched by
    default (measured at or below igemm8: profiles/r06_streaming_class_experiments.txt) but stay in the library behind
    MEGA_IGEMM4 / MEGA_IGEMM2 / MEGA_STREAM1X1; they must give the SAME BITS as the register-staged tiles (same MFMA, same
    ascending K order, same epilogue arithmetic).  Shapes a kernel does not take fall through to igemm8 inside the dispatcher."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def load(name):
        import sys
        sys.path.insert(0, os.path.join(root, "tools", "gpu"))
        spec = importlib.util.spec_from_file_location(name, os.path.join(root, "tools", "gpu", name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    chk = load("igemm8_check")
    cases = list(chk.CASES) + [
        (9, 38, 63, 256, 1024, 1, 1, 0, 1, 1, True, False),       # layer3 conv3 (M = 21546: stream1x1 takes it, 10-row tail tile)
        (9, 38, 63, 256, 1024, 1, 1, 0, 1, 2, False, False),      # LeakyReLU, no residual
        (2, 75, 125, 128, 512, 1, 1, 0, 1, 1, True, False),       # layer2 conv3 (K = 128)
        (8, 38, 63, 256, 2048, 1, 1, 0, 1, 0, True, False),       # eight N tiles
    ]
    for case in cases:
        ref = chk.run(case, "128x128")[0]
        for force in ("4:256", "4:192", "2:128", "s:32"):
            outs = chk.run(case, force, reps=2)
            assert torch.equal(outs[0], outs[1]), "%s %s: run-to-run difference" % (case, force)
            assert torch.equal(outs[0], ref), "%s %s: differs from the 128x128 tile (max |d| %.3g)" % (
                case, force, (outs[0].float() - ref.float()).abs().max().item())


def test_multi_cat_equals_torch_cat(dev):
    """ops.multi_cat (mega_copy_segments: every concatenation of a call in one launch per copy width) == torch.cat,
    bit for bit: row blocks (16-byte rows), f32 boxes, and 2-byte-aligned V^T column blocks (75 keys = 150 bytes)
    taken as views of wider buffers; empty pieces; more segments than one launch holds."""
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    big = torch.randn((4000, 1024), generator=g).to(torch.bfloat16).to(dev)
    vt = torch.randn((1024, 4000), generator=g).to(torch.bfloat16).to(dev)
    boxes = torch.rand((900, 4), generator=g).to(dev)
    rows = [big[0:75], big[300:600], big[75:75], big[1000:1015], big[3000:3999]]
    cols = [vt[:, 0:75], vt[:, 75:150], vt[:, 1000:1015], vt[:, 1875:3750], vt[:, 3:4]]
    bx = [boxes[0:300], boxes[300:375], boxes[890:900]]
    many = [big[i * 7:i * 7 + 5] for i in range(130)]                     # 130 segments: three launches of <= 56
    outs = ops.multi_cat([(rows, 0), (cols, 1), (bx, 0), (many, 0)])
    torch.cuda.synchronize()
    assert torch.equal(outs[0], torch.cat(rows, 0)) and torch.equal(outs[1], torch.cat(cols, 1))
    assert torch.equal(outs[2], torch.cat(bx, 0)) and torch.equal(outs[3], torch.cat(many, 0))
    # copy into views of an existing buffer (destination row stride != row length)
    dst = torch.zeros((64, 300), dtype=torch.bfloat16, device=dev)
    ops.copy_blocks([(dst[:, 10:85], vt[:64, 100:175]), (dst[:, 85:86], vt[:64, 7:8])])
    ref = torch.zeros_like(dst)
    ref[:, 10:85] = vt[:64, 100:175]
    ref[:, 85:86] = vt[:64, 7:8]
    assert torch.equal(dst, ref)
    # round 4 (copy_any_kernel): any geometry in one launch -- odd byte counts, byte tensors, rows shorter than one unit,
    # 2-byte-aligned sources AND destinations, nothing written outside the destination blocks
    u8 = torch.randint(0, 256, (37, 1001), generator=g, dtype=torch.uint8).to(dev)
    d8 = torch.full((37, 1200), 7, dtype=torch.uint8, device=dev)
    d16 = torch.full((1024, 500), -1.0, dtype=torch.bfloat16, device=dev)
    d32 = torch.full((50, 9), -2.0, device=dev)
    ops.copy_blocks([(d8[:, 3:1004], u8), (d16[:, 1:76], vt[:, 75:150]), (d16[:, 77:80], vt[:, 1001:1004]),
                     (d16[:, 301:490], vt[:, 2001:2190]), (d32[:, 2:3], boxes[100:150, 1:2]), (d32[:, 4:8], boxes[200:250])])
    r8, r16, r32 = torch.full_like(d8, 7), torch.full_like(d16, -1.0), torch.full_like(d32, -2.0)
    r8[:, 3:1004] = u8
    r16[:, 1:76] = vt[:, 75:150]
    r16[:, 77:80] = vt[:, 1001:1004]
    r16[:, 301:490] = vt[:, 2001:2190]
    r32[:, 2:3] = boxes[100:150, 1:2]
    r32[:, 4:8] = boxes[200:250]
    assert torch.equal(d8, r8) and torch.equal(d16.view(torch.int16), r16.view(torch.int16)) and torch.equal(d32, r32)


@pytest.mark.parametrize("shape", [(2, 150, 250), (5, 70, 90), (3, 151, 249), (40, 33, 47)])
def test_conv64_persistent_bit_equal_to_generic_tiles(dev, shape, monkeypatch):
    """conv64.hip (layer1's 3x3 64 -> 64 conv + FrozenBN + ReLU as a persistent kernel with the weights resident in LDS
    and an 18 x 18 halo patch per 16 x 16 tile) against the generic implicit-GEMM tile on the same inputs: same MFMA, same
    ascending (r, s, c) K order -> the same bits; and both against F.conv2d in f32.  Shapes with partial edge tiles."""
    ops = _ops()
    N, H, W = shape
    g = torch.Generator().manual_seed(17)
    x = torch.randn((N, H, W, 64), generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn((64, 3, 3, 64), generator=g) * 0.05).to(torch.bfloat16).to(dev)
    sc = (torch.rand((64,), generator=g) + 0.5).to(dev)
    bi = (torch.randn((64,), generator=g) * 0.1).to(dev)
    from mega.pytorch_amd import _lib
    lib = _lib.load()
    assert lib.mega_conv2d_nhwc_plan_ex(N, H, W, 64, 64, 3, 3, 1, 1, 1, 64, 0, 1, 1) // 1000000 == 6      # dispatched to conv64
    # the plan asks the launch path's own predicate: a residual, an f32 output or a stride keep the layer on the generic tiles
    assert lib.mega_conv2d_nhwc_plan_ex(N, H, W, 64, 64, 3, 3, 1, 1, 1, 64, 1, 1, 1) // 1000000 != 6
    assert lib.mega_conv2d_nhwc_plan_ex(N, H, W, 64, 64, 3, 3, 1, 1, 1, 64, 0, 1, 0) // 1000000 != 6
    assert lib.mega_conv2d_nhwc_plan_ex(1, 16, 16, 64, 64, 3, 3, 1, 1, 1, 64, 0, 1, 1) // 1000000 != 6    # one tile: too few
    for relu in (True, False, 2):
        y = ops.conv2d_nhwc(x, w, sc, bi, pad=1, relu=relu)
        monkeypatch.setenv("MEGA_IGEMM_TILE", "128x64")
        y_ref = ops.conv2d_nhwc(x, w, sc, bi, pad=1, relu=relu)
        monkeypatch.delenv("MEGA_IGEMM_TILE")
        torch.cuda.synchronize()
        assert torch.equal(y, y_ref), "relu=%s: %d elements differ" % (relu, (y != y_ref).sum().item())
        # bit level (torch.equal treats -0 and +0 as equal): the ReLU epilogues agree on the sign of zero as well
        assert torch.equal(y.view(torch.int16), y_ref.view(torch.int16)), "relu=%s: bit patterns differ" % (relu,)
    ref = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), w.float().cpu().permute(0, 3, 1, 2), padding=1)
    ref = F.relu(ref * sc.cpu().view(1, -1, 1, 1) + bi.cpu().view(1, -1, 1, 1)).permute(0, 2, 3, 1)
    y = ops.conv2d_nhwc(x, w, sc, bi, pad=1, relu=True)
    assert _relerr(y.float().cpu(), ref) < 1e-2


# ------------------------------------------------------------------------------------------------ f32 head stream (round 4)
@pytest.mark.parametrize("n", [8, 1024 * 675, 1024 * 3 + 5, 7])
def test_cast_f32_to_bf16(dev, n):
    """ops.cast_bf16 == torch's float -> bfloat16 (round to nearest even), bit for bit, incl. a tail that is not a
    multiple of the 8-element vectors."""
    ops = _ops()
    g = torch.Generator().manual_seed(n)
    x = torch.randn((n,), generator=g) * 3
    x[:4] = torch.tensor([0.0, -0.0, 1e-40, 65504.0])[:min(4, n)] if n >= 4 else x[:4]
    got = ops.cast_bf16(x.to(dev))
    assert got.dtype == torch.bfloat16 and torch.equal(got.cpu().view(torch.int16), x.to(torch.bfloat16).view(torch.int16))


@pytest.mark.parametrize("shape", [(300, 750), (129, 70), (40, 1500)])     # last: split keys + combine
def test_relation_attention_f32_stream(dev, shape):
    """io_f32 (cfg.HEAD_STREAM float32): bf16 operands, f32 residual and f32 output.  The attention term is the bf16
    kernel's own (same MFMAs, same softmax): out_f32 - resid == the bare bf16-mode output before its final rounding, so
    rounding it to bf16 reproduces the bare call's bits up to one bf16 ulp of a double rounding; and it is close to the
    f32 formula."""
    ops = _ops()
    Nq, Nk = shape
    g = torch.Generator().manual_seed(Nq + 7 * Nk)
    q = (torch.randn((Nq, 1024), generator=g) * 0.3).to(torch.bfloat16)
    k = (torch.randn((Nk, 1024), generator=g) * 0.3).to(torch.bfloat16)
    ld = (Nk + 31) // 32 * 32
    vt = torch.zeros((1024, ld), dtype=torch.bfloat16)
    vt[:, :Nk] = (torch.randn((1024, Nk), generator=g) + 0.5).to(torch.bfloat16)
    resid = torch.randn((Nq, 1024), generator=g) * 4
    bv = torch.randn((1024,), generator=g) * 0.1
    bare = ops.relation_attention(q.to(dev), k.to(dev), vt.to(dev), Nk, resid=None, bias_v=bv.to(dev))
    out = ops.relation_attention(q.to(dev), k.to(dev), vt.to(dev), Nk, resid=resid.to(dev), bias_v=bv.to(dev))
    assert out.dtype == torch.float32 and bare.dtype == torch.bfloat16
    att = out.cpu() - resid                  # (carries the f32 rounding of resid + att: ~1e-6 relative to |resid|)
    assert (att - bare.float().cpu()).abs().max() <= 1.02 * 2.0 ** -8 * bare.float().abs().max().item() + 4e-6 * resid.abs().max().item()
    # the literal formula on the same (bf16-valued) operands
    qh = q.float().view(Nq, 16, 64).permute(1, 0, 2)
    kh = k.float().view(Nk, 16, 64).permute(1, 0, 2)
    p = F.softmax(torch.bmm(qh, kh.transpose(1, 2)) / 8.0, dim=2)
    ref = torch.bmm(p, vt.float()[:, :Nk].view(16, 64, Nk).transpose(1, 2)).permute(1, 0, 2).reshape(Nq, 1024) + bv + resid
    assert _relerr(out.cpu(), ref) < 2e-3
    # the batched entry gives the same bits
    both = ops.relation_attention_batched([{"q": q.to(dev), "k": k.to(dev), "vt": vt.to(dev), "Nk": Nk, "resid": resid.to(dev),
                                            "bias_v": bv.to(dev)} for _ in range(2)])
    assert torch.equal(both[0], out) and torch.equal(both[1], out)


@pytest.mark.parametrize("dtype", ["bfloat16", "float32"])
@pytest.mark.parametrize("shape", [(300, 1875, 1875), (675, 1875, 1800), (130, 75, 150), (64, 37, 5), (200, 96, 64),
                                   (90, 1, 40), (33, 250, 1)])
def test_attention_two_key_segments_bit_equal(dev, dtype, shape):
    """mega_attn_desc.nk1 / k2 / vt2: keys 0 .. N1-1 read from (k, vt), keys N1 .. Nk-1 from (k2, vt2) -- column / row blocks
    of wider buffers at ELEMENT alignment (what MEGAFeatureExtractor.aggregate_batch hands over: the projections' output
    and the memory tape) -- against the same keys copied into one K / V^T buffer: the same bits, for seams inside a tile,
    inside a 16-byte vector, at tile boundaries, for one-key segments, with and without the tile-ordered position logits."""
    ops = _ops()
    dt = getattr(torch, dtype)
    Nq, N1, N2 = shape
    Nk = N1 + N2
    g = torch.Generator().manual_seed(N1 * 3 + N2)
    q = (torch.randn((Nq, 1024), generator=g) * 0.3).to(dt).to(dev)
    kfull = (torch.randn((Nk, 1024), generator=g) * 0.3).to(dt)
    vfull = (torch.randn((1024, Nk), generator=g) + 0.5).to(dt)
    resid = (torch.randn((Nq, 1024), generator=g)).to(dt).to(dev)
    bv = (torch.randn((1024,), generator=g) * 0.1).to(dev)
    ld = (Nk + 31) // 32 * 32
    vt = torch.zeros((1024, ld), dtype=dt)
    vt[:, :Nk] = vfull
    rq = torch.rand((Nq, 4), generator=g) * 100
    rq[:, 2:] += rq[:, :2] + 5
    rk = torch.rand((Nk, 4), generator=g) * 100
    rk[:, 2:] += rk[:, :2] + 5
    wg = torch.randn((64, 16), generator=g) * 0.3
    bg = torch.randn((16,), generator=g) * 0.1 + 0.3
    dim_mat = torch.full((8,), 1000.0).pow(torch.arange(8) / 8.0)
    fast = dt == torch.bfloat16
    pos = ops.position_logits(rq.to(dev), rk.to(dev), wg.to(dev), bg.to(dev), dim_mat.to(dev), precise=not fast, tiled=fast)
    # the two segments inside wider buffers at odd element offsets (3 and 5 columns in, rows after 2 other rows)
    a_cols, b_cols = 3, 5
    big1 = torch.full((1024, (a_cols + N1 + 9 + 7) // 8 * 8), 7.0, dtype=dt)     # (first segment: 16-byte row pitch, any column)
    big1[:, a_cols:a_cols + N1] = vfull[:, :N1]
    big2 = torch.full((1024, b_cols + N2 + 11), -3.0, dtype=dt)
    big2[:, b_cols:b_cols + N2] = vfull[:, N1:]
    kb1 = torch.zeros((2 + N1 + 1, 1024), dtype=dt)
    kb1[2:2 + N1] = kfull[:N1]
    kb2 = torch.zeros((1 + N2 + 3, 1024), dtype=dt)
    kb2[1:1 + N2] = kfull[N1:]
    big1, big2, kb1, kb2 = big1.to(dev), big2.to(dev), kb1.to(dev), kb2.to(dev)
    for p_ in (None, pos):
        one = ops.relation_attention_batched([{"q": q, "k": kfull.to(dev), "vt": vt.to(dev), "Nk": Nk, "resid": resid,
                                               "bias_v": bv, "pos": p_}])[0]
        two = ops.relation_attention_batched([{"q": q, "k": kb1[2:2 + N1], "vt": big1[:, a_cols:a_cols + N1], "N1": N1,
                                               "k2": kb2[1:1 + N2], "vt2": big2[:, b_cols:b_cols + N2], "Nk": Nk,
                                               "resid": resid, "bias_v": bv, "pos": p_}])[0]
        assert torch.equal(one, two), (dtype, shape, p_ is not None, (one.float() - two.float()).abs().max().item())
    assert torch.isfinite(one.float()).all()


def test_attention_row_sum_uses_the_rounded_p(dev):
    """bf16 mode: softmax weights are rounded to bf16 for the PV MFMA and the row sum is taken over the ROUNDED values,
    so the output is an exact weighted mean of the values: with every value row equal to a constant c the output is c to
    f32 round-off whatever the weights are (with the sum over the unrounded weights it was off by ~1e-3 c / sqrt(keys))."""
    ops = _ops()
    Nq, Nk = 200, 333
    g = torch.Generator().manual_seed(5)
    q = (torch.randn((Nq, 1024), generator=g)).to(torch.bfloat16)
    k = (torch.randn((Nk, 1024), generator=g)).to(torch.bfloat16)
    ld = (Nk + 31) // 32 * 32
    c = (torch.randn((1024,), generator=g) * 3).to(torch.bfloat16)
    vt = torch.zeros((1024, ld), dtype=torch.bfloat16)
    vt[:, :Nk] = c[:, None]
    zero = torch.zeros((Nq, 1024))
    out = ops.relation_attention(q.to(dev), k.to(dev), vt.to(dev), Nk, resid=zero.to(dev))
    assert (out.cpu() - c.float()[None, :]).abs().max() <= 2e-6 * c.float().abs().max()


def test_linear_transposed_residual_and_split_v(dev):
    """ops.linear_transposed(residual=...) and relation.project_v with split weights: V'^T = Wv . ref rounded ONCE per
    element -- the error against the f32 product has no component common to all keys (the mean over the keys of the error
    is ~sqrt(keys) below a single rounding), unlike the one-pass projection with bf16 weights."""
    ops = _ops()
    from types import SimpleNamespace
    from mega.pytorch_amd.relation import project_v
    g = torch.Generator().manual_seed(9)
    M, K, N = 1111, 1024, 1024
    ld = (M + 31) // 32 * 32
    w32 = torch.randn((N, K), generator=g) * 0.02
    x = (torch.rand((M, K), generator=g) + 0.2).to(torch.bfloat16)      # post-ReLU-like: a large common part
    wh = w32.to(torch.bfloat16)
    wl = (w32 - wh.float()).to(torch.bfloat16)
    exact = (x.double() @ w32.double().t()).t()
    r = (torch.randn((N, ld), generator=g)).to(torch.bfloat16)
    got = ops.linear_transposed(wh.to(dev), x.to(dev), ld, residual=r.to(dev)).float().cpu()
    want = ((x.float() @ wh.float().t()).t() + r.float()[:, :M]).to(torch.bfloat16).float()
    assert (got[:, :M] - want).abs().max() <= 2.0 ** -7 * want.abs().max() and torch.equal(got[:, M:], r.float()[:, M:] * 0)
    one = project_v(SimpleNamespace(wv=wh.to(dev), wv_lo=None), x.to(dev), ld).float().cpu()[:, :M]
    two = project_v(SimpleNamespace(wv=wh.to(dev), wv_lo=wl.to(dev)), x.to(dev), ld).float().cpu()[:, :M]
    e1, e2 = (one.double() - exact), (two.double() - exact)
    scale = exact.abs().mean()
    # per-element error: both a bf16 rounding of the result; error of the MEAN over the keys: the split form averages out
    assert e2.abs().max() <= 1.1 * 2.0 ** -8 * exact.abs().max()      # (bf16 half-ulp of the largest element)
    m1, m2 = e1.mean(dim=1).abs().mean() / scale, e2.mean(dim=1).abs().mean() / scale
    print("common-mode error of V'^T over %d keys: one pass %.2e, split %.2e (relative to mean |V|)" % (M, m1, m2))
    assert m2 < 1.5e-4 and m2 < 0.35 * m1


@pytest.mark.parametrize("shape", [(3, 61, 77), (2, 600, 1000), (1, 7, 9)])
@pytest.mark.parametrize("to_bgr", [True, False])
def test_stem_u8_bit_equal_to_preprocess_plus_stem(dev, shape, to_bgr):
    """ops.stem_u8 (the bf16 matrix-core stem reading the uint8 frames, preprocessing on its patch load) ==
    ops.stem(ops.preprocess_frames(frames)) bit for bit: same f32 subtraction of the mean, same f32 -> bf16 conversion,
    same MFMAs (data/transforms/transforms.py:83-129 + backbone/resnet.py:347-366)."""
    ops = _ops()
    N, H, W = shape
    g = torch.Generator().manual_seed(H)
    u8 = torch.randint(0, 256, (N, H, W, 3), generator=g, dtype=torch.uint8).to(dev)
    w = torch.randn((64, 3, 7, 7), generator=g) * 0.05
    sc = (torch.rand((64,), generator=g) + 0.5).to(dev)
    bi = (torch.randn((64,), generator=g) * 0.1).to(dev)
    mean = (102.9801, 115.9465, 122.7717)
    w160 = ops.pack_stem_weight_bf16(w).to(dev)
    wt = w.permute(1, 2, 3, 0).reshape(147, 64).contiguous().to(dev)
    ref = ops.stem(ops.preprocess_frames(u8, mean, to_bgr), wt, sc, bi, torch.bfloat16, w_n160=w160)
    got = ops.stem_u8(u8, w160, sc, bi, mean, to_bgr)
    assert got.shape == ref.shape and torch.equal(got.view(torch.int16), ref.view(torch.int16))


@pytest.mark.parametrize("shape", [(2, 600, 1000), (2, 75, 131), (1, 9, 7), (3, 64, 66), (1, 17, 130), (2, 31, 33)])
def test_stem_pool_bit_equal(dev, shape):
    """ops.stem_pool (stem conv + BN + ReLU + 3x3/2 max-pool in one kernel; the stem's 64-channel map stays in LDS) ==
    ops.maxpool3x3s2(ops.stem[_u8](...)) bit for bit, from the uint8 frames and from the preprocessed f32 image, including
    odd sizes (windows hanging over the map's right / bottom edge), maps smaller than one tile and the padding row /
    column at the top / left (backbone/resnet.py:355-366)."""
    ops = _ops()
    N, H, W = shape
    g = torch.Generator().manual_seed(H * 3 + W)
    u8 = torch.randint(0, 256, (N, H, W, 3), generator=g, dtype=torch.uint8).to(dev)
    w = torch.randn((64, 3, 7, 7), generator=g) * 0.05
    sc = (torch.rand((64,), generator=g) + 0.5).to(dev)
    bi = (torch.randn((64,), generator=g) * 0.1).to(dev)
    mean = (102.9801, 115.9465, 122.7717)
    w160 = ops.pack_stem_weight_bf16(w).to(dev)
    wt = w.permute(1, 2, 3, 0).reshape(147, 64).contiguous().to(dev)
    for to_bgr in (True, False):
        ref = ops.maxpool3x3s2(ops.stem_u8(u8, w160, sc, bi, mean, to_bgr))
        got = ops.stem_pool(u8, w160, sc, bi, mean, to_bgr)
        assert got.shape == ref.shape and torch.equal(got.view(torch.int16), ref.view(torch.int16))
    img = ops.preprocess_frames(u8, mean, True)
    ref = ops.maxpool3x3s2(ops.stem(img, wt, sc, bi, torch.bfloat16, w_n160=w160))
    got = ops.stem_pool(img, w160, sc, bi)
    assert got.shape == ref.shape and torch.equal(got.view(torch.int16), ref.view(torch.int16))
    assert float(got.float().abs().max()) > 0


@pytest.mark.parametrize("shape", [(2, 150, 250), (3, 37, 53), (1, 8, 16), (5, 64, 48), (40, 9, 17)])
def test_fused_bottleneck64_bit_equal_to_unfused(dev, shape):
    """ops.bottleneck64 (bneck64.hip: layer1's identity bottleneck 256 -> 64 -> 64 (3x3) -> 256 + residual in one persistent
    kernel, the 64-channel intermediates in LDS) against the three conv2d_nhwc launches it replaces (backbone/resnet.py:
    324-344): same MFMA, same ascending K order, same bf16 roundings of the intermediates, same epilogue arithmetic ->
    the same BITS, including partial edge tiles, single-tile images and the zero padding of the 3x3 conv at the border
    (t1 = 0 outside the image, not relu(bias)); and close to F.conv2d in f32."""
    ops = _ops()
    N, H, W = shape
    g = torch.Generator().manual_seed(H * 7 + W)
    x = torch.randn((N, H, W, 256), generator=g).relu().to(torch.bfloat16).to(dev)
    w1 = (torch.randn((64, 1, 1, 256), generator=g) * 0.06).to(torch.bfloat16).to(dev)
    w2 = (torch.randn((64, 3, 3, 64), generator=g) * 0.05).to(torch.bfloat16).to(dev)
    w3 = (torch.randn((256, 1, 1, 64), generator=g) * 0.1).to(torch.bfloat16).to(dev)
    sb = [((torch.rand((n,), generator=g) + 0.5).to(dev), (torch.randn((n,), generator=g) * 0.2).to(dev)) for n in (64, 64, 256)]
    t1 = ops.conv2d_nhwc(x, w1, sb[0][0], sb[0][1], relu=True)
    t2 = ops.conv2d_nhwc(t1, w2, sb[1][0], sb[1][1], pad=1, relu=True)
    ref = ops.conv2d_nhwc(t2, w3, sb[2][0], sb[2][1], residual=x, relu=True)
    got = ops.bottleneck64(x, w1, sb[0][0], sb[0][1], w2, sb[1][0], sb[1][1], w3, sb[2][0], sb[2][1])
    torch.cuda.synchronize()
    nd = (got.view(torch.int16) != ref.view(torch.int16)).sum().item()
    assert nd == 0, "%d of %d elements differ (max |d| %.3g)" % (nd, got.numel(), (got.float() - ref.float()).abs().max().item())
    # f32 reference of the whole block
    xf = x.float().cpu().permute(0, 3, 1, 2)
    def bn(y, i):
        return y * sb[i][0].cpu().view(1, -1, 1, 1) + sb[i][1].cpu().view(1, -1, 1, 1)
    y = F.relu(bn(F.conv2d(xf, w1.float().cpu().permute(0, 3, 1, 2)), 0))
    y = F.relu(bn(F.conv2d(y, w2.float().cpu().permute(0, 3, 1, 2), padding=1), 1))
    y = F.relu(bn(F.conv2d(y, w3.float().cpu().permute(0, 3, 1, 2)), 2) + xf).permute(0, 2, 3, 1)
    assert _relerr(got.float().cpu(), y) < 2e-2


def test_cat_rows_cast_bf16(dev):
    """ops.cat_rows_cast_bf16 == torch.cat(pieces, 0).to(bfloat16), bit for bit: contiguous pieces, row slices of wider
    buffers (row stride > K), single-row and empty pieces, more pieces than one launch's 56 segments."""
    ops = _ops()
    g = torch.Generator().manual_seed(2)
    big = (torch.randn((700, 2048), generator=g) * 3).to(dev)
    pieces = [big[:300, :1024], big[300:301, 1024:], torch.zeros((0, 1024), device=dev), big[301:700, 1024:].contiguous(),
              (torch.randn((5, 1024), generator=g)).to(dev)]
    pieces += [big[i:i + 2, :1024] for i in range(0, 140, 2)]         # 70 more segments
    got = ops.cat_rows_cast_bf16(pieces)
    want = torch.cat(pieces, dim=0).to(torch.bfloat16)
    assert got.shape == want.shape and torch.equal(got.view(torch.int16), want.view(torch.int16))


@pytest.mark.parametrize("shape", [(2, 150, 250), (3, 37, 53), (1, 8, 16), (40, 9, 17)])
def test_fused_bottleneck64_ds_bit_equal_to_unfused(dev, shape):
    """ops.bottleneck64_ds (layer1's FIRST block: 64 -> 64 -> 64 (3x3) -> 256 with the 1x1 downsample branch on the residual,
    backbone/resnet.py:266-276,:324-344, in one persistent kernel; the identity branch computed from the x patch in LDS and
    rounded to bf16 before the add) against the four conv2d_nhwc launches it replaces: the same bits."""
    ops = _ops()
    N, H, W = shape
    g = torch.Generator().manual_seed(H * 11 + W)
    x = torch.randn((N, H, W, 64), generator=g).relu().to(torch.bfloat16).to(dev)
    w1 = (torch.randn((64, 1, 1, 64), generator=g) * 0.12).to(torch.bfloat16).to(dev)
    w2 = (torch.randn((64, 3, 3, 64), generator=g) * 0.05).to(torch.bfloat16).to(dev)
    w3 = (torch.randn((256, 1, 1, 64), generator=g) * 0.1).to(torch.bfloat16).to(dev)
    wd = (torch.randn((256, 1, 1, 64), generator=g) * 0.1).to(torch.bfloat16).to(dev)
    sb = [((torch.rand((n,), generator=g) + 0.5).to(dev), (torch.randn((n,), generator=g) * 0.2).to(dev)) for n in (64, 64, 256, 256)]
    ident = ops.conv2d_nhwc(x, wd, sb[3][0], sb[3][1])
    t1 = ops.conv2d_nhwc(x, w1, sb[0][0], sb[0][1], relu=True)
    t2 = ops.conv2d_nhwc(t1, w2, sb[1][0], sb[1][1], pad=1, relu=True)
    ref = ops.conv2d_nhwc(t2, w3, sb[2][0], sb[2][1], residual=ident, relu=True)
    got = ops.bottleneck64_ds(x, w1, sb[0][0], sb[0][1], w2, sb[1][0], sb[1][1], w3, sb[2][0], sb[2][1], wd, sb[3][0], sb[3][1])
    torch.cuda.synchronize()
    nd = (got.view(torch.int16) != ref.view(torch.int16)).sum().item()
    assert nd == 0, "%d of %d elements differ (max |d| %.3g)" % (nd, got.numel(), (got.float() - ref.float()).abs().max().item())


# ------------------------------------------------------------------------------------------------ split-precision planes
SP_CASES = [
    # N, H, W, C, Cout, R, stride, pad, dil, relu, use_res, out_mode
    (2, 38, 63, 256, 1024, 1, 1, 0, 1, True, True, "planes"),     # layer3 conv3 + residual (streaming class)
    (2, 38, 63, 1024, 256, 1, 1, 0, 1, True, False, "planes"),    # layer3 conv1
    (2, 38, 63, 256, 256, 3, 1, 1, 1, True, False, "planes"),     # layer3 conv2
    (1, 75, 125, 256, 128, 1, 2, 0, 1, True, False, "planes"),    # stride-2 1x1, Cout < 256 (column mask)
    (1, 40, 50, 64, 64, 3, 1, 1, 1, True, False, "planes"),       # layer1 conv2: one 64-channel plane per K-tile
    (1, 19, 23, 512, 2048, 1, 1, 0, 1, True, True, "f32"),        # res5 conv3 + residual -> f32 (ROIAlign input)
    (1, 19, 23, 512, 512, 3, 1, 2, 2, True, False, "planes"),     # dilated res5 conv2
    (1, 38, 63, 1024, 1024, 3, 1, 1, 1, True, False, "f32"),      # RPN conv -> f32
    (1, 17, 13, 128, 520, 1, 1, 0, 1, False, True, "planes"),     # Cout % 256 != 0, M tail, no ReLU
]


def _sp_inputs(case):
    N, H, W, C, Cout, R, stride, pad, dil, relu, use_res, out_mode = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn((N, H, W, C), generator=g)
    w = torch.randn((Cout, R, R, C), generator=g) / math.sqrt(C * R * R)
    scale = torch.rand((Cout,), generator=g) + 0.5
    bias = torch.randn((Cout,), generator=g) * 0.1
    Ho = (H + 2 * pad - dil * (R - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (R - 1) - 1) // stride + 1
    res = torch.randn((N, Ho, Wo, Cout), generator=g) if use_res else None
    return x, w, scale, bias, res


@pytest.mark.parametrize("case", SP_CASES)
def test_conv2d_sp_x3_vs_f64(dev, case):
    """The split-precision conv (planes in, [Wh | Wh | Wl] weights, f32 accumulate) against the f64 convolution of the SAME
    f32 tensors: error at f32 level (~2^-16 per product, averaging over K), 100x below the bf16 kernel's."""
    ops = _ops()
    N, H, W, C, Cout, R, stride, pad, dil, relu, use_res, out_mode = case
    x, w, scale, bias, res = _sp_inputs(case)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), stride=stride, padding=pad, dilation=dil)
    ref = ref * scale.double().view(1, -1, 1, 1) + bias.double().view(1, -1, 1, 1)
    if use_res:
        ref = ref + res.permute(0, 3, 1, 2).double()
    if relu:
        ref = ref.clamp(min=0)
    ref = ref.permute(0, 2, 3, 1)
    xp = ops.split_planes(x.to(dev).contiguous())
    assert (xp.float().cpu() - x).abs().max() <= 2.0 ** -16 * x.abs().max()          # the planes hold x to ~2^-17
    rp = ops.split_planes(res.to(dev).contiguous()) if use_res else None
    y = ops.conv2d_sp(xp, ops.split_conv_weight_x3(w).to(dev), scale.to(dev), bias.to(dev), residual=rp, stride=stride,
                      pad=pad, dil=dil, relu=relu, out_mode=out_mode, x3=True)
    got = (y.float() if out_mode == "planes" else y).cpu().double()
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    print("conv2d_sp x3 %s: max err / scale = %.3g" % (case, err))
    assert got.shape == ref.shape and err < 3e-5, err


@pytest.mark.parametrize("case", SP_CASES[:5] + SP_CASES[8:])
def test_conv2d_sp_hi_plane_bit_equal_to_bf16_kernel(dev, case):
    """x3=False (the bf16 mode over a wide residual stream): the hi plane is the bf16 kernel's input, so without a
    residual the rounded output equals the plain bf16 launch bit for bit (out_mode "bf16"; the planes' hi is that value
    too), and with a split residual the sum is taken against hi + lo in f32."""
    ops = _ops()
    N, H, W, C, Cout, R, stride, pad, dil, relu, use_res, out_mode = case
    x, w, scale, bias, res = _sp_inputs(case)
    xp = ops.split_planes(x.to(dev).contiguous())
    wb = w.to(torch.bfloat16).to(dev)
    sc, bi = scale.to(dev), bias.to(dev)
    xh = xp.hi().contiguous()
    plain = ops.conv2d_nhwc(xh, wb, sc, bi, stride=stride, pad=pad, dil=dil, relu=relu)
    y16 = ops.conv2d_sp(xp, wb, sc, bi, stride=stride, pad=pad, dil=dil, relu=relu, out_mode="bf16", x3=False)
    assert torch.equal(y16.view(torch.int16), plain.view(torch.int16))
    yp = ops.conv2d_sp(xp, wb, sc, bi, stride=stride, pad=pad, dil=dil, relu=relu, out_mode="planes", x3=False)
    assert torch.equal(yp.hi().contiguous().view(torch.int16), plain.view(torch.int16))
    if use_res:
        rp = ops.split_planes(res.to(dev).contiguous())
        f32 = ops.conv2d_nhwc(xh, wb, sc, bi, stride=stride, pad=pad, dil=dil, relu=False, out_dtype=torch.float32)
        want = f32 + rp.float()
        if relu:
            want = want.clamp(min=0)
        got = ops.conv2d_sp(xp, wb, sc, bi, residual=rp, stride=stride, pad=pad, dil=dil, relu=relu, out_mode="planes",
                            x3=False).float()
        # (fma(acc, scale, bias) + residual in f32, then the planes hold it to 2^-17)
        assert (got - want).abs().max().item() <= 2.0 ** -15 * want.abs().max().item()


def test_linear_sp_split_k(dev):
    """fc0's shape class (K >= 32768: three K ranges + finalize) through the SP kernels"""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    M, K, Nout = 300, 49 * 1024, 1024
    x = torch.randn((M, K), generator=g)
    w = torch.randn((Nout, K), generator=g) / math.sqrt(K)
    b = torch.randn((Nout,), generator=g) * 0.1
    ref = (x.double() @ w.double().t() + b.double()).clamp(min=0)
    y = ops.linear_sp(ops.split_planes(x.to(dev)), ops.split_conv_weight_x3(w.view(Nout, 1, 1, K)).to(dev).view(Nout, 3 * K),
                      b.to(dev), relu=True).cpu().double()
    err = (y - ref).abs().max().item() / ref.abs().max().item()
    print("linear_sp split-K: max err / scale = %.3g" % err)
    assert err < 3e-5


def test_roi_align_planes_bit_equal_to_split_of_f32(dev):
    """the f32 ROIAlign writing [hi | lo] planes against split_planes of the exact-term-order f32 output: the XCD-sliced hot
    shape (2048 channels) runs the separable form since round 6 (f32 round-off below the planes' own 2^-17), a small unsliced one
    keeps the exact-term-order kernel (bit for bit)"""
    ops = _ops()
    for (B, H, W, C, K) in ((2, 38, 63, 2048, 300), (1, 12, 17, 32, 9)):
        g = torch.Generator().manual_seed(C + K)
        feat = torch.randn((B, H, W, C), generator=g).to(dev)
        rois = _random_rois(g, K, B, W * 16, H * 16).to(dev)
        f32 = ops.roi_align(feat, rois, 1.0 / 16, (7, 7), 0)
        want = ops.split_planes(f32.view(K, -1).contiguous())
        got = ops.roi_align_planes(feat, rois, 1.0 / 16, (7, 7), 0)
        assert got.C == want.C
        if C == 2048:     # the XCD-sliced hot shape runs the SEPARABLE form (round 6): the exact-term-order sums to f32 round-off
            n = got.C
            a = got.t[:, :n].float() + got.t[:, n:].float()
            # (hi + lo carries the planes' own 2^-17 relative rounding; the separable sums' difference is far below it)
            assert (a - f32.view(K, -1)).abs().max().item() <= 2.0 ** -16 * f32.abs().max().item()
            assert (a - f32.view(K, -1)).abs().median().item() <= 2.0 ** -19 * f32.abs().max().item()
        else:             # (small shapes keep the exact-term-order kernel: bit for bit)
            assert torch.equal(got.t.view(torch.int16), want.t.view(torch.int16))


@pytest.mark.parametrize("M", [1875, 300, 37])
def test_x3_weight_linear_and_transposed(dev, M):
    """ops.linear / ops.linear_transposed with an X3Weight (the head's projections in conv_mode "x3") against f64"""
    ops = _ops()
    g = torch.Generator().manual_seed(M)
    K, Nout = 1024, 1024
    x = torch.randn((M, K), generator=g)
    w = torch.randn((Nout, K), generator=g) / math.sqrt(K)
    b = torch.randn((Nout,), generator=g) * 0.1
    xw = ops.X3Weight(w, dev)
    ref = x.double() @ w.double().t() + b.double()
    y = ops.linear(x.to(dev), xw, b.to(dev)).cpu().double()
    assert (y - ref).abs().max().item() / ref.abs().max().item() < 3e-5
    ld = (M + 31) // 32 * 32
    yt = ops.linear_transposed(xw, x.to(dev), ld).cpu().double()
    reft = (x.double() @ w.double().t()).t()
    assert yt.shape == (Nout, ld) and (yt[:, :M] - reft).abs().max().item() / reft.abs().max().item() < 3e-5
    assert (yt[:, M:] == 0).all()


def test_conv_scale_bias_at_unaligned_addresses(dev):
    """The C ABI promises no alignment for the FrozenBN vectors: igemm8's read-out takes 16-byte loads only when both
    pointers are 16-byte aligned and element loads otherwise -- same bits either way (plain and SP kernels)."""
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    N, H, W, C, Cout = 2, 38, 63, 256, 512
    x = torch.randn((N, H, W, C), generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn((Cout, 1, 1, C), generator=g) / 16).to(torch.bfloat16).to(dev)
    sc = (torch.rand((Cout + 8,), generator=g) + 0.5).to(dev)
    bi = (torch.randn((Cout + 8,), generator=g) * 0.1).to(dev)
    a_s, a_b = sc[:Cout].clone(), bi[:Cout].clone()                  # 16-byte aligned copies
    u_s, u_b = sc[1:Cout + 1], bi[3:Cout + 3]                         # 4-byte aligned views
    u_s.copy_(a_s); u_b.copy_(a_b)
    assert u_s.data_ptr() % 16 != 0 and u_b.data_ptr() % 16 != 0 and u_s.is_contiguous()
    ya = ops.conv2d_nhwc(x, w, a_s, a_b, relu=True)
    yu = ops.conv2d_nhwc(x, w, u_s, u_b, relu=True)
    assert torch.equal(ya.view(torch.int16), yu.view(torch.int16))
    xp = ops.split_planes(x.float())
    w3 = ops.split_conv_weight_x3(w.float())
    pa = ops.conv2d_sp(xp, w3, a_s, a_b, relu=True, out_mode="f32")
    pu = ops.conv2d_sp(xp, w3, u_s, u_b, relu=True, out_mode="f32")
    assert torch.equal(pa, pu)


DECONV_CASES = [
    # N, H, W, Cin (padded to the K vector), C, H2, W2 (the skip connection's size), Cs, ksplit
    (2, 5, 8, 1024, 512, 10, 16, 512, 1),        # deconv5 at 600 x 1000 (even target: one row / column cropped)
    (3, 10, 16, 1026, 256, 19, 32, 512, 1),      # deconv4: Cin zero-padded, odd target height
    (2, 7, 9, 770, 128, 13, 17, 256, 3),         # odd x odd target, split-K
    (1, 4, 6, 386, 64, 10, 14, 128, 1),          # target = the full (2H+2) x (2W+2) map: crop_like keeps everything
    (21, 5, 8, 1024, 512, 10, 16, 512, 4),       # the 21-pair window, split-K
    (21, 38, 63, 386, 64, 75, 125, 128, 1),      # deconv2 of a 21-pair key frame: 205 igemm8 tiles -> the LDS-DMA kernel's read-out
    (40, 19, 32, 770, 128, 38, 63, 256, 1),      # igemm8, 192-row tiles, two column tiles
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", DECONV_CASES)
def test_deconv_subpixel_and_level_assemble_vs_torch(dev, case, dtype):
    """ops.deconv4x4s2_into + ops.flow_level_assemble (a FlowNetS refinement level's concatenation,
    mega_core/modeling/backbone/flownet.py:9-13,:40-52,:94-111) against F.conv_transpose2d + crop_like + torch.cat in f32 on the
    rounded operands.  f32: 1e-4 of the output scale (summation order); 16-bit: one output rounding (2^-8 / 2^-10) + the order."""
    ops = _ops()
    N, H, W, Cin, C, H2, W2, Cs, ks = case
    mult = 32 if dtype == torch.float32 else 64
    g = torch.Generator().manual_seed(hash(case) % 997)
    x = torch.randn((N, H, W, Cin), generator=g).to(dtype)
    wt = torch.randn((Cin, C, 4, 4), generator=g) / math.sqrt(Cin * 4.0)
    b = torch.randn((C,), generator=g) * 0.1
    skip = torch.randn((N, H2, W2, Cs), generator=g).to(dtype)
    flow = torch.randn((N, H, W, 2), generator=g).to(dtype)
    wu = torch.randn((2, 2, 4, 4), generator=g) * 0.3
    bu = torch.randn((2,), generator=g) * 0.1
    crop = 0 if (2 * H + 2, 2 * W + 2) == (H2, W2) else 1
    full = F.leaky_relu(F.conv_transpose2d(x.float().permute(0, 3, 1, 2), wt.to(dtype).float(), b, stride=2), 0.1)
    dec = full[:, :, crop:crop + H2, crop:crop + W2].permute(0, 2, 3, 1)
    up = F.conv_transpose2d(flow.float().permute(0, 3, 1, 2), wu, bu, stride=2)[:, :, crop:crop + H2, crop:crop + W2].permute(0, 2, 3, 1)
    cp = (Cin + mult - 1) // mult * mult
    xp = torch.zeros((N, H, W, cp), dtype=dtype)
    xp[..., :Cin] = x
    w4 = ops.pack_deconv4x4s2(wt, dtype, mult)
    ldo = (Cs + C + 2 + mult - 1) // mult * mult
    out = torch.full((N, H2, W2, ldo), float("nan"), dtype=dtype, device=dev)
    ops.deconv4x4s2_into(xp.to(dev), w4.to(dev), b.repeat(4).to(dev), out, Cs, relu=2, ksplit=ks)
    ops.flow_level_assemble(skip.to(dev), flow.to(dev), wu.to(dev), bu.to(dev), out, C)
    got = out.float().cpu()
    assert torch.equal(got[..., :Cs], skip.float()), "skip connection not copied bit for bit"
    assert torch.equal(got[..., Cs + C + 2:], torch.zeros((N, H2, W2, ldo - Cs - C - 2))), "padding channels not zero"
    eps = {torch.float32: 1e-4, torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -9}[dtype]
    for name, a, r in (("deconv", got[..., Cs:Cs + C], dec), ("flow up-sampling", got[..., Cs + C:Cs + C + 2], up)):
        err = (a - r).abs().max().item()
        assert err <= eps * r.abs().max().item(), "%s: max |d| %.3g against scale %.3g" % (name, err, r.abs().max().item())
    if dtype != torch.float32 and ks == 1 and (4 * C) % 256 == 0 and N * (H + 1) * (W + 1) >= 128 * 192:
        # this launch ran on igemm8 (LDS-DMA tiles, ABL = 6 read-out): the register-staged tiles give the same bits
        import os
        os.environ["MEGA_IGEMM_TILE"] = "128x128"
        try:
            out2 = torch.full((N, H2, W2, ldo), float("nan"), dtype=dtype, device=dev)
            ops.deconv4x4s2_into(xp.to(dev), w4.to(dev), b.repeat(4).to(dev), out2, Cs, relu=2, ksplit=1)
        finally:
            del os.environ["MEGA_IGEMM_TILE"]
        assert torch.equal(out2[..., Cs:Cs + C].view(torch.int16), out[..., Cs:Cs + C].view(torch.int16)), "igemm8 vs igemm tiles"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv2d_caller_split_k(dev, dtype):
    """mega_conv2d_nhwc_ks: a small-M / long-K conv with 1, 2, 5 and more K ranges than K-tiles (clamped) against F.conv2d."""
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    x = torch.randn((2, 5, 8, 512), generator=g).to(dtype)
    w = (torch.randn((96, 3, 3, 512), generator=g) / math.sqrt(512 * 9.0)).to(dtype)
    bias = torch.randn((96,), generator=g) * 0.1
    ref = F.leaky_relu(F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), bias, padding=1), 0.1).permute(0, 2, 3, 1)
    eps = 1e-4 if dtype == torch.float32 else 2.0 ** -7
    for ks in (1, 2, 5, 1000):
        y = ops.conv2d_nhwc(x.to(dev), w.to(dev), None, bias.to(dev), pad=1, relu=2, ksplit=ks).float().cpu()
        assert (y - ref).abs().max().item() <= eps * ref.abs().max().item(), "ksplit %d" % ks

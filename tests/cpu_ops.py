"""TEST-ONLY stand-in for mega.pytorch_amd.ops built from the CPU oracle, so that the HOST logic of the
product (weight packing / permutations, the per-video state machine, pool assembly, sharding) can be
exercised by the `-m "not gpu"` suite on a box without a GPU.  It is injected by monkeypatching inside
tests only; the product package never imports it and has no CPU path of its own.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from oracle import mega_oracle as mo
from oracle import native


def conv2d_nhwc(x, w, scale=None, bias=None, residual=None, stride=1, pad=0, dil=1, relu=False, out_dtype=None,
                out=None, ksplit=None):
    y = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), stride=stride, padding=pad, dilation=dil)
    if scale is not None:
        y = y * scale.view(1, -1, 1, 1)
    if bias is not None:
        y = y + bias.view(1, -1, 1, 1)
    y = y.permute(0, 2, 3, 1)
    if residual is not None:
        y = y + residual.float()
    if int(relu) == 1:
        y = F.relu(y)
    elif int(relu) == 2:
        y = F.leaky_relu(y, 0.1)
    return y.contiguous().to(out_dtype or x.dtype)


def _w32(w):
    """an ops.X3Weight as its f32 matrix Wh + Wl"""
    from mega.pytorch_amd import ops
    if isinstance(w, ops.X3Weight):
        K = w.shape[1]
        return w.w3[:, :K].float() + w.w3[:, 2 * K:].float()
    return w


def linear(x, w, bias=None, relu=False, residual=None, out_dtype=None, scale=None, ksplit=None):
    w = _w32(w)
    M, K = x.shape
    y = conv2d_nhwc(x.view(M, 1, 1, K), w.view(w.shape[0], 1, 1, K), scale=scale, bias=bias,
                    residual=None if residual is None else residual.view(M, 1, 1, -1), relu=relu, out_dtype=out_dtype)
    return y.view(M, w.shape[0])


def linear_transposed(w, x, ld, residual=None):
    w = _w32(w)
    out = torch.zeros((w.shape[0], ld), dtype=x.dtype)
    y = (x.float() @ w.float().t()).t()
    if residual is not None:
        y = y + residual.float()[:, :x.shape[0]]
    out[:, :x.shape[0]] = y.to(x.dtype)
    return out


def stem(x_nchw, w_tap64, scale, bias, out_dtype, w_n160=None):
    w = w_tap64.view(3, 7, 7, 64).permute(3, 0, 1, 2)
    y = F.conv2d(x_nchw, w, stride=2, padding=3) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)
    return F.relu(y).permute(0, 2, 3, 1).contiguous().to(out_dtype)


def stem_pool(x, w_n160, scale, bias, mean=None, to_bgr=True):
    raise AssertionError("the CPU twins run float32: the bf16 stem + pool kernel has no twin")


def maxpool3x3s2(x):
    return F.max_pool2d(x.float().permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1).contiguous().to(x.dtype)


def roi_align(feat, rois, spatial_scale, pooled, sampling_ratio, in_nhwc=True, out_nhwc=True):
    f = feat.float().permute(0, 3, 1, 2).contiguous() if in_nhwc else feat.float()
    out = torch.from_numpy(native.roi_align(f.numpy(), rois.numpy(), spatial_scale, pooled[0], pooled[1], sampling_ratio))
    if out_nhwc:
        K, C = out.shape[:2]
        out = out.permute(0, 2, 3, 1).reshape(K, pooled[0] * pooled[1], C)
    return out.contiguous().to(feat.dtype)


def nms(dets, scores, thr, strict_gt=True):
    return torch.from_numpy(native.nms(dets.numpy(), scores.numpy(), thr, strict_gt))


def rpn_select(rpn_out, cell_anchors, Hf, Wf, anchor_stride, pre_nms, post_nms, nms_thresh, min_size, im_w, im_h,
               strict_gt=True, want_index=False, hold=None):
    B = rpn_out.shape[0]
    A = cell_anchors.shape[0]
    anchors = mo.grid_anchors(cell_anchors, Hf, Wf, anchor_stride)
    props = torch.zeros((B, post_nms, 4))
    scores = torch.zeros((B, post_nms))
    cnt = torch.zeros((B,), dtype=torch.int32)
    index = torch.full((B, post_nms), -1, dtype=torch.int32)
    for b in range(B):
        o = rpn_out[b].view(Hf, Wf, 5 * A).permute(2, 0, 1)
        pb, ps, pi = mo.rpn_select(o[:A], o[A:], anchors, im_w, im_h, pre_nms, post_nms, nms_thresh, min_size, strict_gt,
                                   want_index=True)
        n = pb.shape[0]
        props[b, :n], scores[b, :n], cnt[b], index[b, :n] = pb, ps, n, pi.to(torch.int32)
    return (props, scores, cnt, index) if want_index else (props, scores, cnt)


def postprocess(logits, deltas, props, nprop, weights, im_w, im_h, score_thresh, nms_thresh, max_det,
                strict_gt=True, want_probs=False):
    cfg = mo.OracleCfg(score_thresh=score_thresh, nms=nms_thresh, detections_per_img=max_det,
                       bbox_reg_weights=tuple(weights), nms_strict_gt=strict_gt, num_classes=logits.shape[1])
    if nprop is not None:        # rows past the (device-side) proposal count are not proposals
        n_live = min(int(nprop.reshape(-1)[0]), logits.shape[0])
        b, s, l = mo.postprocess(logits[:n_live], deltas[:n_live], props[:n_live], im_w, im_h, cfg)
    else:
        b, s, l = mo.postprocess(logits, deltas, props, im_w, im_h, cfg)
    cap = (logits.shape[1] - 1) * logits.shape[0]
    ob, os_, ol = torch.zeros((cap, 4)), torch.zeros((cap,)), torch.zeros((cap,), dtype=torch.int64)
    n = b.shape[0]
    ob[:n], os_[:n], ol[:n] = b, s, l
    return ob, os_, ol, torch.tensor([n], dtype=torch.int32)


def postprocess_batched(logits, deltas, props, B, weights, im_w, im_h, score_thresh, nms_thresh, max_det, strict_gt=True,
                        nprop=None):
    R = logits.shape[0] // B
    res = [postprocess(logits[b * R:(b + 1) * R], deltas[b * R:(b + 1) * R], props[b * R:(b + 1) * R],
                       None if nprop is None else nprop[b:b + 1], weights,
                       im_w, im_h, score_thresh, nms_thresh, max_det, strict_gt) for b in range(B)]
    return tuple(torch.stack([r[i] for r in res]) for i in range(3)) + (torch.cat([r[3] for r in res]),)


def position_logits(rois_q, rois_k, wg_t, bg, dim_mat, precise=True, tiled=False):   # the twin always returns f32 rows
    pe = mo.cal_position_embedding(rois_q, rois_k)
    w = wg_t.t().contiguous().view(16, 64, 1, 1)
    out = (F.relu(F.conv2d(pe, w, bg)) + 1e-6).log()[0]
    ldp = (rois_k.shape[0] + 31) // 32 * 32
    full = torch.zeros((16, rois_q.shape[0], ldp))
    full[:, :, :rois_k.shape[0]] = out
    return full


def relation_attention(q, k, vt, Nk, pos=None, resid=None, bias_v=None, groups=16):
    Nq = q.shape[0]
    qh = q.float().view(Nq, groups, 64).permute(1, 0, 2)
    kh = k.float().view(Nk, groups, 64).permute(1, 0, 2)
    s = torch.bmm(qh, kh.transpose(1, 2)) / math.sqrt(64.0)
    if pos is not None:
        s = s + pos[:, :, :Nk]
    if q.dtype in (torch.bfloat16, torch.float16):   # the kernel's 16-bit modes: P rounded for the PV MFMA, row sum over the rounded values
        e = torch.exp(s - s.max(dim=2, keepdim=True).values).to(q.dtype).float()
        p = e / e.sum(dim=2, keepdim=True)
    else:
        p = F.softmax(s, dim=2)
    v = vt.float()[:, :Nk].view(groups, 64, Nk)
    o = torch.bmm(p, v.transpose(1, 2)).permute(1, 0, 2).reshape(Nq, groups * 64)
    if bias_v is not None:
        o = o + bias_v
    if resid is not None:
        o = o + resid.float()
    return o.to(q.dtype if resid is None else resid.dtype)     # (f32 activation stream over bf16 operands: io_f32)


def cast_bf16(x):
    return x.to(torch.bfloat16)


def cast_half(x, dtype):
    return x.to(dtype)


def cat_rows_cast_bf16(pieces, dtype=torch.bfloat16):
    return torch.cat([p for p in pieces if p.shape[0] > 0], dim=0).to(dtype)


def split_bf16x3(x):
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return torch.cat([hi, lo, hi], dim=1).contiguous()


def split_weight_bf16x3(w32):
    wh = w32.to(torch.bfloat16)
    wl = (w32 - wh.float()).to(torch.bfloat16)
    return torch.cat([wh, wh, wl], dim=1).contiguous()


def position_logits_batched(rois_qs, rois_ks, wg_t, bg, dim_mat, precise=True, tiled=False):
    return [position_logits(a, b, wg_t, bg, dim_mat) for a, b in zip(rois_qs, rois_ks)]


def relation_attention_batched(items, groups=16):
    outs = []
    for it in items:
        k, vt = it["k"], it["vt"]
        if it.get("k2") is not None:        # second key segment: keys N1 .. Nk-1 (the kernel reads both in place)
            N1 = it["N1"]
            k = torch.cat([k[:N1], it["k2"]], dim=0)
            vt = torch.cat([vt[:, :N1], it["vt2"][:, :it["Nk"] - N1]], dim=1)
        outs.append(relation_attention(it["q"], k, vt, it["Nk"], pos=it.get("pos"), resid=it.get("resid"),
                                       bias_v=it.get("bias_v"), groups=groups))
    return outs


import contextlib


@contextlib.contextmanager
def launch_on(stream):
    yield


def profiling():
    return False


def multi_cat(groups):
    return [torch.cat(list(p), dim=d) for p, d in groups]


def copy_blocks(pairs):
    for d, s_ in pairs:
        d.copy_(s_)


def preprocess_frames(frames_u8, mean, to_bgr=True, out=None):
    x = frames_u8.permute(0, 3, 1, 2).float() / 255.0
    if to_bgr:
        x = x[:, [2, 1, 0]] * 255.0
    x = x - torch.tensor(mean).view(1, 3, 1, 1)
    if out is not None:
        out.copy_(x)
        return out
    return x


def avgpool2x2_ceil(x):
    return F.avg_pool2d(x.float().permute(0, 3, 1, 2), 2, 2, ceil_mode=True).permute(0, 2, 3, 1).contiguous().to(x.dtype)


def fgfa_pair_taps(refs, cur=None, order=None, dtype=torch.bfloat16):
    """the arithmetic of pair_taps_kernel: every f32 pixel rounded to `dtype`, the in-bounds taps of a 2 x 2 window summed in
    f32 in (dy, dx) order and divided by their count, rounded once; the seven horizontal taps of a pooled pixel in 8-channel
    groups of its 64-channel row, three zero rows above and below"""
    T, _, H, W = refs.shape
    if cur is None:
        cur = refs[int(order[0]):int(order[0]) + 1]
    pair = torch.cat([cur.expand(T, -1, -1, -1), refs], dim=1).to(dtype).float()       # [T,6,H,W]
    Hp, Wp = (H + 1) // 2, (W + 1) // 2
    acc = torch.zeros((T, 6, Hp, Wp))
    cnt = torch.zeros((1, 1, Hp, Wp))
    for dy in range(2):
        for dx in range(2):
            v = pair[:, :, dy::2, dx::2]
            acc[:, :, :v.shape[2], :v.shape[3]] = acc[:, :, :v.shape[2], :v.shape[3]] + v
            cnt[:, :, :v.shape[2], :v.shape[3]] += 1
    pooled = (acc / cnt).to(dtype)                                                     # [T,6,Hp,Wp]
    out = torch.zeros((T, Hp + 6, Wp, 64), dtype=dtype)
    for s in range(7):
        lo, hi = max(0, 3 - s), min(Wp, Wp + 3 - s)                                    # w with 0 <= w - 3 + s < Wp
        out[:, 3:3 + Hp, lo:hi, s * 8:s * 8 + 6] = pooled[:, :, :, lo - 3 + s:hi - 3 + s].permute(0, 2, 3, 1)
    return out


def fgfa_warp_aggregate(feats, flow, Cf, key, want_weights=False, order=None, flow_pos=None):
    if order is not None:          # ring form: window position t lives in slot order[1 + t], the key frame in order[0]
        slots = order[1:].long()
        key = int((slots == int(order[0])).nonzero()[0]) if flow_pos is None else int(flow_pos)
        feats = feats.index_select(0, slots)
        if flow_pos is None:       # (flow by slot; with flow_pos it already is in window order)
            flow = flow.index_select(0, slots)
    out, w = mo.fgfa_aggregate(feats.float().permute(0, 3, 1, 2), flow, key, nfeat=Cf)
    out = out[0].permute(1, 2, 0).contiguous().to(feats.dtype)
    return (out, w[:, 0]) if want_weights else out


# ---- split-precision planes (conv_mode "x3" / "wide"): the twins compute on hi + lo (or hi) in f32 and re-split
def _planes(x32, dtype=torch.bfloat16):
    from mega.pytorch_amd import ops
    hi = x32.to(dtype)
    lo = (x32 - hi.float()).to(dtype)
    return ops.Planes(torch.cat([hi, lo], dim=-1).contiguous(), x32.shape[-1])


def split_planes(x, dtype=torch.bfloat16):
    return _planes(x.float(), dtype)


def split_conv_weight_h2(w_ohwi):
    w = w_ohwi.float().to(torch.float16)
    return torch.cat([w, w], dim=-1).contiguous()


def conv2d_sp(x, w, scale=None, bias=None, residual=None, stride=1, pad=0, dil=1, relu=False, out_mode="planes", x3=True,
              out=None):
    from mega.pytorch_amd import ops
    if isinstance(x, ops.Planes):
        C = x.C
        xin = x.float() if x3 else x.hi().float()
    else:
        C, xin = x.shape[-1], x.float()
    h2 = isinstance(x3, str) and x3 == "h2"
    pdt = torch.float16 if h2 else torch.bfloat16
    if h2:
        w32 = w[..., :C].float()                                                    # [W | W] -> W (rounded to fp16 once)
        assert w.shape[-1] == 2 * C
    else:
        w32 = (w[..., :C].float() + w[..., 2 * C:].float()) if x3 else w.float()    # [Wh | Wh | Wl] -> Wh + Wl
    assert w32.shape[-1] == C
    y = conv2d_nhwc(xin, w32, scale, bias, None if residual is None else residual.float(), stride, pad, dil, relu,
                    out_dtype=torch.float32)
    if out is not None:
        out[:, :y.shape[-1]] = y.reshape(out.shape[0], -1).to(out.dtype)
        return out
    return _planes(y, pdt) if out_mode == "planes" else y.to(torch.float32 if out_mode == "f32" else pdt)


def linear_sp(x, w3, bias=None, relu=False):
    M, K = x.shape
    w32 = w3[:, :K].float() + w3[:, 2 * K:].float() if w3.shape[1] == 3 * K else w3[:, :K].float()
    y = x.float() @ w32.t()
    if bias is not None:
        y = y + bias
    return F.relu(y) if relu else y


def roi_align_planes(feat, rois, spatial_scale, pooled, sampling_ratio, dtype=torch.bfloat16):
    y = roi_align(feat, rois, spatial_scale, pooled, sampling_ratio)
    return _planes(y.reshape(y.shape[0], -1).float(), dtype)


ALL = ["split_planes", "split_conv_weight_h2", "conv2d_sp", "linear_sp", "roi_align_planes", "cast_bf16", "cast_half", "cat_rows_cast_bf16", "split_bf16x3", "split_weight_bf16x3", "multi_cat", "copy_blocks", "pack_stem_weight_bf16", "dff_warp_scale", "resize_bilinear_u8", "avgpool2x2_ceil", "fgfa_pair_taps", "fgfa_warp_aggregate", "conv2d_nhwc", "linear", "linear_transposed", "stem", "maxpool3x3s2", "roi_align", "nms", "rpn_select",
       "postprocess", "position_logits", "relation_attention", "preprocess_frames", "position_logits_batched",
       "relation_attention_batched", "postprocess_batched"]


def dff_warp_scale(feats, flow, scale):
    f = feats.float().permute(2, 0, 1)[None]
    grid = mo.fgfa_get_grid(flow[None])
    w = F.grid_sample(f, grid, mode="bilinear", padding_mode="border", align_corners=False)
    return (w[0].permute(1, 2, 0) * scale.float()).to(feats.dtype).contiguous()


def resize_bilinear_u8(frames_u8, out_hw, tables=None):
    from oracle import pil_resize
    a = frames_u8.numpy()
    return torch.from_numpy(np.stack([pil_resize.resize_bilinear_u8(f, out_hw[0], out_hw[1]) for f in a]))


def pack_stem_weight_bf16(w_oihw, dtype=torch.bfloat16):
    return torch.zeros((64, 176), dtype=dtype)      # (the twin's stem() reads the f32 taps; a placeholder operand)


def install(monkeypatch):
    """Replace every kernel wrapper of mega.pytorch_amd.ops with its oracle-backed CPU twin."""
    from mega.pytorch_amd import ops
    g = globals()
    for name in ALL:
        monkeypatch.setattr(ops, name, g[name])


def deconv4x4s2_into(x, w4, bias4, out, coff, relu=0, ksplit=1):
    """the sub-pixel GEMM's arithmetic: a 2 x 2 / pad 1 conv with (a, b, co) output columns, scattered to (2m + a - crop, 2n + b - crop)"""
    N, H, W, _ = x.shape
    C = w4.shape[0] // 4
    H2, W2 = out.shape[1:3]
    crop = 0 if (2 * H + 2, 2 * W + 2) == (H2, W2) else 1
    y = conv2d_nhwc(x, w4, None, bias4, pad=1, relu=relu, out_dtype=torch.float32)          # [N,H+1,W+1,4C]
    full = y.view(N, H + 1, W + 1, 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(N, 2 * H + 2, 2 * W + 2, C)
    out[..., coff:coff + C] = full[:, crop:crop + H2, crop:crop + W2].to(out.dtype)
    return out


def flow_level_assemble(skip, flow, w_up, b_up, out, C):
    N, H2, W2, Cs = skip.shape
    h, w = flow.shape[1:3]
    crop = 0 if (2 * h + 2, 2 * w + 2) == (H2, W2) else 1
    up = F.conv_transpose2d(flow.float().permute(0, 3, 1, 2), w_up, b_up, stride=2).permute(0, 2, 3, 1)
    out[..., :Cs] = skip
    out[..., Cs + C:Cs + C + 2] = up[:, crop:crop + H2, crop:crop + W2].to(out.dtype)
    out[..., Cs + C + 2:] = 0
    return out


def flow_conv1_combine(ab, bias, dtype, key=None, order=None, T=None, out=None, nwin=0):
    S = ab.shape[0]
    if nwin > 0:
        ys = []
        for g in range(order.shape[0]):
            slots = order[g, 1:].long()
            ys.append(ab[int(order[g, 0]):int(order[g, 0]) + 1, :, :, :64] + ab.index_select(0, slots)[:, :, :, 64:])
        y = torch.cat(ys, dim=0) + bias.view(1, 1, 1, 64)
    else:
        T = S if T is None else T
        ks = int(order[0]) if order is not None else int(key)
        y = ab[ks:ks + 1, :, :, :64] + ab[:T, :, :, 64:] + bias.view(1, 1, 1, 64)
    y = F.leaky_relu(y, 0.1).to(dtype)
    if out is not None:
        out.copy_(y)
        return out
    return y


def fgfa_warp_aggregate_group(feats, flow, Cf, orders, key_pos):
    G, T = orders.shape[0], orders.shape[1] - 1
    return torch.stack([fgfa_warp_aggregate(feats, flow[g * T:(g + 1) * T], Cf, 0, order=orders[g], flow_pos=key_pos) for g in range(G)], 0)


def flow_pred_finish(z, bias, scale, out_dtype):
    N, H, W, _ = z.shape
    zp = F.pad(z, (0, 0, 1, 1, 1, 1))
    acc = torch.zeros((N, H, W, 2))
    for r in range(3):
        for s in range(3):
            acc = acc + zp[:, r:r + H, s:s + W, (r * 3 + s) * 2:(r * 3 + s) * 2 + 2]
    return (acc * scale + bias.view(1, 1, 1, 2)).to(out_dtype)


ALL += ["deconv4x4s2_into", "flow_level_assemble", "flow_pred_finish", "flow_conv1_combine", "fgfa_warp_aggregate_group"]

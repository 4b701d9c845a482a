"""Tensor-level wrappers over the C ABI (include/mega_hip.h).

torch is used here only for device memory and the current HIP stream; every computation is a
hand-written gfx950 kernel in libmega_hip.so.  All wrappers raise if the library is missing, a
tensor is not on a HIP device, or a kernel call returns non-zero -- there is no CPU fallback.
"""
import contextlib
import ctypes
import threading
import math

import torch

from . import _lib

F32, BF16, F16 = 0, 1, 2
_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}
_HALF = (torch.bfloat16, torch.float16)      # 16-bit matrix-core operand types


def _dt(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError("unsupported dtype %s (float32 / bfloat16 / float16 only)" % t.dtype)


_LAUNCH = threading.local()


def _stream():
    s = getattr(_LAUNCH, "s", None)
    return torch.cuda.current_stream().cuda_stream if s is None else s.cuda_stream


@contextlib.contextmanager
def launch_on(stream):
    """Inside the block this module's kernels are launched on `stream` (a torch.cuda.Stream; None = no change) while
    tensors are still ALLOCATED under torch's current stream: the way to run an op beside the current stream's work without
    handing the caching allocator blocks that belong to another stream (the caller orders the two streams with events
    and keeps the op's inputs alive until the streams have joined)."""
    prev = getattr(_LAUNCH, "s", None)
    _LAUNCH.s = stream if stream is not None else prev
    try:
        yield
    finally:
        _LAUNCH.s = prev


def profiling():
    return _PROF is not None


def _gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("mega.pytorch_amd ops need HIP device tensors (no CPU path)")


def _ptr(t):
    return None if t is None else t.data_ptr()


def _ws(nbytes, device):
    return torch.empty(int(nbytes), dtype=torch.uint8, device=device)


class Profiler(object):
    """Per-kernel-family GPU time from HIP event pairs recorded on the launch stream (torch's current stream,
    the one every wrapper launches on), with the algorithmic FLOPs / bytes each launch stands for.  Used by
    bench.py for the roofline line; never enabled in the timed region."""

    def __init__(self):
        self.items = []

    def begin(self, family, flops=0.0, nbytes=0.0, detail=None):
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        return [family, float(flops), float(nbytes), e0, detail]

    def end(self, tok):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        tok.append(e1)
        self.items.append(tok)

    def summary(self, by_detail=False):
        """{family: launches, ms, flops, bytes}; by_detail=True keys on (family, detail) -- detail = the GEMM shape
        "M x Cout x K (RxS)" of a conv / linear launch -- so that one layer can be told from another of the same tile."""
        torch.cuda.synchronize()
        out = {}
        for fam, fl, nb, e0, det, e1 in self.items:
            d = out.setdefault((fam, det) if by_detail else fam, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += fl
            d["bytes"] += nb
        return out


_PROF = None


def set_profiler(p):
    global _PROF
    _PROF = p


def _pb(family, flops=0.0, nbytes=0.0, detail=None):
    return None if _PROF is None else _PROF.begin(family, flops, nbytes, detail)


def _pe(tok):
    if tok is not None:
        _PROF.end(tok)


# ------------------------------------------------------------------------------------------------ conv / linear
def _igemm_family(lib, M, Cout, K, dtype, shape=None):
    """name of the kernel instantiation a conv / linear is dispatched to (profiler families).  shape = (N, H, W, Cin, R,
    S, stride, pad, dil, ldo, has_residual, out dtype): the layer itself, asked through the launch path's own predicates;
    without it the (M, Cout, K) GEMM shape of a 1x1 layer / linear."""
    if shape is not None:
        N, H, W, Cin, R, S, stride, pad, dil, ldo, has_res, odt = shape
        t = lib.mega_conv2d_nhwc_plan_ex(N, H, W, Cin, Cout, R, S, stride, pad, dil, ldo, int(has_res), _DT[dtype], _DT[odt])
    else:
        t = lib.mega_conv2d_nhwc_plan(M, Cout, K, _DT[dtype])
    kind, t = t // 1000000, t % 1000000
    if kind == 6:
        return "conv64_bf16_3x3"
    # one family = one rocprofv3 symbol: igemm8_kernel<OT, ...> is instantiated per OUTPUT type, so a bf16 launch that writes
    # f32 (fc0's split-K partial sums, the RPN head's f32 logits) is a different symbol from the bf16-output launches of the
    # same tile ("_f32out")
    f32out = shape is not None and dtype in _HALF and (shape[-1] == torch.float32 or (kind in (2, 3, 4, 7, 8) and lib.mega_conv2d_nhwc_workspace_bytes(M, Cout, K) > 0))
    return "igemm%s_%s_%dx%d%s" % ({8: "8", 7: "8s", 4: "4", 3: "4s", 2: "2", 1: "s"}.get(kind, ""), {torch.bfloat16: "bf16", torch.float16: "f16"}.get(dtype, "f32"),
                                   t // 1000, t % 1000, "_f32out" if f32out else "")


def conv2d_nhwc(x, w, scale=None, bias=None, residual=None, stride=1, pad=0, dil=1, relu=False, out_dtype=None,
                out=None, ksplit=None):
    """x [N,H,W,Cin] (contiguous), w [Cout,R,S,Cin] -> [N,Ho,Wo,Cout].  y = act(conv*scale + bias (+res));
    relu: False/0 none, True/1 ReLU, 2 LeakyReLU(0.1).  ksplit (int >= 1): the caller's split-K count
    (mega_conv2d_nhwc_ks: small-M / long-K layers of FlowNetS and the FGFA box head); None: the library's K-only rule."""
    _gpu(x, w, scale, bias, residual)
    lib = _lib.load()
    N, H, W, Cin = x.shape
    Cout, R, S, Cin2 = w.shape
    assert Cin == Cin2 and x.is_contiguous() and w.is_contiguous() and x.dtype == w.dtype
    Ho = (H + 2 * pad - dil * (R - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (S - 1) - 1) // stride + 1
    odt = x.dtype if out_dtype is None else out_dtype
    if out is None:
        out = torch.empty((N, Ho, Wo, Cout), dtype=odt, device=x.device)
    else:
        assert out.dtype == odt and out.shape == (N, Ho, Wo, Cout) and out.is_contiguous()
    if residual is not None:
        assert residual.shape == out.shape and residual.dtype == x.dtype and residual.is_contiguous()
    for v in (scale, bias):
        assert v is None or (v.dtype == torch.float32 and v.numel() == Cout and v.is_contiguous())
    _tok = None
    if _PROF is not None:      # family = the kernel symbol rocprofv3 would report for this launch
        _tok = _pb(_igemm_family(lib, N * Ho * Wo, Cout, R * S * Cin, x.dtype,
                                 (N, H, W, Cin, R, S, stride, pad, dil, Cout, residual is not None, odt)),
                   2.0 * N * Ho * Wo * Cout * R * S * Cin,
                   x.numel() * x.element_size() + w.numel() * w.element_size() + out.numel() * out.element_size()
                   + (0 if residual is None else residual.numel() * residual.element_size()),
                   detail="%dx%dx%d (%dx%d)" % (N * Ho * Wo, Cout, R * S * Cin, R, S))
    if x.numel() * x.element_size() >= 0x7FF00000 or w.numel() * w.element_size() >= 0x7FF00000:
        raise ValueError("conv2d_nhwc: operand of %.2f GiB; the kernels use 32-bit buffer offsets (< 2 GiB per operand): "
                         "lower the frame-stage batch (ClipEngine steps_per_batch) or split the call over M"
                         % (max(x.numel() * x.element_size(), w.numel() * w.element_size()) / 2.0 ** 30))
    if ksplit is not None:
        nb = lib.mega_conv2d_nhwc_ks_workspace_bytes(N * Ho * Wo, Cout, R * S * Cin, _dt(x), int(ksplit))
        ws = _ws(nb, x.device) if nb else None
        rc = lib.mega_conv2d_nhwc_ks(_ptr(x), _ptr(w), _ptr(scale), _ptr(bias), _ptr(residual), _ptr(out), N, H, W, Cin,
                                     Cout, R, S, stride, pad, dil, int(relu), Cout, Cout, _dt(x), _DT[odt], int(ksplit),
                                     _ptr(ws), nb, _stream())
        _pe(_tok)
        _lib.check(rc, "mega_conv2d_nhwc_ks")
        return out
    nb = lib.mega_conv2d_nhwc_workspace_bytes(N * Ho * Wo, Cout, R * S * Cin)     # > 0: long-K layer, split-K
    ws = _ws(nb, x.device) if nb else None
    rc = lib.mega_conv2d_nhwc_ws(_ptr(x), _ptr(w), _ptr(scale), _ptr(bias), _ptr(residual), _ptr(out), N, H, W, Cin,
                                 Cout, R, S, stride, pad, dil, int(relu), Cout, Cout, _dt(x), _DT[odt], _ptr(ws), nb,
                                 _stream())
    _pe(_tok)
    _lib.check(rc, "mega_conv2d_nhwc")
    return out


def pack_deconv4x4s2(wt, dtype, cin_mult):
    """nn.ConvTranspose2d(Cin, C, 4, stride=2).weight [Cin,C,4,4] -> the sub-pixel conv's operand [4 C, 2, 2, Cin padded to
    cin_mult]: w4[(a*2 + b)*C + co, r, s, ci] = wt[ci, co, a + 2 (1 - r), b + 2 (1 - s)]  (mega_conv2d_nhwc_subpixel)."""
    Cin, C = wt.shape[:2]
    cp = (Cin + cin_mult - 1) // cin_mult * cin_mult
    w4 = torch.zeros((2, 2, C, 2, 2, cp), dtype=torch.float32)
    wf = wt.detach().float().cpu()
    for a in range(2):
        for b in range(2):
            for r in range(2):
                for s_ in range(2):
                    w4[a, b, :, r, s_, :Cin] = wf[:, :, a + 2 * (1 - r), b + 2 * (1 - s_)].t()
    return w4.view(4 * C, 2, 2, cp).to(dtype).contiguous()


def deconv4x4s2_into(x, w4, bias4, out, coff, relu=0, ksplit=1):
    """crop_like(act(ConvTranspose2d(Cin, C, 4, stride=2)(x))) written to out[..., coff:coff + C] (flownet.py:9-13,:40-52,:94-111):
    x [N,H,W,Cin] NHWC, w4 = pack_deconv4x4s2(weight), bias4 f32 [4 C] (the bias repeated four times) or None, out
    [N,H2,W2,ldo] contiguous (the level's concatenation buffer); crop_like's rule: the full map is (2H+2) x (2W+2); when its
    size differs from (H2, W2) one row / column is dropped at the top / left."""
    _gpu(x, w4, bias4, out)
    lib = _lib.load()
    N, H, W, Cin = x.shape
    C = w4.shape[0] // 4
    assert tuple(w4.shape) == (4 * C, 2, 2, Cin) and x.is_contiguous() and w4.is_contiguous() and out.is_contiguous()
    assert x.dtype == w4.dtype == out.dtype and out.shape[0] == N
    H2, W2, ldo = out.shape[1:]
    crop = 0 if (2 * H + 2, 2 * W + 2) == (H2, W2) else 1
    assert H2 + crop <= 2 * H + 2 and W2 + crop <= 2 * W + 2
    M = N * (H + 1) * (W + 1)
    _tok = _pb("deconv_subpixel", 2.0 * M * 4 * C * 4 * Cin, (x.numel() + w4.numel() + N * H2 * W2 * C) * x.element_size(),
               detail="%dx%dx%d (2x2 sub-pixel)" % (M, 4 * C, 4 * Cin))
    nb = lib.mega_conv2d_nhwc_ks_workspace_bytes(M, 4 * C, 4 * Cin, _dt(x), int(ksplit))
    ws = _ws(nb, x.device) if nb else None
    rc = lib.mega_conv2d_nhwc_subpixel(_ptr(x), _ptr(w4), _ptr(bias4), _ptr(out), N, H, W, Cin, C, int(relu), H2, W2, crop, ldo,
                                       coff, _dt(x), int(ksplit), _ptr(ws), nb, _stream())
    _pe(_tok)
    _lib.check(rc, "mega_conv2d_nhwc_subpixel")
    return out


def flow_conv1_combine(ab, bias, dtype, key=None, order=None, T=None, out=None, nwin=0):
    """FlowNetS flow_conv1 of the pairs (key frame, frame t) from the per-frame halves (mega_flow_conv1_combine): ab f32
    [S,h,w,128] = [A | B] of S frames, bias f32 [64] -> `dtype` [T,h,w,64] = leaky(A[key] + B[t] + bias), t = 0 .. T-1 (T = S by
    default).  The key frame's slot is `key` (host int) or order[0] (device i32: the engine's ring).
    nwin > 0: order i32 [G, 1 + nwin] = rows [key slot, slot of window position 0 .. nwin-1]: -> [G * nwin,h,w,64], pair
    g * nwin + t = (key frame of row g, the frame at window position t) -- exactly the pairs of G key frames, in window order."""
    _gpu(ab, bias, order)
    lib = _lib.load()
    S, h, w, c = ab.shape
    if nwin > 0:
        assert order is not None and order.dim() == 2 and order.shape[1] == nwin + 1
        T = order.shape[0] * nwin
    else:
        T = S if T is None else T
        assert T <= S
    assert c == 128 and ab.dtype == torch.float32 and ab.is_contiguous() and bias.dtype == torch.float32 and bias.numel() == 64
    assert dtype in _HALF and (key is not None or order is not None)
    assert order is None or (order.dtype == torch.int32 and order.is_contiguous())
    if out is None:
        out = torch.empty((T, h, w, 64), dtype=dtype, device=ab.device)
    assert out.dtype == dtype and tuple(out.shape) == (T, h, w, 64) and out.is_contiguous()
    _tok = _pb("flow_conv1_combine", 0.0, T * h * w * 64 * 6.0)
    rc = lib.mega_flow_conv1_combine(_ptr(ab), _ptr(bias), _ptr(order), -1 if key is None else int(key), _ptr(out), T, h * w,
                                     int(nwin), _DT[dtype], _stream())
    _pe(_tok)
    _lib.check(rc, "mega_flow_conv1_combine")
    return out


def flow_pred_finish(z, bias, scale, out_dtype):
    """second half of a FlowNetS flow prediction (mega_flow_pred_finish): z f32 [N,H,W,ldz] = the 1 x 1 conv of the level's map
    with the 18 (tap, channel) columns of Conv2d(Cin, 2, 3, padding=1) -> [N,H,W,2] = (sum of the shifted taps) * scale + bias."""
    _gpu(z, bias)
    lib = _lib.load()
    N, H, W, ldz = z.shape
    assert z.dtype == torch.float32 and z.is_contiguous() and bias.dtype == torch.float32 and bias.numel() == 2
    out = torch.empty((N, H, W, 2), dtype=out_dtype, device=z.device)
    _tok = _pb("flow_pred_finish", 0.0, N * H * W * (18 * 4.0 + 2 * out.element_size()))
    rc = lib.mega_flow_pred_finish(_ptr(z), ldz, _ptr(bias), float(scale), _ptr(out), N, H, W, _DT[out_dtype], _stream())
    _pe(_tok)
    _lib.check(rc, "mega_flow_pred_finish")
    return out


def flow_level_assemble(skip, flow, w_up, b_up, out, C):
    """the rest of a FlowNetS refinement level's concatenation (mega_flow_level_assemble): out[..., :Cs] = skip,
    out[..., Cs+C:Cs+C+2] = crop_like(ConvTranspose2d(2, 2, 4, stride=2)(flow)) from the f32 weights w_up [2,2,4,4] / b_up [2],
    zeros up to out's channel stride; out[..., Cs:Cs+C] (the deconvolution's slice) is left alone."""
    _gpu(skip, flow, w_up, b_up, out)
    lib = _lib.load()
    N, H2, W2, Cs = skip.shape
    _, h, w, two = flow.shape
    assert two == 2 and out.shape[:3] == skip.shape[:3] and flow.shape[0] == N
    assert skip.is_contiguous() and flow.is_contiguous() and out.is_contiguous() and skip.dtype == flow.dtype == out.dtype
    assert w_up.dtype == torch.float32 and tuple(w_up.shape) == (2, 2, 4, 4) and w_up.is_contiguous() and b_up.dtype == torch.float32
    crop = 0 if (2 * h + 2, 2 * w + 2) == (H2, W2) else 1
    _tok = _pb("flow_level_assemble", 0.0, (skip.numel() + N * H2 * W2 * (out.shape[3] - C)) * skip.element_size())
    rc = lib.mega_flow_level_assemble(_ptr(skip), _ptr(flow), _ptr(w_up), _ptr(b_up), _ptr(out), N, H2, W2, Cs, C, out.shape[3],
                                      h, w, crop, _dt(skip), _stream())
    _pe(_tok)
    _lib.check(rc, "mega_flow_level_assemble")
    return out


def bottleneck64(x, w1, s1, b1, w2, s2, b2, w3, s3, b3):
    """The fused identity bottleneck 256 -> 64 -> 64 (3x3) -> 256 (layer1 blocks 1, 2): x NHWC bf16 [N,H,W,256] ->
    relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1 x))))))) + x) in one persistent kernel; same bits as the three conv2d_nhwc
    launches.  w1 [64,1,1,256], w2 [64,3,3,64], w3 [256,1,1,64] (OHWI bf16), s / b f32 FrozenBN scale / bias."""
    _gpu(x, w1, s1, b1, w2, s2, b2, w3, s3, b3)
    lib = _lib.load()
    N, H, W, C = x.shape
    assert C == 256 and x.dtype in _HALF and x.is_contiguous()
    assert tuple(w1.shape) == (64, 1, 1, 256) and tuple(w2.shape) == (64, 3, 3, 64) and tuple(w3.shape) == (256, 1, 1, 64)
    for t in (w1, w2, w3):
        assert t.dtype == x.dtype and t.is_contiguous()
    for v, n in ((s1, 64), (b1, 64), (s2, 64), (b2, 64), (s3, 256), (b3, 256)):
        assert v.dtype == torch.float32 and v.numel() == n and v.is_contiguous()
    if x.numel() * 2 >= 0x7FF00000:
        raise ValueError("bottleneck64: operand of %.2f GiB; 32-bit buffer offsets (< 2 GiB)" % (x.numel() * 2 / 2.0 ** 30))
    out = torch.empty_like(x)
    px = float(N * H * W)
    _tok = _pb("bneck64_fused", 2.0 * px * (256 * 64 + 576 * 64 + 64 * 256), px * 1024.0)
    rc = lib.mega_bottleneck64_fwd_dt(_ptr(x), _ptr(w1), _ptr(s1), _ptr(b1), _ptr(w2), _ptr(s2), _ptr(b2), _ptr(w3), _ptr(s3),
                                      _ptr(b3), _ptr(out), N, H, W, _dt(x), _stream())
    _pe(_tok)
    _lib.check(rc, "mega_bottleneck64_fwd_dt")
    return out


def bottleneck64_ds(x, w1, s1, b1, w2, s2, b2, w3, s3, b3, wd, sd, bd):
    """The fused FIRST block of layer1 (64 -> 64 -> 64 (3x3) -> 256 with the 1x1 downsample branch): x NHWC bf16 [N,H,W,64] ->
    relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1 x))))))) + bnd(convd(x))) [N,H,W,256]; same bits as the four conv2d_nhwc
    launches.  w1 [64,1,1,64], w2 [64,3,3,64], w3 / wd [256,1,1,64]."""
    _gpu(x, w1, s1, b1, w2, s2, b2, w3, s3, b3, wd, sd, bd)
    lib = _lib.load()
    N, H, W, C = x.shape
    assert C == 64 and x.dtype in _HALF and x.is_contiguous()
    assert tuple(w1.shape) == (64, 1, 1, 64) and tuple(w2.shape) == (64, 3, 3, 64)
    assert tuple(w3.shape) == (256, 1, 1, 64) and tuple(wd.shape) == (256, 1, 1, 64)
    for t in (w1, w2, w3, wd):
        assert t.dtype == x.dtype and t.is_contiguous()
    for v, n in ((s1, 64), (b1, 64), (s2, 64), (b2, 64), (s3, 256), (b3, 256), (sd, 256), (bd, 256)):
        assert v.dtype == torch.float32 and v.numel() == n and v.is_contiguous()
    out = torch.empty((N, H, W, 256), dtype=x.dtype, device=x.device)
    if out.numel() * 2 >= 0x7FF00000:
        raise ValueError("bottleneck64_ds: output of %.2f GiB; 32-bit buffer offsets (< 2 GiB)" % (out.numel() * 2 / 2.0 ** 30))
    px = float(N * H * W)
    _tok = _pb("bneck64_fused", 2.0 * px * (64 * 64 + 576 * 64 + 2 * 64 * 256), px * 640.0)
    rc = lib.mega_bottleneck64_ds_fwd_dt(_ptr(x), _ptr(w1), _ptr(s1), _ptr(b1), _ptr(w2), _ptr(s2), _ptr(b2), _ptr(w3), _ptr(s3),
                                         _ptr(b3), _ptr(wd), _ptr(sd), _ptr(bd), _ptr(out), N, H, W, _dt(x), _stream())
    _pe(_tok)
    _lib.check(rc, "mega_bottleneck64_ds_fwd_dt")
    return out


def linear(x, w, bias=None, relu=False, residual=None, out_dtype=None, scale=None, ksplit=None):
    """x [M,K], w [Nout,K] (nn.Linear layout) -> [M,Nout].  ksplit: the caller's split-K count (conv2d_nhwc)."""
    if isinstance(w, X3Weight):      # split-precision: f32 in, f32 out
        assert x.dtype == torch.float32 and residual is None and scale is None and out_dtype in (None, torch.float32)
        return linear_sp(split_planes(x.contiguous()), w.w3, bias, relu=relu)
    M, K = x.shape
    y = conv2d_nhwc(x.view(M, 1, 1, K), w.view(w.shape[0], 1, 1, K), scale=scale, bias=bias,
                    residual=None if residual is None else residual.view(M, 1, 1, -1), relu=relu,
                    out_dtype=out_dtype, ksplit=ksplit)
    return y.view(M, w.shape[0])


def stem(x_nchw, w_tap64, scale, bias, out_dtype, w_n160=None):
    """x [N,3,H,W] f32 -> conv7x7 s2 + BN + ReLU -> NHWC [N,Ho,Wo,64].  f32: direct conv with w_tap64 [147,64] f32;
    bf16: matrix-core kernel with w_n160 = bf16 [64,176] (see pack_stem_weight_bf16)."""
    _gpu(x_nchw, w_tap64, scale, bias)
    lib = _lib.load()
    N, C, H, W = x_nchw.shape
    assert C == 3 and x_nchw.dtype == torch.float32 and x_nchw.is_contiguous()
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty((N, Ho, Wo, 64), dtype=out_dtype, device=x_nchw.device)
    _tok = _pb("stem", 2.0 * N * Ho * Wo * 64 * 147, x_nchw.numel() * 4 + out.numel() * out.element_size())
    if out_dtype == torch.bfloat16 and w_n160 is not None:
        _gpu(w_n160)
        assert w_n160.dtype == torch.bfloat16 and tuple(w_n160.shape) == (64, 176) and w_n160.is_contiguous()
        rc = lib.mega_stem_conv_bn_relu_bf16(_ptr(x_nchw), _ptr(w_n160), _ptr(scale), _ptr(bias), _ptr(out), N, H, W,
                                             _stream())
    else:
        rc = lib.mega_stem_conv_bn_relu(_ptr(x_nchw), _ptr(w_tap64), _ptr(scale), _ptr(bias), _ptr(out), N, H, W,
                                        _DT[out_dtype], _stream())
    _pe(_tok)
    _lib.check(rc, "mega_stem_conv_bn_relu")
    return out


def stem_u8(frames_u8, w_n160, scale, bias, mean, to_bgr=True):
    """uint8 frames [N,H,W,3] RGB -> preprocess (BGR*255 - mean) + conv7x7 s2 + BN + ReLU -> NHWC bf16 [N,Ho,Wo,64] in ONE
    kernel (the bf16 matrix-core stem with the preprocessing on its patch load): same bits as
    stem(preprocess_frames(frames_u8), ...)."""
    _gpu(frames_u8, w_n160, scale, bias)
    lib = _lib.load()
    N, H, W, C = frames_u8.shape
    assert C == 3 and frames_u8.dtype == torch.uint8 and frames_u8.is_contiguous()
    assert w_n160.dtype == torch.bfloat16 and tuple(w_n160.shape) == (64, 176) and w_n160.is_contiguous()
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty((N, Ho, Wo, 64), dtype=torch.bfloat16, device=frames_u8.device)
    _tok = _pb("stem", 2.0 * N * Ho * Wo * 64 * 147, frames_u8.numel() + out.numel() * 2)
    rc = lib.mega_stem_conv_bn_relu_bf16_u8(_ptr(frames_u8), _ptr(w_n160), _ptr(scale), _ptr(bias), _ptr(out), N, H, W,
                                            float(mean[0]), float(mean[1]), float(mean[2]), int(to_bgr), _stream())
    _pe(_tok)
    _lib.check(rc, "mega_stem_conv_bn_relu_bf16_u8")
    return out


def stem_pool(x, w_n160, scale, bias, mean=None, to_bgr=True):
    """resnet.py:355-366 in ONE kernel (bf16): conv7x7 s2 + BN + ReLU + max_pool2d(3, 2, 1) -> NHWC bf16 [N,Hp,Wp,64].  x =
    uint8 frames [N,H,W,3] RGB (preprocessing on the patch load, `mean` required) or the preprocessed f32 NCHW image.  Same
    bits as maxpool3x3s2(stem[_u8](...)); the stem's own map never reaches HBM."""
    _gpu(x, w_n160, scale, bias)
    lib = _lib.load()
    u8 = x.dtype == torch.uint8
    if u8:
        N, H, W, C = x.shape
        assert mean is not None
    else:
        N, C, H, W = x.shape
        assert x.dtype == torch.float32
        mean = (0.0, 0.0, 0.0)
    assert C == 3 and x.is_contiguous()
    assert w_n160.dtype in _HALF and tuple(w_n160.shape) == (64, 176) and w_n160.is_contiguous()
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Hp, Wp = (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1
    out = torch.empty((N, Hp, Wp, 64), dtype=w_n160.dtype, device=x.device)      # (the weights' 16-bit type is the output's)
    _tok = _pb("stem", 2.0 * N * Ho * Wo * 64 * 147, x.numel() * x.element_size() + out.numel() * 2)
    rc = lib.mega_stem_pool_dt(_ptr(x), int(u8), _ptr(w_n160), _ptr(scale), _ptr(bias), _ptr(out), N, H, W,
                               float(mean[0]), float(mean[1]), float(mean[2]), int(to_bgr), _dt(w_n160), _stream())
    _pe(_tok)
    _lib.check(rc, "mega_stem_pool_dt")
    return out


def pack_stem_weight_bf16(w_oihw, dtype=torch.bfloat16):
    """conv1.weight [64,3,7,7] -> bf16 (or f16) [64,176]: column k = ((c*7+r)*8 + s for the 7 taps s of kernel row (c, r); the
    8th column of every group and columns 168..175 are zero (the kernel reads 8 consecutive patch pixels per group)."""
    w = torch.zeros((64, 22, 8), dtype=torch.float32, device=w_oihw.device)
    w[:, :21, :7] = w_oihw.detach().float().reshape(64, 21, 7)
    w = w.reshape(64, 176)
    return w.to(dtype).contiguous()


def maxpool3x3s2(x):
    _gpu(x)
    lib = _lib.load()
    N, H, W, C = x.shape
    assert x.is_contiguous()
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty((N, Ho, Wo, C), dtype=x.dtype, device=x.device)
    _tok = _pb("maxpool", 0.0, (x.numel() + out.numel()) * x.element_size())
    rc = lib.mega_maxpool3x3s2_nhwc(_ptr(x), _ptr(out), N, H, W, C, _dt(x), _stream())
    _pe(_tok)
    _lib.check(rc, "mega_maxpool3x3s2_nhwc")
    return out


# ------------------------------------------------------------------------------------------------ ROIAlign
def roi_align(feat, rois, spatial_scale, pooled, sampling_ratio, in_nhwc=True, out_nhwc=True):
    """feat NHWC [B,H,W,C] (or NCHW when in_nhwc=False); rois [K,5] f32 -> [K,ph*pw,C] (or [K,C,ph,pw])."""
    _gpu(feat, rois)
    lib = _lib.load()
    if in_nhwc:
        B, H, W, C = feat.shape
    else:
        B, C, H, W = feat.shape
    ph, pw = pooled
    K = rois.shape[0]
    assert feat.is_contiguous() and rois.dtype == torch.float32 and rois.is_contiguous() and rois.shape[1] == 5
    shape = (K, ph * pw, C) if out_nhwc else (K, C, ph, pw)
    out = torch.empty(shape, dtype=feat.dtype, device=feat.device)
    _tok = _pb("roi_align", 0.0, out.numel() * out.element_size() + feat.numel() * feat.element_size())
    rc = lib.mega_roi_align_fwd(_ptr(feat), _ptr(rois), _ptr(out), K, C, H, W, float(spatial_scale), ph, pw,
                                int(sampling_ratio), int(in_nhwc), int(out_nhwc), _dt(feat), _dt(feat), _stream())
    _pe(_tok)
    _lib.check(rc, "mega_roi_align_fwd")
    return out


def roi_align_planes(feat, rois, spatial_scale, pooled, sampling_ratio, dtype=torch.bfloat16):
    """roi_align on f32 NHWC features, the pooled rows as split-precision planes: -> Planes [K, ph*pw*C] (t = bf16 / f16
    [K, 2*ph*pw*C]), without the f32 tensor in between.  Large launches (C / 4 a multiple of 8 channel vectors, adaptive grid) run
    the separable per-ROI form of the 16-bit kernels: split_planes(roi_align(...)) to f32 round-off; small ones the
    exact-term-order kernel: bit for bit (MEGA_ROI_NO_SEPARABLE=1: always)."""
    _gpu(feat, rois)
    lib = _lib.load()
    B, H, W, C = feat.shape
    ph, pw = pooled
    K = rois.shape[0]
    assert feat.dtype == torch.float32 and feat.is_contiguous() and rois.dtype == torch.float32 and rois.is_contiguous()
    assert rois.shape[1] == 5 and C % 4 == 0
    out = torch.empty((K, 2 * ph * pw * C), dtype=dtype, device=feat.device)
    _tok = _pb("roi_align", 0.0, out.numel() * 2 + feat.numel() * 4)
    rc = lib.mega_roi_align_fwd_planes_dt(_ptr(feat), _ptr(rois), _ptr(out), K, C, H, W, float(spatial_scale), ph, pw,
                                          int(sampling_ratio), _DT[dtype], _stream())
    _pe(_tok)
    _lib.check(rc, "mega_roi_align_fwd_planes_dt")
    return Planes(out, ph * pw * C)


# ------------------------------------------------------------------------------------------------ NMS
def nms(dets, scores, thr, strict_gt=True):
    """Kept ORIGINAL indices ascending (int64), reference mega_core._C.nms semantics.  One host sync (size)."""
    _gpu(dets, scores)
    lib = _lib.load()
    n = dets.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=dets.device)
    if n > 8192:
        raise ValueError("nms: %d boxes; the single-block sort / scan kernels take at most 8192 per problem" % n)
    dets = dets.contiguous().float()
    scores = scores.contiguous().float()
    keep = torch.empty((n,), dtype=torch.int64, device=dets.device)
    cnt = torch.zeros((1,), dtype=torch.int32, device=dets.device)
    nb = lib.mega_nms_full_workspace_bytes(n)
    ws = _ws(nb, dets.device)
    rc = lib.mega_nms(_ptr(dets), _ptr(scores), n, float(thr), int(strict_gt), _ptr(keep), _ptr(cnt), _ptr(ws), nb,
                      _stream())
    _lib.check(rc, "mega_nms")
    return keep[: int(cnt.item())]


def rpn_select(rpn_out, cell_anchors, Hf, Wf, anchor_stride, pre_nms, post_nms, nms_thresh, min_size, im_w, im_h,
               strict_gt=True, want_index=False, hold=None):
    """rpn_out [B,Hf*Wf,5A] f32 -> proposals [B,post_nms,4], scores [B,post_nms], counts [B] (all on device).
    want_index: also the kept proposals' flat anchor indices (y*Wf + x)*A + a, [B,post_nms] i32, -1 past the count.
    hold (a list): the workspace is appended to it -- a caller that launches this on a side stream (launch_on) keeps it
    alive until the streams have joined, so that the allocator cannot hand the block to the current stream's next op."""
    _gpu(rpn_out, cell_anchors)
    lib = _lib.load()
    B = rpn_out.shape[0]
    A = cell_anchors.shape[0]
    ldc = rpn_out.shape[-1]
    assert rpn_out.dtype == torch.float32 and rpn_out.is_contiguous() and rpn_out.numel() == B * Hf * Wf * ldc
    k = min(pre_nms, Hf * Wf * A)
    if k > 8192:
        raise ValueError("rpn_select: PRE_NMS_TOP_N = %d; the on-chip sort takes at most 8192 candidates per frame "
                         "(MODEL.RPN.PRE_NMS_TOP_N_TEST / MODEL.VID.RPN.REF_PRE_NMS_TOP_N)" % k)
    props = torch.empty((B, post_nms, 4), dtype=torch.float32, device=rpn_out.device)
    scores = torch.empty((B, post_nms), dtype=torch.float32, device=rpn_out.device)
    cnt = torch.empty((B,), dtype=torch.int32, device=rpn_out.device)
    nb = lib.mega_rpn_select_workspace_bytes(B, k)
    ws = _ws(nb, rpn_out.device)
    _tok = _pb("rpn_select", 0.0, rpn_out.numel() * 4)
    index = torch.empty((B, post_nms), dtype=torch.int32, device=rpn_out.device) if want_index else None
    rc = lib.mega_rpn_select_idx(_ptr(rpn_out), _ptr(cell_anchors), B, Hf, Wf, A, ldc, anchor_stride, pre_nms, post_nms,
                                 float(nms_thresh), int(strict_gt), float(min_size), float(im_w), float(im_h),
                                 _ptr(props), _ptr(scores), _ptr(cnt), _ptr(index), _ptr(ws), nb, _stream())
    _pe(_tok)
    _lib.check(rc, "mega_rpn_select")
    if hold is not None:
        hold.append(ws)
    return (props, scores, cnt, index) if want_index else (props, scores, cnt)


def postprocess(logits, deltas, props, nprop, weights, im_w, im_h, score_thresh, nms_thresh, max_det,
                strict_gt=True, want_probs=False):
    """One image: returns (boxes [cap,4], scores [cap], labels [cap] i64, count [1] i32 device[, probs])."""
    _gpu(logits, deltas, props)
    lib = _lib.load()
    R, NC = logits.shape
    if R > 1024:
        raise ValueError("postprocess: %d proposals; the per-class sort takes at most 1024 per image "
                         "(MODEL.RPN.POST_NMS_TOP_N_TEST)" % R)
    cap = (NC - 1) * R
    dev = logits.device
    ob = torch.empty((cap, 4), dtype=torch.float32, device=dev)
    os_ = torch.empty((cap,), dtype=torch.float32, device=dev)
    ol = torch.empty((cap,), dtype=torch.int64, device=dev)
    oc = torch.zeros((1,), dtype=torch.int32, device=dev)
    probs = torch.empty((R, NC), dtype=torch.float32, device=dev) if want_probs else None
    nb = lib.mega_postprocess_workspace_bytes(R, NC)
    ws = _ws(nb, dev)
    assert logits.dtype == torch.float32 and deltas.dtype == torch.float32 and props.dtype == torch.float32
    assert logits.is_contiguous() and deltas.is_contiguous() and props.is_contiguous()
    wx, wy, ww, wh = weights
    _tok = _pb("postprocess", 0.0, (logits.numel() + deltas.numel()) * 4)
    rc = lib.mega_postprocess(_ptr(logits), _ptr(deltas), _ptr(props), _ptr(nprop), R, NC, wx, wy, ww, wh,
                              float(im_w), float(im_h), float(score_thresh), float(nms_thresh), int(strict_gt),
                              int(max_det), _ptr(ob), _ptr(os_), _ptr(ol), _ptr(oc), _ptr(probs), _ptr(ws), nb,
                              _stream())
    _pe(_tok)
    _lib.check(rc, "mega_postprocess")
    return (ob, os_, ol, oc, probs) if want_probs else (ob, os_, ol, oc)


def postprocess_batched(logits, deltas, props, B, weights, im_w, im_h, score_thresh, nms_thresh, max_det,
                        strict_gt=True, nprop=None):
    """B images of R = rows / B proposals each (logits [B*R,NC], deltas [B*R,NC*4], props [B*R,4]) in one launch chain.
    Returns (boxes [B,cap,4], scores [B,cap], labels [B,cap] i64, counts [B] i32 device); image b has the bits of
    postprocess() on its own rows."""
    _gpu(logits, deltas, props)
    lib = _lib.load()
    NC = logits.shape[1]
    assert logits.shape[0] % B == 0
    R = logits.shape[0] // B
    if R > 1024:
        raise ValueError("postprocess: %d proposals; the per-class sort takes at most 1024 per image "
                         "(MODEL.RPN.POST_NMS_TOP_N_TEST)" % R)
    cap = (NC - 1) * R
    dev = logits.device
    ob = torch.empty((B, cap, 4), dtype=torch.float32, device=dev)
    os_ = torch.empty((B, cap), dtype=torch.float32, device=dev)
    ol = torch.empty((B, cap), dtype=torch.int64, device=dev)
    oc = torch.zeros((B,), dtype=torch.int32, device=dev)
    nb = lib.mega_postprocess_batched_workspace_bytes(B, R, NC)
    ws = _ws(nb, dev)
    assert logits.dtype == torch.float32 and deltas.dtype == torch.float32 and props.dtype == torch.float32
    assert logits.is_contiguous() and deltas.is_contiguous() and props.is_contiguous()
    assert deltas.shape == (B * R, NC * 4) and props.shape == (B * R, 4)
    wx, wy, ww, wh = weights
    _tok = _pb("postprocess", 0.0, (logits.numel() + deltas.numel()) * 4)
    if nprop is not None:      # device i32 [B]: the valid proposal rows of each image (rows past it are ignored)
        _gpu(nprop)
        assert nprop.dtype == torch.int32 and nprop.numel() == B and nprop.is_contiguous()
    rc = lib.mega_postprocess_batched(_ptr(logits), _ptr(deltas), _ptr(props), _ptr(nprop), B, R, NC, wx, wy, ww, wh,
                                      float(im_w), float(im_h), float(score_thresh), float(nms_thresh), int(strict_gt),
                                      int(max_det), _ptr(ob), _ptr(os_), _ptr(ol), _ptr(oc), None, _ptr(ws), nb,
                                      _stream())
    _pe(_tok)
    _lib.check(rc, "mega_postprocess_batched")
    return ob, os_, ol, oc


# ------------------------------------------------------------------------------------------------ relation module
def position_logits(rois_q, rois_k, wg_t, bg, dim_mat, precise=True, tiled=False):
    """-> [16, Nq, ldp] f32 with ldp = roundup(Nk, 32).  precise=False: fast sin/cos (bf16 mode).
    tiled=True (bf16 mode only): -> bf16 [16, ceil(Nk/32), Nq, 32] in the attention kernel's tile order (half the
    bytes; relation_attention recognises it by its dtype)."""
    _gpu(rois_q, rois_k, wg_t, bg, dim_mat)
    lib = _lib.load()
    Nq, Nk = rois_q.shape[0], rois_k.shape[0]
    if tiled:
        tdt = torch.bfloat16 if tiled is True else tiled      # the head's 16-bit operand type (bf16, or f16: round 6)
        assert not precise and tdt in _HALF, "the tile-ordered position logits are bf16 / f16"
        kt = (Nk + 31) // 32
        out = torch.empty((16, kt, Nq, 32), dtype=tdt, device=rois_q.device)
        _tok = _pb("pos_logits", 2.0 * Nq * Nk * 1024, out.numel() * 2.0)
        rc = lib.mega_position_logits_tiled_dt(_ptr(rois_q.contiguous()), _ptr(rois_k.contiguous()), _ptr(wg_t), _ptr(bg),
                                               _ptr(dim_mat), _ptr(out), Nq, Nk, _DT[tdt], _stream())
        _pe(_tok)
        _lib.check(rc, "mega_position_logits_tiled_dt")
        return out
    ldp = (Nk + 31) // 32 * 32
    out = torch.empty((16, Nq, ldp), dtype=torch.float32, device=rois_q.device)
    _tok = _pb("pos_logits", 2.0 * Nq * Nk * 1024, 16.0 * Nq * ldp * 4)
    rc = lib.mega_position_logits(_ptr(rois_q.contiguous()), _ptr(rois_k.contiguous()), _ptr(wg_t), _ptr(bg),
                                  _ptr(dim_mat), _ptr(out), Nq, Nk, ldp, int(precise), _stream())
    _pe(_tok)
    _lib.check(rc, "mega_position_logits")
    return out


def relation_attention(q, k, vt, Nk, pos=None, resid=None, bias_v=None, groups=16):
    """q [Nq,G*64] (u folded in), k [Nk,G*64], vt [G*64, ldv] key-contiguous projected V -> [Nq, G*64]."""
    _gpu(q, k, vt, pos, resid, bias_v)
    if resid is not None and resid.dtype != q.dtype:     # f32 activation stream over bf16 operands: same kernel, same bits
        return relation_attention_batched([{"q": q, "k": k, "vt": vt, "Nk": Nk, "pos": pos, "resid": resid,
                                            "bias_v": bias_v}], groups)[0]
    lib = _lib.load()
    Nq = q.shape[0]
    assert q.dtype == k.dtype == vt.dtype and q.is_contiguous() and k.is_contiguous() and vt.is_contiguous()
    out = torch.empty((Nq, groups * 64), dtype=q.dtype, device=q.device)
    nb = lib.mega_relation_attention_workspace_bytes(Nq, Nk, groups)
    ws = _ws(nb, q.device) if nb else None
    _tok = _pb("attention_" + _ATT_NAME.get(q.dtype, "f32"), 4.0 * Nq * Nk * 64 * groups, (q.numel() + k.numel() + vt.numel() + out.numel()) * q.element_size() + (0 if pos is None else pos.numel() * pos.element_size()))
    if pos is not None and pos.dtype in _HALF:       # tile-ordered 16-bit logits (position_logits(tiled=dtype))
        assert q.dtype == pos.dtype and pos.is_contiguous() and tuple(pos.shape) == (groups, (Nk + 31) // 32, Nq, 32)
        rc = lib.mega_relation_attention_tiled_pos_dt(_ptr(q), q.shape[1], _ptr(k), k.shape[1], _ptr(vt), vt.shape[1],
                                                      _ptr(pos), _ptr(resid), 0 if resid is None else resid.shape[1],
                                                      _ptr(bias_v), _ptr(out), groups * 64, Nq, Nk, groups,
                                                      1.0 / math.sqrt(64.0), _DT[q.dtype], _ptr(ws), nb, _stream())
        _pe(_tok)
        _lib.check(rc, "mega_relation_attention_tiled_pos_dt")
        return out
    rc = lib.mega_relation_attention(_ptr(q), q.shape[1], _ptr(k), k.shape[1], _ptr(vt), vt.shape[1], _ptr(pos),
                                     0 if pos is None else pos.shape[2], _ptr(resid),
                                     0 if resid is None else resid.shape[1], _ptr(bias_v), _ptr(out), groups * 64,
                                     Nq, Nk, groups, 1.0 / math.sqrt(64.0), _dt(q), _ptr(ws), nb, _stream())
    _pe(_tok)
    _lib.check(rc, "mega_relation_attention")
    return out


_ATT_NAME = {torch.bfloat16: "bf16", torch.float16: "f16"}      # attention kernel family names (bench.py's per-family roofline)
_MAX_BATCHED = 20    # problems per batched position-logit / attention launch (POS_MAXB / ATTN_MAXB in relation.hip)


def _even_chunks(n, cap):
    """problems per launch when n problems go out in launches of at most `cap`: as even as possible (20 -> 10 + 10, not
    16 + 4: a 4-problem launch leaves most of the chip idle)"""
    launches = -(-n // cap)
    return -(-n // launches) if launches else cap


class _AttnDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("q", "k", "vt", "pos", "pos_tiled", "resid", "bias_v", "out", "ws")] + \
               [("ws_bytes", ctypes.c_size_t)] + \
               [(n, ctypes.c_int) for n in ("ldq", "ldk", "ldv", "ldp", "ldr", "ldo", "Nq", "Nk", "io_f32", "nk1")] + \
               [("k2", ctypes.c_void_p), ("vt2", ctypes.c_void_p), ("ldv2", ctypes.c_int), ("reserved", ctypes.c_int)]


class _PosDesc(ctypes.Structure):
    _fields_ = [("rois_q", ctypes.c_void_p), ("rois_k", ctypes.c_void_p), ("out", ctypes.c_void_p),
                ("Nq", ctypes.c_int), ("Nk", ctypes.c_int)]


def position_logits_batched(rois_qs, rois_ks, wg_t, bg, dim_mat, precise=True, tiled=False):
    """position_logits for several (query boxes, key boxes) problems; the tile-ordered bf16 form (bf16 mode) runs as
    ONE launch per 20 problems, the f32 forms fall back to one call per problem."""
    if not tiled:
        return [position_logits(a, b, wg_t, bg, dim_mat, precise=precise, tiled=False) for a, b in zip(rois_qs, rois_ks)]
    tdt = torch.bfloat16 if tiled is True else tiled
    assert tdt in _HALF, "the tile-ordered position logits are bf16 / f16"
    _gpu(wg_t, bg, dim_mat, *rois_qs, *rois_ks)
    lib = _lib.load()
    outs, keep = [], []
    for a, b in zip(rois_qs, rois_ks):
        a, b = a.contiguous(), b.contiguous()
        keep.append((a, b))
        outs.append(torch.empty((16, (b.shape[0] + 31) // 32, a.shape[0], 32), dtype=tdt, device=a.device))
    per = _even_chunks(len(outs), _MAX_BATCHED)
    for o in range(0, len(outs), per):
        n = min(per, len(outs) - o)
        arr = (_PosDesc * n)()
        for i in range(n):
            a, b = keep[o + i]
            arr[i].rois_q, arr[i].rois_k, arr[i].out = a.data_ptr(), b.data_ptr(), outs[o + i].data_ptr()
            arr[i].Nq, arr[i].Nk = a.shape[0], b.shape[0]
        _tok = _pb("pos_logits", sum(2.0 * arr[i].Nq * arr[i].Nk * 1024 for i in range(n)),
                   sum(outs[o + i].numel() * 2.0 for i in range(n)))
        rc = lib.mega_position_logits_tiled_batched_dt(ctypes.addressof(arr), n, _ptr(wg_t), _ptr(bg), _ptr(dim_mat), _DT[tdt], _stream())
        _pe(_tok)
        _lib.check(rc, "mega_position_logits_tiled_batched_dt")
    return outs


def relation_attention_batched(items, groups=16):
    """relation_attention for a list of independent problems, each a dict(q, k, vt, Nk, pos, resid, bias_v), in ONE launch
    per 20 problems (+ one combine launch).  Every problem gets the bits of its own relation_attention call."""
    if not items:
        return []
    lib = _lib.load()
    dt = items[0]["q"].dtype
    # stream dtype: the residual's (the head's f32 activation stream over bf16 operands), else the operands'
    r0 = items[0].get("resid")
    sdt = dt if r0 is None else r0.dtype
    io_f32 = int(sdt == torch.float32 and dt != torch.float32)
    outs, wss = [], []
    # one output buffer, the problems' row blocks in order (the next batched GEMM reads it without a concatenation)
    out_all = torch.empty((sum(it["q"].shape[0] for it in items), groups * 64), dtype=sdt, device=items[0]["q"].device)
    o = 0
    for it in items:
        q, k, vt, pos = it["q"], it["k"], it["vt"], it.get("pos")
        _gpu(q, k, vt, pos, it.get("resid"), it.get("bias_v"))
        # q / k / resid: row blocks (or column-complete views) of wider buffers are fine -- unit column stride, the row
        # stride is what the kernel gets as the leading dimension, 16-byte aligned rows
        resid = it.get("resid")
        assert resid is None or resid.dtype == sdt
        for t_ in (q, k, resid):
            assert t_ is None or (t_.dtype in (dt, sdt) and t_.stride(1) == 1 and t_.data_ptr() % 16 == 0 and
                                  (t_.stride(0) * t_.element_size()) % 16 == 0)
        assert q.dtype == k.dtype == vt.dtype == dt
        # vt: [G*64, >= ceil32(Nk)] with unit column stride; a column block of a wider matrix is fine (16-B aligned).
        # With a second key segment (k2 / vt2 / N1) the blocks are read in place at element alignment: no such constraint.
        if it.get("k2") is None:
            assert vt.stride(1) == 1 and vt.shape[1] >= (it["Nk"] + 31) // 32 * 32 and vt.data_ptr() % 16 == 0 and \
                (vt.stride(0) * vt.element_size()) % 16 == 0
            assert k.shape[0] >= it["Nk"]
        else:
            _gpu(it["k2"], it["vt2"])
            assert it["k2"].data_ptr() % 16 == 0 and (vt.stride(0) * vt.element_size()) % 16 == 0
        assert (pos is None) == (items[0].get("pos") is None) and (pos is None or pos.dtype == items[0]["pos"].dtype)
        outs.append(out_all[o:o + q.shape[0]])
        o += q.shape[0]
        nb = lib.mega_relation_attention_workspace_bytes(q.shape[0], it["Nk"], groups)
        wss.append(_ws(nb, q.device) if nb else None)
    per = _even_chunks(len(items), _MAX_BATCHED)
    for o in range(0, len(items), per):
        n = min(per, len(items) - o)
        arr = (_AttnDesc * n)()
        fl = by = 0.0
        for i in range(n):
            it, d = items[o + i], arr[i]
            q, k, vt, pos, resid = it["q"], it["k"], it["vt"], it.get("pos"), it.get("resid")
            Nq, Nk = q.shape[0], it["Nk"]
            d.q, d.k, d.vt, d.out = q.data_ptr(), k.data_ptr(), vt.data_ptr(), outs[o + i].data_ptr()
            d.ldq, d.ldk, d.ldv, d.ldo, d.Nq, d.Nk = q.stride(0), k.stride(0), vt.stride(0), groups * 64, Nq, Nk
            k2, vt2 = it.get("k2"), it.get("vt2")
            if k2 is not None:          # second key segment (keys N1 .. Nk-1), read where it lies
                N1 = it["N1"]
                assert 0 < N1 < Nk and k.shape[0] >= N1 and k2.shape[0] == Nk - N1 and vt2.shape[1] >= Nk - N1
                assert k2.dtype == k.dtype and vt2.dtype == vt.dtype and k2.stride(0) == k.stride(0)
                assert k2.stride(1) == 1 and vt2.stride(1) == 1 and vt.shape[1] >= N1
                d.nk1, d.k2, d.vt2, d.ldv2 = N1, k2.data_ptr(), vt2.data_ptr(), vt2.stride(0)
            d.resid, d.ldr = _ptr(resid), 0 if resid is None else resid.stride(0)
            d.io_f32 = io_f32
            d.bias_v = _ptr(it.get("bias_v"))
            if pos is not None and pos.dtype in _HALF:
                assert pos.dtype == dt and pos.is_contiguous() and tuple(pos.shape) == (groups, (Nk + 31) // 32, Nq, 32)
                d.pos_tiled = pos.data_ptr()
            elif pos is not None:
                d.pos, d.ldp = pos.data_ptr(), pos.shape[2]
            w = wss[o + i]
            d.ws, d.ws_bytes = _ptr(w), 0 if w is None else w.numel()
            fl += 4.0 * Nq * Nk * 64 * groups
            by += (q.numel() + k.numel() + vt.numel()) * q.element_size() + outs[o + i].numel() * outs[o + i].element_size() + \
                (0 if pos is None else pos.numel() * pos.element_size())
        _tok = _pb("attention_" + _ATT_NAME.get(dt, "f32"), fl, by)
        rc = lib.mega_relation_attention_batched(ctypes.addressof(arr), n, groups, 1.0 / math.sqrt(64.0), _DT[dt], _stream())
        _pe(_tok)
        _lib.check(rc, "mega_relation_attention_batched")
    return outs


class _CopySeg(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("src_stride", ctypes.c_longlong),
                ("dst_stride", ctypes.c_longlong), ("rows", ctypes.c_int), ("row_bytes", ctypes.c_int)]


def copy_blocks(pairs):
    """[(dst, src)]: 2-D tensors of equal shape / dtype with unit column stride (row / column blocks of wider buffers
    are fine) -> all copies in ONE launch (mega_copy_segments)."""
    pairs = [(d, s_) for d, s_ in pairs if d.numel() > 0]
    if not pairs:
        return
    lib = _lib.load()
    arr = (_CopySeg * len(pairs))()
    nbytes = 0
    for i, (d, s_) in enumerate(pairs):
        _gpu(d, s_)
        assert d.dim() == 2 and d.shape == s_.shape and d.dtype == s_.dtype, (d.shape, s_.shape, d.dtype, s_.dtype)
        assert (d.shape[1] == 1 or (d.stride(1) == 1 and s_.stride(1) == 1))
        es = d.element_size()
        arr[i].src, arr[i].dst = s_.data_ptr(), d.data_ptr()
        arr[i].src_stride, arr[i].dst_stride = s_.stride(0) * es, d.stride(0) * es
        arr[i].rows, arr[i].row_bytes = d.shape[0], d.shape[1] * es
        nbytes += 2 * d.numel() * es
    _tok = _pb("assemble", 0.0, nbytes)
    rc = lib.mega_copy_segments(ctypes.addressof(arr), len(pairs), _stream())
    _pe(_tok)
    _lib.check(rc, "mega_copy_segments")


def multi_cat(groups):
    """Several independent torch.cat calls as ONE launch: groups = [(pieces, dim)], pieces = 2-D tensors (same dtype and
    device, unit column stride, equal sizes along the other dimension) -> [torch.cat(pieces, dim) for each group], same
    bits.  The concatenations must not depend on each other's results."""
    outs, pairs = [], []
    for pieces, dim in groups:
        pieces = list(pieces)
        p0 = pieces[0]
        if dim == 0:
            out = torch.empty((sum(p.shape[0] for p in pieces), p0.shape[1]), dtype=p0.dtype, device=p0.device)
        else:
            out = torch.empty((p0.shape[0], sum(p.shape[1] for p in pieces)), dtype=p0.dtype, device=p0.device)
        o = 0
        for p in pieces:
            assert p.dtype == p0.dtype and p.shape[1 - dim] == p0.shape[1 - dim]
            n = p.shape[dim]
            pairs.append((out.narrow(dim, o, n), p))
            o += n
        outs.append(out)
    copy_blocks(pairs)
    return outs


def cat_rows_cast_bf16(pieces, dtype=torch.bfloat16):
    """torch.cat(pieces, 0).to(dtype) (bfloat16 / float16) for f32 row blocks [n_i, K] (K % 8 == 0, unit column stride) in
    ONE launch that never writes the f32 concatenation: the K / V sources of the relation modules under an f32 activation
    stream."""
    pieces = [p for p in pieces if p.shape[0] > 0]
    lib = _lib.load()
    K = pieces[0].shape[1]
    assert dtype in _HALF
    out = torch.empty((sum(p.shape[0] for p in pieces), K), dtype=dtype, device=pieces[0].device)
    arr = (_CopySeg * len(pieces))()
    o, nbytes = 0, 0
    for i, p in enumerate(pieces):
        _gpu(p)
        assert p.dtype == torch.float32 and p.dim() == 2 and p.shape[1] == K and K % 8 == 0 and p.stride(1) == 1
        d = out[o:o + p.shape[0]]
        arr[i].src, arr[i].dst = p.data_ptr(), d.data_ptr()
        arr[i].src_stride, arr[i].dst_stride = p.stride(0) * 4, K * 2
        arr[i].rows, arr[i].row_bytes = p.shape[0], K * 4
        o += p.shape[0]
        nbytes += p.numel() * 6
    _tok = _pb("assemble", 0.0, nbytes)
    rc = lib.mega_copy_cast_segments_dt(ctypes.addressof(arr), len(pieces), _DT[dtype], _stream())
    _pe(_tok)
    _lib.check(rc, "mega_copy_cast_segments_dt")
    return out


def resize_bilinear_u8(frames_u8, out_hw, tables):
    """uint8 [N,Hi,Wi,3] -> [N,Ho,Wo,3], Pillow-exact BILINEAR.  tables = feed.ResizeTables (device coefficient tables)."""
    _gpu(frames_u8)
    lib = _lib.load()
    N, Hi, Wi, C = frames_u8.shape
    Ho, Wo = out_hw
    assert C == 3 and frames_u8.dtype == torch.uint8 and frames_u8.is_contiguous()
    assert tables.in_hw == (Hi, Wi) and tables.out_hw == (Ho, Wo)
    out = torch.empty((N, Ho, Wo, 3), dtype=torch.uint8, device=frames_u8.device)
    tmp = torch.empty((N, Hi, Wo, 3), dtype=torch.uint8, device=frames_u8.device) if (Hi != Ho and Wi != Wo) else None
    _tok = _pb("resize", 0.0, frames_u8.numel() + out.numel() * 3)
    rc = lib.mega_resize_bilinear_u8(_ptr(frames_u8), _ptr(out), _ptr(tmp), N, Hi, Wi, Ho, Wo, _ptr(tables.bounds_h),
                                     _ptr(tables.coef_h), tables.ksize_h, _ptr(tables.bounds_v), _ptr(tables.coef_v),
                                     tables.ksize_v, _stream())
    _pe(_tok)
    _lib.check(rc, "mega_resize_bilinear_u8")
    return out


def preprocess_frames(frames_u8, mean, to_bgr=True, out=None):
    """uint8 [N,H,W,3] RGB -> f32 [N,3,H,W] (BGR*255 - mean); out: optional contiguous f32 [N,3,H,W] to write into."""
    _gpu(frames_u8, out)
    lib = _lib.load()
    N, H, W, C = frames_u8.shape
    assert C == 3 and frames_u8.dtype == torch.uint8 and frames_u8.is_contiguous()
    if out is None:
        out = torch.empty((N, 3, H, W), dtype=torch.float32, device=frames_u8.device)
    else:
        assert out.shape == (N, 3, H, W) and out.dtype == torch.float32 and out.is_contiguous()
    _tok = _pb("preprocess", 0.0, frames_u8.numel() * 5)
    rc = lib.mega_preprocess_frames(_ptr(frames_u8), _ptr(out), N, H, W, float(mean[0]), float(mean[1]),
                                    float(mean[2]), int(to_bgr), _stream())
    _pe(_tok)
    _lib.check(rc, "mega_preprocess_frames")
    return out


def cast_half(x, dtype):
    """f32 -> bf16 / f16 copy (round to nearest even) of a contiguous tensor; an input of that dtype is returned as it is."""
    if x.dtype == dtype:
        return x
    _gpu(x)
    lib = _lib.load()
    assert x.dtype == torch.float32 and x.is_contiguous() and dtype in _HALF
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    _tok = _pb("assemble", 0.0, x.numel() * 6.0)
    rc = lib.mega_cast_f32_to_half(_ptr(x), _ptr(out), x.numel(), _DT[dtype], _stream())
    _pe(_tok)
    _lib.check(rc, "mega_cast_f32_to_half")
    return out


def cast_bf16(x):
    return cast_half(x, torch.bfloat16)


def split_bf16x3(x):
    """f32 [M,K] -> bf16 [M,3K] = [hi | lo | hi] (hi = bf16(x), lo = bf16(x - hi)): A operand of a split-precision bf16
    GEMM against weights packed by split_weight_bf16x3."""
    _gpu(x)
    lib = _lib.load()
    M, K = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous() and K % 8 == 0
    out = torch.empty((M, 3 * K), dtype=torch.bfloat16, device=x.device)
    _tok = _pb("assemble", 0.0, x.numel() * 10.0)
    rc = lib.mega_split_f32_to_bf16x3(_ptr(x), _ptr(out), M, K, _stream())
    _pe(_tok)
    _lib.check(rc, "mega_split_f32_to_bf16x3")
    return out


def split_weight_bf16x3(w32):
    """f32 [N,K] -> bf16 [N,3K] = [Wh | Wh | Wl]: with split_bf16x3(x) as the other operand, the K = 3K contraction is
    x_hi.Wh + x_lo.Wh + x_hi.Wl = x.W to ~2^-16 (host-side packing, once per model)."""
    wh = w32.to(torch.bfloat16)
    wl = (w32 - wh.float()).to(torch.bfloat16)
    return torch.cat([wh, wh, wl], dim=1).contiguous()


# ------------------------------------------------------------------------------------------------ split-precision planes
class Planes(object):
    """An f32 activation [..., C] held as bf16 [..., 2C] = [hi | lo] planes (hi = bf16(x), lo = bf16(x - hi); x = hi + lo to
    ~2^-17): what the split-precision ("bf16 x 3") frame stage and the wide residual stream of the bf16 mode keep in HBM.
    `t` is the contiguous bf16 tensor, `C` the logical channel count."""
    __slots__ = ("t", "C")

    def __init__(self, t, C):
        assert t.dtype in _HALF and t.shape[-1] == 2 * C and t.is_contiguous()     # (float16 pairs: the fp16 two-pass mode)
        self.t, self.C = t, C

    @property
    def dtype(self):
        return self.t.dtype

    @property
    def shape(self):
        return tuple(self.t.shape[:-1]) + (self.C,)

    @property
    def device(self):
        return self.t.device

    def float(self):
        """hi + lo as an f32 tensor (torch ops; tests and seams only)"""
        return self.t[..., :self.C].float() + self.t[..., self.C:].float()

    def hi(self):
        return self.t[..., :self.C]


def split_planes(x, dtype=torch.bfloat16):
    """f32 [..., C] (contiguous, C % 8 == 0) -> Planes of `dtype` pairs (one launch)."""
    _gpu(x)
    lib = _lib.load()
    C = x.shape[-1]
    assert x.dtype == torch.float32 and x.is_contiguous() and C % 8 == 0 and dtype in _HALF
    out = torch.empty(tuple(x.shape[:-1]) + (2 * C,), dtype=dtype, device=x.device)
    _tok = _pb("assemble", 0.0, x.numel() * 8.0)
    rc = lib.mega_split_f32_to_planes_dt(_ptr(x), _ptr(out), x.numel() // C, C, _DT[dtype], _stream())
    _pe(_tok)
    _lib.check(rc, "mega_split_f32_to_planes_dt")
    return Planes(out, C)


def split_conv_weight_h2(w_ohwi):
    """f32 conv weight [Cout,R,S,C] (OHWI) -> float16 [Cout,R,S,2C] = [W | W] per tap, W rounded to fp16 once: the operand of
    conv2d_sp(x3="h2") -- against fp16 [hi | lo] planes the K = 2C contraction is x_hi.W + x_lo.W (the two-pass fp16 mode)."""
    w = w_ohwi.float().to(torch.float16)
    return torch.cat([w, w], dim=-1).contiguous()


def split_conv_weight_x3(w_ohwi):
    """f32 conv weight [Cout,R,S,C] (OHWI) -> bf16 [Cout,R,S,3C] = [Wh | Wh | Wl] per tap: the operand of conv2d_sp(x3=True)
    (host-side packing, once per model)."""
    w32 = w_ohwi.float()
    wh = w32.to(torch.bfloat16)
    wl = (w32 - wh.float()).to(torch.bfloat16)
    return torch.cat([wh, wh, wl], dim=-1).contiguous()


def conv2d_sp(x, w, scale=None, bias=None, residual=None, stride=1, pad=0, dil=1, relu=False, out_mode="planes", x3=True,
              out=None):
    """conv + FrozenBN (+ residual) + activation on split-precision planes (mega_conv2d_nhwc_sp, igemm8 SP kernels).
    x: Planes [N,H,W,C].  x3=True: w = split_conv_weight_x3(W) [Cout,R,S,3C]; the contraction reads the planes as
    [hi | lo | hi] -- x.W to ~2^-16 with f32 accumulation.  x3=False: w plain bf16 [Cout,R,S,C], only the hi plane is read
    (bf16 compute over a wide residual stream).  residual: Planes [N,Ho,Wo,Cout] (hi + lo added in f32).
    out_mode: "planes" -> Planes, "f32" -> f32 tensor, "bf16" -> bf16 tensor  [N,Ho,Wo,Cout]."""
    assert residual is None or isinstance(residual, Planes)
    # x3 == "h2": the two-pass fp16 form -- float16 planes read as [hi | lo] (K = 2C, no wrap) against w = split_conv_weight_h2(W)
    h2 = isinstance(x3, str) and x3 == "h2"
    kmul = 2 if h2 else (3 if x3 else 1)
    pdt = torch.float16 if h2 else torch.bfloat16
    if not isinstance(x, Planes):      # a plain bf16 tensor as the input (pixel stride C): x3=False only
        assert x.dtype == torch.bfloat16 and x.is_contiguous() and not x3
        xt, ldi = x, x.shape[-1]
    else:
        xt, ldi = x.t, 2 * x.C
    assert xt.dtype == pdt and (residual is None or residual.t.dtype == pdt)
    _gpu(xt, w, scale, bias, None if residual is None else residual.t)
    lib = _lib.load()
    N, H, W, C = x.shape
    Cout, R, S, Cw = w.shape
    assert Cw == kmul * C and w.dtype == pdt and w.is_contiguous() and C % 64 == 0 and Cout % 8 == 0
    Ho = (H + 2 * pad - dil * (R - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (S - 1) - 1) // stride + 1
    mode = {"bf16": 0, "planes": 1, "f32": 2}[out_mode]
    ldo = 0
    if out is not None:      # caller's [N*Ho*Wo, ldo >= Cout] buffer (plain output modes): row stride = its width
        assert mode != 1 and out.dim() == 2 and out.shape[0] == N * Ho * Wo and out.shape[1] >= Cout and out.is_contiguous()
        assert out.dtype == (torch.float32 if mode == 2 else pdt)
        ldo = out.shape[1]
    elif mode == 1:
        out = torch.empty((N, Ho, Wo, 2 * Cout), dtype=pdt, device=x.device)
    else:
        out = torch.empty((N, Ho, Wo, Cout), dtype=torch.float32 if mode == 2 else pdt, device=x.device)
    if residual is not None:
        assert residual.shape == (N, Ho, Wo, Cout)
    for v in (scale, bias):
        assert v is None or (v.dtype == torch.float32 and v.numel() == Cout and v.is_contiguous())
    if max(xt.numel(), w.numel(), out.numel() * (2 if mode == 2 else 1)) * 2 >= 0x7FF00000:
        raise ValueError("conv2d_sp: an operand of 2 GiB or more; the kernels use 32-bit buffer offsets: lower the "
                         "frame-stage batch (ClipEngine steps_per_batch) or split the call over M")
    K = R * S * Cw
    M = N * Ho * Wo
    _tok = None
    if _PROF is not None:
        _tok = _pb("igemm8_sp_" + ("h2" if h2 else ("x3" if x3 else "hi")), 2.0 * M * Cout * K,
                   xt.numel() * (2.0 if x3 or xt is x else 1.0) + w.numel() * 2.0 + out.numel() * out.element_size()
                   + (0 if residual is None else residual.t.numel() * 2.0), detail="%dx%dx%d (%dx%d)" % (M, Cout, K, R, S))
    nb = lib.mega_conv2d_nhwc_workspace_bytes(M, Cout, K) if mode == 2 and residual is None else 0
    ws = _ws(nb, x.device) if nb else None
    rc = lib.mega_conv2d_nhwc_sp_dt(_ptr(xt), ldi, 2 * C if kmul == 3 else 0, _ptr(w), _ptr(scale), _ptr(bias),
                                    None if residual is None else _ptr(residual.t), 0, _ptr(out), ldo, mode, N, H, W, Cw, Cout,
                                    R, S, stride, pad, dil, int(relu), _DT[pdt], _ptr(ws), nb, _stream())
    _pe(_tok)
    _lib.check(rc, "mega_conv2d_nhwc_sp_dt")
    return Planes(out, Cout) if mode == 1 else out


class X3Weight(object):
    """An f32 linear weight [Nout, K] as the split-precision operand [Wh | Wh | Wl] (bf16 [Nout, 3K]).  ops.linear /
    ops.linear_transposed take it in place of the f32 matrix and run x.W on the bf16 matrix cores to ~2^-16 with f32
    accumulation (the aggregation head's projections and stage FCs in conv_mode "x3").  `dtype` reports float32: callers
    that ask the weight for the dtype of its operands hand over f32 activations."""
    dtype = torch.float32

    def __init__(self, w32, device):
        w32 = w32.detach().float()
        self.shape = tuple(w32.shape)
        self.w3 = split_conv_weight_x3(w32.contiguous().view(w32.shape[0], 1, 1, w32.shape[1])).view(w32.shape[0], -1).to(device)
        self.device = self.w3.device


def linear_transposed_x3(w, x, ld):
    """linear_transposed for an X3Weight: out[n][m] = sum_k W[n][k] x[m][k] as f32 [Nout, ld], pad columns zero.  The weight
    rows are the GEMM's A operand ([Wh | Wh | Wl], plain bf16, K x 3), the activations its B operand as [xh | xl | xh]
    (ops.split_bf16x3; rows padded with zeros to a multiple of 8: the SP kernels store whole 16-byte vectors along m)."""
    Nout, K = w.shape
    M = x.shape[0]
    assert x.dtype == torch.float32 and x.shape[1] == K and ld >= (M + 7) // 8 * 8
    M8 = (M + 7) // 8 * 8
    xs = torch.empty((M8, 3 * K), dtype=torch.bfloat16, device=x.device)
    if M8 > M:
        xs[M:].zero_()
    lib = _lib.load()
    _gpu(x)
    rc = lib.mega_split_f32_to_bf16x3(_ptr(x.contiguous()), _ptr(xs), M, K, _stream())
    _lib.check(rc, "mega_split_f32_to_bf16x3")
    out = torch.empty((Nout, ld), dtype=torch.float32, device=x.device)
    if ld > M8:
        out[:, M8:].zero_()
    conv2d_sp(w.w3.view(Nout, 1, 1, 3 * K), xs.view(M8, 1, 1, 3 * K), out_mode="f32", x3=False, out=out)
    return out


def linear_sp(x, w3, bias=None, relu=False):
    """x: Planes [M,K]; w3 = split_conv_weight_x3 of [Nout,K] viewed [Nout,1,1,K] -> f32 [M,Nout] (x.W to ~2^-16); or, for
    float16 planes, w3 = split_conv_weight_h2 of it ([Nout, 2K]: the two-pass fp16 form)."""
    M, K = x.shape
    h2 = x.t.dtype == torch.float16
    y = conv2d_sp(Planes(x.t.view(M, 1, 1, 2 * K), K), w3.view(w3.shape[0], 1, 1, (2 if h2 else 3) * K), None, bias, relu=relu,
                  out_mode="f32", x3="h2" if h2 else True)
    return y.view(M, w3.shape[0])


def linear_transposed(w, x, ld, residual=None):
    """Returns (x @ w^T)^T (+ residual) laid out [Nout, ld] with ld >= M and the pad columns zero:
    out[n][m] = sum_k w[n][k] * x[m][k].  The weight matrix plays the GEMM 'A rows' role, so the
    projected values of one output feature are contiguous over rows m (keys).  residual: [Nout, ld] of the same dtype,
    added in the epilogue (the second pass of a split-weight projection, relation.project_v)."""
    if isinstance(w, X3Weight):
        assert residual is None
        return linear_transposed_x3(w, x, ld)
    _gpu(w, x, residual)
    lib = _lib.load()
    Nout, K = w.shape
    M = x.shape[0]
    assert x.shape[1] == K and ld >= M and w.dtype == x.dtype and w.is_contiguous() and x.is_contiguous()
    assert residual is None or (residual.shape == (Nout, ld) and residual.dtype == x.dtype and residual.is_contiguous())
    # (only the pad columns are zeroed: zero-filling the whole [Nout, ld] buffer -- 155 MB at stage 0 -- before the GEMM
    #  overwrote it was 11 fill launches / 0.08 ms per 20 key frames)
    out = torch.empty((Nout, ld), dtype=x.dtype, device=x.device)
    if ld > M:
        out[:, M:].zero_()
    _tok = None
    if _PROF is not None:
        _tok = _pb(_igemm_family(lib, Nout, M, K, x.dtype),
                   2.0 * Nout * M * K, (w.numel() + x.numel() + out.numel()) * x.element_size())
    rc = lib.mega_conv2d_nhwc(_ptr(w), _ptr(x), None, None, _ptr(residual), _ptr(out), Nout, 1, 1, K, M, 1, 1, 1, 0, 1, 0,
                              ld, ld, _dt(x), _dt(x), _stream())
    _pe(_tok)
    _lib.check(rc, "mega_conv2d_nhwc(transposed)")
    return out


def dff_warp_scale(feats, flow, scale):
    """feats NHWC [H,W,C] (key frame), flow [2,H,W] f32, scale [H,W,C] -> warp(feats, flow) * scale  [H,W,C]."""
    _gpu(feats, flow, scale)
    lib = _lib.load()
    H, W, C = feats.shape
    assert feats.is_contiguous() and flow.is_contiguous() and scale.is_contiguous()
    assert flow.dtype == torch.float32 and flow.shape == (2, H, W) and scale.shape == feats.shape and scale.dtype == feats.dtype
    out = torch.empty_like(feats)
    _tok = _pb("dff_warp", 0.0, 6.0 * feats.numel() * feats.element_size())
    rc = lib.mega_dff_warp_scale(_ptr(feats), _ptr(flow), _ptr(scale), _ptr(out), H, W, C, _dt(feats), _stream())
    _pe(_tok)
    _lib.check(rc, "mega_dff_warp_scale")
    return out


def fgfa_warp_aggregate(feats, flow, Cf, key, want_weights=False, order=None, flow_pos=None):
    """feats NHWC [T,H,W,Cf+Ce], flow [T,2,H,W] f32 -> aggregated key-frame features [H,W,Cf] (+ weights [T,H,W]).
    order (i32 [1 + T] on the device): feats is a ring of S >= T slots, order[0] = the key frame's slot, order[1 + t] = the
    slot of window position t (`key` is ignored); same bits as the call on the frames in window order.  flow is indexed by
    slot like feats ([S,2,H,W]) or, with flow_pos = the key frame's window position, by window position ([T,2,H,W]: exactly
    the window's pairs, in window order)."""
    _gpu(feats, flow, order)
    lib = _lib.load()
    S, H, W, C = feats.shape
    T = S if order is None else order.numel() - 1
    assert feats.is_contiguous() and flow.is_contiguous() and flow.dtype == torch.float32
    assert flow.shape == ((T if flow_pos is not None else S), 2, H, W) and (flow_pos is None or order is not None)
    out = torch.empty((H, W, Cf), dtype=feats.dtype, device=feats.device)
    wts = torch.empty((T, H, W), dtype=torch.float32, device=feats.device) if want_weights else None
    _tok = _pb("fgfa_warp", 0.0, 4.0 * feats.numel() * feats.element_size())
    if order is not None:
        assert order.dtype == torch.int32 and order.is_contiguous()
        if flow_pos is not None:
            rc = lib.mega_fgfa_warp_aggregate_ring_pos(_ptr(feats), _ptr(flow), _ptr(out), _ptr(wts), T, H, W, Cf, C - Cf,
                                                       _ptr(order), int(flow_pos), _dt(feats), _stream())
            _pe(_tok)
            _lib.check(rc, "mega_fgfa_warp_aggregate_ring_pos")
            return (out, wts) if want_weights else out
        rc = lib.mega_fgfa_warp_aggregate_ring(_ptr(feats), _ptr(flow), _ptr(out), _ptr(wts), T, H, W, Cf, C - Cf,
                                               _ptr(order), _dt(feats), _stream())
    else:
        rc = lib.mega_fgfa_warp_aggregate(_ptr(feats), _ptr(flow), _ptr(out), _ptr(wts), T, H, W, Cf, C - Cf, key,
                                          _dt(feats), _stream())
    _pe(_tok)
    _lib.check(rc, "mega_fgfa_warp_aggregate")
    return (out, wts) if want_weights else out


def fgfa_warp_aggregate_group(feats, flow, Cf, orders, key_pos):
    """fgfa_warp_aggregate(..., order=orders[g], flow_pos=key_pos) for the G key frames of a group in ONE launch: feats ring
    [S,H,W,Cf+Ce], flow f32 [G*T,2,H,W] (window order per key frame), orders i32 [G,1+T] -> [G,H,W,Cf]; same bits per key frame."""
    _gpu(feats, flow, orders)
    lib = _lib.load()
    S, H, W, C = feats.shape
    G, T1 = orders.shape
    T = T1 - 1
    assert feats.is_contiguous() and flow.is_contiguous() and flow.dtype == torch.float32 and tuple(flow.shape) == (G * T, 2, H, W)
    assert orders.dtype == torch.int32 and orders.is_contiguous()
    out = torch.empty((G, H, W, Cf), dtype=feats.dtype, device=feats.device)
    _tok = _pb("fgfa_warp", 0.0, 4.0 * G * T * H * W * C * feats.element_size())
    rc = lib.mega_fgfa_warp_aggregate_ring_pos_batched(_ptr(feats), _ptr(flow), _ptr(out), None, T, H, W, Cf, C - Cf, _ptr(orders),
                                                       int(key_pos), G, _dt(feats), _stream())
    _pe(_tok)
    _lib.check(rc, "mega_fgfa_warp_aggregate_ring_pos_batched")
    return out


def fgfa_pair_taps(refs, cur=None, order=None, dtype=torch.bfloat16):
    """FlowNetS's first-conv operand from the f32 NCHW frames in one kernel (mega_fgfa_pair_taps): refs [T,3,H,W] f32, the
    key frame `cur` [1,3,H,W] (one for all pairs) or [T,3,H,W] (one per pair), or cur=None with `order` (i32 on the device):
    the key frame is refs[order[0]] (the engine's image ring).  -> `dtype` [T, ceil(H/2) + 6, ceil(W/2), 64]:
    out[t, 3 + h, w, s * 8 + c] = avgpool2x2_ceil(cat([cur, refs[t]]))[h, w - 3 + s, c], zeros elsewhere."""
    _gpu(refs, cur, order)
    lib = _lib.load()
    T, C, H, W = refs.shape
    assert C == 3 and refs.dtype == torch.float32 and refs.stride(1) == H * W and refs.stride(2) == W and refs.stride(3) == 1
    assert dtype in _HALF and (cur is not None or order is not None)
    cs = 0
    if cur is not None:
        assert cur.dtype == torch.float32 and cur.shape[1:] == (3, H, W) and cur.shape[0] in (1, T)
        assert cur.stride(1) == H * W and cur.stride(2) == W and cur.stride(3) == 1
        cs = cur.stride(0) if cur.shape[0] == T else 0
    else:
        assert order.dtype == torch.int32 and order.is_contiguous()
    out = torch.empty((T, (H + 1) // 2 + 6, (W + 1) // 2, 64), dtype=dtype, device=refs.device)
    _tok = _pb("fgfa_pair_taps", 0.0, refs.numel() * 4.0 + out.numel() * 2.0)
    rc = lib.mega_fgfa_pair_taps(_ptr(refs), refs.stride(0), _ptr(cur), cs, _ptr(order), _ptr(out), T, H, W, _DT[dtype], _stream())
    _pe(_tok)
    _lib.check(rc, "mega_fgfa_pair_taps")
    return out


def avgpool2x2_ceil(x):
    """nn.AvgPool2d(2, 2, ceil_mode=True) on NHWC."""
    _gpu(x)
    lib = _lib.load()
    N, H, W, C = x.shape
    assert x.is_contiguous()
    out = torch.empty((N, (H + 1) // 2, (W + 1) // 2, C), dtype=x.dtype, device=x.device)
    rc = lib.mega_avgpool2x2_ceil_nhwc(_ptr(x), _ptr(out), N, H, W, C, _dt(x), _stream())
    _lib.check(rc, "mega_avgpool2x2_ceil_nhwc")
    return out

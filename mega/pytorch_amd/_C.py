"""Drop-in for the two ``mega_core._C`` entry points on MEGA's inference path, with the reference's
pybind signatures (mega_core/csrc/vision.cpp:9-12, csrc/nms.h:10-28, csrc/ROIAlign.h:11-25), backed by the
HIP kernels.  ``mega_core/layers/nms.py`` / ``layers/roi_align.py`` can bind to this module unchanged
(INTEGRATION.md shows the one-line patch).

Error behaviour mirrors the reference: non-device tensors raise RuntimeError (the reference's CUDA branch
asserts ``is_cuda``; its CPU branch has no counterpart here by design), an empty ``dets`` returns an empty
int64 tensor (csrc/nms.h:17-18).
"""
import torch

from . import ops


def nms(dets, scores, threshold):
    """at::Tensor nms(const at::Tensor& dets [N,4], const at::Tensor& scores [N], float threshold)
    -> int64 [K], kept ORIGINAL indices in ascending order (cuda/nms.cu:127-130)."""
    if not dets.is_cuda or not scores.is_cuda:
        raise RuntimeError("mega.pytorch_amd._C.nms: dets/scores must be HIP device tensors")
    if dets.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device="cpu")          # csrc/nms.h:17-18
    if dets.dim() != 2 or dets.shape[1] != 4 or scores.shape[0] != dets.shape[0]:
        raise RuntimeError("nms: dets must be [N,4] and scores [N]")
    return ops.nms(dets, scores, float(threshold), strict_gt=True)


def roi_align_forward(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio):
    """at::Tensor ROIAlign_forward(input [B,C,H,W], rois [K,5], float spatial_scale, int pooled_height,
    int pooled_width, int sampling_ratio) -> [K,C,pooled_height,pooled_width] (input dtype)."""
    if not input.is_cuda or not rois.is_cuda:
        raise RuntimeError("mega.pytorch_amd._C.roi_align_forward: input/rois must be HIP device tensors")
    if input.dim() != 4 or rois.dim() != 2 or rois.shape[1] != 5:
        raise RuntimeError("roi_align_forward: input must be [B,C,H,W] and rois [K,5]")
    x = input
    if x.dtype not in (torch.float32, torch.bfloat16):
        x = x.float()                                                       # amp.float_function (layers/roi_align.py:57)
    # a channels-last tensor is consumed in place as NHWC; a plain NCHW tensor is read with the NCHW index map
    nhwc = x.permute(0, 2, 3, 1)
    if nhwc.is_contiguous() and not x.is_contiguous():
        out = ops.roi_align(nhwc, rois.float().contiguous(), spatial_scale, (pooled_height, pooled_width),
                            sampling_ratio, in_nhwc=True, out_nhwc=False)
    else:
        out = ops.roi_align(x.contiguous(), rois.float().contiguous(), spatial_scale,
                            (pooled_height, pooled_width), sampling_ratio, in_nhwc=False, out_nhwc=False)
    return out

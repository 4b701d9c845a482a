"""Size distribution of the proposals ROIAlign sees on the bench's synthetic model (how many of them fit an LDS-staged patch?)
and the ROIAlign launch time on exactly those boxes."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from mega.pytorch_amd import ops

dev = torch.device("cuda", 0)
with torch.no_grad():
    cfg, model, sd = bench.build_model("R-101", "bfloat16", dev)
    clip = bench.make_clip(20, 600, 1000, dev)
    x = ops.preprocess_frames(clip[:20].contiguous(), tuple(cfg.INPUT.PIXEL_MEAN), True)
    from mega.pytorch_amd.modeling import _nhwc
    feat = _nhwc(model.backbone(x)[0])
    rpn = model.rpn
    rpn_out = rpn.head.run(feat)
    B, H, W, _ = feat.shape
    cell = next(iter(rpn.anchor_generator.cell_anchors)).to(dev).float().contiguous()
    for post in (300, 75):
        p, s, c = ops.rpn_select(rpn_out, cell, H, W, rpn.anchor_generator.strides[0], 6000, post, rpn.nms_thresh, rpn.min_size, 1000, 600, rpn.strict_gt)
        boxes = torch.cat([p[i, :int(c[i])] for i in range(B)], 0).float()
        w = (boxes[:, 2] - boxes[:, 0]) / 16.0
        h = (boxes[:, 3] - boxes[:, 1]) / 16.0
        # patch extent of the separable kernel: whole ROI spans floor(start) .. floor(end) + 1 feature pixels per axis
        pw = (torch.floor(boxes[:, 2] / 16.0) - torch.floor(boxes[:, 0] / 16.0) + 2).clamp(max=W)
        ph = (torch.floor(boxes[:, 3] / 16.0) - torch.floor(boxes[:, 1] / 16.0) + 2).clamp(max=H)
        px = pw * ph
        print("post %d: %d boxes; feature-pixel width median %.1f p90 %.1f, height median %.1f p90 %.1f" % (
            post, boxes.shape[0], w.median(), w.quantile(0.9), h.median(), h.quantile(0.9)))
        for cap in (36, 48, 64, 72, 96, 128, 144, 192, 256):
            print("   patch <= %3d pixels: %5.1f %% of the boxes" % (cap, 100.0 * (px <= cap).float().mean()))
    feat5 = torch.randn((B, H, W, 2048), device=dev).to(torch.bfloat16)
    p, s, c = ops.rpn_select(rpn_out, cell, H, W, 16, 6000, 300, rpn.nms_thresh, rpn.min_size, 1000, 600, rpn.strict_gt)
    rois = torch.cat([torch.cat([torch.full((int(c[i]), 1), float(i), device=dev), p[i, :int(c[i])].float()], 1) for i in range(B)], 0).contiguous()
    for _ in range(3):
        o = ops.roi_align(feat5, rois, 1.0 / 16, (7, 7), 0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        o = ops.roi_align(feat5, rois, 1.0 / 16, (7, 7), 0)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("roi_align on %d boxes x 2048 channels: %.3f ms (%.2f TB/s of output)" % (rois.shape[0], ms, o.numel() * 2 / ms / 1e9))

"""RPN selection timing probe: how rpn_select scales with pre / post NMS top-n on the synthetic model's RPN output."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from mega.pytorch_amd import ops

dev = torch.device("cuda", 0)
with torch.no_grad():
    cfg, model, sd = bench.build_model("R-101", "bfloat16", dev)
    clip = bench.make_clip(20, 600, 1000, dev)
    mean = tuple(cfg.INPUT.PIXEL_MEAN)
    x = ops.preprocess_frames(clip[:20].contiguous(), mean, True)
    from mega.pytorch_amd.modeling import _nhwc
    feat = _nhwc(model.backbone(x)[0])
    print("feat", tuple(feat.shape), feat.dtype)
    rpn = model.rpn
    rpn_out = rpn.head.run(feat)
    B, H, W, _ = feat.shape
    cell = next(iter(rpn.anchor_generator.cell_anchors)).to(dev).float().contiguous()

    def run(pre, post):
        return ops.rpn_select(rpn_out, cell, H, W, rpn.anchor_generator.strides[0], pre, post, rpn.nms_thresh, rpn.min_size, 1000, 600, rpn.strict_gt)
    for pre, post in [(6000, 300), (6000, 75), (6000, 2000), (3000, 300), (1000, 300), (6000, 10)]:
        p, s, c = run(pre, post)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run(pre, post)
        e1.record()
        torch.cuda.synchronize()
        print("pre %5d post %5d: %.3f ms per call; kept per image: %s" % (pre, post, e0.elapsed_time(e1) / 10, c.tolist()))

"""stem (7x7/2 conv + BN + ReLU) and max-pool timing on 20 frames of 600x1000, bf16; checks the bf16 stem against itself
run on the f32 direct kernel (loose) so that a broken build shows."""
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
    stem = model.backbone.body.stem
    pk = stem._packed(torch.bfloat16, dev)

    def t(fn, n=10):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    y = ops.stem(x, pk["w"], pk["s"], pk["b"], torch.bfloat16, w_n160=pk["w160"])
    yf = ops.stem(x, pk["w"], pk["s"], pk["b"], torch.float32)
    err = (y.float() - yf).abs().max().item()
    print("stem bf16 vs f32 direct: max abs diff %.4f (scale %.2f)" % (err, yf.abs().max().item()))
    print("stem      %.3f ms" % t(lambda: ops.stem(x, pk["w"], pk["s"], pk["b"], torch.bfloat16, w_n160=pk["w160"])))
    print("maxpool   %.3f ms" % t(lambda: ops.maxpool3x3s2(y)))
    print("stem.run  %.3f ms" % t(lambda: stem.run(x, torch.bfloat16)))
    ref = ops.maxpool3x3s2(y)
    out = stem.run(x, torch.bfloat16)
    print("stem.run == stem + maxpool:", torch.equal(ref, out), tuple(out.shape))

"""FlowNetS (trunk after flow_conv1) replayed from a hipGraph on 21 / 42 / 84 pairs: ms per 21 pairs.  Would batching the
FlowNetS of 2 / 4 key frames per graph pay?"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mega.pytorch_amd import config, modeling, synth  # noqa: E402
import mega.pytorch_amd.fgfa  # noqa: F401,E402

dev = torch.device("cuda:0")
cfg = config.get_cfg("R-101", "fgfa")
cfg.MODEL.DEVICE = "cuda:0"
cfg.DTYPE = "bfloat16"
model = modeling.build_detection_model(cfg)
model.load_state_dict(synth.make_fgfa_state_dict(blocks=(3, 4, 23), reduce_channel=False, seed=0))
model.to(dev)
fn = model.flownet
with torch.no_grad():
    for S in (21, 42, 84):
        frames = (torch.rand((S, 3, 600, 1000), device=dev) * 255.0 - 110.0)
        ab = fn.conv1_parts(frames, torch.bfloat16)
        for _ in range(2):
            fn.run_parts(ab, torch.bfloat16, key=0)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = fn.run_parts(ab, torch.bfloat16, key=0)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print("FlowNetS on %3d pairs: %.3f ms per replay = %.3f ms per 21 pairs" % (S, ms, ms * 21 / S), flush=True)
        del ab, frames, g, out

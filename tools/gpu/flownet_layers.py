"""Per-launch time of FlowNetS on the 21 pairs of a config-5 key frame (600 x 1000), and of the per-key-frame RPN / box head:
where do config 5's 4.2 ms go?  (ops.Profiler: HIP events around every wrapper call; torch's own kernels show up as gaps)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from mega.pytorch_amd import config, modeling, ops, synth  # noqa: E402
import mega.pytorch_amd.fgfa  # noqa: F401,E402

dev = torch.device("cuda:0")
cfg = config.get_cfg("R-101", "fgfa")
cfg.MODEL.DEVICE = "cuda:0"
cfg.DTYPE = "bfloat16"
model = modeling.build_detection_model(cfg)
model.load_state_dict(synth.make_fgfa_state_dict(blocks=(3, 4, 23), reduce_channel=False, seed=0))
model.to(dev)
fn = model.flownet
refs = (torch.rand((21, 3, 600, 1000), device=dev) * 255.0 - 110.0)
with torch.no_grad():
    for _ in range(2):
        fn.run_pairs(refs, refs[10:11], None, torch.bfloat16)
    torch.cuda.synchronize()
    p = ops.Profiler()
    ops.set_profiler(p)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn.run_pairs(refs, refs[10:11], None, torch.bfloat16)
    e1.record()
    torch.cuda.synchronize()
    ops.set_profiler(None)
    tot = 0.0
    for fam, fl, nb, a, det, b in p.items:
        ms = a.elapsed_time(b)
        tot += ms
        print("%-28s %-28s %.4f ms  %7.1f TF/s %6.2f TB/s" % (fam, det or "", ms, fl / ms / 1e9 if ms > 0 else 0, nb / ms / 1e9 if ms > 0 else 0))
    print("FlowNetS on 21 pairs: %.3f ms wall, %.3f ms inside wrapped kernels (the rest: torch cat / zero-stuffing / pads)" % (e0.elapsed_time(e1), tot))

# ---- the whole key frame in the reference call convention (steady state): every wrapped launch with its GEMM shape
clip = synth.make_clip(16, 600, 1000, seed=0).to(dev)
mean = tuple(cfg.INPUT.PIXEL_MEAN)


def frame(i):
    return ops.preprocess_frames(clip[i % 16:i % 16 + 1].contiguous(), mean, True)[0]


def step(i):
    if i == 0:
        images = {"cur": frame(0), "frame_category": 0, "seg_len": 100000, "ref_init": [frame(j) for j in range(1, 11)]}
    else:
        images = {"cur": frame(i), "ref": [frame(i + 10)], "frame_category": 1, "seg_len": 100000}
    return model(images)


with torch.no_grad():
    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    p = ops.Profiler()
    ops.set_profiler(p)
    step(3)
    torch.cuda.synchronize()
    ops.set_profiler(None)
    print("---- one key frame, reference call convention (includes the single-frame backbone + EmbedNet)")
    tot = 0.0
    for fam, fl, nb, a, det, b in p.items:
        ms = a.elapsed_time(b)
        tot += ms
        print("%-28s %-28s %.4f ms  %7.1f TF/s %6.2f TB/s" % (fam, det or "", ms, fl / ms / 1e9 if ms > 0 else 0, nb / ms / 1e9 if ms > 0 else 0))
    print("sum of wrapped kernels: %.3f ms" % tot)

"""How does the CPU baseline bench.py reports (oracle/mega_oracle.py, kind "port") compare with the UNMODIFIED reference
run through oracle/ref_shim.py on the same host cores?  Only possible where /root/reference exists (this build
container, not the GPU box).  R-101 MEGA, 600x1000, same seeded clip / weights as bench.py: cold start + 2 key frames.

  python tools/cpu_port_vs_reference.py [threads]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_shim  # noqa: E402
from mega.pytorch_amd import synth  # noqa: E402
from oracle import mega_oracle as mo  # noqa: E402


def main():
    nthr = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 8)
    torch.set_num_threads(nthr)
    H, W, T, nkey = 600, 1000, 20, 3
    sd = synth.make_state_dict(blocks=(3, 4, 23), reduce_channel=False, global_res_stage=1, seed=0)
    clip = synth.make_clip(8, H, W, seed=0)
    frames = synth.preprocess_cpu(clip[torch.arange(T) % 8])
    _, gfor = mo.global_frame_schedule(T, 10, seed=0)
    # ---- the port
    orc = mo.MegaOracle(sd, mo.OracleCfg(blocks=(3, 4, 23), reduce_channel=False, global_res_stage=1, nms_strict_gt=False))
    t_port = []
    with torch.no_grad():
        for idx in range(nkey):
            t0 = time.perf_counter()
            orc.forward_frame(frames[idx:idx + 1], 0 if idx == 0 else 1, ref_l=frames[min(T - 1, idx + 12)][None],
                              ref_g=[frames[g][None] for g in gfor(idx)], seg_len=T, frame_loader=lambda i: frames[i][None])
            t_port.append(time.perf_counter() - t0)
    # ---- the reference itself
    cfg = ref_shim.make_cfg("configs/MEGA/vid_R_101_C4_MEGA_1x.yaml")
    model = ref_shim.build_model(cfg)
    model.load_state_dict(sd, strict=True)
    import mega_core.modeling.detector.generalized_rcnn_mega as gm

    class _FakeImg(object):
        def __init__(self, i): self.i = i
        def convert(self, m): return self

    class _FakeImage(object):
        @staticmethod
        def open(path): return _FakeImg(int(path))
    gm.Image = _FakeImage
    t_ref = []
    with torch.no_grad():
        for idx in range(nkey):
            images = {"cur": frames[idx], "ref_l": [frames[min(T - 1, idx + 12)]], "ref_g": [frames[g] for g in gfor(idx)],
                      "frame_category": 0 if idx == 0 else 1, "seg_len": T, "pattern": "%d", "img_dir": "%s",
                      "transforms": lambda im: frames[im.i]}
            t0 = time.perf_counter()
            model(images)
            t_ref.append(time.perf_counter() - t0)
    print("threads %d  port (oracle): cold %.1f s, steady %s s   reference (shim): cold %.1f s, steady %s s" % (
        nthr, t_port[0], ["%.2f" % t for t in t_port[1:]], t_ref[0], ["%.2f" % t for t in t_ref[1:]]))
    print("steady ratio port/reference = %.2f" % (np.mean(t_port[1:]) / np.mean(t_ref[1:])))


if __name__ == "__main__":
    main()

"""How does the CPU baseline bench.py reports (oracle/mega_oracle.py, kind "port") compare with the UNMODIFIED reference
run through oracle/ref_shim.py on the same host cores?  Only possible where /root/reference exists (this build
container, not the GPU box).  R-101 MEGA, 600x1000, same seeded clip / weights as bench.py: cold start + 2 key frames.

  python tools/cpu_port_vs_reference.py [threads] [--json out.json]

Also times BASELINE configs[0] -- the single-frame R-50-C4 detector (configs/vid_R_50_C4_1x.yaml), one 600x1000 frame,
"CPU-only PyTorch reference path" -- with the unmodified reference and with the port (oracle BaseOracle).  With --json
the numbers are written as a small record (committed under profiles/: bench.py and tools/bench_configs.py quote the
port / reference ratio from it next to the port they time on the GPU box, where /root/reference does not exist).
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


def config1(nthr):
    """BASELINE configs[0]: single-frame R-50-C4, 600x1000, reference vs port, seconds per frame"""
    H, W = 600, 1000
    sd = {k: v for k, v in synth.make_fgfa_state_dict(seed=0).items() if not k.startswith(("flownet.", "embednet."))}
    frames = synth.preprocess_cpu(synth.make_clip(4, H, W, seed=0))
    orc = mo.BaseOracle(sd, mo.OracleCfg(blocks=(3, 4, 6), reduce_channel=True, nms_strict_gt=False))
    t_port, t_ref = [], []
    with torch.no_grad():
        for i in range(4):
            t0 = time.perf_counter()
            orc.forward_frame(frames[i:i + 1])
            t_port.append(time.perf_counter() - t0)
    cfg = ref_shim.make_cfg("configs/vid_R_50_C4_1x.yaml")
    model = ref_shim.build_model(cfg)
    model.load_state_dict(sd, strict=True)
    with torch.no_grad():
        for i in range(4):
            t0 = time.perf_counter()
            model(frames[i])
            t_ref.append(time.perf_counter() - t0)
    return float(np.mean(t_port[1:])), float(np.mean(t_ref[1:]))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    nthr = int(args[0]) if args else (os.cpu_count() or 8)
    json_out = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    torch.set_num_threads(nthr)
    H, W, T, nkey = 600, 1000, 20, 4
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
    c1_port, c1_ref = config1(nthr)
    print("config 1 (single-frame R-50-C4, 600x1000): port %.2f s/frame, reference %.2f s/frame" % (c1_port, c1_ref))
    if json_out:
        import json
        rec = {"host_threads": nthr, "cpu_count": os.cpu_count(), "frame": "%dx%d" % (W, H),
               "mega_r101": {"port_cold_s": round(t_port[0], 2), "reference_cold_s": round(t_ref[0], 2),
                             "port_steady_s": round(float(np.mean(t_port[1:])), 3),
                             "reference_steady_s": round(float(np.mean(t_ref[1:])), 3),
                             "port_over_reference_time": round(float(np.mean(t_port[1:]) / np.mean(t_ref[1:])), 3),
                             "note": "steady = key frames 1..3 (memory not yet full: both sides run the same pools)"},
               "config1_single_frame_r50": {"port_s_per_frame": round(c1_port, 3), "reference_s_per_frame": round(c1_ref, 3),
                                            "reference_fps": round(1.0 / c1_ref, 3), "port_fps": round(1.0 / c1_port, 3),
                                            "port_over_reference_time": round(c1_port / c1_ref, 3)},
               "how": "tools/cpu_port_vs_reference.py in the build container (the unmodified reference through "
                      "oracle/ref_shim.py vs oracle/mega_oracle.py, same weights / frames / threads)"}
        with open(json_out, "w") as f:
            json.dump(rec, f, indent=1)


if __name__ == "__main__":
    main()

"""Generates the oracle fixtures of the BASELINE configurations that are not the headline, AT THEIR STATED SIZES
(600x1000 frames), through oracle/mega_oracle.py (torch-CPU fp32, pinned to the unmodified reference by
tests/test_oracle_golden.py):

  oracle_cfg2_r50_600x1000.npz    BASELINE configs[1]: MEGA R-50, fp32, "10 local + 10 global": ALL_FRAME_INTERVAL 11,
                                  KEY_FRAME_LOCATION 5, GLOBAL.SIZE 10 (memory deques of 11) -- 14 key frames, so the
                                  window, the global pool and the memory wrap (roi_box_feature_extractors.py:657-688)
  oracle_cfg5_fgfa_r101_600x1000.npz   BASELINE configs[4]: FGFA R-101, 21-frame window (ALL_FRAME_INTERVAL 21,
                                  KEY_FRAME_LOCATION 10): 24 key frames of a 40-frame clip -- cold start, the window sliding
                                  and WRAPPED (every slot of the 21-frame ring re-used from key frame 11 on); traces kept for
                                  key frames 0, 1, 22, 23 (detector/generalized_rcnn_fgfa.py:144-219)
  oracle_cfg1_base_r50_600x1000.npz    BASELINE configs[0] at its stated size: the single-frame R-50-C4 detector
                                  (configs/vid_R_50_C4_1x.yaml, detector/generalized_rcnn.py:33-65), two 600x1000 frames

  python tests/golden/make_oracle_configs.py [cfg1] [cfg2] [cfg5]       (minutes of CPU; the GPU tests only read the .npz)
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from mega.pytorch_amd import synth  # noqa: E402
from oracle import mega_oracle as mo  # noqa: E402

CFG2 = dict(H=600, W=1000, T=24, nkey=14, seed_w=0, seed_clip=0, unique=8, global_seed=0, afi=11, key=5, gsize=10)
KEEP2 = (0, 1, 12, 13)
CFG5 = dict(H=600, W=1000, T=40, nkey=24, seed_w=0, seed_clip=0, unique=8, afi=21, key=10)
KEEP5 = (0, 1, 22, 23)            # cold start, first steady key frame, and two after the 21-frame window has wrapped
CFG1 = dict(H=600, W=1000, T=2, seed_w=0, seed_clip=0)


def inputs_cfg2(c=CFG2):
    sd = synth.make_state_dict(blocks=(3, 4, 6), reduce_channel=True, global_res_stage=0, seed=c["seed_w"])
    clip = synth.make_clip(c["unique"], c["H"], c["W"], seed=c["seed_clip"])
    clip = clip[torch.arange(c["T"]) % c["unique"]]
    _, gfor = mo.global_frame_schedule(c["T"], c["gsize"], seed=c["global_seed"])
    return sd, clip, gfor


def inputs_cfg5(c=CFG5):
    sd = synth.make_fgfa_state_dict(blocks=(3, 4, 23), reduce_channel=False, seed=c["seed_w"])
    clip = synth.make_clip(c["unique"], c["H"], c["W"], seed=c["seed_clip"])
    clip = clip[torch.arange(c["T"]) % c["unique"]]
    return sd, clip


def make_cfg2():
    c = CFG2
    sd, clip, gfor = inputs_cfg2(c)
    frames = synth.preprocess_cpu(clip)
    T = c["T"]
    orc = mo.MegaOracle(sd, mo.OracleCfg(blocks=(3, 4, 6), reduce_channel=True, global_res_stage=0, nms_strict_gt=True,
                                         all_frame_interval=c["afi"], key_frame_location=c["key"], global_size=c["gsize"]))
    out, mem_len = {}, []
    ahead = c["afi"] - c["key"] - 1
    for idx in range(c["nkey"]):
        orc.trace = {}
        t0 = time.time()
        with torch.no_grad():
            b, s, l = orc.forward_frame(frames[idx:idx + 1], 0 if idx == 0 else 1, ref_l=frames[min(T - 1, idx + ahead)][None],
                                        ref_g=[frames[g][None] for g in gfor(idx)], seg_len=T,
                                        frame_loader=lambda i: frames[i][None])
        mem_len.append(min(len(q["rois"]) for q in orc.mem_queue))
        if idx in KEEP2:
            out["boxes%d" % idx], out["scores%d" % idx], out["labels%d" % idx] = b.numpy(), s.numpy(), l.numpy()
            out["logits%d" % idx] = orc.trace["logits"].numpy()
            out["deltas%d" % idx] = orc.trace["deltas"].numpy()[:, :8]
            out["proposals%d" % idx] = orc.trace["proposals"].numpy()
        print("cfg2 key frame %d: %d detections, memory %d, %.1fs" % (idx, b.shape[0], mem_len[-1], time.time() - t0), flush=True)
    for k, v in c.items():
        out["cfg_" + k] = np.int64(v)
    out["keep"] = np.asarray(KEEP2, dtype=np.int64)
    out["mem_len"] = np.asarray(mem_len, dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "oracle_cfg2_r50_600x1000.npz"), **out)


def make_cfg5():
    c = CFG5
    sd, clip = inputs_cfg5(c)
    frames = synth.preprocess_cpu(clip)
    T = c["T"]
    orc = mo.FgfaOracle(sd, mo.OracleCfg(blocks=(3, 4, 23), reduce_channel=False, nms_strict_gt=True),
                        all_frame_interval=c["afi"], key_frame_location=c["key"])
    out = {}
    ahead = c["afi"] - c["key"] - 1
    for idx in range(c["nkey"]):
        orc.trace = {}
        t0 = time.time()
        with torch.no_grad():
            b, s, l = orc.forward_frame(frames[idx:idx + 1], 0 if idx == 0 else 1, ref=frames[min(T - 1, idx + ahead)][None],
                                        seg_len=T, frame_loader=lambda i: frames[i][None])
        print("cfg5 key frame %d: %d detections, %d proposals, %.1fs" % (idx, b.shape[0], orc.trace["proposals"].shape[0],
                                                                          time.time() - t0), flush=True)
        if idx not in KEEP5:
            continue
        out["boxes%d" % idx], out["scores%d" % idx], out["labels%d" % idx] = b.numpy(), s.numpy(), l.numpy()
        out["logits%d" % idx] = orc.trace["logits"].numpy()
        out["proposals%d" % idx] = orc.trace["proposals"].numpy()
        out["flow%d" % idx] = orc.trace["flow"].numpy().astype(np.float32)              # [21,2,38,63]
        out["feats%d" % idx] = orc.trace["feats"].numpy()[0, ::64].astype(np.float32)    # every 64th channel of the map
        out["weights%d" % idx] = orc.trace["weights"].numpy().astype(np.float32)
    for k, v in c.items():
        out["cfg_" + k] = np.int64(v)
    out["keep"] = np.asarray(KEEP5, dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "oracle_cfg5_fgfa_r101_600x1000.npz"), **out)


def inputs_cfg1(c=CFG1):
    sd = synth.make_fgfa_state_dict(blocks=(3, 4, 6), reduce_channel=True, seed=c["seed_w"])
    sd = {k: v for k, v in sd.items() if not k.startswith(("flownet.", "embednet."))}
    clip = synth.make_clip(c["T"], c["H"], c["W"], seed=c["seed_clip"])
    return sd, clip


def make_cfg1():
    c = CFG1
    sd, clip = inputs_cfg1(c)
    frames = synth.preprocess_cpu(clip)
    orc = mo.BaseOracle(sd, mo.OracleCfg(blocks=(3, 4, 6), reduce_channel=True, nms_strict_gt=True))
    out = {}
    for idx in range(c["T"]):
        orc.trace = {}
        t0 = time.time()
        with torch.no_grad():
            b, s, l = orc.forward_frame(frames[idx:idx + 1])
        out["boxes%d" % idx], out["scores%d" % idx], out["labels%d" % idx] = b.numpy(), s.numpy(), l.numpy()
        out["logits%d" % idx] = orc.trace["logits"].numpy()
        out["proposals%d" % idx] = orc.trace["proposals"].numpy()
        print("cfg1 frame %d: %d detections, %d proposals, %.1fs" % (idx, b.shape[0], orc.trace["proposals"].shape[0],
                                                                     time.time() - t0), flush=True)
    for k, v in c.items():
        out["cfg_" + k] = np.int64(v)
    np.savez_compressed(os.path.join(HERE, "oracle_cfg1_base_r50_600x1000.npz"), **out)


if __name__ == "__main__":
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    what = sys.argv[1:] or ["cfg1", "cfg2", "cfg5"]
    if "cfg1" in what:
        make_cfg1()
    if "cfg2" in what:
        make_cfg2()
    if "cfg5" in what:
        make_cfg5()

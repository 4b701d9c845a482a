"""Generates tests/golden/oracle_r101_600x1000.npz: the BENCHMARKED configuration (MEGA R-101-C4, 600x1000 frames,
25 local / 10 global / 25 memory, BASELINE.json configs[2]) through oracle/mega_oracle.py (torch-CPU fp32; itself
pinned to the unmodified reference by tests/test_oracle_golden.py) for key frame 0 (cold start: 13 local + 10 global
frames) and NKEY-1 steady key frames, on the seeded synthetic clip / weights bench.py uses.  NKEY = 28: from key frame
25 on all three 25-entry memory deques are full and evicting (roi_box_feature_extractors.py:657-688), the local window
and the global pool wrapped long before -- the regime bench.py times.  Full traces (logits, deltas, proposals,
detections) are kept for the key frames in KEEP (the first three and the last three); `mem_len` records every frame's
memory fill.

  python tests/golden/make_oracle_r101.py          (~10 minutes of CPU on 8 cores; the GPU tests only read the .npz)

The GPU parity tests (tests/test_e2e_gpu.py::test_r101_600x1000_*) run the HIP path on the SAME seeded inputs in f32
(north_star tolerance 1e-3) and in bf16 (the performance mode: proposal-set overlap, matched-box deltas and logit
error percentiles are reported and bounded).  NMS comparisons use `IoU > thr` (the CUDA semantics north_star names,
csrc/cuda/nms.cu:60), the HIP path's default.
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

CFG = dict(H=600, W=1000, T=44, nkey=28, seed_w=0, seed_clip=0, unique=8, global_seed=0)
KEEP = (0, 1, 2, 25, 26, 27)


def inputs(c=CFG):
    sd = synth.make_state_dict(blocks=(3, 4, 23), reduce_channel=False, global_res_stage=1, seed=c["seed_w"])
    clip = synth.make_clip(c["unique"], c["H"], c["W"], seed=c["seed_clip"])
    clip = clip[torch.arange(c["T"]) % c["unique"]]
    _, gfor = mo.global_frame_schedule(c["T"], 10, seed=c["global_seed"])
    return sd, clip, gfor


def main():
    c = CFG
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    sd, clip, gfor = inputs(c)
    frames = synth.preprocess_cpu(clip)
    T = c["T"]
    orc = mo.MegaOracle(sd, mo.OracleCfg(blocks=(3, 4, 23), reduce_channel=False, global_res_stage=1, nms_strict_gt=True))
    out, mem_len = {}, []
    for idx in range(c["nkey"]):
        orc.trace = {}
        t0 = time.time()
        with torch.no_grad():
            b, s, l = orc.forward_frame(frames[idx:idx + 1], 0 if idx == 0 else 1, ref_l=frames[min(T - 1, idx + 12)][None],
                                        ref_g=[frames[g][None] for g in gfor(idx)], seg_len=T,
                                        frame_loader=lambda i: frames[i][None])
        mem_len.append(min(len(q["rois"]) for q in orc.mem_queue))
        if idx in KEEP:
            out["boxes%d" % idx], out["scores%d" % idx], out["labels%d" % idx] = b.numpy(), s.numpy(), l.numpy()
            out["logits%d" % idx] = orc.trace["logits"].numpy()
            out["deltas%d" % idx] = orc.trace["deltas"].numpy()[:, :8]
            out["proposals%d" % idx] = orc.trace["proposals"].numpy()
            out["prop_scores%d" % idx] = orc.trace["prop_scores"].numpy()
            out["x%d" % idx] = orc.trace["x"].numpy()[:32]
        print("key frame %d: %d detections, %d proposals, %.1fs" % (idx, b.shape[0], orc.trace["proposals"].shape[0],
                                                                     time.time() - t0), flush=True)
    for k, v in c.items():
        out["cfg_" + k] = np.int64(v)
    out["keep"] = np.asarray(KEEP, dtype=np.int64)
    out["mem_len"] = np.asarray(mem_len, dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "oracle_r101_600x1000.npz"), **out)


if __name__ == "__main__":
    main()

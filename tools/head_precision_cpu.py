#!/usr/bin/env python
"""CPU emulation of run (H) of tests/test_e2e_gpu.py::test_r101_bf16_attribution: exact-f32 frame stage + the
aggregation head in the product's bf16 mode, on the oracle-backed CPU twins of the kernels (tests/cpu_ops.py: f32
arithmetic on the bf16-rounded operands, outputs rounded to the dtype the HIP kernel writes -- the same hand-off
roundings as the MFMA path, a different summation order).  Compares the logits / detections of the kept key frames with
tests/golden/oracle_r101_600x1000.npz for several head variants, so that precision work on the head can be judged
before GPU time is spent.  The f32 frame-stage records are computed once and cached in /tmp.

  python tools/head_precision_cpu.py [--nkey 28] [--variants bf16,f32stream]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cpu_ops  # noqa: E402
from mega.pytorch_amd import config, engine, modeling, ops  # noqa: E402
from test_e2e_gpu import _bf16_metrics, _fmt, _r101_fixture  # noqa: E402


def install_twins():
    for name in cpu_ops.ALL:
        setattr(ops, name, getattr(cpu_ops, name))
    for name in getattr(cpu_ops, "EXTRA", []):
        setattr(ops, name, getattr(cpu_ops, name))


def build(dtype, sd, **flags):
    cfg = config.get_cfg("R-101")
    cfg.DTYPE = dtype
    cfg.MODEL.DEVICE = "cpu"
    cfg.NMS_STRICT_GT = True
    for k, v in flags.items():
        setattr(cfg, k, v)
    m = modeling.build_detection_model(cfg)
    m.load_state_dict(sd)
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nkey", type=int, default=28)
    ap.add_argument("--variants", default="bf16,f32stream")
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--cache", default="/tmp/head_precision_records.pt")
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    install_twins()
    d, gen = _r101_fixture()
    sd, clip, gfor = gen.inputs()
    nkey, T = min(args.nkey, int(d["cfg_nkey"])), int(d["cfg_T"])
    keep = [int(k) for k in d["keep"] if int(k) < nkey]
    frame = build("float32", sd)
    memo = torch.load(args.cache) if os.path.exists(args.cache) else {}
    dirty = [False]

    for variant in args.variants.split(","):
        flags = {}
        if variant == "f32stream":
            flags["HEAD_STREAM"] = "float32"
        elif variant == "bf16":
            flags["HEAD_STREAM"] = "bfloat16"
        else:
            raise SystemExit("unknown variant " + variant)
        head = build("bfloat16", sd, **flags)
        eng = engine.ClipEngine(head, steps_per_batch=4, keep_logits=True, frame_model=frame)
        orig = eng.records_async

        def cached(clip_, jobs, on_counts=None, _orig=orig):
            key = tuple((int(j[0]), int(j[1])) for j in jobs)
            if key not in memo:
                t0 = time.time()
                memo[key] = _orig(clip_, jobs, None)
                dirty[0] = True
                print("  frame stage of %d frames: %.1fs" % (len(jobs), time.time() - t0), flush=True)
            h = memo[key]
            return {"st": {k: (v.clone() if torch.is_tensor(v) else v) for k, v in h["st"].items()}}
        eng.records_async = cached
        t0 = time.time()
        with torch.no_grad():
            dets = eng.run(clip, T, gfor, first=0, last=nkey)
        if dirty[0]:
            torch.save(memo, args.cache)
            dirty[0] = False
        print("== variant %s (%.0fs)" % (variant, time.time() - t0))
        for idx in keep:
            m = _bf16_metrics(d, idx, eng.key_boxes_log[idx].numpy(), eng.logits_log[idx].numpy(), dets[idx])
            print(_fmt(variant, idx, m), flush=True)


if __name__ == "__main__":
    main()

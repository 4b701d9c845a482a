"""Extra bench lines for the other BASELINE.json configurations (the headline, configs[2], is bench.py's default):

  python tools/bench_configs.py --config 1     single-frame R-50-C4 detector, 600x1000 (configs/vid_R_50_C4_1x.yaml)
  python tools/bench_configs.py --config 2     MEGA R-50 fp32, 10 local + 10 global frames (11-frame window)
  python tools/bench_configs.py --config 5     FGFA R-101, 21-frame window, 600x1000, bf16

Each prints ONE JSON line (same field names as bench.py; `roofline` describes that configuration's own dominant
kernel).  Synthetic clip and seeded calibrated weights, frames resident in HBM as uint8, preprocessing included.
Configs 1 and 5 are driven in the reference's call convention, one key frame per model(...) call (the reference has
no batching there either); config 2 runs through ClipEngine like the headline.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def log(msg):
    sys.stderr.write("[bench_configs] %s\n" % msg)
    sys.stderr.flush()


def families(ops, fn, steps):
    p = ops.Profiler()
    ops.set_profiler(p)
    fn()
    summ = p.summary()
    ops.set_profiler(None)
    fam = {}
    for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
        fam[k] = {"ms_per_step": round(v["ms"] / steps, 4), "launches_per_step": round(v["launches"] / steps, 1),
                  "tflops": round(v["flops"] / (v["ms"] * 1e9), 2) if v["ms"] > 0 else 0.0,
                  "gbps": round(v["bytes"] / (v["ms"] * 1e6), 1) if v["ms"] > 0 else 0.0}
    return fam, summ


def timed(fn, steps, min_seconds=1.0, max_blocks=40):
    """blocks of `steps` calls of fn(i), each bracketed by synchronize; median block."""
    blocks, i = [], 0
    while True:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn(i)
            i += 1
        torch.cuda.synchronize()
        blocks.append(time.perf_counter() - t0)
        if sum(blocks) >= min_seconds or len(blocks) >= max_blocks:
            break
    s = sorted(blocks)
    med = s[len(s) // 2] if len(s) % 2 else 0.5 * (s[len(s) // 2 - 1] + s[len(s) // 2])
    return med, blocks


def config1(args, dev):
    from mega.pytorch_amd import config, modeling, ops, synth
    import mega.pytorch_amd.fgfa  # noqa: F401  (registers GeneralizedRCNN)
    cfg = config.get_cfg("R-50", "base")
    cfg.DTYPE = args.dtype
    cfg.MODEL.DEVICE = str(dev)
    sd = {k: v for k, v in synth.make_fgfa_state_dict(seed=0).items() if not k.startswith(("flownet.", "embednet."))}
    model = modeling.build_detection_model(cfg)
    model.load_state_dict(sd)
    model.to(dev)
    clip = synth.make_clip(8, args.height, args.width, seed=0).to(dev)
    mean = tuple(cfg.INPUT.PIXEL_MEAN)

    def step(i):
        x = ops.preprocess_frames(clip[i % 8:i % 8 + 1].contiguous(), mean, True)
        return model(x[0])
    for i in range(4):
        step(i)
    med, blocks = timed(step, args.steps)
    fam, _ = families(ops, lambda: [step(i) for i in range(4)], 4)
    # ---- the clip engine (fgfa.BaseClipEngine): the backbone on 20 frames per launch chain, the box head's graph per frame
    from mega.pytorch_amd import fgfa as fgfa_mod
    Lv = 120 + 40 * args.steps
    video = ops.preprocess_frames(clip.contiguous(), mean, True)[torch.arange(Lv, device=dev) % 8].contiguous()
    eng = fgfa_mod.BaseClipEngine(model, group=20, batch_head=args.batch_head != 0)
    eng.run(video, first=0, last=80)
    eb, pos = [], 80
    while pos + 40 <= Lv and sum(eb) < 1.0:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.run(video, first=pos, last=pos + 40, sync_every=40)
        torch.cuda.synchronize()
        eb.append(time.perf_counter() - t0)
        pos += 40
    em = sorted(eb)[len(eb) // 2]
    engine_line = {"fps": round(40 / em, 2), "ms_per_frame": round(1e3 * em / 40, 3), "blocks_of_40_frames_ms": [round(1e3 * b, 2) for b in eb[:8]],
                   "driver": "fgfa.BaseClipEngine: the backbone on 20 frames per launch chain (graph A), RPN + box head + post-processing of "
                             "those 20 frames as one batched launch chain (graph B) on a second stream beside the next group's backbone "
                             "(identical detections to the per-call path: test_base_engine_equals_model)"}
    cpu = None
    if not args.no_cpu_baseline:
        # BASELINE configs[0] IS the CPU reference run: the port (oracle BaseOracle = the reference's GeneralizedRCNN
        # restated on torch-CPU) timed on this box's host cores, next to the unmodified reference's time measured in the
        # build container (profiles/r03_cpu_port_vs_reference.json; /root/reference does not exist here)
        import time as _t
        from oracle import mega_oracle as mo
        cores = min(len(os.sched_getaffinity(0)), 64)
        torch.set_num_threads(cores)
        orc = mo.BaseOracle({k: v.cpu() for k, v in sd.items()}, mo.OracleCfg(blocks=(3, 4, 6), reduce_channel=True,
                                                                              nms_strict_gt=False))
        frames = synth.preprocess_cpu(clip[:4].cpu())
        ts = []
        for i in range(4):
            t0 = _t.perf_counter()
            orc.forward_frame(frames[i:i + 1])
            ts.append(_t.perf_counter() - t0)
        s_frame = sum(ts[1:]) / 3
        cpu = {"value": round(1.0 / s_frame, 3), "unit": "frames/s", "cores": cores, "kind": "port",
               "sample": "oracle BaseOracle, 3 frames of %dx%d after one warm-up frame (%.2f s each)" % (args.width, args.height, s_frame)}
        rec = os.path.join(ROOT, "profiles", "r03_cpu_port_vs_reference.json")
        if os.path.exists(rec):
            r = json.load(open(rec))["config1_single_frame_r50"]
            cpu["vs_unmodified_reference"] = {"port_over_reference_time": r["port_over_reference_time"],
                                              "reference_fps_build_container": r["reference_fps"],
                                              "source": "profiles/r03_cpu_port_vs_reference.json"}
    return {"metric": "frames/sec single-frame R-50-C4 detector, %dx%d frames" % (args.width, args.height),
            "cpu_baseline": cpu,
            "value": round(args.steps / med, 2), "unit": "frames/s", "ms_per_step": round(1e3 * med / args.steps, 3),
            "config": {"workload": "GeneralizedRCNN R-50-C4 + ResNetConv52MLPFeatureExtractor, 300 proposals, one frame "
                                   "per call with a host read of the detection count (BASELINE configs[0])",
                       "clip_engine": engine_line},
            "kernel_families": fam, "blocks_ms": [round(1e3 * b, 2) for b in blocks]}


def config5(args, dev):
    from mega.pytorch_amd import config, modeling, ops, synth
    import mega.pytorch_amd.fgfa  # noqa: F401
    cfg = config.get_cfg("R-101", "fgfa")
    cfg.DTYPE = args.dtype
    cfg.MODEL.DEVICE = str(dev)
    cfg.MODEL.VID.FGFA.ALL_FRAME_INTERVAL, cfg.MODEL.VID.FGFA.KEY_FRAME_LOCATION = 21, 10
    cfg.MODEL.VID.FGFA.MIN_OFFSET, cfg.MODEL.VID.FGFA.MAX_OFFSET = -10, 10
    sd = synth.make_fgfa_state_dict(blocks=(3, 4, 23), reduce_channel=False, seed=0)
    model = modeling.build_detection_model(cfg)
    model.load_state_dict(sd)
    model.to(dev)
    Tc = 16
    clip = synth.make_clip(Tc, args.height, args.width, seed=0).to(dev)
    mean = tuple(cfg.INPUT.PIXEL_MEAN)
    T = 100000

    def frame(i):
        return ops.preprocess_frames(clip[i % Tc:i % Tc + 1].contiguous(), mean, True)[0]

    def step(i):
        if i == 0:
            images = {"cur": frame(0), "frame_category": 0, "seg_len": T, "ref_init": [frame(j) for j in range(1, 11)]}
        else:
            images = {"cur": frame(i), "ref": [frame(i + 10)], "frame_category": 1, "seg_len": T}
        return model(images)
    for i in range(4):
        step(i)
    state = {"i": 4}

    def steady(_):
        step(state["i"])
        state["i"] += 1
    if args.skip_call_convention:
        med_call, blocks_call = float("inf"), []
    else:
        med_call, blocks_call = timed(steady, args.steps, min_seconds=0.5)
    # ---- the clip engine (fgfa.FgfaClipEngine): batched features, ring window, one hipGraph per key frame
    from mega.pytorch_amd import fgfa as fgfa_mod
    L = 64 + 40 * args.steps
    frames = torch.cat([frame(i)[None] for i in range(Tc)], dim=0)
    video = frames[torch.arange(L, device=dev) % Tc].contiguous()
    engine = fgfa_mod.FgfaClipEngine(model, lookahead=args.fgfa_lookahead, group=args.fgfa_group, pipeline=not args.fgfa_no_pipeline,
                                     lanes=args.lanes or 1, batch_head=args.batch_head != 0)
    engine.run(video, first=0, last=1 + 3 * 20)            # cold start + the eager / capture / replay warm-up
    pos = [1 + 3 * 20]
    blocks = []
    while True:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        engine.run(video, first=pos[0], last=pos[0] + args.steps, sync_every=args.steps)
        torch.cuda.synchronize()
        blocks.append(time.perf_counter() - t0)
        pos[0] += args.steps
        if sum(blocks) >= 1.0 or pos[0] + args.steps + 12 > L:
            break
    sb = sorted(blocks)
    med = sb[len(sb) // 2]
    fam, summ = ({}, {}) if args.skip_call_convention else families(ops, lambda: [steady(0) for _ in range(4)], 4)
    warp = summ.get("fgfa_warp")
    h, w = (args.height - 1) // 16 + 1, (args.width - 1) // 16 + 1
    esz = 2 if args.dtype == "bfloat16" else 4
    warp_bytes = 21 * 3072 * h * w * esz + 1024 * h * w * esz          # every frame's map read once, output written
    roof = None
    if warp:
        us = 1e3 * warp["ms"] / warp["launches"]
        roof = {"bound": "hbm", "kernel": "fgfa_warp_aggregate_kernel (flow-guided warp + cosine weights + softmax + sum)",
                "achieved": round(warp_bytes / us / 1e3, 1), "peak": 8000.0, "unit": "GB/s",
                "frac": round(warp_bytes / us / 1e3 / 8000.0, 4), "avg_launch_us": round(us, 1),
                "bytes_per_launch": warp_bytes, "traffic": None}
    return {"metric": "frames/sec FGFA R-101 inference, 21-frame window, %dx%d frames" % (args.width, args.height),
            "value": round(args.steps / med, 2), "unit": "frames/s", "ms_per_step": round(1e3 * med / args.steps, 3),
            "config": {"workload": "GeneralizedRCNNFGFA R-101-C4, ALL_FRAME_INTERVAL 21 / KEY_FRAME_LOCATION 10 (BASELINE "
                                   "configs[4]): per key frame 1 backbone + EmbedNet pass (+ FlowNetS's first conv of that frame), "
                                   "FlowNetS on 21 image pairs (refinement levels as sub-pixel GEMMs), fused warp + aggregation, RPN "
                                   "+ conv5 box head",
                       "driver": "fgfa.FgfaClipEngine: backbone + EmbedNet + the per-frame halves of FlowNetS's first conv for 20 "
                                 "upcoming frames per launch, the window in rings addressed through a device index table, ten key "
                                 "frames per FlowNetS pass, two hipGraphs on two streams (the batched box head of a group beside "
                                 "FlowNetS of the next; identical detections to the per-call path)",
                       "reference_call_convention_fps": round(args.steps / med_call, 2),
                       "graph_replays": engine.replays},
            "roofline": roof, "kernel_families": fam, "blocks_ms": [round(1e3 * b, 2) for b in blocks]}


def config2(args, dev):
    """MEGA R-50, exact-f32 (the reference's own precision), 10 local + 10 global frames: an 11-frame window."""
    from mega.pytorch_amd import config, engine as eng, modeling, ops, synth
    cfg = config.get_cfg("R-50")
    cfg.DTYPE = "float32"
    cfg.F32_CONV = args.f32_conv
    cfg.MODEL.DEVICE = str(dev)
    cfg.merge_from_list(["MODEL.VID.MEGA.ALL_FRAME_INTERVAL", 11, "MODEL.VID.MEGA.KEY_FRAME_LOCATION", 5,
                         "MODEL.VID.MEGA.MIN_OFFSET", -5, "MODEL.VID.MEGA.MAX_OFFSET", 5, "MODEL.VID.MEGA.GLOBAL.SIZE", 10])
    sd = synth.make_state_dict(blocks=(3, 4, 6), reduce_channel=True, global_res_stage=0, seed=0)
    model = modeling.build_detection_model(cfg)
    model.load_state_dict(sd)
    model.to(dev)
    x3 = args.f32_conv == "bf16x3"
    K, spb = args.steps, (20 if x3 and args.steps % 20 == 0 else 10)      # (planes: a 40-frame batch fits the 2 GiB operand limit)
    pre = 1 + 4 * spb
    nblk = 12
    T = pre + K * nblk + 8 + 8
    base = synth.make_clip(16, args.height, args.width, seed=0).to(dev)
    clip = base.index_select(0, torch.arange(T, device=dev) % 16).contiguous()
    gfor = eng.global_schedule(T, 10, seed=0)
    runner = eng.ClipEngine(model, steps_per_batch=spb)
    runner.run(clip, T, gfor, first=0, last=pre)
    torch.cuda.synchronize()
    st0 = runner.steady_state()
    blocks, pos = [], pre
    for _ in range(nblk):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        runner.run(clip, T, gfor, first=pos, last=pos + K)
        torch.cuda.synchronize()
        blocks.append(time.perf_counter() - t0)
        pos += K
        if sum(blocks) > 1.5:
            break
    s = sorted(blocks)
    med = s[len(s) // 2]
    runner.use_graphs, runner.overlap = False, False
    fam, summ = families(ops, lambda: runner.run(clip, T, gfor, first=pos, last=pos + 8), 8)
    ig = {k: v for k, v in summ.items() if k.startswith("igemm")}
    dom = max(ig, key=lambda k: ig[k]["ms"])
    ach = ig[dom]["flops"] / (ig[dom]["ms"] * 1e9)
    peak = 2500.0 if x3 else 157.3
    return {"metric": "frames/sec MEGA R-50 fp32 inference, 10 local + 10 global frames, %dx%d" % (args.width, args.height),
            "value": round(K / med, 2), "unit": "frames/s", "ms_per_step": round(1e3 * med / K, 3),
            "dtype": "bf16x3 (fp32 arithmetic from three bf16 MFMA passes per product, f32 accumulation)" if x3 else "f32",
            "config": {"workload": "MEGA R-50-C4, %s, ALL_FRAME_INTERVAL 11 / KEY_FRAME_LOCATION 5, GLOBAL.SIZE 10 "
                                   "(BASELINE configs[1])" % ("split-precision planes (cfg.F32_CONV = bf16x3; parity: "
                                   "test_cfg2_mega_r50_f32_600x1000_vs_oracle[bf16x3])" if x3 else
                                   "exact-f32 MFMA (v_mfma_f32_32x32x2_f32)"), "engine_state": st0, "steps_per_batch": spb},
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(ach / peak, 4), "traffic": None},
            "kernel_families": fam, "blocks_ms": [round(1e3 * b, 2) for b in blocks]}


def method_line(args, dev, method):
    """SURVEY 8f rows 3 / 4 (the reference's other video detectors) to the measurement bar: RDN (R-101, base + advanced stage,
    configs/RDN/vid_R_101_C4_RDN_1x.yaml) and DFF (R-101, key frame every 10 frames, configs/DFF) driven frame by frame in the
    reference's call convention on the reference's own test feed (inference.frame_feed), frames resident in HBM."""
    from mega.pytorch_amd import config, inference, modeling, ops, synth
    import mega.pytorch_amd.fgfa  # noqa: F401
    import mega.pytorch_amd.rdn  # noqa: F401
    cfg = config.get_cfg("R-101", method)
    cfg.DTYPE = args.dtype
    cfg.MODEL.DEVICE = str(dev)
    if method == "rdn":
        sd = synth.make_rdn_state_dict(blocks=(3, 4, 23), reduce_channel=False, advanced_stage=1, seed=0)
    else:
        sd = synth.make_dff_state_dict(blocks=(3, 4, 23), reduce_channel=False, seed=0)
    model = modeling.build_detection_model(cfg)
    model.load_state_dict(sd)
    model.to(dev)
    Tc = 16
    clip = synth.make_clip(Tc, args.height, args.width, seed=0).to(dev)
    mean = tuple(cfg.INPUT.PIXEL_MEAN)
    L = 120 + 4 + 40 * args.steps
    base = ops.preprocess_frames(clip.contiguous(), mean, True)
    video = base[torch.arange(L, device=dev) % Tc].contiguous()
    state = {"i": 0}

    def step(_):
        out = model(inference.frame_feed(cfg, video, state["i"]))
        state["i"] += 1
        return out
    for i in range(24 if method == "rdn" else 12):      # cold start + window / key-frame state
        step(i)
    med, blocks = timed(step, args.steps, min_seconds=1.0, max_blocks=25)
    fam, _ = families(ops, lambda: [step(0) for _ in range(10)], 10)
    engine_line = None
    if method == "rdn":       # ClipEngine drives this detector too (rdn.py: MEGA's frame stage, no memory / global pools)
        from mega.pytorch_amd import engine as eng_mod
        runner = eng_mod.ClipEngine(model, steps_per_batch=20)
        u8 = clip[torch.arange(L, device=dev) % Tc].contiguous()
        runner.run(u8, L, first=0, last=81)
        eb, pos = [], 81
        while pos + 40 <= L and sum(eb) < 1.5:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            runner.run(u8, L, first=pos, last=pos + 40)
            torch.cuda.synchronize()
            eb.append(time.perf_counter() - t0)
            pos += 40
        em = sorted(eb)[len(eb) // 2]
        engine_line = {"fps": round(40 / em, 2), "ms_per_frame": round(1e3 * em / 40, 3), "blocks_of_40_frames_ms": [round(1e3 * b, 2) for b in eb[:8]],
                       "driver": "engine.ClipEngine(steps_per_batch=20): the batched, graph-captured frame stage of the MEGA path on the "
                                 "new local frame of every step + the relation stages per key frame (parity: "
                                 "tests/test_e2e_gpu.py::test_rdn_f32_end_to_end_vs_reference_fixture)"}
    if method == "dff":       # the clip engine (fgfa.DffClipEngine): one FlowNetS pass per key-frame interval, two graphs / streams
        from mega.pytorch_amd import fgfa as fgfa_mod
        eng = fgfa_mod.DffClipEngine(model, interval=10, lookahead=8, lanes=args.lanes or 2, batch_head=args.batch_head != 0)
        if os.environ.get("MEGA_NO_FORK_SELECT") == "0":      # (experiments: the forked selection on several lanes)
            eng.fork_select = True
        eng.run(video, first=0, last=80)
        eb, pos = [], 80
        while pos + 40 <= L and sum(eb) < 1.0:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.run(video, first=pos, last=pos + 40, sync_every=40)
            torch.cuda.synchronize()
            eb.append(time.perf_counter() - t0)
            pos += 40
        em = sorted(eb)[len(eb) // 2]
        engine_line = {"fps": round(40 / em, 2), "ms_per_frame": round(1e3 * em / 40, 3), "blocks_of_40_frames_ms": [round(1e3 * b, 2) for b in eb[:8]],
                       "driver": "fgfa.DffClipEngine: backbone of 8 upcoming key frames per launch, FlowNetS on the 10 pairs of a key-frame "
                                 "interval in one pass (graph A), RPN + box head + post-processing of the interval's 10 frames as one "
                                 "batched launch chain (graph B) on a second stream (identical detections to the per-call path: "
                                 "tests/test_e2e_gpu.py::test_dff_engine_equals_model)", "graph_replays": eng.replays}
    what = {"rdn": "GeneralizedRCNNRDN R-101-C4 (relation distillation: base stage over 300 key + 37 x 75 reference proposals, "
                   "advanced stage on the distilled 20 %), one model(images) call per frame (SURVEY 8f row 3)",
            "dff": "GeneralizedRCNNDFF R-101-C4: the backbone on every 10th frame, FlowNetS (1 pair) + warp x scale on the others, "
                   "RPN + conv5 box head per frame; one model(images) call per frame (SURVEY 8f row 4)"}[method]
    return {"metric": "frames/sec %s R-101 inference, %dx%d frames" % (method.upper(), args.width, args.height),
            "value": round(args.steps / med, 2), "unit": "frames/s", "ms_per_step": round(1e3 * med / args.steps, 3),
            "config": {"workload": what, "driver": "the reference's call convention: host-bound at batch 1, the detection count is read "
                                                   "back every frame", "clip_engine": engine_line},
            "kernel_families": fam, "blocks_ms": [round(1e3 * b, 2) for b in blocks]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=0, choices=[0, 1, 2, 5])
    ap.add_argument("--method", default=None, choices=["rdn", "dff"], help="instead of --config: the line of SURVEY 8f row 3 / 4")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--dtype", default="bfloat16", choices=["bfloat16", "float32"])
    ap.add_argument("--height", type=int, default=600)
    ap.add_argument("--width", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fgfa-group", type=int, default=10, help="config 5: key frames per FlowNetS pass (FgfaClipEngine group)")
    ap.add_argument("--fgfa-no-pipeline", action="store_true", help="config 5: both graphs of a key frame on one stream")
    ap.add_argument("--fgfa-lookahead", type=int, default=20, help="config 5: frames per backbone + EmbedNet batch")
    ap.add_argument("--lanes", type=int, default=0, help="config 5 / --method dff: box-head graph lanes (0: the engine's default)")
    ap.add_argument("--batch-head", type=int, default=-1, help="clip engines: the box head of a group as one batched graph (1) or per frame (0); -1: the engine's default")
    ap.add_argument("--skip-call-convention", action="store_true", help="config 5: do not time the reference call convention")
    ap.add_argument("--f32-conv", default="exact", choices=["exact", "bf16x3"],
                    help="config 2: exact-f32 MFMA, or the split-precision mode (cfg.F32_CONV) -- fp32 arithmetic from three bf16 "
                         "matrix-core passes per product")
    args = ap.parse_args()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    with torch.no_grad():
        if args.method:
            line = method_line(args, dev, args.method)
        else:
            line = {1: config1, 2: config2, 5: config5}[args.config](args, dev)
    line.setdefault("dtype", "bf16" if args.dtype == "bfloat16" else "f32")
    line.update({"n_gpus": 1, "steps": args.steps, "higher_is_better": True, "data": "synthetic", "vs_baseline": None})
    os.write(json_fd, (json.dumps(line) + "\n").encode())


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""MEGA R-101 per-key-frame inference benchmark on MI355X (BASELINE.json metric: frames/sec, synthetic
1000x600 VID clip, MEGA R-101 25 local / 10 global / 25-frame memory, bf16).

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank/GPU)

A "step" is one steady-state key frame: one new local frame + one new global-pool frame through
backbone/RPN/res5/ROIAlign/fc0 (preprocessing included, uint8 frames resident in HBM), then the relation
aggregation (7 attention calls), predictor and post-processing NMS -- the reference's per-key-frame work,
nothing skipped.  Warm-up covers the cold start (frame 0: 13 local + 10 global frames) and fills the window,
the global pool and the 25-frame memory when W >= 25.  N>1: the frame stage of each batch of steps is sharded
over the ranks and the fixed-size frame records are exchanged with one RCCL all-gather ("strong" scaling:
the clip is the same for every N).

Prints ONE JSON line (rank 0).  Besides the driver's contract fields it carries
  roofline     -- the dominant kernel family (implicit-GEMM conv/linear on MFMA): algorithmic FLOPs of its launches
                  / their HIP-event durations measured in a separate, untimed, instrumented pass
  cpu_baseline -- oracle/ (a CPU restatement of the reference path, kind "port") timed on the host cores over a
                  bounded sample of the SAME workload (N=1, rank 0 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

T_START = time.perf_counter()


def log(msg):
    sys.stderr.write("[bench %7.1fs] %s\n" % (time.perf_counter() - T_START, msg))
    sys.stderr.flush()


def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


ALGO_GFLOP_PER_FRAME = 729.0   # minimal algorithmic work per steady-state key frame, R-101 MEGA


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--arch", default="R-101")
    ap.add_argument("--dtype", default="bfloat16", choices=["bfloat16", "float32"])
    ap.add_argument("--steps-per-batch", type=int, default=10)
    ap.add_argument("--height", type=int, default=600)
    ap.add_argument("--width", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="run frame stage and aggregation on one stream")
    ap.add_argument("--no-graphs", action="store_true", help="launch the frame stage kernel by kernel (no hipGraph)")
    ap.add_argument("--reuse-records", action="store_true",
                    help="compute each frame's record once per video (engine option; NOT the headline configuration)")
    ap.add_argument("--static-aggregation", action="store_true",
                    help="experimental: steady-state aggregation steps replayed from one hipGraph (engine option)")
    ap.add_argument("--cpu-seconds", type=float, default=25.0)
    return ap.parse_args()


def build_model(arch, dtype, device):
    from mega.pytorch_amd import config, modeling, synth
    cfg = config.get_cfg(arch)
    cfg.DTYPE = dtype
    cfg.MODEL.DEVICE = str(device)
    r50 = arch.startswith("R-50")
    sd = synth.make_state_dict(blocks=(3, 4, 6) if r50 else (3, 4, 23), reduce_channel=r50,
                               global_res_stage=0 if r50 else 1, seed=0)
    model = modeling.build_detection_model(cfg)
    model.load_state_dict(sd)
    model.to(device)
    return cfg, model, sd


def make_clip(T, H, W, device, unique=16):
    from mega.pytorch_amd import synth
    base = synth.make_clip(min(unique, T), H, W, seed=0).to(device)
    idx = torch.arange(T, device=device) % base.shape[0]
    return base.index_select(0, idx).contiguous()


def cpu_baseline(arch, sd, H, W, budget_s):
    """oracle (kind 'port') on the host cores: cold start + a few steady frames, bounded by budget_s."""
    import numpy as np
    from oracle import mega_oracle as mo
    from mega.pytorch_amd import synth
    cores = min(host_cores(), 64)
    torch.set_num_threads(cores)
    log("cpu baseline on %d threads (affinity %d, cpu_count %s)" % (cores, host_cores(), os.cpu_count()))
    r50 = arch.startswith("R-50")
    ocfg = mo.OracleCfg(blocks=(3, 4, 6) if r50 else (3, 4, 23), reduce_channel=r50, global_res_stage=0 if r50 else 1)
    T = 40
    frames = synth.preprocess_cpu(synth.make_clip(8, H, W, seed=0))
    frames = frames[torch.arange(T) % frames.shape[0]]
    _, gfor = mo.global_frame_schedule(T, ocfg.global_size, seed=0)
    orc = mo.MegaOracle({k: v.cpu() for k, v in sd.items()}, ocfg)
    times = []
    t_all = time.perf_counter()
    with torch.no_grad():
        for idx in range(T):
            t0 = time.perf_counter()
            orc.forward_frame(frames[idx:idx + 1], 0 if idx == 0 else 1, ref_l=frames[min(T - 1, idx + 12)][None],
                              ref_g=[frames[g][None] for g in gfor(idx)], seg_len=T,
                              frame_loader=lambda i: frames[i][None])
            times.append(time.perf_counter() - t0)
            log("cpu baseline frame %d: %.2fs" % (idx, times[-1]))
            if idx >= 1 and time.perf_counter() - t_all > budget_s:
                break
    steady = times[1:]
    fps = len(steady) / sum(steady)
    return {"value": round(fps, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "oracle/mega_oracle.py (torch-CPU fp32 restatement of the reference path), same weights and "
                      "frame size, %d steady key frames after a %.1f s cold-start frame (memory pool still filling: "
                      "%d of 25 frames)" % (len(steady), times[0], len(steady))}


def main():
    # stdout carries exactly ONE line (the JSON): everything else any library writes to fd 1 -- RCCL prints a
    # version banner there at exit -- is sent to stderr.
    json_fd = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device (the MEGA hot path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    group = None
    if world > 1 or os.environ.get("MEGA_FORCE_SHARDED") == "1":   # the latter: exercise the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=device)
        group = dist.group.WORLD
    from mega.pytorch_amd import engine as eng, ops

    log("building model")
    cfg, model, sd = build_model(args.arch, args.dtype, device)
    log("model ready")
    K, Wm = args.steps, max(args.warmup, 1)
    prof_steps = 0 if args.no_roofline else 8
    T = 1 + Wm + K + prof_steps + 13
    clip = make_clip(T, args.height, args.width, device)
    gfor = eng.global_schedule(T, cfg.MODEL.VID.MEGA.GLOBAL.SIZE, seed=0)
    runner = eng.ClipEngine(model, steps_per_batch=args.steps_per_batch, dist_group=group, overlap=not args.no_overlap,
                            graphs=not args.no_graphs, reuse_records=args.reuse_records,
                            static_aggregation=args.static_aggregation)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warm-up: cold start (key frame 0) + W - 1 steady frames
    log("clip ready (%d frames); warm-up" % T)
    runner.run(clip, T, gfor, first=0, last=1)
    barrier()
    log("cold start done")
    runner.run(clip, T, gfor, first=1, last=Wm)
    barrier()
    log("warm-up done; timing %d steps" % K)
    for k in runner.host_times:
        runner.host_times[k] = 0
    fc_before = runner.frames_computed
    t0 = time.perf_counter()
    dets = runner.run(clip, T, gfor, first=Wm, last=Wm + K)
    barrier()
    elapsed = time.perf_counter() - t0
    runner_frames_after = runner.frames_computed
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    fps = K / elapsed
    log("timed region: %.3fs (%.2f frames/s)" % (elapsed, fps))
    log("frame-stage batches so far: %s; static aggregation steps: %d" % (runner.graph_stats, runner.static_steps))
    ht = dict(runner.host_times)
    log("host ms/step: frame-stage enqueue %.3f, aggregation enqueue %.3f, waiting for results %.3f" % tuple(
        1e3 * ht[k] / max(ht["steps"], 1) for k in ("frame_enqueue", "aggregate_enqueue", "finish_wait")))

    roofline = None
    fam = {}
    if prof_steps:
        p = ops.Profiler()
        ops.set_profiler(p)
        runner.use_graphs = False
        runner.overlap = False     # per-kernel event pairs are only meaningful without cross-stream concurrency
        runner.run(clip, T, gfor, first=Wm + K, last=Wm + K + prof_steps)
        summ = p.summary()
        ops.set_profiler(None)
        tot_ms = sum(v["ms"] for v in summ.values())
        for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
            fam[k] = {"ms_per_step": round(v["ms"] / prof_steps, 4), "launches_per_step": round(v["launches"] / prof_steps, 1),
                      "tflops": round(v["flops"] / (v["ms"] * 1e9), 2) if v["ms"] > 0 else 0.0,
                      "gbps": round(v["bytes"] / (v["ms"] * 1e6), 1) if v["ms"] > 0 else 0.0}
        # dominant kernel = the igemm instantiation (one rocprofv3 kernel symbol) with the largest share of GPU time
        tag = "bf16" if args.dtype == "bfloat16" else "f32"
        igemms = {k: v for k, v in summ.items() if k.startswith("igemm_" + tag)}
        dom = max(igemms, key=lambda k: igemms[k]["ms"])
        tile = dom.rsplit("_", 1)[1].split("x")
        peak = 2500.0 if args.dtype == "bfloat16" else 157.3
        d = summ[dom]
        ach = d["flops"] / (d["ms"] * 1e9)
        fam_ms = sum(v["ms"] for v in igemms.values())
        fam_fl = sum(v["flops"] for v in igemms.values())
        # HBM traffic per launch of the dominant variant, from the committed rocprofv3 PMC passes of this same
        # command (FETCH_SIZE and WRITE_SIZE in separate passes, gfx950 x2 read correction; tools/pmc_summary.py)
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
        sym = "igemm_kernel<%s, %s, %s, %s>" % (tag if tag == "bf16" else "float", tag if tag == "bf16" else "float",
                                                tile[0], tile[1])
        if os.path.exists(pmc):
            k = json.load(open(pmc))["kernels"].get(sym)
            if k:
                traffic, traffic_src = round(k["hbm_bytes_per_launch_corrected"]), "profiles/r01_pmc_summary.json"
        roofline = {"bound": "mfma", "kernel": sym + " (implicit-GEMM conv / linear)",
                    "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                    "traffic": traffic, "traffic_source": traffic_src, "avg_launch_us": round(1e3 * d["ms"] / d["launches"], 2),
                    "flops_per_launch": round(d["flops"] / d["launches"], 0),
                    "share_of_gpu_time": round(d["ms"] / tot_ms, 3),
                    "all_igemm_variants": {"achieved": round(fam_fl / (fam_ms * 1e9), 2),
                                           "frac": round(fam_fl / (fam_ms * 1e9) / peak, 4),
                                           "share_of_gpu_time": round(fam_ms / tot_ms, 3)},
                    "whole_path_frac": round(ALGO_GFLOP_PER_FRAME * 1e9 * fps / (peak * 1e12), 5)
                    if args.arch == "R-101" else None}

    # whole-clip rate (SURVEY 8d ii): a fresh video -- cold start (13 local + 10 global frames, eager aggregation
    # while the pools fill) plus the same K steady key frames -- on the warmed-up engine.  Reported beside `value`.
    whole_clip = None
    try:
        runner.use_graphs, runner.overlap = not args.no_graphs, not args.no_overlap   # (the instrumented pass turned them off)
        barrier()
        t0 = time.perf_counter()
        runner.run(clip, T, gfor, first=0, last=1 + K)
        barrier()
        wc = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([wc], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wc = float(t.item())
        whole_clip = {"key_frames": 1 + K, "seconds": round(wc, 4), "frames_per_s": round((1 + K) / wc, 2)}
        log("whole clip incl. cold start: %d key frames in %.3fs (%.1f frames/s)" % (1 + K, wc, (1 + K) / wc))
    except Exception as e:  # noqa: BLE001  (an optional extra must never cost the headline line)
        log("whole-clip measurement skipped: %r" % (e,))

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.arch, sd, args.height, args.width, args.cpu_seconds)

    if rank == 0:
        ndet = sum(len(d) for d in dets) / max(len(dets), 1)
        line = {
            "metric": "frames/sec MEGA %s inference, %dx%d VID clip" % (args.arch, args.width, args.height),
            "value": round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": round(1e3 * elapsed / K, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "bf16" if args.dtype == "bfloat16" else "f32", "data": "synthetic",
            "config": {"workload": "MEGA %s-C4, %dx%d frames, 25 local + 10 global frames + 25-frame memory, "
                                   "300 key / 75 ref proposals, 3 attention stages (BASELINE configs[2]%s)"
                                   % (args.arch, args.width, args.height, "" if world == 1 else " sharded = configs[3]"),
                       "steps_per_batch": args.steps_per_batch, "parallelism": "frame-sharded x%d" % world,
                       "frame_record_reuse": bool(args.reuse_records),
                       "static_aggregation": bool(args.static_aggregation),
                       "frames_through_frame_stage_per_step": round((runner_frames_after - fc_before) / K, 2),
                       "avg_detections": round(ndet, 1),
                       "key_proposals_last_frame": int(model.records[model.key_frame_location]["boxes"].shape[0]),
                       "whole_clip_incl_cold_start": whole_clip},
            "roofline": roofline, "cpu_baseline": cpu, "kernel_families": fam,
        }
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if world > 1 or os.environ.get("MEGA_FORCE_SHARDED") == "1":
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

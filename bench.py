#!/usr/bin/env python
"""MEGA R-101 per-key-frame inference benchmark on MI355X (BASELINE.json metric: frames/sec, synthetic
1000x600 VID clip, MEGA R-101 25 local / 10 global / 25-frame memory, bf16).

  python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU.  Either the caller launches the ranks (`python -m torch.distributed.run --nproc-per-node N
... bench.py --gpus N`: WORLD_SIZE is set and must equal --gpus) or, when WORLD_SIZE is not set, bench.py launches them
itself by re-executing under torch.distributed.run on 127.0.0.1 (the reference starts one process per GPU the same way:
tools/test_net.py:69-75).  A box with fewer than N devices, or a WORLD_SIZE that differs from --gpus, is an error
(exit code 2): the line never reports an N that did not run.

A "step" is one steady-state key frame: one new local frame + one new global-pool frame through
backbone/RPN/res5/ROIAlign/fc0 (preprocessing included, uint8 frames resident in HBM), then the relation
aggregation (7 attention calls), predictor and post-processing NMS -- the reference's per-key-frame work,
nothing skipped.

What is timed is ALWAYS the steady state of SURVEY.md 8d (mirrors mega_core/engine/inference.py:24-42), whatever
--warmup says: an internal pre-roll runs the cold start (frame 0: 13 local + 10 global frames) and then key frames
until (i) the local window, the global pool and the 25-entry memory deques of all three stages are full and (ii) the
steady frame-stage / aggregation hipGraphs have been captured AND replayed; --warmup only lengthens it.  The timed
region is then R blocks of EXACTLY --steps key frames, each block bracketed by barrier + synchronize on both sides
(R chosen so that the blocks cover >= ~1 s); `value` / `ms_per_step` are the MEDIAN block (all block times are in
`config.timed_blocks_ms`).  The engine's graph statistics must not change inside the timed region (asserted).
N>1: a step is N key frames of the video (one per rank: "weak" scaling, per-GPU work fixed); the frame stage of each
step-batch (20 N key frames = 40 N frames, 40 per rank) is sharded over the ranks and the fixed-size frame records are
exchanged with one RCCL all-gather per row-count group; the key frames' aggregation is dealt to the ranks, the memory
entries all-gathered once per stage (engine.KeyFrameShard).  `value` = all key frames of a block / its time.

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


HALF_MODES = ("bfloat16", "wide", "float16", "f16x2")      # --dtype values whose frame stage runs on 16-bit matrix-core operands
ALGO_GFLOP_PER_FRAME = 729.0   # minimal algorithmic work per steady-state key frame, R-101 MEGA


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)     # three step-batches of 20 key frames
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--arch", default="R-101")
    ap.add_argument("--dtype", default="bfloat16", choices=["bfloat16", "float16", "f16x2", "float32", "bf16x3", "wide"],
                    help="bfloat16 (BASELINE configs[2], the headline) | float16 (the same kernels on IEEE-half operands: the "
                         "frame stage in fp16, 11 significant bits at the bf16 rate) | float32 (exact-f32 MFMA parity mode) | bf16x3 (float32 "
                         "with the split-precision frame stage, cfg.F32_CONV) | wide (bfloat16 with the residual trunk as "
                         "[hi | lo] planes, cfg.RESIDUAL_STREAM)")
    ap.add_argument("--steps-per-batch", type=int, default=0,
                    help="key frames per engine step-batch; default 20 N on N GPUs when --steps allows (a timed block is "
                         "--steps x N key frames, so a rank's frame-stage launch holds 40 frames: a 2-5 frame launch leaves "
                         "most of a rank's CUs idle)")
    ap.add_argument("--height", type=int, default=600)
    ap.add_argument("--width", type=int, default=1000)
    ap.add_argument("--head-stream", default=None, choices=["float32", "bfloat16"],
                    help="cfg.HEAD_STREAM of the bf16 mode (default: the product's default, float32 = the parity-preserving head; "
                         "bfloat16 = the round-3 head with 8 bf16 hand-offs, ~2 %% faster)")
    ap.add_argument("--no-f32-leg", action="store_true", help="skip the parity-mode legs (config.f32_parity_mode, "
                    "config.bf16x3_parity_mode)")
    ap.add_argument("--no-h2d-leg", action="store_true", help="skip the leg that feeds the frames from pinned host memory "
                    "(config.with_h2d)")
    ap.add_argument("--no-whole-clip", action="store_true", help="skip the whole-clip (cold start + K key frames) measurement "
                    "after the timed region (kernel traces: the tail of the stream is then a steady timed block)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="run frame stage and aggregation on one stream")
    ap.add_argument("--no-graphs", action="store_true", help="launch the frame stage kernel by kernel (no hipGraph)")
    ap.add_argument("--reuse-records", action="store_true",
                    help="compute each frame's record once per video (engine option; NOT the headline configuration)")
    ap.add_argument("--aggregation", default="batched", choices=["batched", "batched-eager", "static", "per-frame"],
                    help="batched (default): the aggregation of a step-batch runs stage by stage over all its key "
                         "frames, replayed from one hipGraph on fixed-address state once the pools are full; batched-eager: "
                         "the same launched kernel by kernel (round 3); static: one hipGraph replay per key frame on "
                         "fixed-address pools; per-frame: eager steps")
    ap.add_argument("--ramp", action="store_true",
                    help="short first / last step-batch inside a timed block (ClipEngine ramp; measured slower: 555 vs 585 FPS)")
    ap.add_argument("--cpu-frames", type=int, default=3, help="steady key frames timed by the CPU baseline (memory full)")
    ap.add_argument("--min-seconds", type=float, default=5.0, help="the timed blocks cover at least this long")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch / rendezvous / reporting only (no model, no kernels; gloo when there is no HIP device): "
                         "proves the --gpus N self-launch and the n_gpus bookkeeping on a CPU box (tests/test_bench_cli.py)")
    ap.add_argument("--max-blocks", type=int, default=300)
    return ap.parse_args()


def key_frames_per_block(steps, world):
    """A step is `world` key frames (one per rank): a timed block of --steps steps covers steps x world key frames."""
    return steps * max(world, 1)


def default_steps_per_batch(steps, world):
    """Key frames per engine step-batch when --steps-per-batch is not given: the largest divisor of the block's key
    frames (steps x world) that is <= 20 x world.  One GPU: 20 key frames = a 40-frame frame stage (M = 95760 rows per
    layer-3 launch = 1.95 rounds of 192-row tiles on the 256 CUs), the aggregation's small kernels amortised over 20 key
    frames.  N GPUs: 20 N whenever --steps is a multiple of 20 -- every rank's slice of the frame stage is then the same
    40 frames as on one GPU.
    The driver's --steps 20 block is therefore ONE batch: frame stage, then aggregation, nothing overlapped inside the
    synchronised block.  Cutting it into two batches of 10 (the second frame stage beside the first aggregation) was
    re-measured in round 4 with the aggregation replayed from a hipGraph: +1.5 % on one box (890.5 : 877.0), -1.4 % and
    -2.4 % on two others (858.7 : 871.1, 867.5 : 887.9; profiles/r04_step_batch_structure.txt) -- how much of the first
    aggregation gets CUs beside igemm8 blocks that own a CU's whole register file is up to the hardware scheduler, and two
    aggregations of 10 cost 7.2 ms of GPU time against 5.3 for one of 20.  [20] stays."""
    world = max(world, 1)
    kf = key_frames_per_block(steps, world)
    return max(d for d in range(1, kf + 1) if kf % d == 0 and d <= 20 * world)


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args, argv):
    """--gpus N > 1 and no launcher around us: start one rank per GPU under torch.distributed.run (rendezvous on
    127.0.0.1) and hand on its exit code.  stdout is inherited, so rank 0's JSON line is this process's JSON line."""
    import subprocess
    if not args.dry_run:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            sys.stderr.write("bench.py: --gpus %d but this box has %d HIP device(s); refusing to report a smaller N\n"
                             % (args.gpus, have))
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    log("self-launch: " + " ".join(cmd))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def dry_run(args, world, rank, local_rank, json_fd):
    """The launch / rendezvous / reporting skeleton without model or kernels: process group (nccl when every rank has a
    device, else gloo), the barrier + MAX-over-ranks timing bracket, the sharded frame-stage plan of one steady
    step-batch, and the JSON line with n_gpus taken from the LIVE process group."""
    import torch.distributed as dist
    from mega.pytorch_amd import engine as eng
    live = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        use_nccl = torch.cuda.is_available() and torch.cuda.device_count() >= world
        if use_nccl:
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl" if use_nccl else "gloo")
        live = dist.get_world_size()
    assert live == args.gpus, "process group has %d ranks, --gpus %d" % (live, args.gpus)
    K = args.steps
    KF = key_frames_per_block(K, live)
    spb = args.steps_per_batch if args.steps_per_batch > 0 else default_steps_per_batch(K, live)

    class _M(object):        # the schedule only needs these constants (MEGA R-101 defaults)
        all_frame_interval, key_frame_location, key_num, base_num, global_enable = 25, 12, 300, 75, True
        cfg = type("C", (), {"INPUT": type("I", (), {"PIXEL_MEAN": (0, 0, 0), "TO_BGR255": True})})
    e = eng.ClipEngine(_M(), steps_per_batch=spb)
    e.rank, e.world = rank, live
    e.owner_aligned = True        # (the product's dealing with the batched aggregation: frame f -> rank f mod world)
    T = 64 + 2 * KF
    gfor = eng.global_schedule(T, 10, seed=0)
    jobs = [j for i in range(40, 40 + spb) for j in e.jobs_for_step(i, T, gfor)]
    mine = e.shard_plan(jobs)[1] if live > 1 else list(range(len(jobs)))
    t0 = time.perf_counter()
    if live > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if live > 1:
        t = torch.tensor([dt, float(len(mine))], dtype=torch.float64)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        mx = t.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dt, most = float(mx[0]), int(mx[1])
    else:
        most = len(mine)
    if rank == 0:
        line = {"metric": "frames/sec MEGA %s inference, %dx%d VID clip" % (args.arch, args.width, args.height),
                "value": None, "unit": "frames/s", "n_gpus": live, "steps": K, "warmup": args.warmup, "ms_per_step": None,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dry_run": True,
                "config": {"key_frames_per_step": live, "key_frames_per_block": KF, "steps_per_batch": spb,
                           "batch_sizes_in_a_block": e.batch_sizes(KF), "frames_per_batch": len(jobs), "frames_per_rank_per_batch": most,
                           "backend": dist.get_backend() if live > 1 else None}}
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if live > 1:
        dist.destroy_process_group()


def build_model(arch, dtype, device, head_stream=None):
    from mega.pytorch_amd import config, modeling, synth
    cfg = config.get_cfg(arch)
    if dtype == "bf16x3":
        dtype, cfg.F32_CONV = "float32", "bf16x3"
    elif dtype == "wide":
        dtype, cfg.RESIDUAL_STREAM = "bfloat16", "planes"
    elif dtype == "f16x2":       # float16, two-pass form: fp16 [hi | lo] activation planes against weights rounded to fp16 once
        dtype, cfg.F16_CONV = "float16", "x2"
    elif dtype == "f16head":     # float16 with the aggregation head on IEEE-half operands too (cfg.HEAD_DTYPE; as a head_mode:
        dtype, cfg.HEAD_DTYPE = "float16", "float16"      # the fp16 head behind another mode's frame stage)
    cfg.DTYPE = dtype
    if head_stream is not None:
        cfg.HEAD_STREAM = head_stream
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


def cpu_baseline(arch, sd, H, W, n_timed):
    """oracle (kind 'port') on the host cores.  The oracle is stepped through the cold start and key frames 1..25
    UNTIMED so that its 25-entry memory deques are full (the regime `value` is measured in), then `n_timed` steady
    key frames are timed."""
    from oracle import mega_oracle as mo
    from mega.pytorch_amd import synth
    avail = host_cores()
    cores = min(avail, 16)       # the untimed fill: torch-CPU on this path is fastest at 8-32 threads (sweep below)
    torch.set_num_threads(cores)
    log("cpu baseline: fill on %d threads (affinity %d, cpu_count %s)" % (cores, avail, os.cpu_count()))
    r50 = arch.startswith("R-50")
    ocfg = mo.OracleCfg(blocks=(3, 4, 6) if r50 else (3, 4, 23), reduce_channel=r50, global_res_stage=0 if r50 else 1)
    fill = ocfg.all_frame_interval + 1            # key frames 0..25: every memory deque holds 25 entries afterwards
    # never more than 64: on the 256-thread MI355X hosts ONE steady key frame took 247 s at 256 threads (2.6 s at 8, 2.7 s at
    # 32, 4.0 s at 64: profiles/r05_bench_n1.json) -- torch's intra-op pool thrashes on this path's many small ops
    sweep = sorted(set(c for c in (8, 16, 32, 64) if c <= avail)) or [avail]
    T = fill + len(sweep) + n_timed + 13
    frames = synth.preprocess_cpu(synth.make_clip(8, H, W, seed=0))
    frames = frames[torch.arange(T) % frames.shape[0]]
    _, gfor = mo.global_frame_schedule(T, ocfg.global_size, seed=0)
    orc = mo.MegaOracle({k: v.cpu() for k, v in sd.items()}, ocfg)
    times = []

    def one(idx):
        t0 = time.perf_counter()
        orc.forward_frame(frames[idx:idx + 1], 0 if idx == 0 else 1, ref_l=frames[min(T - 1, idx + 12)][None],
                          ref_g=[frames[g][None] for g in gfor(idx)], seg_len=T,
                          frame_loader=lambda i: frames[i][None])
        return time.perf_counter() - t0
    with torch.no_grad():
        for idx in range(fill):
            times.append(one(idx))
            if idx < 2 or idx >= fill - 1 or idx % 8 == 0:
                log("cpu baseline frame %d: %.2fs (memory %d/25)" % (idx, times[-1], len(orc.mem_queue[0]["rois"])))
        # thread sweep on steady key frames (memory full): the count that is fastest is the one the baseline is quoted at
        # (64 threads on a 256-thread host measured slower than the survey's 8-core number)
        swept = {}
        idx = fill
        for c in sweep:
            torch.set_num_threads(c)
            swept[c] = one(idx)
            log("cpu baseline thread sweep: %d threads -> %.2fs per steady key frame" % (c, swept[c]))
            idx += 1
        cores = min(swept, key=lambda c: swept[c])
        torch.set_num_threads(cores)
        steady = [one(idx + i) for i in range(n_timed)]
    times += list(swept.values()) + steady
    mem = [len(q["rois"]) for q in orc.mem_queue]
    fps = len(steady) / sum(steady)
    # The unmodified reference cannot run on the GPU box (/root/reference is not there), so the timed thing is the port;
    # how the port compares with the reference on the SAME host cores was measured in the build container
    # (tools/cpu_port_vs_reference.py) and is quoted from the committed record.
    vs_ref = None
    rec = os.path.join(ROOT, "profiles", "r03_cpu_port_vs_reference.json")
    if os.path.exists(rec) and not r50:
        r = json.load(open(rec))
        ratio = r["mega_r101"]["port_over_reference_time"]
        vs_ref = {"port_over_reference_time": ratio, "reference_equivalent_frames_per_s": round(fps * ratio, 4),
                  "measured": "build container, %d threads: reference %.2f s / port %.2f s per steady key frame"
                              % (r["host_threads"], r["mega_r101"]["reference_steady_s"], r["mega_r101"]["port_steady_s"]),
                  "source": "profiles/r03_cpu_port_vs_reference.json"}
    ref_eq = None
    if vs_ref is not None:     # the reference's own CPU path on these cores = the port's rate x the measured factor
        ref_eq = {"value": vs_ref["reference_equivalent_frames_per_s"], "unit": "frames/s",
                  "factor": vs_ref["port_over_reference_time"], "factor_is": "port seconds / reference seconds per steady key frame "
                  "(unmodified mega_core through oracle/ref_shim.py against oracle/mega_oracle.py, same weights, frames and threads)",
                  "factor_source": vs_ref["source"], "factor_measured_on": vs_ref["measured"], "cores": cores}
    return {"value": round(fps, 4), "unit": "frames/s", "cores": cores, "kind": "port", "memory_frames": min(mem),
            "reference_equivalent": ref_eq,
            "host_threads_available": avail,
            "thread_sweep_s_per_key_frame": {str(c): round(v, 3) for c, v in swept.items()},
            "vs_unmodified_reference": vs_ref,
            "sample": "oracle/mega_oracle.py (torch-CPU fp32 restatement of the reference path), same weights and "
                      "frame size: %d steady key frames timed (%.2f s each) on %d threads -- the fastest of a sweep over %s "
                      "threads, one steady key frame each -- AFTER an untimed fill of %d key frames (cold start %.1f s + "
                      "%.1f s) that leaves all memory deques full (%s of 25)"
                      % (len(steady), sum(steady) / len(steady), cores, sorted(swept), fill, times[0], sum(times[1:fill]),
                         min(mem))}


# what tests/test_e2e_gpu.py::test_r101_600x1000_f16_vs_oracle / test_r101_calibrated_f16_agreement assert for the fp16 mode
F16_PARITY = ("against the f32 oracle, R-101 600x1000, 28 key frames incl. the memory-full regime: see "
              "tests/test_e2e_gpu.py::test_r101_600x1000_f16_vs_oracle (seeded fixture) and test_r101_calibrated_f16_agreement "
              "(fixture with margins); predicted on the CPU twins before the kernels existed: profiles/r06_fp16_prediction.txt")


F16X2_PARITY = ("against the f32 oracle, R-101 600x1000, 28 key frames incl. the memory-full regime: "
                "tests/test_e2e_gpu.py::test_r101_600x1000_f16x2_vs_oracle; predicted on the CPU twins (variant 2pass-bound of "
                "profiles/r06_fp16_prediction.txt)")


def igemm_symbol(famname):
    """profiler family -> the rocprofv3 symbol of its launches"""
    if famname.startswith("igemm8_sp"):
        return "igemm8_kernel<*, *, *, 0, 1> (split-precision planes)"
    parts = famname.split("_")
    t_ = parts[2].split("x")
    f32o = famname.endswith("_f32out")
    et = {"bf16": "unsigned short", "f16": "_Float16"}.get(parts[1], "float")
    if parts[0] in ("igemm8", "igemm8s"):      # <OT, MF1, CLS, ABL, SP, HT> (HT: the 16-bit operand type, round 6)
        return "igemm8_kernel<%s, %d, %d, 0, 0, %s>" % ("float" if f32o else et, 2 if t_[0] == "256" else 1,
                                                        1 if parts[0] == "igemm8s" else 0, et)
    return "igemm_kernel<%s, %s, %s, %s>" % (et, "float" if (f32o or parts[1] == "f32") else et, t_[0], t_[1])


def leg_roofline(runner, clip, T, gfor, pos, spb, mode):
    """The headline's `roofline` block for a parity-mode leg (VERDICT r05 item 6 iii): one instrumented step-batch of the leg's
    own engine -- kernel by kernel, no graph replays, one stream, a HIP event pair around every launch (ops.Profiler) -- after
    one untimed eager batch; dominant symbol = the matrix-core GEMM family with the largest share of GPU time.  FLOPs are the
    MATRIX-CORE work of a launch (a split-precision launch contracts 3 K, a two-pass fp16 launch 2 K); `traffic` = HBM bytes
    per launch of that symbol from the newest committed PMC summary of the mode (null when there is none)."""
    from mega.pytorch_amd import ops
    try:
        if pos + 2 * spb + 13 >= T:
            return None
        ug, us, ov = runner.use_graphs, runner.use_static, runner.overlap
        runner.use_graphs, runner.use_static, runner.overlap = False, False, False
        runner.run(clip, T, gfor, first=pos, last=pos + spb)
        torch.cuda.synchronize()
        p = ops.Profiler()
        ops.set_profiler(p)
        runner.run(clip, T, gfor, first=pos + spb, last=pos + 2 * spb)
        summ = p.summary()
        ops.set_profiler(None)
        runner.use_graphs, runner.use_static, runner.overlap = ug, us, ov
        tot_ms = sum(v["ms"] for v in summ.values())
        mm = {k: v for k, v in summ.items() if k.startswith("igemm") and not k.startswith("igemm8s_") and v["ms"] > 0}
        if not mm or tot_ms <= 0:
            return None
        dom = max(mm, key=lambda k: mm[k]["ms"])
        d = mm[dom]
        peak = 157.3 if mode == "float32" else 2500.0
        ach = d["flops"] / (d["ms"] * 1e9)
        allg = {k: v for k, v in summ.items() if k.startswith("igemm")}
        sym = {"igemm8_sp_x3": "igemm8_kernel<*, *, *, 0, 1, unsigned short> (split-precision planes, 3 K contraction)",
               "igemm8_sp_h2": "igemm8_kernel<*, *, *, 0, 1, _Float16> (fp16 [hi | lo] planes, 2 K contraction)",
               "igemm8_sp_hi": "igemm8_kernel<*, *, *, 0, 1, unsigned short> (hi plane only)"}.get(dom) or igemm_symbol(dom)
        traffic = src = None
        tagf = {"bf16x3": "r06_x3", "float32": "r06_f32", "float16": "r06_f16", "f16x2": "r06_f16x2"}.get(mode)
        for cand in ([tagf] if tagf else []) + (["r05_x3"] if mode == "bf16x3" else []):
            pmc = os.path.join(ROOT, "profiles", cand + "_pmc_summary.json")
            if os.path.exists(pmc):
                ks = json.load(open(pmc))["kernels"]
                want = "igemm8_kernel" if dom.startswith("igemm8") else "igemm_kernel"

                def is_sp(name):       # igemm8_kernel<OT, MF1, CLS, ABL, SP[, HT]>
                    a = [x.strip() for x in name[name.index("<") + 1:name.rindex(">")].split(",")] if "<" in name else []
                    return len(a) >= 5 and a[4] == "1"
                if dom.startswith("igemm8_sp"):
                    rows = [v for name, v in ks.items() if name.startswith(want) and is_sp(name)]
                else:       # a plain family is ONE symbol: its own row
                    w_ = sym.replace("unsigned short", "bf16").replace(" ", "")
                    rows = [v for name, v in ks.items() if name.replace(" ", "").startswith(w_)]
                if rows:
                    k = max(rows, key=lambda v: v.get("launches", 0) * v.get("hbm_bytes_per_launch_corrected", 0))
                    traffic, src = round(k["hbm_bytes_per_launch_corrected"]), "profiles/%s_pmc_summary.json%s" % (cand, " (the symbol of this family with the most HBM bytes)" if dom.startswith("igemm8_sp") else "")
                    break
        return {"bound": "mfma", "kernel": sym, "family": dom, "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": src,
                "avg_launch_us": round(1e3 * d["ms"] / d["launches"], 2), "flops_per_launch": round(d["flops"] / d["launches"], 0),
                "algorithmic_bytes_per_launch": round(d["bytes"] / d["launches"], 0),
                "recomputed_from": {"launches": d["launches"], "sum_gflop": round(d["flops"] / 1e9, 2), "sum_us": round(1e3 * d["ms"], 1),
                                    "key_frames": spb, "how": "achieved = sum_gflop / sum_us (HIP events around every launch of this "
                                                              "family in one instrumented step-batch of this leg's engine)"},
                "share_of_gpu_time": round(d["ms"] / tot_ms, 3),
                "all_igemm_variants": {"achieved": round(sum(v["flops"] for v in allg.values()) / (sum(v["ms"] for v in allg.values()) * 1e9), 2),
                                       "share_of_gpu_time": round(sum(v["ms"] for v in allg.values()) / tot_ms, 3)},
                "kernel_families_ms_per_key_frame": {k: round(v["ms"] / spb, 4) for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:12]}}
    except Exception as e:  # noqa: BLE001  (an extra must never cost the leg)
        try:
            ops.set_profiler(None)
        except Exception:  # noqa: BLE001
            pass
        log("leg roofline skipped: %r" % (e,))
        return None


def f32_parity_leg(args, device, clip, gfor, T, spb, mode="float32", head_mode=None):
    """mode "bf16x3": the split-precision parity mode (cfg.F32_CONV = "bf16x3": the frame stage's convs / fc0 as bf16
    matrix-core GEMMs over [hi | lo | hi] . [Wh | Wh | Wl], f32 accumulation; aggregation head exact f32) -- pinned by
    tests/test_e2e_gpu.py::test_r101_600x1000_bf16x3_vs_oracle with the f32 test's bounds.  Otherwise:
    the SAME workload in the exact-f32 mode (cfg.DTYPE float32: v_mfma_f32_32x32x2_f32 everywhere) -- the mode whose
    outputs meet north_star's 1e-3 / bit-exact-index tolerance against the reference (tests/test_e2e_gpu.py::
    test_r101_600x1000_f32_vs_oracle, test_f32_long_clip_vs_reference_fixture).  Same engine, same steady-state rules
    (pools full, graphs captured and replayed before the timed blocks), a short timed region: blocks of `spb` key frames
    between synchronizes, >= 1.5 s.  Reported inside the headline line as config.f32_parity_mode."""
    from mega.pytorch_amd import engine as eng
    cfg, model, _ = build_model(args.arch, mode, device)
    frame_model = None
    if head_mode is not None:      # the frame stage in `mode`, the aggregation head as its own model in `head_mode`
        frame_model = model
        cfg, model, _ = build_model(args.arch, head_mode, device)
    if mode == "float32":
        spb = min(spb, 10)    # f32 activations: a 40-frame batch would exceed the kernels' 2 GiB-per-operand limit
    # (bf16x3: an activation is two bf16 planes = the bytes of an f32 map, but layer1's largest operand of a 40-frame batch
    #  -- [40,150,250,2 x 256] bf16 = 1.54 GB -- stays below the limit and fc0 runs in row chunks)
    runner = eng.ClipEngine(model, steps_per_batch=spb, overlap=not args.no_overlap, graphs=not args.no_graphs,
                            frame_model=frame_model)
    afi = cfg.MODEL.VID.MEGA.ALL_FRAME_INTERVAL
    pre = max(afi + 12 + 1, 3 * spb + 1)
    pre = 1 + -(-(pre - 1) // spb) * spb
    runner.run(clip, T, gfor, first=0, last=1)
    pos = 1
    while pos < pre:
        runner.run(clip, T, gfor, first=pos, last=pos + spb)
        pos += spb
    for _ in range(6):
        before = dict(runner.graph_stats)
        runner.run(clip, T, gfor, first=pos, last=pos + spb)
        torch.cuda.synchronize()
        pos += spb
        if runner.steady_state()["steady"] and runner.graph_stats["eager"] == before["eager"] and \
                runner.graph_stats["captured"] == before["captured"] and \
                runner.graph_stats.get("agg_captured", 0) == before.get("agg_captured", 0):
            break
    st0 = runner.steady_state()
    g0 = dict(runner.graph_stats)
    blocks = []
    while sum(blocks) < 1.5 and len(blocks) < 40 and pos + spb + 13 < T:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        runner.run(clip, T, gfor, first=pos, last=pos + spb)
        torch.cuda.synchronize()
        blocks.append(time.perf_counter() - t0)
        pos += spb
    g1 = dict(runner.graph_stats)
    srt = sorted(blocks)
    el = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
    fps = spb / el
    x3 = mode == "bf16x3"
    leg_roof = leg_roofline(runner, clip, T, gfor, pos, spb, mode) if not args.no_roofline else None
    if mode in ("float16", "f16x2"):
        out = {"dtype": ("f16 (frame stage on IEEE-half operands: v_mfma_f32_32x32x16_f16, 11 significant bits at the bf16 MFMA rate "
                         "and bytes; f32 accumulation; the head is the bf16 head on an f32 activation stream)") if mode == "float16" else
                        ("f16x2 (frame stage in the two-pass fp16 form: activations as float16 [hi | lo] planes against weights rounded "
                         "to fp16 once, K x 2 per conv / fc0, f32 accumulation; exact-f32 stem / RPN logits / ROIAlign; the bf16 head)"),
               "fps": round(fps, 2), "ms_per_key_frame": round(1e3 * el / spb, 4),
               "frac_of_2500TF": round(ALGO_GFLOP_PER_FRAME * 1e9 * fps / 2500e12, 4) if args.arch == "R-101" else None,
               "peak_tflops": 2500.0, "key_frames_per_block": spb, "timed_blocks": len(blocks),
               "timed_blocks_ms": [round(1e3 * b, 2) for b in blocks], "pools_full": bool(st0["pools_full"]),
               "graph_captures_in_timed_region": (g1["captured"] - g0["captured"]) + (g1["eager"] - g0["eager"])
               + (g1.get("agg_captured", 0) - g0.get("agg_captured", 0)),
               "parity": F16_PARITY if mode == "float16" else F16X2_PARITY}
        if mode == "f16x2":
            out["frac_of_2500TF_at_2x_flops"] = round(2 * ALGO_GFLOP_PER_FRAME * 1e9 * fps / 2500e12, 4) if args.arch == "R-101" else None
        out["roofline"] = leg_roof
        del runner, model, frame_model
        torch.cuda.empty_cache()
        return out
    out = {"dtype": ("bf16x3 (f32 activations as bf16 [hi | lo] planes, 3 bf16 MFMA passes per product, f32 accumulation; %s)"
                     % ("f32 head" if head_mode is None else ("fp16 head (Q K^T, P V, position term and projections on IEEE-half operands) "
                                                              "on an f32 activation stream") if head_mode == "f16head" else
                        "bf16 head on an f32 activation stream")) if x3 else "f32", "fps": round(fps, 2), "ms_per_key_frame": round(1e3 * el / spb, 4),
           "frac_of_157TF": round(ALGO_GFLOP_PER_FRAME * 1e9 * fps / 157.3e12, 4) if args.arch == "R-101" else None,
           "frac_of_2500TF_at_3x_flops": round(3 * ALGO_GFLOP_PER_FRAME * 1e9 * fps / 2500e12, 4) if x3 and args.arch == "R-101" else None,
           "peak_tflops": 157.3, "key_frames_per_block": spb, "timed_blocks": len(blocks),
           "timed_blocks_ms": [round(1e3 * b, 2) for b in blocks], "pools_full": bool(st0["pools_full"]),
           "graph_captures_in_timed_region": (g1["captured"] - g0["captured"]) + (g1["eager"] - g0["eager"])
           + (g1.get("agg_captured", 0) - g0.get("agg_captured", 0)),
           "parity": ("frame stage: the oracle's proposals (100 %% IoU-matched, as in the bf16x3 mode); head on fp16 operands: logit error "
                      "median 1.3-2.5e-5, p99 <= 1.5e-4 against the f32 oracle (the bf16 head: 1.0-1.9e-4 / <= 7e-4), 100 %% of the "
                      "detections (tests/test_e2e_gpu.py::test_r101_600x1000_f16_head_vs_oracle, run X3H)") if x3 and head_mode == "f16head" else
                     ("frame stage: the oracle's proposals (100 %% IoU-matched); head: logit error median 1.0-1.9e-4, p99 <= 7e-4 "
                      "against the f32 oracle, 98.7-100 %% of the detections (tests/test_e2e_gpu.py::test_r101_bf16_attribution, "
                      "run X)") if x3 and head_mode is not None else
                     ("kept anchor indices bit for bit and every detection on the fixture with margins; seeded fixture: 100 %% of "
                      "the proposals IoU-matched, logit error median 6-7e-6 / p99 <= 1.3e-4, 100 %% of the detections "
                      "(tests/test_e2e_gpu.py: test_r101_calibrated_bf16_agreement leg X3, test_r101_600x1000_bf16x3_vs_oracle)")
                     if x3 else
                     ("logits within 1e-3 of the reference / oracle, identical detections (tests/test_e2e_gpu.py: "
                      "test_r101_600x1000_f32_vs_oracle, test_f32_long_clip_vs_reference_fixture)")}
    out["roofline"] = leg_roof
    del runner, model, frame_model
    torch.cuda.empty_cache()
    return out


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:        # no launcher around us: start the ranks ourselves
        sys.exit(self_launch(args, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: launched with WORLD_SIZE=%d but --gpus %d; they must agree\n" % (world, args.gpus))
        sys.exit(2)
    # stdout carries exactly ONE line (the JSON): everything else any library writes to fd 1 -- RCCL prints a
    # version banner there at exit -- is sent to stderr.
    json_fd = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)
    if args.dry_run:
        return dry_run(args, world, rank, local_rank, json_fd)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device (the MEGA hot path has no CPU fallback)")
    if torch.cuda.device_count() <= local_rank:
        sys.stderr.write("bench.py: rank %d has no HIP device (%d visible)\n" % (local_rank, torch.cuda.device_count()))
        sys.exit(2)
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
        assert dist.get_world_size() == world, "process group has %d ranks, WORLD_SIZE %d" % (dist.get_world_size(), world)
    from mega.pytorch_amd import engine as eng, ops

    log("building model")
    cfg, model, sd = build_model(args.arch, args.dtype, device, args.head_stream)
    from mega.pytorch_amd import modeling as _modeling
    modeling_conv_mode = _modeling.conv_mode(cfg)
    log("model ready")
    K = args.steps
    KF = key_frames_per_block(K, world)           # key frames per timed block (a step = `world` key frames)
    if args.steps_per_batch <= 0:
        args.steps_per_batch = default_steps_per_batch(K, world)
    spb = args.steps_per_batch
    afi = cfg.MODEL.VID.MEGA.ALL_FRAME_INTERVAL
    # pre-roll (untimed): cold start + enough key frames to fill the 25-entry memory deques (frames >= 37)
    # and to see the steady frame-stage batch shape three times (eager -> captured -> replayed); --warmup can only
    # lengthen it.  Rounded so that the timed region starts on a batch boundary.
    pre = max(args.warmup, afi + 12 + 1, 3 * spb + 1)
    pre = 1 + -(-(pre - 1) // spb) * spb
    extra_cap = 6 * max(spb, KF)                  # further pre-roll blocks if the engine is not yet in steady state
    max_blocks = max(1, min(args.max_blocks, -(-6000 // KF)))
    prof_steps = 0 if args.no_roofline else spb       # the instrumented pass runs the steady batch shape
    # (+ the with-H2D leg's blocks: one GPU only -- the clip is resident on every rank, 1.8 MB per frame)
    h2d_T = (62 * KF + 40 * max(spb, 1) + 8) if (world == 1 and not args.no_h2d_leg and args.dtype in HALF_MODES) else 0
    T = pre + KF + extra_cap + KF * max_blocks + 2 * prof_steps + 1 + KF + 13 + h2d_T
    clip = make_clip(T, args.height, args.width, device)
    gfor = eng.global_schedule(T, cfg.MODEL.VID.MEGA.GLOBAL.SIZE, seed=0)
    runner = eng.ClipEngine(model, steps_per_batch=spb, dist_group=group, overlap=not args.no_overlap,
                            graphs=not args.no_graphs, reuse_records=args.reuse_records,
                            static_aggregation=args.aggregation == "static",
                            batch_aggregation=args.aggregation in ("batched", "batched-eager"), ramp=args.ramp,
                            graph_aggregation=args.aggregation == "batched")

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def engine_state():
        fe = model.roi_heads.box.feature_extractor
        st = runner.steady_state()
        st["graph_stats"] = dict(runner.graph_stats)
        st["static_graph_replays"] = runner.static_replays()
        return st

    # ---- pre-roll
    log("clip ready (%d frames); pre-roll: cold start + %d key frames" % (T, pre - 1))
    runner.run(clip, T, gfor, first=0, last=1)
    barrier()
    log("cold start done")
    pos = 1
    while pos < pre:                       # same call pattern as the timed blocks: the same batch shapes get captured
        runner.run(clip, T, gfor, first=pos, last=pos + KF)
        barrier()
        pos += KF
    prev_stats = None
    while pos < pre + extra_cap:           # until a whole block runs without an eager batch or a capture
        before = dict(runner.graph_stats)
        runner.run(clip, T, gfor, first=pos, last=pos + KF)
        barrier()
        pos += KF
        after = runner.graph_stats
        if runner.steady_state()["steady"] and (args.no_graphs or (
                after["eager"] == before["eager"] and after["captured"] == before["captured"]
                and after.get("agg_captured", 0) == before.get("agg_captured", 0))):
            break
    st0 = engine_state()
    log("pre-roll done at key frame %d: %s" % (pos, st0))
    if not st0["steady"]:
        log("WARNING: engine did not reach the steady state in the pre-roll (%s)" % (st0,))
    for k in runner.host_times:
        runner.host_times[k] = 0
    fc_before = runner.frames_computed

    # ---- timed region: blocks of EXACTLY K key frames, each bracketed by barrier + synchronize
    def timed_block(first):
        barrier()
        t0 = time.perf_counter()
        d = runner.run(clip, T, gfor, first=first, last=first + KF)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, d

    blocks = []
    dt, dets = timed_block(pos)
    blocks.append(dt)
    pos += KF
    nblk = max(1, min(max_blocks, int(-(-args.min_seconds // dt))))
    if world > 1:      # every rank must run the same number of blocks
        t = torch.tensor([nblk], dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        nblk = min(max_blocks, int(t.item()))
    for _ in range(nblk - 1):
        dt, dets = timed_block(pos)
        blocks.append(dt)
        pos += KF
    st1 = engine_state()
    runner_frames_after = runner.frames_computed
    srt = sorted(blocks)
    elapsed = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
    fps = KF / elapsed
    log("timed region: %d blocks of %d steps (%d key frames), median %.4fs (%.2f frames/s), min %.4f max %.4f, total %.2fs" % (
        len(blocks), K, KF, elapsed, fps, srt[0], srt[-1], sum(blocks)))
    captures_in_region = (st1["graph_stats"]["captured"] - st0["graph_stats"]["captured"]
                          + st1["graph_stats"]["eager"] - st0["graph_stats"]["eager"]
                          + st1["graph_stats"].get("agg_captured", 0) - st0["graph_stats"].get("agg_captured", 0))
    log("engine state after the timed region: %s" % (st1,))
    if not args.no_graphs:
        assert captures_in_region == 0, "a frame-stage batch ran eagerly / was captured inside the timed region: %s -> %s" % (st0, st1)
    assert st1["pools_full"], "pools not full in the timed region: %s" % (st1,)
    ht = dict(runner.host_times)
    log("host ms/step: frame-stage enqueue %.3f, aggregation enqueue %.3f (of which waiting for the frame stage %.3f), "
        "waiting for results %.3f" % tuple(1e3 * ht[k] / max(ht["steps"], 1)
                                            for k in ("frame_enqueue", "aggregate_enqueue", "frame_wait", "finish_wait")))
    Wm = pos - KF * len(blocks)       # key frames processed before the first timed block

    # ---- the same blocks with the frames ARRIVING FROM HOST MEMORY (the reference's timer starts before the H2D copy:
    #      mega_core/engine/inference.py:24-42, SURVEY 8d "tic before H2D+forward"; `value` above times a clip resident in
    #      HBM).  The synthetic clip lives in host memory as uint8 frames; feed.FrameSource (the reference's
    #      test-time feed) copies each step-batch's frames into pinned staging buffers and issues ONE async H2D copy per
    #      batch; the engine, its graphs and the video state are the ones of the timed region.
    h2d_leg = None
    if world == 1 and not args.no_h2d_leg and args.dtype in HALF_MODES:
        try:
            from mega.pytorch_amd import feed
            host = clip[:16].cpu().numpy()
            src = feed.FrameSource(None, None, T, device, min_size=args.height, max_size=max(args.width, args.height),
                                   opener=lambda f: host[f % 16], workers=8, cache_frames=4 * spb + 32)
            assert tuple(src.out_hw) == (args.height, args.width), src.out_hw
            hb = []
            runner.run(src, T, gfor, first=pos, last=pos + KF)       # (untimed: allocates the pinned staging ring)
            barrier()
            pos += KF
            while sum(hb) < 1.0 and len(hb) < 60:
                barrier()
                t0 = time.perf_counter()
                runner.run(src, T, gfor, first=pos, last=pos + KF)
                barrier()
                hb.append(time.perf_counter() - t0)
                pos += KF
            hs = sorted(hb)
            hel = hs[len(hs) // 2] if len(hs) % 2 else 0.5 * (hs[len(hs) // 2 - 1] + hs[len(hs) // 2])
            # the same comparison in 5-batch blocks: the host copies and the H2D transfer of batch i + 1 then run beside the
            # GPU work of batch i (the engine enqueues a batch's frame stage while the previous aggregation is in flight)
            long_blk = {}
            for name, source in (("resident", clip), ("with_h2d", src)):
                tl = []
                for _ in range(4):
                    barrier()
                    t0 = time.perf_counter()
                    runner.run(source, T, gfor, first=pos, last=pos + 5 * spb)
                    barrier()
                    tl.append(time.perf_counter() - t0)
                    pos += 5 * spb
                long_blk[name] = round(5 * spb / sorted(tl[1:])[1], 2)
            st_h = engine_state()
            h2d_leg = {"fps": round(KF / hel, 2), "ms_per_key_frame": round(1e3 * hel / KF, 4), "vs_resident": round(elapsed / hel, 4),
                       "timed_blocks": len(hb), "timed_blocks_ms": [round(1e3 * b, 2) for b in hb],
                       "bytes_h2d_per_key_frame": 2 * args.height * args.width * 3,
                       "blocks_of_%d_key_frames_fps" % (5 * spb): long_blk,
                       "graph_captures_in_leg": (st_h["graph_stats"]["captured"] - st1["graph_stats"]["captured"]
                                                 + st_h["graph_stats"]["eager"] - st1["graph_stats"]["eager"]),
                       "how": "uint8 frames in host memory -> feed.FrameSource (memcpy into a ring of pinned staging buffers, one "
                              "async H2D copy per step-batch) -> the same engine and hipGraphs; blocks of --steps key frames between "
                              "barrier + synchronize, median"}
            log("with H2D: %.1f frames/s (%.4f of the resident-clip rate)" % (h2d_leg["fps"], h2d_leg["vs_resident"]))
            src.close()
        except Exception as e:  # noqa: BLE001  (an extra leg must never cost the headline line)
            log("with-H2D leg skipped: %r" % (e,))

    roofline = None
    roofline_hbm = []
    roofline_mfma = []
    fam = {}
    if prof_steps:
        # Instrumented pass: kernel by kernel (no hipGraph replays, one stream), every launch between a HIP event pair on
        # the launch stream.  One UNTIMED eager batch of the same shape goes first: the first eager launches after the
        # graphs are switched off pay one-time costs (allocator growth, kernel attribute calls) that belong to no kernel.
        runner.use_graphs = False
        runner.use_static = False
        runner.overlap = False     # per-kernel event pairs are only meaningful without cross-stream concurrency
        runner.run(clip, T, gfor, first=pos, last=pos + prof_steps)
        barrier()
        pos += prof_steps
        p = ops.Profiler()
        ops.set_profiler(p)
        runner.run(clip, T, gfor, first=pos, last=pos + prof_steps)
        summ = p.summary()
        detail = p.summary(by_detail=True)
        ops.set_profiler(None)
        pos += prof_steps
        tot_ms = sum(v["ms"] for v in summ.values())
        for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
            fam[k] = {"ms_per_step": round(v["ms"] / prof_steps, 4), "launches_per_step": round(v["launches"] / prof_steps, 1),
                      "tflops": round(v["flops"] / (v["ms"] * 1e9), 2) if v["ms"] > 0 else 0.0,
                      "gbps": round(v["bytes"] / (v["ms"] * 1e6), 1) if v["ms"] > 0 else 0.0}
        # dominant kernel = the matrix-core-bound igemm instantiation -- exactly ONE rocprofv3 kernel symbol per family
        # (ops._igemm_family keys on tile, launch class AND output type) -- with the largest share of GPU time in this
        # instrumented pass of the timed configuration.  igemm8's streaming launch class ("igemm8s_*": 1x1 layers with
        # K <= 512, bound by HBM / the CU fetch rate) is reported in roofline_hbm; every other igemm symbol (fc0's f32-output
        # split-K launch among them) has its own row in roofline_mfma.
        tag = "f16" if args.dtype in ("float16", "f16x2") else ("bf16" if args.dtype in ("bfloat16", "wide") else "f32")
        igemms = {k: v for k, v in summ.items() if k.startswith(("igemm_" + tag, "igemm8_" + tag, "igemm8s_" + tag, "igemm8_sp"))}
        mm = {k: v for k, v in igemms.items() if not k.startswith("igemm8s_")}
        peak = 2500.0 if args.dtype in HALF_MODES + ("bf16x3",) else 157.3

        symbol_of = igemm_symbol
        for k, v in sorted(mm.items(), key=lambda kv: -kv[1]["ms"]):
            a_ = v["flops"] / (v["ms"] * 1e9) if v["ms"] > 0 else 0.0
            roofline_mfma.append({"kernel": symbol_of(k), "family": k, "bound": "mfma", "achieved": round(a_, 2), "peak": peak,
                                  "unit": "TFLOP/s", "frac": round(a_ / peak, 4), "launches": v["launches"],
                                  "sum_gflop": round(v["flops"] / 1e9, 2), "sum_us": round(1e3 * v["ms"], 1),
                                  "ms_per_key_frame": round(v["ms"] / prof_steps, 4), "share_of_gpu_time": round(v["ms"] / tot_ms, 3)})
        dom = max(mm, key=lambda k: mm[k]["ms"])
        tile = dom.split("_")[2].split("x") if not dom.startswith("igemm8_sp") else ["256", "256"]
        is8 = dom.startswith("igemm8")
        d = summ[dom]
        ach = d["flops"] / (d["ms"] * 1e9)
        fam_ms = sum(v["ms"] for v in igemms.values())
        fam_fl = sum(v["flops"] for v in igemms.values())
        # HBM traffic per launch of the dominant symbol: the rocprofv3 PMC passes of this same command (FETCH_SIZE and
        # WRITE_SIZE in separate passes, gfx950 x2 read correction; tools/pmc_summary.py) cannot run inside bench.py, so
        # the number is read from the newest committed summary and labelled with its source -- it is a property of that
        # profiled run of this code, not of this run.
        traffic, traffic_src = None, None
        sym = symbol_of(dom)
        for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
            pmc = os.path.join(ROOT, "profiles", rnd + {"float16": "_f16", "f16x2": "_f16x2", "bf16x3": "_x3"}.get(args.dtype, "") + "_pmc_summary.json")
            if os.path.exists(pmc):
                ks = json.load(open(pmc))["kernels"]
                want_ = sym.replace("unsigned short", "bf16").replace(" ", "")      # (tools/pmc_summary.py's short names)
                k = next((v for name, v in ks.items() if name.replace(" ", "").startswith(want_)), None)
                if k is None:      # summaries of round 5: <OT, MF1, CLS, ABL, SP> without the operand-type parameter
                    w5 = want_[:want_.rindex(",")] + ">"
                    k = next((v for name, v in ks.items() if name.replace(" ", "").startswith(w5)), None)
                if k is None:      # summaries of rounds 1-4: <OT, MF1, CLS, ABL> without the SP parameter
                    w4 = want_[:want_.rindex(",")]
                    w4 = w4[:w4.rindex(",")]
                    k = next((v for name, v in ks.items() if name.replace(" ", "").startswith(w4)), None)
                if k:
                    traffic, traffic_src = round(k["hbm_bytes_per_launch_corrected"]), "profiles/%s_pmc_summary.json" % rnd
                    break
        roofline = {"bound": "mfma", "kernel": sym + " (implicit-GEMM conv / linear%s)" % (
                        ", %sx256 tile" % tile[0] if is8 else ""),
                    "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                    "traffic": traffic, "traffic_source": traffic_src, "avg_launch_us": round(1e3 * d["ms"] / d["launches"], 2),
                    "flops_per_launch": round(d["flops"] / d["launches"], 0),
                    "recomputed_from": {"launches": d["launches"], "sum_gflop": round(d["flops"] / 1e9, 2),
                                        "sum_us": round(1e3 * d["ms"], 1), "key_frames": prof_steps,
                                        "how": "achieved = sum_gflop / sum_us (HIP events around every launch of this symbol "
                                               "in one instrumented step-batch)"},
                    "algorithmic_bytes_per_launch": round(d["bytes"] / d["launches"], 0),
                    "share_of_gpu_time": round(d["ms"] / tot_ms, 3),
                    "all_igemm_variants": {"achieved": round(fam_fl / (fam_ms * 1e9), 2),
                                           "frac": round(fam_fl / (fam_ms * 1e9) / peak, 4),
                                           "share_of_gpu_time": round(fam_ms / tot_ms, 3)},
                    "whole_path_frac": round(ALGO_GFLOP_PER_FRAME * 1e9 * fps / (peak * 1e12), 5)
                    if args.arch == "R-101" else None}

        # the HBM-bound sub-steps SURVEY 8d asks for separately: algorithmic bytes / event time against 8 TB/s
        def hbm_entry(name, sel):
            rows = [v for k, v in detail.items() if sel(k[0], k[1] or "")]
            ms = sum(v["ms"] for v in rows)
            n = sum(v["launches"] for v in rows)
            if not n or ms <= 0:
                return
            by = sum(v["bytes"] for v in rows)
            roofline_hbm.append({"kernel": name, "bound": "hbm", "achieved": round(by / (ms * 1e6), 1), "peak": 8000.0,
                                 "unit": "GB/s", "frac": round(by / (ms * 1e6) / 8000.0, 4), "launches": n,
                                 "algorithmic_bytes_per_launch": round(by / n), "avg_launch_us": round(1e3 * ms / n, 2),
                                 "ms_per_key_frame": round(ms / prof_steps, 4)})

        def shape_of(det):       # "MxCoutxK (RxS)" -> (M, Cout, K, R)
            try:
                a, b2 = det.split(" ")
                M_, C_, K_ = (int(x) for x in a.split("x"))
                return M_, C_, K_, int(b2[1])
            except Exception:    # noqa: BLE001
                return None
        conv = lambda f: f.startswith("igemm")      # noqa: E731
        hbm_entry("igemm8 streaming class (1x1 layers with K <= 512: layer3 conv3, res5 conv3, layer2 / layer3 projections)",
                  lambda f, d_: f.startswith("igemm8s_"))
        hbm_entry("stem: 7x7/2 conv + BN + ReLU + 3x3/2 max-pool in one kernel (uint8 frames in, pooled map out)"
                  if os.environ.get("MEGA_STEM_POOL", "1") != "0" else "stem 7x7/2 conv + BN + ReLU", lambda f, d_: f == "stem")
        hbm_entry("max-pool 3x3/2", lambda f, d_: f == "maxpool")
        hbm_entry("layer1 convs (Cin or Cout = 64)", lambda f, d_: conv(f) and shape_of(d_) is not None
                  and (shape_of(d_)[2] in (64, 576) or shape_of(d_)[1] == 64))
        hbm_entry("layer3 1x1 256->1024 + residual", lambda f, d_: conv(f) and shape_of(d_) is not None
                  and shape_of(d_)[1:] == (1024, 256, 1))
        hbm_entry("layer3 1x1 1024->256", lambda f, d_: conv(f) and shape_of(d_) is not None
                  and shape_of(d_)[1:] == (256, 1024, 1) and shape_of(d_)[0] > 10000)
        hbm_entry("ROIAlign (gather + [K,49,C] write)", lambda f, d_: f == "roi_align")
        hbm_entry("fc0 (K = 100352 weight + activation stream)", lambda f, d_: conv(f) and shape_of(d_) is not None
                  and shape_of(d_)[2] >= 50000)
        hbm_entry("position logits (write of the [16,Nq,Nk] bf16 logits)", lambda f, d_: f == "pos_logits")

    # whole-clip rate (SURVEY 8d ii): a fresh video -- cold start (13 local + 10 global frames, eager aggregation
    # while the pools fill) plus the same K steady key frames -- on the warmed-up engine.  Reported beside `value`.
    whole_clip = None
    try:
        if args.no_whole_clip:
            raise RuntimeError("--no-whole-clip")
        runner.use_graphs, runner.overlap = not args.no_graphs, not args.no_overlap   # (the instrumented pass turned them off)
        runner.use_static = True
        barrier()
        t0 = time.perf_counter()
        runner.run(clip, T, gfor, first=0, last=1 + KF)
        barrier()
        wc = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([wc], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wc = float(t.item())
        whole_clip = {"key_frames": 1 + KF, "seconds": round(wc, 4), "frames_per_s": round((1 + KF) / wc, 2)}
        log("whole clip incl. cold start: %d key frames in %.3fs (%.1f frames/s)" % (1 + KF, wc, (1 + KF) / wc))
    except Exception as e:  # noqa: BLE001  (an optional extra must never cost the headline line)
        log("whole-clip measurement skipped: %r" % (e,))

    f32_leg = x3_leg = f16_leg = None
    if world == 1 and args.dtype == "bfloat16" and not args.no_f32_leg:
        try:
            f16_leg = f32_parity_leg(args, device, clip, gfor, T, spb, mode="float16")
            log("fp16-mode leg: %.1f frames/s (%.3f ms per key frame)" % (f16_leg["fps"], f16_leg["ms_per_key_frame"]))
            f16_leg["two_pass"] = f32_parity_leg(args, device, clip, gfor, T, spb, mode="f16x2")
            log("fp16 two-pass leg: %.1f frames/s" % f16_leg["two_pass"]["fps"])
        except Exception as e:  # noqa: BLE001  (an extra leg must never cost the headline line)
            log("fp16-mode leg skipped: %r" % (e,))
        try:
            f32_leg = f32_parity_leg(args, device, clip, gfor, T, spb)
            log("f32 parity-mode leg: %.1f frames/s (%.3f ms per key frame, %.3f of the 157 TF/s f32 MFMA peak)" % (
                f32_leg["fps"], f32_leg["ms_per_key_frame"], f32_leg["frac_of_157TF"] or 0.0))
        except Exception as e:  # noqa: BLE001  (an extra leg must never cost the headline line)
            log("f32 parity-mode leg skipped: %r" % (e,))
        try:
            x3_leg = f32_parity_leg(args, device, clip, gfor, T, spb, mode="bf16x3")
            log("bf16x3 parity-mode leg: %.1f frames/s (%.3f ms per key frame)" % (x3_leg["fps"], x3_leg["ms_per_key_frame"]))
            x3_leg["with_bf16_head"] = f32_parity_leg(args, device, clip, gfor, T, spb, mode="bf16x3", head_mode="bfloat16")
            log("bf16x3 frame stage + bf16 head (f32 stream): %.1f frames/s" % x3_leg["with_bf16_head"]["fps"])
            x3_leg["with_f16_head"] = f32_parity_leg(args, device, clip, gfor, T, spb, mode="bf16x3", head_mode="f16head")
            log("bf16x3 frame stage + fp16 head (f32 stream): %.1f frames/s" % x3_leg["with_f16_head"]["fps"])
        except Exception as e:  # noqa: BLE001
            log("bf16x3 parity-mode leg skipped: %r" % (e,))

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.arch, sd, args.height, args.width, args.cpu_frames)

    live_world = dist.get_world_size() if group is not None else 1
    if rank == 0:
        ndet = sum(len(d) for d in dets) / max(len(dets), 1)
        line = {
            "metric": "frames/sec MEGA %s inference, %dx%d VID clip" % (args.arch, args.width, args.height),
            "value": round(fps, 3), "unit": "frames/s", "n_gpus": live_world, "steps": K, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / K, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {"bfloat16": "bf16", "wide": "bf16", "bf16x3": "bf16x3", "float16": "f16", "f16x2": "f16x2"}.get(args.dtype, "f32"), "data": "synthetic",
            "config": {"workload": "MEGA %s-C4, %dx%d frames, 25 local + 10 global frames + 25-frame memory, "
                                   "300 key / 75 ref proposals, 3 attention stages (BASELINE configs[2]%s)"
                                   % (args.arch, args.width, args.height, "" if world == 1 else " sharded = configs[3]"),
                       "step": "%d key frame%s" % (world, "" if world == 1 else "s of the video (one per rank; the frame stage of "
                                                   "a step-batch is sharded, its records all-gathered over RCCL)"),
                       "key_frames_per_block": KF, "ms_per_key_frame": round(1e3 * elapsed / KF, 4),
                       "steps_per_batch": args.steps_per_batch, "batch_sizes_in_a_block": runner.batch_sizes(KF),
                       "frames_per_rank_per_batch": -(-2 * args.steps_per_batch // world),
                       "parallelism": "frame-sharded x%d" % world,
                       "frame_record_reuse": bool(args.reuse_records),
                       "aggregation": args.aggregation,
                       "pre_roll_key_frames": Wm, "pools_full": bool(st0["pools_full"] and st1["pools_full"]),
                       "graph_captures_in_timed_region": captures_in_region,
                       "engine_state_before": st0, "engine_state_after": st1,
                       "timed_blocks": len(blocks), "timed_blocks_ms": [round(1e3 * b, 2) for b in blocks],
                       "value_is": "median block (each block = exactly --steps steps between barrier+synchronize)",
                       "clip": "16 unique synthetic frames repeated (only the RPN NMS's early stop depends on frame content)",
                       "frames_through_frame_stage_per_key_frame": round(
                           (runner_frames_after - fc_before) / (KF * len(blocks)), 2),
                       "avg_detections": round(ndet, 1),
                       "key_proposals_last_frame": int(model.records[model.key_frame_location]["boxes"].shape[0]),
                       "whole_clip_incl_cold_start": whole_clip,
                       "head_stream": str(getattr(cfg, "HEAD_STREAM", None)) if args.dtype in HALF_MODES else "float32",
                       "conv_mode": modeling_conv_mode,
                       "fp16_mode": f16_leg, "f32_parity_mode": f32_leg, "bf16x3_parity_mode": x3_leg, "with_h2d": h2d_leg},
            "roofline": roofline, "roofline_mfma": roofline_mfma, "roofline_hbm": roofline_hbm, "cpu_baseline": cpu,
            "kernel_families": fam,
        }
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if world > 1 or os.environ.get("MEGA_FORCE_SHARDED") == "1":
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""CPU, world_size 2 and 4 over gloo: the sharded ClipEngine must reproduce the single-process result.

What is sharded (mega/pytorch_amd/engine.py): (1) the frame stage -- each rank computes a slice of every frame-stage
batch, the fixed-size frame records travel in one packed all-gather per row-count group; (2) the aggregation -- the
key frames of a step-batch are dealt round-robin to the ranks (KeyFrameShard), the memory entries of every key
frame are all-gathered once per stage and the padded detections at the end; nothing of the per-key-frame step is
replicated.  The kernels are replaced by the oracle-backed CPU twins (tests/cpu_ops.py); on the GPU box the same
code path runs over RCCL (backend "nccl") -- see bench.py --gpus N.
"""
import os
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
T, NKEY = 22, 14


def _install_cpu_ops():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import cpu_ops
    from mega.pytorch_amd import ops
    for name in cpu_ops.ALL:
        setattr(ops, name, getattr(cpu_ops, name))


def _build(T=T):
    """R-50 MEGA with a 7-frame window / memory and a 3-frame global pool: 14 key frames wrap every deque."""
    from mega.pytorch_amd import config, modeling, synth
    cfg = config.get_cfg("R-50")
    cfg.MODEL.DEVICE = "cpu"
    cfg.merge_from_list(["MODEL.VID.MEGA.ALL_FRAME_INTERVAL", 7, "MODEL.VID.MEGA.KEY_FRAME_LOCATION", 3,
                         "MODEL.VID.MEGA.MIN_OFFSET", -3, "MODEL.VID.MEGA.MAX_OFFSET", 3, "MODEL.VID.MEGA.GLOBAL.SIZE", 3,
                         "MODEL.RPN.POST_NMS_TOP_N_TEST", 40, "MODEL.VID.RPN.REF_POST_NMS_TOP_N", 10])
    model = modeling.build_detection_model(cfg)
    model.load_state_dict(synth.make_state_dict(blocks=(3, 4, 6), reduce_channel=True, global_res_stage=0, seed=5))
    frames = synth.preprocess_cpu(synth.make_clip(T, 96, 128, seed=2))
    return cfg, model, frames


def _worker(rank, world, port, outdir, spb, T=T, NKEY=NKEY):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _install_cpu_ops()
    from mega.pytorch_amd import engine
    cfg, model, frames = _build(T)
    gfor = engine.global_schedule(T, 3, seed=0)
    eng = engine.ClipEngine(model, steps_per_batch=spb, dist_group=dist.group.WORLD, keep_logits=True)
    dets = eng.run(frames, T, gfor, first=0, last=NKEY)
    fe = model.roi_heads.box.feature_extractor
    torch.save({"dets": [(d.bbox, d.get_field("scores"), d.get_field("labels")) for d in dets],
                "mem": [fe.mem[i]["k"].clone() for i in range(fe.stage)], "frames_computed": eng.frames_computed,
                "wire": dict(eng.wire), "wire_order": list(eng.wire_order),
                "own_logits": sum(1 for x in eng.logits_log if x is not None)},
               os.path.join(outdir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("world,spb,T,NKEY", [(2, 3, 22, 14), (4, 5, 22, 14), (8, 16, 46, 34)])
def test_sharded_engine_matches_single_process(world, spb, T, NKEY):
    """world 8 = BASELINE configs[3]'s rank count with bench.py's step-batch rule scaled down (S = 2 N key frames per
    step-batch instead of 20 N): two full 16-key-frame batches after the cold start, every rank owning two key frames of
    each."""
    port = 29500 + (os.getpid() * 7 + world) % 2000
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, port, d, spb, T, NKEY), nprocs=world, join=True)
        ranks = [torch.load(os.path.join(d, "rank%d.pt" % r)) for r in range(world)]
    _install_cpu_ops()
    torch.set_num_threads(4)
    from mega.pytorch_amd import engine
    cfg, model, frames = _build(T)
    single_eng = engine.ClipEngine(model, steps_per_batch=spb)
    single = single_eng.run(frames, T, engine.global_schedule(T, 3, seed=0), first=0, last=NKEY)
    assert all(len(r["dets"]) == NKEY for r in ranks) and len(single) == NKEY
    for k in range(NKEY):
        b0, s0, l0 = ranks[0]["dets"][k]
        n = len(single[k])
        for r in ranks[1:]:                        # every rank returns the detections of ALL key frames, identical
            b1, s1, l1 = r["dets"][k]
            assert torch.equal(l0, l1) and torch.equal(b0, b1) and torch.equal(s0, s1)
        # and they equal the un-sharded run (CPU twins are not batch-invariant to the last bit: MKL)
        assert b0.shape[0] == n and torch.equal(l0, single[k].get_field("labels")), k
        assert (b0 - single[k].bbox).abs().max() < 5e-3 and (s0 - single[k].get_field("scores")).abs().max() < 1e-5
    # the memory pools stay replicated: same rows on every rank (every rank replays all pushes in frame order)
    for r in ranks[1:]:
        for a, b in zip(ranks[0]["mem"], r["mem"]):
            assert a.shape == b.shape and (a.float() - b.float()).abs().max() < 1e-4
    # frame-stage work is divided: a rank computes about 1/world of the frames
    assert ranks[0]["frames_computed"] <= single_eng.frames_computed
    # ---- bytes on the wire: a frame travels as ONE record of the REF_POST_NMS_TOP_N (here 10) rows every
    # rank's window reads -- boxes, scores, feature rows, count -- never as its 40-row key-role record (rows 10-39 stay on
    # the rank that owns key frame f); a key frame's aggregation runs on exactly one rank
    bn, D, esz = 10, 1024, 4                                   # (CPU twins: f32 stream)
    rec = bn * 16 + (bn * 4 + 15) // 16 * 16 + bn * D * esz + 16
    nbatch = 1 + -(-(NKEY - 1) // spb)
    njobs = 13 + 3 + 2 * (NKEY - 1)                            # cold start: 13 local + 3 global frames; then 1 + 1 per key frame
    slots = sum(r["wire"]["frame_records"] for r in ranks) / float(rec)
    assert slots == int(slots) and njobs <= slots <= njobs + nbatch * 2 * (world - 1), (slots, njobs)
    per_kf = sum(r["wire"]["frame_records"] for r in ranks) / NKEY
    print("world %d: %.0f bytes of frame records per key frame on the wire (2 records = %d), memory rows %.0f B, detections %.0f B"
          % (world, per_kf, 2 * rec, sum(r["wire"].get("memory_rows", 0) for r in ranks) / NKEY,
             sum(r["wire"].get("detections", 0) for r in ranks) / NKEY))
    # memory entries: (10 + 2 + 2) feature rows per key frame and stage set, padded to the batch's per-rank slot count
    mem_rows = sum(r["wire"]["memory_rows"] for r in ranks) / (D * esz)
    assert mem_rows <= nbatch * world * -(-spb // world) * (10 + 10 + 10), mem_rows
    assert sum(r["own_logits"] for r in ranks) == NKEY         # every key frame aggregated by exactly ONE rank
    # ---- order of the collectives (identical on every rank, or the ranks would deadlock / mix buffers): per step-batch the
    # frame records are gathered BEFORE its aggregation starts, then one memory-row gather per stage set, the detections last
    for r in ranks[1:]:
        assert r["wire_order"] == ranks[0]["wire_order"]
    order = ranks[0]["wire_order"]
    assert order.count("aggregate") == nbatch
    pos = [i for i, k in enumerate(order) if k == "aggregate"]
    for bi, p0 in enumerate(pos):
        # (the engine enqueues the frame stage of batch i + 1 -- and its record gather -- before it aggregates batch i: what
        #  must hold is that batch i's records were gathered before ITS aggregation starts)
        assert order[:p0].count("frame_records") >= bi + 1, (bi, order[:p0])
        seg = [k for k in order[p0 + 1:pos[bi + 1] if bi + 1 < len(pos) else len(order)] if k != "frame_records"]
        assert seg.count("detections") == 1 and "memory_rows" in seg[:seg.index("detections")], (bi, seg)
        assert all(k != "memory_rows" for k in seg[seg.index("detections"):])   # ... and its detections last

"""CPU, world_size 2 over gloo: the frame-sharded ClipEngine (each rank computes a slice of every frame-stage
batch, fixed-size frame records exchanged with ONE all-gather, aggregation replicated) must reproduce the
single-process result.  The kernels are replaced by the oracle-backed CPU twins (tests/cpu_ops.py); on the
GPU box the same code path runs over RCCL (backend "nccl") -- see bench.py --gpus N.
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


def _install_cpu_ops():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import cpu_ops
    from mega.pytorch_amd import ops
    for name in cpu_ops.ALL:
        setattr(ops, name, getattr(cpu_ops, name))


def _build():
    from mega.pytorch_amd import config, modeling, synth
    cfg = config.get_cfg("R-50")
    cfg.MODEL.DEVICE = "cpu"
    model = modeling.build_detection_model(cfg)
    model.load_state_dict(synth.make_state_dict(blocks=(3, 4, 6), reduce_channel=True, global_res_stage=0, seed=5))
    frames = synth.preprocess_cpu(synth.make_clip(16, 96, 128, seed=2))
    return cfg, model, frames


def _worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _install_cpu_ops()
    from mega.pytorch_amd import engine
    cfg, model, frames = _build()
    gfor = engine.global_schedule(16, 10, seed=0)
    eng = engine.ClipEngine(model, steps_per_batch=2, dist_group=dist.group.WORLD)
    dets = eng.run(frames, 16, gfor, first=0, last=4)
    torch.save([(d.bbox, d.get_field("scores"), d.get_field("labels")) for d in dets],
               os.path.join(outdir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_frame_sharded_engine_matches_single_process():
    port = 29500 + os.getpid() % 2000
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, port, d), nprocs=2, join=True)
        r0 = torch.load(os.path.join(d, "rank0.pt"))
        r1 = torch.load(os.path.join(d, "rank1.pt"))
    _install_cpu_ops()
    torch.set_num_threads(4)
    from mega.pytorch_amd import engine
    cfg, model, frames = _build()
    single = engine.ClipEngine(model, steps_per_batch=2).run(frames, 16, engine.global_schedule(16, 10, seed=0),
                                                            first=0, last=4)
    assert len(r0) == len(r1) == len(single) == 4
    for (b0, s0, l0), (b1, s1, l1), det in zip(r0, r1, single):
        # replicated aggregation: every rank holds the same detections
        assert torch.equal(l0, l1) and torch.allclose(b0, b1, atol=1e-4) and torch.allclose(s0, s1, atol=1e-6)
        # and they equal the un-sharded run (CPU twins are not batch-invariant to the last bit: MKL)
        assert torch.equal(l0, det.get_field("labels"))
        assert (b0 - det.bbox).abs().max() < 5e-3 and (s0 - det.get_field("scores")).abs().max() < 1e-5


def test_job_schedule_matches_reference_feed():
    """ClipEngine.jobs_for_step reproduces VIDMEGADataset._get_test (data/datasets/vid_mega.py:95-142) +
    the frame-0 fill of generalized_rcnn_mega.py:180-193."""
    _install_cpu_ops()
    from mega.pytorch_amd import engine

    class M(object):
        all_frame_interval, key_frame_location, key_num, base_num, global_enable = 25, 12, 300, 75, True
        cfg = type("C", (), {"INPUT": type("I", (), {"PIXEL_MEAN": (0, 0, 0), "TO_BGR255": True})})
    eng = engine.ClipEngine(M())
    gfor = engine.global_schedule(40, 10, seed=0)
    j0 = eng.jobs_for_step(0, 40, gfor)
    assert [j[0] for j in j0 if j[2] == "l"] == list(range(13)) and len([j for j in j0 if j[2] == "g"]) == 10
    assert all(j[1] == 300 for j in j0 if j[2] == "l") and all(j[1] == 75 for j in j0 if j[2] == "g")
    assert [j[0] for j in eng.jobs_for_step(5, 40, gfor) if j[2] == "l"] == [17]
    assert [j[0] for j in eng.jobs_for_step(35, 40, gfor) if j[2] == "l"] == [39]      # clamped to seg_len - 1
    short = eng.jobs_for_step(0, 5, engine.global_schedule(5, 10, seed=0))
    assert [j[0] for j in short if j[2] == "l"] == [0, 1, 2, 3, 4] + [4] * 8              # short video: tail repeats

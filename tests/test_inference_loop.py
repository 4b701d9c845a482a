"""Test-time loop + result gathering: index parsing, video-per-rank partition, predictions.pth in
the reference's on-disk format, and the world-size-2 gather over gloo."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from mega.pytorch_amd import config, engine, inference, modeling, synth  # noqa: E402
from oracle import pil_resize  # noqa: E402

VIDEOS = [("val/vid_a", 7), ("val/vid_b", 5)]
H0, W0, MIN_S, MAX_S = 60, 100, 96, 160
GSIZE = 3


def _make_dataset(root):
    """two tiny videos as <root>/Data/<video>/<%06d>.JPEG (lossless PNG payload: Pillow sniffs the content) plus the
    4-column index file the reference's VIDDataset reads (vid.py:55-66)."""
    from PIL import Image
    lines, clips, n = [], {}, 1
    for vi, (name, L) in enumerate(VIDEOS):
        os.makedirs(os.path.join(root, "Data", name))
        clip = synth.make_clip(L, H0, W0, seed=20 + vi).numpy()
        clips[name] = clip
        for t in range(L):
            Image.fromarray(clip[t]).save(os.path.join(root, "Data", name, "%06d.JPEG" % t), format="PNG")
            lines.append("%s %d %d %d" % (name, n, t, L))
            n += 1
    idx = os.path.join(root, "index.txt")
    with open(idx, "w") as f:
        f.write("\n".join(lines) + "\n")
    return os.path.join(root, "Data"), idx, clips


def _model():
    cfg = config.get_cfg("R-50")
    cfg.MODEL.DEVICE = "cpu"
    cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST = MIN_S, MAX_S
    # a short window keeps the cold start of each tiny video cheap (3 local + 3 global frames instead of 13 + 10)
    cfg.merge_from_list(["MODEL.VID.MEGA.ALL_FRAME_INTERVAL", 5, "MODEL.VID.MEGA.KEY_FRAME_LOCATION", 2,
                         "MODEL.VID.MEGA.MIN_OFFSET", -2, "MODEL.VID.MEGA.MAX_OFFSET", 2, "MODEL.VID.MEGA.GLOBAL.SIZE", GSIZE])
    m = modeling.build_detection_model(cfg)
    m.load_state_dict(synth.make_state_dict(blocks=(3, 4, 6), reduce_channel=True, global_res_stage=0, seed=5))
    return cfg, m


def _install_cpu_ops():
    sys.path.insert(0, HERE)
    import cpu_ops
    from mega.pytorch_amd import ops
    for name in cpu_ops.ALL:
        setattr(ops, name, getattr(cpu_ops, name))


def test_index_and_rank_partition(tmp_path):
    _, idx, _ = _make_dataset(str(tmp_path))
    index = inference.VIDTestIndex(idx)
    assert len(index) == 12 and [v["start"] for v in index.videos] == [0, 7]
    assert index.image_set_index[8] == "val/vid_b/000001" and index.videos[1]["pattern"] == "val/vid_b/%06d"
    parts = [inference.videos_for_rank(index.videos, r, 2) for r in range(2)]
    assert sorted(v["start"] for p in parts for v in p) == [0, 7] and all(len(p) == 1 for p in parts)
    bad = tmp_path / "bad.txt"
    bad.write_text("a 1 1 3\na 2 0 3\n")
    with pytest.raises(ValueError):
        inference.VIDTestIndex(str(bad))


def test_inference_writes_reference_format_predictions(monkeypatch, tmp_path):
    import cpu_ops
    cpu_ops.install(monkeypatch)
    torch.set_num_threads(8)
    img_dir, idx, clips = _make_dataset(str(tmp_path))
    cfg, model = _model()
    out = str(tmp_path / "out")
    preds = inference.inference(cfg, model, img_dir, idx, output_folder=out, steps_per_batch=3,
                                engine_kwargs={"overlap": False, "graphs": False, "reuse_records": False}, source_kwargs={"workers": 2})
    assert len(preds) == 12
    # same detections as the engine on the pre-resized resident clip of each video
    _, model2 = _model()
    start = 0
    for name, L in VIDEOS:
        clip = torch.from_numpy(np.stack([pil_resize.resize_bilinear_u8(f, MIN_S, MAX_S) for f in clips[name]]))
        # (same steps_per_batch as inference(): the CPU twins' GEMMs are not batch-invariant, the HIP kernels are --
        # tests/test_e2e_gpu.py::test_inference_loop_on_gpu_from_files compares different batchings bit for bit)
        ref = engine.ClipEngine(model2, steps_per_batch=3, overlap=False, graphs=False).run(
            clip, L, engine.global_schedule(L, GSIZE, seed=start))
        for i, r in enumerate(ref):
            p = preds[start + i]
            assert p.size == (MAX_S, MIN_S) and torch.equal(p.bbox, r.bbox)
            assert torch.equal(p.get_field("scores"), r.get_field("scores"))
            assert torch.equal(p.get_field("labels"), r.get_field("labels")) and p.get_field("labels").dtype == torch.int64
        start += L
    # round trip through the file
    back = inference.load_predictions(os.path.join(out, "predictions.pth"))
    assert len(back) == 12 and all(torch.equal(a.bbox, b.bbox) for a, b in zip(back, preds))
    # ... and the reference itself unpickles it as ITS BoxList (the reference cannot travel to the GPU box)
    import ref_shim
    if ref_shim.available():
        ref_shim.install()
        from mega_core.structures.bounding_box import BoxList as RefBoxList
        loaded = torch.load(os.path.join(out, "predictions.pth"), weights_only=False)
        assert all(type(b) is RefBoxList for b in loaded)
        assert loaded[3].mode == "xyxy" and loaded[3].size == (MAX_S, MIN_S)
        assert torch.equal(loaded[3].get_field("scores"), preds[3].get_field("scores"))
        assert len(loaded[3]) == len(preds[3]) and loaded[3].resize((W0, H0)).size == (W0, H0)


def _worker(rank, world, port, root, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _install_cpu_ops()
    cfg, model = _model()
    preds = inference.inference(cfg, model, os.path.join(root, "Data"), os.path.join(root, "index.txt"),
                                output_folder=outdir, steps_per_batch=3,
                                engine_kwargs={"overlap": False, "graphs": False, "reuse_records": False}, source_kwargs={"workers": 1})
    assert (preds is None) == (rank != 0)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_video_sharded_inference_world2(monkeypatch, tmp_path):
    """2 ranks x 1 video each, no data-path collective; rank 0 gathers and writes the same file a single process does."""
    root = str(tmp_path)
    _make_dataset(root)
    port = 31500 + os.getpid() % 2000
    out2 = os.path.join(root, "out2")
    mp.spawn(_worker, args=(2, port, root, out2), nprocs=2, join=True)
    import cpu_ops
    cpu_ops.install(monkeypatch)
    cfg, model = _model()
    single = inference.inference(cfg, model, os.path.join(root, "Data"), os.path.join(root, "index.txt"),
                                 steps_per_batch=3, engine_kwargs={"overlap": False, "graphs": False, "reuse_records": False})
    both = inference.load_predictions(os.path.join(out2, "predictions.pth"))
    assert len(both) == len(single) == 12
    for a, b in zip(both, single):
        assert torch.equal(a.bbox, b.bbox) and torch.equal(a.get_field("labels"), b.get_field("labels"))


def test_video_partition_equals_reference_sampler():
    """videos_for_rank == the frame ranges mega_core.data.samplers.VIDTestDistributedSampler hands out (executed from
    the reference tree when it is present: the build container), on random video-length lists and world sizes."""
    import ref_shim
    if not ref_shim.available():
        pytest.skip("needs /root/reference")
    ref_shim.install()
    from mega_core.data.samplers.distributed import VIDTestDistributedSampler

    class _DS(object):
        def __init__(self, lens):
            self.start_index, n = [], 0
            for l in lens:
                self.start_index.append(n)
                n += l
            self.n = n

        def __len__(self):
            return self.n
    rng = np.random.RandomState(0)
    for trial in range(40):
        lens = rng.randint(1, 60, size=rng.randint(1, 14)).tolist()
        ds = _DS(lens)
        videos = [{"start": s, "seg_len": l, "pattern": str(i)} for i, (s, l) in enumerate(zip(ds.start_index, lens))]
        for world in (2, 3, 4, 8):
            covered = []
            for rank in range(world):
                smp = VIDTestDistributedSampler(ds, num_replicas=world, rank=rank)
                if smp.start is None:
                    # the reference's find_zero fell off its list (offset inside the LAST video) and returned None;
                    # its slice [None:end] would then hand this rank the whole dataset again.  Here: nothing.
                    # (an end of None slices to the end of the dataset, which is what is meant: compared below)
                    assert inference.videos_for_rank(videos, rank, world) == []
                    continue
                want = list(iter(smp))
                mine = inference.videos_for_rank(videos, rank, world)
                got = [i for v in mine for i in range(v["start"], v["start"] + v["seg_len"])]
                assert got == want, (lens, world, rank)
                covered += got
            assert len(covered) == len(set(covered))


@pytest.mark.parametrize("method", ["fgfa", "base", "dff", "rdn", "rdn_engine"])
def test_inference_loop_drives_the_other_meta_architectures(monkeypatch, tmp_path, method):
    """engine/inference.py:17-47 for MODEL.VID.METHOD fgfa / base / dff: compute_on_dataset feeds every video through
    feed.FrameSource -> resident preprocessed frames -> FgfaClipEngine (fgfa) or the detector frame by frame on the
    reference's own test feed (inference.frame_feed = vid_fgfa.py / vid.py / vid_dff.py `_get_test`), and the predictions
    equal driving the model by hand on the pre-resized clip."""
    import cpu_ops
    cpu_ops.install(monkeypatch)
    torch.set_num_threads(8)
    import mega.pytorch_amd.fgfa  # noqa: F401
    import mega.pytorch_amd.rdn  # noqa: F401
    img_dir, idx, clips = _make_dataset(str(tmp_path))
    via_engine = method == "rdn_engine"            # RDN through ClipEngine (the default) instead of frame by frame
    method = "rdn" if via_engine else method
    cfg = config.get_cfg("R-50", method)
    cfg.MODEL.DEVICE = "cpu"
    cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST = MIN_S, MAX_S
    cfg.MODEL.RPN.POST_NMS_TOP_N_TEST = 30
    if method == "fgfa":
        cfg.MODEL.VID.FGFA.ALL_FRAME_INTERVAL, cfg.MODEL.VID.FGFA.KEY_FRAME_LOCATION = 5, 2
        cfg.MODEL.VID.FGFA.MIN_OFFSET, cfg.MODEL.VID.FGFA.MAX_OFFSET = -2, 2
    if method == "rdn":
        cfg.MODEL.VID.RDN.ALL_FRAME_INTERVAL, cfg.MODEL.VID.RDN.KEY_FRAME_LOCATION = 5, 2
        cfg.MODEL.VID.RDN.MIN_OFFSET, cfg.MODEL.VID.RDN.MAX_OFFSET = -2, 2
        cfg.MODEL.VID.RPN.REF_POST_NMS_TOP_N = 20
    sd = synth.make_dff_state_dict(seed=3) if method == "dff" else (
        synth.make_rdn_state_dict(advanced_stage=1, seed=3) if method == "rdn" else synth.make_fgfa_state_dict(seed=3))
    if method == "base":
        sd = {k: v for k, v in sd.items() if not k.startswith(("flownet.", "embednet."))}

    def build():
        m = modeling.build_detection_model(cfg)
        m.load_state_dict(sd)
        return m
    ek = {"graphs": False, "lookahead": 3} if method in ("fgfa", "dff") else ({"graphs": False, "group": 3} if method == "base" else
                                                                                {"per_frame": True})
    if via_engine:
        ek = {"overlap": False, "graphs": False}
    preds = inference.inference(cfg, build(), img_dir, idx, output_folder=str(tmp_path / "out"), engine_kwargs=ek,
                                source_kwargs={"workers": 2}, **({"steps_per_batch": 3} if via_engine else {}))
    assert len(preds) == 12 and all(p.size == (MAX_S, MIN_S) for p in preds)
    # by hand: the same detector in the reference's call convention on the pre-resized, preprocessed clip
    model = build()
    start = 0
    for name, L in VIDEOS:
        clip = torch.from_numpy(np.stack([pil_resize.resize_bilinear_u8(f, MIN_S, MAX_S) for f in clips[name]]))
        frames = synth.preprocess_cpu(clip)
        for i in range(L):
            with torch.no_grad():
                out = model(inference.frame_feed(cfg, frames, i))
            ref = out[0] if isinstance(out, (list, tuple)) else out
            p = preds[start + i]
            if method in ("fgfa", "dff", "base") or via_engine:      # (the engines batch the backbone / FlowNetS: MKL is not batch-invariant on
                assert abs(len(p) - len(ref)) <= 2, (name, i, len(p), len(ref))      # the CPU; the GPU test is bit-exact)
            else:
                assert torch.equal(p.bbox, ref.bbox) and torch.equal(p.get_field("scores"), ref.get_field("scores"))
                assert torch.equal(p.get_field("labels"), ref.get_field("labels"))
        start += L
    back = inference.load_predictions(os.path.join(str(tmp_path / "out"), "predictions.pth"))
    assert len(back) == 12 and all(torch.equal(a.bbox, b.bbox) for a, b in zip(back, preds))

"""Frame feed: Resize rule, Pillow-exact resampler, FrameSource through the engine."""
import os

import numpy as np
import pytest
import torch

from mega.pytorch_amd import config, engine, feed, modeling, synth
from oracle import pil_resize
import cpu_ops

SIZES = [(720, 1280), (360, 480), (1080, 1920), (600, 1000), (500, 375), (97, 211), (333, 1000)]


def test_resize_rule_matches_reference_table():
    """transforms.py:35-55 on the frame sizes ImageNet VID ships (values produced by the reference's Resize.get_size,
    tests/golden/make_golden.py::golden_feed)."""
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_feed.npz"))
    for (w, h), want in zip(d["sizes_wh"], d["get_size_hw"]):
        assert feed.get_size((int(w), int(h))) == tuple(int(v) for v in want)


@pytest.mark.parametrize("hw", SIZES)
def test_resampler_oracle_and_host_tables_equal_pillow(hw):
    """the numpy restatement (oracle/pil_resize.py) and the host coefficient tables the kernel consumes, against Pillow."""
    from PIL import Image
    H, W = hw
    a = np.random.RandomState(H + W).randint(0, 256, (H, W, 3)).astype(np.uint8)
    oh, ow = feed.get_size((W, H))
    ref = np.asarray(Image.fromarray(a).resize((ow, oh), Image.BILINEAR))
    assert np.array_equal(pil_resize.resize_bilinear_u8(a, oh, ow), ref)
    for insz, outsz in ((W, ow), (H, oh)):
        b, k, ks = feed.pil_bilinear_coeffs(insz, outsz)
        ob, ok, oks = pil_resize.coeffs(insz, outsz)
        assert ks == oks and [tuple(x) for x in b.tolist()] == ob
        for row, (lo, n), taps in zip(k, ob, ok):
            assert row[:n].tolist() == taps and not row[n:].any()


def test_totensor_times_255_is_the_identity_on_uint8():
    """the preprocess kernel relies on it: (x / 255) * 255 == x in IEEE f32 for all 256 byte values."""
    x = np.arange(256, dtype=np.float32)
    assert np.array_equal((x / np.float32(255)) * np.float32(255), x)
    t = torch.arange(256, dtype=torch.uint8).float()
    assert torch.equal(t.div(255) * 255, t)


def test_reference_transform_chain_fixture():
    """Resize + ToTensor + to_bgr255 + Normalize of the reference's own transforms (run in the build container on a
    seeded 360x480 frame, committed in ref_feed.npz) == oracle resize + synth.preprocess_cpu."""
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_feed.npz"))
    img = d["img"]
    oh, ow = feed.get_size((img.shape[1], img.shape[0]))
    r = pil_resize.resize_bilinear_u8(img, oh, ow)
    got = synth.preprocess_cpu(torch.from_numpy(r)[None])[0].numpy()
    assert got.shape == d["transformed"].shape
    assert np.array_equal(got, d["transformed"])


def test_frame_source_through_engine_equals_resident_clip(monkeypatch, tmp_path):
    """ClipEngine fed by a FrameSource (files on disk -> host decode -> resize twin) == the same engine on the
    pre-resized resident clip; every frame is decoded, none more than a handful of times."""
    from PIL import Image
    cpu_ops.install(monkeypatch)
    torch.set_num_threads(8)
    T, H0, W0 = 15, 60, 100          # 60x100 -> resize rule gives 600x1000: far too big for a CPU test, so use a
    clip0 = synth.make_clip(T, H0, W0, seed=9).numpy()   # small min/max size with the same aspect logic
    for t in range(T):
        Image.fromarray(clip0[t]).save(str(tmp_path / ("%06d.png" % t)))
    src = feed.FrameSource(str(tmp_path / "%s.png"), "%06d", T, "cpu", min_size=96, max_size=160, workers=2)
    assert src.out_hw == (96, 160) and src.shape == (T, 96, 160, 3)
    resident = torch.from_numpy(np.stack([pil_resize.resize_bilinear_u8(f, 96, 160) for f in clip0]))
    cfg = config.get_cfg("R-50")
    cfg.MODEL.DEVICE = "cpu"
    sd = synth.make_state_dict(blocks=(3, 4, 6), reduce_channel=True, global_res_stage=0, seed=5)
    outs = []
    for clip in (src, resident):
        model = modeling.build_detection_model(cfg)
        model.load_state_dict(sd)
        outs.append(engine.ClipEngine(model, steps_per_batch=2, overlap=False, graphs=False).run(clip, T, last=3))
    for a, b in zip(*outs):
        assert torch.equal(a.bbox, b.bbox) and torch.equal(a.get_field("scores"), b.get_field("scores"))
    assert src.decoded >= 15 - 1 and src.decoded <= 3 * T
    src.close()


def test_frame_source_errors(tmp_path):
    with pytest.raises(FileNotFoundError):
        feed.FrameSource(str(tmp_path / "%s.JPEG"), "%06d", 4, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("hw", SIZES)
def test_resize_kernel_equals_pillow(dev, hw):
    from PIL import Image
    from mega.pytorch_amd import ops
    H, W = hw
    a = np.random.RandomState(H * 7 + W).randint(0, 256, (3, H, W, 3)).astype(np.uint8)
    oh, ow = feed.get_size((W, H))
    ref = np.stack([np.asarray(Image.fromarray(f).resize((ow, oh), Image.BILINEAR)) for f in a])
    tables = feed.ResizeTables((H, W), (oh, ow), dev)
    got = ops.resize_bilinear_u8(torch.from_numpy(a).to(dev), (oh, ow), tables).cpu().numpy()
    assert np.array_equal(got, ref), "max |diff| %d" % np.abs(got.astype(int) - ref.astype(int)).max()


@pytest.mark.gpu
def test_frame_source_gpu_equals_pillow_chain(dev, tmp_path):
    """files -> FrameSource.fetch (decode, pinned H2D, resize kernel) -> preprocess kernel == Pillow resize +
    the reference's ToTensor/BGR/mean arithmetic."""
    from PIL import Image
    from mega.pytorch_amd import ops
    T, H0, W0 = 6, 180, 320
    clip0 = synth.make_clip(T, H0, W0, seed=3).numpy()
    for t in range(T):
        Image.fromarray(clip0[t]).save(str(tmp_path / ("%06d.png" % t)))
    src = feed.FrameSource(str(tmp_path / "%s.png"), "%06d", T, dev, workers=4)
    assert src.out_hw == (562, 999)
    ids = [4, 0, 5, 4]
    got = ops.preprocess_frames(src.fetch(ids), synth.PIXEL_MEAN, True).cpu()
    want = synth.preprocess_cpu(torch.from_numpy(np.stack(
        [np.asarray(Image.fromarray(clip0[i]).resize((999, 562), Image.BILINEAR)) for i in ids])))
    assert torch.equal(got, want)
    src.close()


@pytest.mark.gpu
def test_frame_source_read_ahead_is_the_same_batch(dev):
    """FrameSource.stage_ahead(ids): the batch brought over by the background thread on the copy stream is, bit for bit, the
    batch fetch(ids) delivers on its own -- for batches that go through the chunked staging ring (n >= 8) and small ones, with
    and without the resize -- a fetch of OTHER ids drops the read-ahead, and ClipEngine.run() walking a video block by block
    finds every block after the first already staged (and returns the detections of the resident clip)."""
    from mega.pytorch_amd import engine
    T, H0, W0 = 64, 96, 160
    host = synth.make_clip(16, H0, W0, seed=5).numpy()
    for size in ((96, 160), (120, 200)):
        src = feed.FrameSource(None, None, T, dev, min_size=size[0], max_size=size[1], opener=lambda f: host[f % 16], workers=4)
        for ids in ([3, 9, 1, 1, 15, 40, 22, 8, 63, 2], [5, 6]):
            want = src.fetch(ids).clone()
            src.stage_ahead(ids)
            got = src.fetch(ids)
            torch.cuda.synchronize()
            assert src.ahead_hits >= 1 and torch.equal(got, want)
        n0 = src.ahead_hits
        src.stage_ahead([1, 2, 3])
        other = src.fetch([4, 5, 6])              # not what was staged: dropped, fetched the ordinary way
        assert src.ahead_hits == n0 and torch.equal(other.cpu(), src.fetch([4, 5, 6]).cpu())
        src.close()
    cfg = config.get_cfg("R-50")
    cfg.DTYPE = "bfloat16"
    cfg.MODEL.DEVICE = str(dev)
    sd = synth.make_state_dict(blocks=(3, 4, 6), reduce_channel=True, global_res_stage=0, seed=5)
    src = feed.FrameSource(None, None, T, dev, min_size=H0, max_size=W0, opener=lambda f: host[f % 16], workers=4)
    clip = torch.from_numpy(np.stack([host[f % 16] for f in range(T)])).to(dev)
    gfor = engine.global_schedule(T, cfg.MODEL.VID.MEGA.GLOBAL.SIZE, seed=0)
    outs = {}
    for name, c in (("resident", clip), ("source", src)):
        model = modeling.build_detection_model(cfg)
        model.load_state_dict(sd)
        eng = engine.ClipEngine(model.to(dev), steps_per_batch=4)
        dets = eng.run(c, T, gfor, first=0, last=1)
        for first in (1, 5, 9, 13):                 # the video block by block: every block after the cold start is read ahead
            dets += eng.run(c, T, gfor, first=first, last=first + 4)
        outs[name] = dets
    assert src.ahead_hits >= 3, src.ahead_hits
    for a, b in zip(outs["resident"], outs["source"]):
        assert torch.equal(a.bbox, b.bbox) and torch.equal(a.get_field("scores"), b.get_field("scores"))
    src.close()

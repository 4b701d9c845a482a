"""Test-time loop and result gathering: the step AFTER the hot path.

Mirror of  mega_core/engine/inference.py:17-47   compute_on_dataset
           mega_core/engine/inference.py:50-69   _accumulate_predictions_from_multiple_gpus
           mega_core/engine/inference.py:72-134  inference  (writes <output_folder>/predictions.pth, :119)
           mega_core/data/datasets/vid.py:55-66  the 4-column VID index file  ("dir  global_id  seg_id  seg_len")
           mega_core/data/samplers/distributed.py VIDTestDistributedSampler (whole videos per rank)

What changes on MI355X: a video is not fed key frame by key frame through a DataLoader; each video becomes a
feed.FrameSource driven by ClipEngine (batched frame stage, two streams, hipGraph).  Detections stay on the device
until the video is finished, then move to the host in one go (the reference does a blocking .to(cpu) per frame).

What does NOT change: the result.  `predictions` is a list[BoxList] indexed by dataset index (line number of the
index file), boxes in the resized frame's coordinates with fields "scores" / "labels", and predictions.pth unpickles
inside the reference as mega_core.structures.bounding_box.BoxList objects, so tools/test_prediction.py and the VID
evaluation keep working on it.

Multi-GPU: videos are independent, so ranks take whole videos (the reference's sampler policy) with NO collective
on the data path; the only exchange is the final gather of the per-rank {index: BoxList} dicts.
"""
import contextlib
import logging
import os
import sys
import time
import types

import numpy as np
import torch

from . import engine as _engine
from . import feed
from .structures import BoxList

_REF_MODULE = "mega_core.structures.bounding_box"


class VIDTestIndex(object):
    """vid.py:55-66 for the test split: parses the index file into per-video records; dataset index = line number."""

    def __init__(self, img_index):
        with open(img_index) as f:
            lines = [x.strip().split(" ") for x in f.readlines() if x.strip()]
        if not lines or len(lines[0]) != 4:
            raise ValueError("expected the 4-column VID index format: '<video dir> <id> <frame_seg_id> <frame_seg_len>'")
        self.image_set_index = ["%s/%06d" % (x[0], int(x[2])) for x in lines]
        self.pattern = [x[0] + "/%06d" for x in lines]
        self.frame_seg_id = [int(x[2]) for x in lines]
        self.frame_seg_len = [int(x[3]) for x in lines]
        self.videos = []        # {"start": dataset index of frame 0, "pattern", "seg_len"}
        for idx, sid in enumerate(self.frame_seg_id):
            if sid == 0:
                self.videos.append({"start": idx, "pattern": self.pattern[idx], "seg_len": self.frame_seg_len[idx]})
        for v in self.videos:   # the reference's sampler and state machine both assume contiguous, complete videos
            ids = self.frame_seg_id[v["start"]:v["start"] + v["seg_len"]]
            if ids != list(range(v["seg_len"])):
                raise ValueError("video %s is not listed contiguously from frame 0" % v["pattern"])

    def __len__(self):
        return len(self.image_set_index)


def videos_for_rank(videos, rank, world, num_frames=None):
    """VIDTestDistributedSampler (data/samplers/distributed.py:83-95): the frame range of rank r starts at the first
    VIDEO START at or after r * ceil(N / world) and ends at the first video start at or after (r + 1) * ceil(N / world)
    (N = frames in the dataset) -- a video never straddles two ranks; a rank whose range holds no video start gets
    nothing.  (Where the reference's find_zero runs off the end of its list and returns None, the end of the dataset is
    meant and used here.)"""
    if world == 1:
        return list(videos)
    n = sum(v["seg_len"] for v in videos) if num_frames is None else num_frames
    per = -(-n // world)
    starts = [v["start"] for v in videos]

    def find_zero(offset):
        if offset >= n:
            return n
        for s in starts:
            if s >= offset:
                return s
        return n
    lo, hi = find_zero(rank * per), find_zero((rank + 1) * per)
    return [v for v in videos if lo <= v["start"] < hi]


def resident_video(src, cfg, chunk=32):
    """the whole video of a feed.FrameSource as preprocessed f32 frames [L,3,H,W] on the device (decode -> resize ->
    ToTensor / BGR255 / Normalize, data/transforms/build.py:6-37), fetched in chunks: what the per-key-frame detectors and
    FgfaClipEngine index by frame id (7.2 MB per 600 x 1000 frame)."""
    from . import ops
    mean, to_bgr = tuple(cfg.INPUT.PIXEL_MEAN), bool(cfg.INPUT.TO_BGR255)
    L = src.seg_len
    parts = []
    for o in range(0, L, chunk):
        ids = list(range(o, min(L, o + chunk)))
        parts.append(ops.preprocess_frames(src.fetch(ids).contiguous(), mean, to_bgr))
    return torch.cat(parts, dim=0) if len(parts) > 1 else parts[0]


def frame_feed(cfg, frames, idx):
    """What the reference's test datasets hand the detector for frame `idx` of a video (frames = resident_video):
      base  vid.py                the image
      dff   vid_dff.py:53-71      {"cur", "is_key_frame": frame_id % 10 == 0}
      rdn   vid_rdn.py:49-83      {"cur", "ref": [frame min(L-1, id + MAX_OFFSET)], "frame_category", "seg_len", ...}
      fgfa  vid_fgfa.py:49-83     the same with FGFA.MAX_OFFSET
    (for frame_category 0 the reference's detector reads frames 1 .. MAX_OFFSET itself through "img_dir" / "pattern" /
    "transforms"; here they travel as the extension "ref_init": already preprocessed, already on the device)."""
    method = cfg.MODEL.VID.METHOD
    L = frames.shape[0]
    if method == "base":
        return frames[idx]
    if method == "dff":
        return {"cur": frames[idx], "is_key_frame": idx % 10 == 0}
    if method in ("rdn", "fgfa"):
        mo = (cfg.MODEL.VID.RDN if method == "rdn" else cfg.MODEL.VID.FGFA).MAX_OFFSET
        d = {"cur": frames[idx], "ref": [frames[min(L - 1, idx + mo)]], "frame_category": 0 if idx == 0 else 1, "seg_len": L}
        if idx == 0:
            d["ref_init"] = [frames[i] for i in range(1, min(mo, L - 1) + 1)]
        return d
    raise ValueError("frame_feed: MODEL.VID.METHOD = %r (mega runs through ClipEngine)" % method)


def compute_on_dataset(model, index, img_dir, device, videos=None, steps_per_batch=10, seed=0, timer=None,
                       source_kwargs=None, engine_kwargs=None):
    """inference.py:17-47: -> {dataset index: BoxList on the host}, for every MODEL.VID.METHOD of the reference:
      mega / rdn   ClipEngine on the video's FrameSource (RDN is the MEGA detector without memory / global pools: rdn.py).  The
                   engine runs with reuse_records=True unless engine_kwargs says otherwise: every frame of a video goes
                   through the frame stage once and serves both its local-window and its global-pool role (bit-identical
                   detections on the GPU, ~half the backbone work); engine_kwargs={"per_frame": True}: RDN frame by frame;
      fgfa / dff / base   fgfa.FgfaClipEngine / DffClipEngine / BaseClipEngine on the resident video (engine_kwargs: lookahead,
                   graphs, pipeline, group / interval, lanes); engine_kwargs={"per_frame": True} runs the reference's call
                   convention instead;
      (per_frame)  the detector frame by frame on the reference's own test feed (frame_feed)."""
    model.eval()
    results = {}
    videos = index.videos if videos is None else videos
    method = model.cfg.MODEL.VID.METHOD
    if method == "rdn" and not (engine_kwargs or {}).get("per_frame"):
        method = "mega"
    if method != "mega":
        ek = dict(engine_kwargs or {})
        per_frame = bool(ek.pop("per_frame", False)) or method not in ("fgfa", "dff", "base")
        eng = None
        if not per_frame:
            from . import fgfa as _fgfa
            if method == "fgfa":
                eng = _fgfa.FgfaClipEngine(model, **{k: v for k, v in ek.items() if k in ("lookahead", "graphs", "pipeline", "group", "lanes")})
            elif method == "dff":
                eng = _fgfa.DffClipEngine(model, **{k: v for k, v in ek.items() if k in ("lookahead", "graphs", "pipeline", "interval", "lanes")})
            else:
                eng = _fgfa.BaseClipEngine(model, **{k: v for k, v in ek.items() if k in ("group", "graphs", "pipeline", "lanes")})
        for v in videos:
            src = feed.FrameSource(os.path.join(img_dir, "%s.JPEG"), v["pattern"], v["seg_len"], device,
                                   min_size=model.cfg.INPUT.MIN_SIZE_TEST, max_size=model.cfg.INPUT.MAX_SIZE_TEST,
                                   **(source_kwargs or {}))
            t0 = time.perf_counter()
            with torch.no_grad():
                frames = resident_video(src, model.cfg)
                if eng is not None:
                    dets = eng.run(frames, first=0, last=v["seg_len"])
                else:
                    dets = []
                    for i in range(v["seg_len"]):
                        out = model(frame_feed(model.cfg, frames, i))
                        dets.append(out[0] if isinstance(out, (list, tuple)) else out)
            if device.type == "cuda":
                torch.cuda.synchronize(device)
            if timer is not None:
                timer["inference_s"] = timer.get("inference_s", 0.0) + time.perf_counter() - t0
            for i, det in enumerate(dets):
                results[v["start"] + i] = det.to("cpu")
            src.close()
        return results
    ek = dict(reuse_records=model.cfg.MODEL.VID.METHOD == "mega")      # (RDN: a frame has one role, the local window)
    ek.update(engine_kwargs or {})
    ek.pop("per_frame", None)
    eng = _engine.ClipEngine(model, steps_per_batch=steps_per_batch, **ek)
    gsize = model.cfg.MODEL.VID.MEGA.GLOBAL.SIZE
    for vi, v in enumerate(videos):
        src = feed.FrameSource(os.path.join(img_dir, "%s.JPEG"), v["pattern"], v["seg_len"], device,
                               min_size=model.cfg.INPUT.MIN_SIZE_TEST, max_size=model.cfg.INPUT.MAX_SIZE_TEST,
                               **(source_kwargs or {}))
        t0 = time.perf_counter()
        # vid_mega.py:21-24 shuffles with numpy's global RNG; seeded per video here so runs are reproducible
        gfor = _engine.global_schedule(v["seg_len"], gsize, seed=seed + v["start"],
                                       shuffle=bool(model.cfg.MODEL.VID.MEGA.GLOBAL.SHUFFLE))
        dets = eng.run(src, v["seg_len"], gfor)
        if device.type == "cuda":
            torch.cuda.synchronize(device)
        if timer is not None:
            timer["inference_s"] = timer.get("inference_s", 0.0) + time.perf_counter() - t0
        for i, det in enumerate(dets):
            results[v["start"] + i] = det.to("cpu")
        src.close()
    return results


def accumulate_predictions(predictions_per_rank, group=None):
    """inference.py:50-69: gather the per-rank dicts, merge, order by dataset index (main process only)."""
    dist = torch.distributed
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        gathered = [None] * dist.get_world_size(group)
        dist.all_gather_object(gathered, predictions_per_rank, group=group)
        if dist.get_rank(group) != 0:
            return None
    else:
        gathered = [predictions_per_rank]
    predictions = {}
    for p in gathered:
        predictions.update(p)
    image_ids = list(sorted(predictions.keys()))
    if image_ids and len(image_ids) != image_ids[-1] + 1:
        logging.getLogger("mega.pytorch_amd.inference").warning(
            "Number of images that were gathered from multiple processes is not a contiguous set. "
            "Some images might be missing from the evaluation")
    return [predictions[i] for i in image_ids]


@contextlib.contextmanager
def _reference_boxlist_module(cls):
    """Make `mega_core.structures.bounding_box.BoxList` resolvable for the duration of a pickle dump / load when the
    reference package itself is not importable (it is a plain attribute container on disk: bbox, size, mode,
    extra_fields -- bounding_box.py:20-36)."""
    if _REF_MODULE in sys.modules:
        yield getattr(sys.modules[_REF_MODULE], "BoxList")
        return
    added = []
    parts = _REF_MODULE.split(".")
    for i in range(1, len(parts) + 1):
        name = ".".join(parts[:i])
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
            added.append(name)
    sys.modules[_REF_MODULE].BoxList = cls
    try:
        yield cls
    finally:
        for name in added:
            sys.modules.pop(name, None)


def save_predictions(predictions, path):
    """torch.save(list[BoxList]) whose pickled class is the reference's BoxList (inference.py:119)."""
    shadow = type("BoxList", (object,), {"__module__": _REF_MODULE})
    with _reference_boxlist_module(shadow) as ref_cls:
        out = []
        for p in predictions:
            o = ref_cls.__new__(ref_cls)
            o.__dict__.update({"bbox": p.bbox, "size": tuple(p.size), "mode": p.mode,
                               "extra_fields": dict(p.extra_fields)})
            out.append(o)
        torch.save(out, path)


def load_predictions(path):
    """Read a predictions.pth written by this package OR by the reference -> list of this package's BoxList."""
    shadow = type("BoxList", (object,), {"__module__": _REF_MODULE})
    with _reference_boxlist_module(shadow):
        raw = torch.load(path, map_location="cpu", weights_only=False)
    out = []
    for r in raw:
        b = BoxList(r.bbox, tuple(r.size), r.mode)
        for k, v in r.extra_fields.items():
            b.add_field(k, v)
        out.append(b)
    return out


def inference(cfg, model, img_dir, img_index, output_folder=None, device=None, group=None, **kw):
    """inference.py:72-134 up to (and including) predictions.pth; evaluation itself (datasets/evaluation) is the
    reference's and consumes the returned list / the file unchanged."""
    logger = logging.getLogger("mega.pytorch_amd.inference")
    device = torch.device(cfg.MODEL.DEVICE if device is None else device)
    dist = torch.distributed
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    index = VIDTestIndex(img_index)
    mine = videos_for_rank(index.videos, rank, world)
    timer = {}
    t0 = time.perf_counter()
    preds = compute_on_dataset(model, index, img_dir, device, videos=mine, timer=timer, **kw)
    if world > 1:
        dist.barrier(group=group)
    total = time.perf_counter() - t0
    logger.info("Total run time: %.1f s (%.4f s / img per device, on %d devices)", total,
                total * world / max(1, len(index)), world)
    logger.info("Model inference time: %.1f s", timer.get("inference_s", 0.0))
    predictions = accumulate_predictions(preds, group)
    if predictions is None:
        return None
    if output_folder:
        os.makedirs(output_folder, exist_ok=True)
        save_predictions(predictions, os.path.join(output_folder, "predictions.pth"))
    return predictions

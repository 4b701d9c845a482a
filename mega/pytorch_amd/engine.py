"""Clip-level inference engine: the MI355X-first way to drive GeneralizedRCNNMEGA.

The reference's loop (mega_core/engine/inference.py:17-47) feeds one key frame per call and the model
runs 2-3 single-frame backbone passes per call.  Here the frame-independent stage (backbone -> RPN ->
res5 -> ROIAlign -> fc0; ~90 % of the FLOPs) of many upcoming frames is run as ONE batch
(large GEMM M, full chip), optionally sharded over the GPUs of a node (each rank computes a slice of the
batch, one RCCL all-gather of the fixed-size frame records over xGMI), and only then the inherently
sequential per-key-frame aggregation (global/local/memory relation attention, predictor, NMS) is stepped.
Results are identical to calling ``model(images)`` frame by frame: every kernel is batch-invariant.

Frame schedule = the reference's test-time feed (mega_core/data/datasets/vid_mega.py:95-142):
key frame 0 consumes local frames 0..12 and GLOBAL.SIZE shuffled global frames, key frame t > 0 consumes
local frame min(T-1, t+12) and one more global frame.
"""
import numpy as np
import torch

from . import ops
from .structures import BoxList


def global_schedule(seg_len, global_size, seed=0):
    """Shuffled frame order of a video and the per-key-frame global frame ids (vid_mega.py:21-24,:112-120)."""
    rng = np.random.RandomState(seed)
    shuffled = np.arange(seg_len)
    rng.shuffle(shuffled)

    def for_frame(idx):
        size = global_size if idx == 0 else 1
        return [int(shuffled[(idx + global_size - i - 1) % seg_len]) for i in range(size)]
    return for_frame


class ClipEngine(object):
    def __init__(self, model, steps_per_batch=8, dist_group=None, overlap=True):
        """steps_per_batch: key-frame steps whose frame jobs are computed in one frame-stage batch
        (steady state: 2 frames per step).  dist_group: torch.distributed group to shard the frame stage over
        (None = single process).  overlap: run the frame stage of the next batch and the aggregation of the
        current one on two HIP streams (see run())."""
        self.model = model
        self.steps_per_batch = steps_per_batch
        self.group = dist_group
        if dist_group is not None:
            import torch.distributed as dist
            self.dist = dist
            self.rank, self.world = dist.get_rank(dist_group), dist.get_world_size(dist_group)
        else:
            self.dist, self.rank, self.world = None, 0, 1
        self.mean = tuple(model.cfg.INPUT.PIXEL_MEAN)
        self.to_bgr = bool(model.cfg.INPUT.TO_BGR255)
        self.overlap = overlap
        self._streams = None

    # ------------------------------------------------------------------ schedule
    def jobs_for_step(self, idx, T, gfor):
        """[(frame_id, want, role)] consumed by key frame idx, in consumption order."""
        m = self.model
        if idx == 0:
            loc = [0]
            end = 0
            for _ in range(m.all_frame_interval - m.key_frame_location - 1):
                end = min(end + 1, T - 1)
                loc.append(end)
        else:
            loc = [min(T - 1, idx + m.all_frame_interval - m.key_frame_location - 1)]
        jobs = [(f, m.key_num, "l") for f in loc]
        if m.global_enable:
            jobs += [(f, m.base_num, "g") for f in gfor(idx)]
        return jobs

    # ------------------------------------------------------------------ frame stage (optionally sharded)
    def _frames(self, clip, ids):
        """clip: uint8 [T,H,W,3] (device) -> preprocessed f32 [n,3,H,W]; or already-preprocessed f32 [T,3,H,W]."""
        idx = torch.as_tensor(ids, device=clip.device)
        sel = clip.index_select(0, idx)
        if clip.dtype == torch.uint8:
            return ops.preprocess_frames(sel.contiguous(), self.mean, self.to_bgr)
        return sel.contiguous()

    def compute_records(self, clip, jobs):
        """Run the frame stage for jobs [(frame_id, want, role)] -> list of records (same order)."""
        m = self.model
        if self.world == 1:
            return m.frame_stage(self._frames(clip, [j[0] for j in jobs]), [j[1] for j in jobs])
        # ---- sharded: contiguous slices of the (padded) job list per rank, fixed-size records, one all-gather
        n = len(jobs)
        per = (n + self.world - 1) // self.world
        padded = jobs + [jobs[-1]] * (per * self.world - n)
        mine = padded[self.rank * per:(self.rank + 1) * per]
        recs = m.frame_stage(self._frames(clip, [j[0] for j in mine]), [j[1] for j in mine])
        K, dev = m.key_num, clip.device
        fdt = recs[0]["feats"].dtype
        boxes = torch.zeros((per, K, 4), dtype=torch.float32, device=dev)
        scores = torch.zeros((per, K), dtype=torch.float32, device=dev)
        feats = torch.zeros((per, K, recs[0]["feats"].shape[1]), dtype=fdt, device=dev)
        cnt = torch.zeros((per,), dtype=torch.int32, device=dev)
        for i, r in enumerate(recs):
            k = r["boxes"].shape[0]
            boxes[i, :k], scores[i, :k], feats[i, :k], cnt[i] = r["boxes"], r["scores"], r["feats"], k
        g = {}
        for name, t in (("boxes", boxes), ("scores", scores), ("feats", feats), ("cnt", cnt)):
            out = torch.empty((self.world * per,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
            self.dist.all_gather_into_tensor(out, t.contiguous(), group=self.group)
            g[name] = out
        counts = g["cnt"].tolist()
        return [{"boxes": g["boxes"][i, :counts[i]], "scores": g["scores"][i, :counts[i]],
                 "feats": g["feats"][i, :counts[i]]} for i in range(n)]

    # ------------------------------------------------------------------ driver
    @torch.no_grad()
    def run(self, clip, T=None, gfor=None, first=0, last=None, on_step=None):
        """Process key frames first..last-1 of a T-frame video (first must be 0 for a fresh video, or the
        continuation of the previous call).  Returns the list of detections (BoxList, on device)."""
        m = self.model
        T = clip.shape[0] if T is None else T
        if gfor is None:
            gfor = global_schedule(T, m.cfg.MODEL.VID.MEGA.GLOBAL.SIZE)
        last = T if last is None else last
        H, W = (clip.shape[1], clip.shape[2]) if clip.dtype == torch.uint8 else (clip.shape[2], clip.shape[3])
        # ---- software pipeline over batches of steps, on two HIP streams:
        #        frame stage of batch b+1 (big MFMA kernels)  ||  aggregation steps of batch b (many small kernels)
        #      The aggregation of batch b only needs the frame records of batch b (an event on the frame stream);
        #      its small kernels fill the CUs the frame-stage tails leave idle.  Host order per iteration: enqueue
        #      aggregation(b) [async] -> enqueue frame stage(b+1) [blocks on the proposal counts] -> read the
        #      detection counts of batch b.  Results are identical to the sequential order.
        use_streams = clip.is_cuda and self.overlap
        if use_streams:
            if self._streams is None:
                self._streams = (torch.cuda.Stream(device=clip.device), torch.cuda.Stream(device=clip.device))
            sF, sB = self._streams
            cur = torch.cuda.current_stream(clip.device)
            sF.wait_stream(cur)
            sB.wait_stream(cur)
        batches = []
        idx = first
        while idx < last:
            hi = min(last, idx + (1 if idx == 0 else self.steps_per_batch))
            batches.append((idx, hi))
            idx = hi
        out = []
        pp = m.roi_heads.box.post_processor

        def frame_stage(b):
            per_step = [self.jobs_for_step(i, T, gfor) for i in range(b[0], b[1])]
            flat = [j for js in per_step for j in js]
            if use_streams:
                with torch.cuda.stream(sF):
                    recs = self.compute_records(clip, flat)
                    ev = torch.cuda.Event()
                    ev.record(sF)
                for r in recs:                       # produced on sF, consumed on sB
                    for t in r.values():
                        t.record_stream(sB)
            else:
                recs, ev = self.compute_records(clip, flat), None
            return per_step, recs, ev

        def aggregate(b, per_step, recs, ev):
            pending, o = [], 0
            for i, js in zip(range(b[0], b[1]), per_step):
                r = recs[o:o + len(js)]
                o += len(js)
                loc = [x for x, j in zip(r, js) if j[2] == "l"]
                glob = [x for x, j in zip(r, js) if j[2] == "g"]
                if i == 0:
                    m._reset(T)
                    for _ in range(m.key_frame_location + 1):
                        m.records.append(loc[0])
                    for x in loc[1:]:
                        m.records.append(x)
                    pending.append((i, m.step(None, glob, (W, H), defer=True)))
                else:
                    pending.append((i, m.step(loc[0], glob, (W, H), defer=True)))
            return pending

        def finish(pending):
            # one host sync per batch of steps: read all detection counts, then cut the padded outputs
            counts = torch.cat([pd[3] for _, pd in pending]).tolist()
            for (i, pd), n in zip(pending, counts):
                det = pp.materialize(pd, int(n), (W, H))
                out.append(det)
                if on_step is not None:
                    on_step(i, det)

        staged = frame_stage(batches[0]) if batches else None
        for bi, b in enumerate(batches):
            per_step, recs, ev = staged
            if use_streams:
                with torch.cuda.stream(sB):
                    sB.wait_event(ev)
                    pending = aggregate(b, per_step, recs, ev)
                staged = frame_stage(batches[bi + 1]) if bi + 1 < len(batches) else None
                with torch.cuda.stream(sB):
                    finish(pending)
            else:
                pending = aggregate(b, per_step, recs, ev)
                staged = frame_stage(batches[bi + 1]) if bi + 1 < len(batches) else None
                finish(pending)
        if use_streams:
            cur.wait_stream(sF)
            cur.wait_stream(sB)
        return out

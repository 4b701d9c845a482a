"""bench.py's launch contract on a CPU box (no kernels: `--dry-run` keeps only launch / rendezvous / reporting).

  * `python bench.py --gpus 2 --dry-run` with no launcher around it starts its own two ranks under torch.distributed.run
    (gloo here, nccl = RCCL when every rank has a device) and prints ONE line whose n_gpus comes from the live process group;
  * launched BY torch.distributed.run (the round driver's form) it must agree with WORLD_SIZE, else exit code 2;
  * `--gpus 2` on a box with fewer than 2 devices is refused (exit code 2) instead of silently reporting N = 1;
  * the host-side prefetch of a sharded run asks for exactly the frames records_async computes on that rank.
Reference: one process per GPU, tools/test_net.py:41,:69-75.
"""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    return env


def _run(cmd, env=None, timeout=600):
    pr = subprocess.run(cmd, env=env or _env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, cwd=ROOT)
    return pr.returncode, pr.stdout.decode(), pr.stderr.decode()


@pytest.mark.timeout(900)
def test_self_launch_reports_the_live_world_size():
    rc, out, err = _run([sys.executable, BENCH, "--gpus", "2", "--steps", "20", "--warmup", "5", "--dry-run"])
    assert rc == 0, err[-2000:]
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, out                      # exactly one JSON line on stdout, from rank 0
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["dry_run"] is True and line["steps"] == 20 and line["scaling"] == "weak"
    c = line["config"]
    # a step = 2 key frames; a block = 40; the default step-batch = 40 key frames = 80 frames, 40 per rank
    assert c["key_frames_per_step"] == 2 and c["key_frames_per_block"] == 40 and c["steps_per_batch"] == 40
    assert c["batch_sizes_in_a_block"] == [40]
    assert c["frames_per_batch"] == 80 and c["frames_per_rank_per_batch"] == 40 and c["backend"] == "gloo"


@pytest.mark.timeout(900)
def test_driver_form_and_world_size_mismatch():
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
            "127.0.0.1", "--master-port", str(29600 + os.getpid() % 300), BENCH]
    rc, out, err = _run(base + ["--gpus", "2", "--steps", "10", "--dry-run"])
    assert rc == 0, err[-2000:]
    line = json.loads([l for l in out.splitlines() if l.strip()][0])
    assert line["n_gpus"] == 2 and line["config"]["key_frames_per_block"] == 20
    rc, out, err = _run(base + ["--gpus", "4", "--steps", "10", "--dry-run"])       # WORLD_SIZE 2, --gpus 4
    assert rc != 0 and "must agree" in err and not out.strip()


def test_more_gpus_than_devices_is_refused():
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 devices")
    rc, out, err = _run([sys.executable, BENCH, "--gpus", "2", "--steps", "4"])
    assert rc == 2 and "refusing" in err and not out.strip()


def test_prefetch_slice_is_the_frame_stage_slice():
    """ClipEngine.run's prefetch and records_async use ONE dealing (shard_plan).  Owner-aligned form: local
    frame f is computed by rank f mod world -- the owner of KEY FRAME f (KeyFrameShard.owner) -- global frames round-robin;
    every rank's launch is nl locals then ng globals; every job has exactly one (rank, slot).  The legacy form (per-frame
    aggregation) keeps contiguous slices per row-count group."""
    sys.path.insert(0, ROOT)
    from mega.pytorch_amd import engine

    class M(object):
        all_frame_interval, key_frame_location, key_num, base_num, global_enable = 25, 12, 300, 75, True
        cfg = type("C", (), {"INPUT": type("I", (), {"PIXEL_MEAN": (0, 0, 0), "TO_BGR255": True})})
        roi_heads = type("R", (), {"box": type("B", (), {"post_processor": None, "feature_extractor": None})})
    T = 200
    gfor = engine.global_schedule(T, 10, seed=0)
    for world in (2, 3, 4, 8):
        for spb in (5, 10, 7, 16):
            eng = engine.ClipEngine(M(), steps_per_batch=spb)
            eng.world = world
            for first in (40, 0, T - 6):
                jobs = [j for i in range(first, min(T, first + spb)) for j in eng.jobs_for_step(i, T, gfor)]
                # ---- owner-aligned
                eng.owner_aligned = True
                plans = [eng.shard_plan(jobs, rank=r, world=world) for r in range(world)]
                plan = plans[0][0]
                nl, ng = plan["nl"], plan["ng"]
                assert all(p[0] == plan for p in plans)                           # the placement does not depend on the rank
                assert sorted(set(plan["place"])) == sorted(plan["place"])         # one (rank, slot) per job
                for pos, (r, slot) in enumerate(plan["place"]):
                    mine = plans[r][1]
                    assert len(mine) == nl + ng and mine[slot] == pos              # the owner's launch computes it, at that slot
                    if jobs[pos][2] == "l":
                        assert r == jobs[pos][0] % world and slot < nl             # frame f -> rank f mod world = owner of key frame f
                        ks = engine.KeyFrameShard(None, None, r, world, base=first)
                        if first <= jobs[pos][0] < first + spb:                    # (when key frame f is in this very batch)
                            assert ks.owner(jobs[pos][0] - first) == r
                    else:
                        assert slot >= nl
                for r in range(world):                                             # padding repeats a job of the same kind
                    mine = plans[r][1]
                    assert all(jobs[p][2] != "g" for p in mine[:nl]) and all(jobs[p][2] == "g" for p in mine[nl:])
                # the early count all-gather: rank r sends the counts of ITS launch (here: the job position itself, as a
                # stand-in); count_index must pick every job's own count out of the concatenation of the ranks' vectors
                flat = [x for r in range(world) for x in plans[r][1]]
                idx = eng.count_index(plan, len(jobs), nl + ng)
                assert [flat[i] for i in idx] == list(range(len(jobs)))
                # ---- legacy (whole records everywhere: per-frame aggregation)
                eng.owner_aligned = False
                covered = {}
                for rank in range(world):
                    plan_l, mine = eng.shard_plan(jobs, rank=rank, world=world)
                    assert len(mine) == sum(per for _, _, per, _ in plan_l)
                    for want, poss, per, first_slot in plan_l:
                        sl = mine[first_slot:first_slot + per]
                        assert all(int(jobs[p][1]) == want for p in sl)
                        for slot, p in enumerate(sl):
                            covered.setdefault(want, {})[rank * per + slot] = p
                for want, slots in covered.items():
                    poss = [p for p, j in enumerate(jobs) if int(j[1]) == want]
                    got = [slots[i] for i in range(len(slots))]
                    assert got[:len(poss)] == poss and all(p == poss[-1] for p in got[len(poss):])
                vecs = [list(eng.shard_plan(jobs, rank=r, world=world)[1]) for r in range(world)]
                nm = len(vecs[0])
                flat = [x for v in vecs for x in v]
                idx = eng.count_index(eng.shard_plan(jobs, rank=0, world=world)[0], len(jobs), nm)
                assert [flat[i] for i in idx] == list(range(len(jobs)))
    # and run()'s prefetch asks the source for exactly those frames
    asked = []

    class Src(object):
        is_cuda, dtype, shape = False, None, (T, 8, 8, 3)

        def prefetch(self, ids):
            asked.append(list(ids))

        def fetch(self, ids):
            raise StopIteration          # the test stops at the first frame-stage launch

    eng = engine.ClipEngine(M(), steps_per_batch=10)
    eng.world, eng.rank = 4, 1
    eng.owner_aligned = True
    import torch
    Src.dtype = torch.uint8
    with pytest.raises(StopIteration):
        eng.run(Src(), T, gfor, first=41, last=61)
    jobs = [j for i in range(41, 51) for j in eng.jobs_for_step(i, T, gfor)]
    assert asked[0] == [jobs[p][0] for p in eng.shard_plan(jobs, rank=1, world=4)[1]]

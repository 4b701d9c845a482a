"""Generates tests/golden/*.npz FROM THE REFERENCE ITSELF (run in the build container, where
/root/reference exists; the fixtures are committed because the reference cannot travel to the GPU box).

  python tests/golden/make_golden.py

1. ref_tests_nms.npz / ref_tests_box_coder.npz -- the known-answer vectors held by the reference's own tests
   (tests/test_nms.py:11-217, tests/test_box_coder.py:11-105).  They are extracted by EXECUTING those test
   modules under oracle/ref_shim.py with the op under test and numpy's assert functions wrapped by recorders, so
   no literal is transcribed by hand; the reference's ops must pass their own asserts while recording.
2. ref_ops.npz  -- seeded inputs/outputs of the reference's native ops (mega_core._C.nms / roi_align_forward,
   i.e. csrc/cpu/*.cpp compiled by oracle/build_ref.py) and python pieces (anchors, BoxCoder.decode,
   position embedding, attention_module_multi_head, RPNPostProcessor, PostProcessor).
3. ref_e2e_r50.npz -- GeneralizedRCNNMEGA (R-50 MEGA config, calibrated synthetic weights from
   mega.pytorch_amd.synth, seeded synthetic 160x256 clip) run for a few key frames on CPU: final detections and
   the intermediates named in SURVEY.md 8a.
5. ref_base_r50.npz / ref_rdn_r50.npz / ref_feed.npz -- see golden_base / golden_rdn / golden_feed.
4. ref_fgfa_r50.npz -- GeneralizedRCNNFGFA (configs/FGFA/vid_R_50_C4_FGFA_1x.yaml) on a 128x192 clip: flow field,
   aggregated feature map, predictor logits and final detections per key frame.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_shim  # noqa: E402
from mega.pytorch_amd import synth  # noqa: E402

E2E = dict(H=160, W=256, T=30, nkey=4, seed_w=1, seed_clip=3, global_seed=0)
# long clip: 44 key frames -> the 25-entry memory deques of all three stages and the 10-entry global deque WRAP
# (roi_box_feature_extractors.py:657-688, read-before-push :914-917): the regime bench.py times
E2E_LONG = dict(H=160, W=256, T=64, nkey=44, seed_w=1, seed_clip=6, global_seed=2)
FGFA = dict(H=128, W=192, T=14, nkey=3, seed_w=2, seed_clip=4)
RDN = dict(H=128, W=192, T=24, nkey=3, seed_w=2, seed_clip=4)


def _load_ref_test(name):
    spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(ref_shim.REF_ROOT, "tests", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def golden_from_reference_tests():
    ref_shim.install()
    # ---- tests/test_nms.py
    mod = _load_ref_test("test_nms")
    calls, expects = [], []
    real_nms = mod.box_nms

    def rec_nms(boxes, scores, thresh):
        keep = real_nms(boxes, scores, thresh)
        calls.append((boxes.numpy().copy(), scores.numpy().copy(), float(thresh)))
        return keep
    mod.box_nms = rec_nms
    real_assert = np.testing.assert_array_equal

    def rec_assert(a, b, *args, **kw):
        real_assert(a, b, *args, **kw)          # the reference op must satisfy the reference's own assert
        expects.append(np.asarray(b).copy())
    np.testing.assert_array_equal = rec_assert
    try:
        t = mod.TestNMS()
        t.test_nms_cpu()
        t.test_nms1_cpu()
    finally:
        np.testing.assert_array_equal = real_assert
    assert len(calls) == len(expects) == 6
    out = {}
    for i, ((b, s, thr), e) in enumerate(zip(calls, expects)):
        out["boxes%d" % i], out["scores%d" % i], out["thr%d" % i], out["keep%d" % i] = b, s, np.float32(thr), np.sort(e)
    out["n"] = np.int64(len(calls))
    np.savez(os.path.join(HERE, "ref_tests_nms.npz"), **out)
    # ---- tests/test_box_coder.py
    mod = _load_ref_test("test_box_coder")
    rec = {}
    real_decode = mod.BoxCoder.decode

    def rec_decode(self, rel_codes, boxes):
        r = real_decode(self, rel_codes, boxes)
        rec.update(weights=np.array(self.weights, dtype=np.float32), rel_codes=rel_codes.numpy().copy(),
                   boxes=boxes.numpy().copy(), out=r.numpy().copy())
        return r
    mod.BoxCoder.decode = rec_decode
    real_allclose = np.testing.assert_allclose

    def rec_allclose(a, b, *args, **kw):
        real_allclose(a, b, *args, **kw)
        rec["expected"] = np.asarray(b).copy()
        rec["atol"] = np.float32(kw.get("atol", 0))
    np.testing.assert_allclose = rec_allclose
    try:
        mod.TestBoxCoder().test_box_decoder()
    finally:
        np.testing.assert_allclose = real_allclose
        mod.BoxCoder.decode = real_decode
    np.savez(os.path.join(HERE, "ref_tests_box_coder.npz"), **rec)
    print("reference test vectors: %d nms cases, box_coder %s" % (len(calls), rec["out"].shape))


def golden_ops():
    ref_shim.install()
    from mega_core import _C
    from mega_core.modeling.box_coder import BoxCoder
    from mega_core.modeling.rpn.anchor_generator import generate_anchors, AnchorGenerator
    from mega_core.modeling.rpn.inference import RPNPostProcessor
    from mega_core.modeling.roi_heads.box_head.inference import PostProcessor
    from mega_core.modeling.roi_heads.box_head.roi_box_feature_extractors import AttentionExtractor
    from mega_core.structures.bounding_box import BoxList
    from mega_core.structures.image_list import ImageList
    g = torch.Generator().manual_seed(123)
    out = {}
    # native ops
    n = 400
    ctr = torch.rand((n, 2), generator=g) * 300
    wh = torch.rand((n, 2), generator=g) * 90 + 4
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], dim=1)
    scores = torch.rand((n,), generator=g)
    out["nms_boxes"], out["nms_scores"] = boxes.numpy(), scores.numpy()
    for thr in (0.3, 0.5, 0.7):
        out["nms_keep_%d" % int(thr * 10)] = _C.nms(boxes, scores, thr).numpy()
    feat = torch.randn((2, 24, 20, 30), generator=g)
    rois = torch.tensor([[0, 0, 0, 479, 319], [1, 35.2, 17.9, 200.4, 180.1], [0, 100, 50, 104, 53], [1, 300, 200, 470, 310],
                         [0, -20, -30, 50, 60], [1, 450, 300, 520, 400]], dtype=torch.float32)
    out["ra_feat"], out["ra_rois"] = feat.numpy(), rois.numpy()
    out["ra_out_sr0"] = _C.roi_align_forward(feat, rois, 1 / 16., 7, 7, 0).numpy()
    out["ra_out_sr2"] = _C.roi_align_forward(feat, rois, 1 / 16., 7, 7, 2).numpy()
    # anchors
    out["cell_anchors"] = generate_anchors(16, (64, 128, 256, 512), (0.5, 1.0, 2.0)).float().numpy()
    ag = AnchorGenerator((64, 128, 256, 512), (0.5, 1.0, 2.0), (16,), 0)
    out["grid_anchors_5x7"] = ag.grid_anchors([(5, 7)])[0].numpy()
    # box coder
    rel = torch.randn((50, 8), generator=g)
    rel[0, 2] = 9.0
    bx = boxes[:50]
    out["bc_rel"], out["bc_boxes"] = rel.numpy(), bx.numpy()
    out["bc_out_1111"] = BoxCoder((1., 1., 1., 1.)).decode(rel, bx).numpy()
    out["bc_out_10_5"] = BoxCoder((10., 10., 5., 5.)).decode(rel, bx).numpy()
    # position embedding + attention
    sd = synth.make_state_dict(blocks=(1, 1, 1), seed=4)
    bq, bk = boxes[:23], boxes[30:71]
    pe = AttentionExtractor.extract_position_embedding(AttentionExtractor.extract_position_matrix(bq, bk), 64)
    out["pe_bq"], out["pe_bk"], out["pe_out"] = bq.numpy(), bk.numpy(), pe.numpy()
    cfg = ref_shim.make_cfg("configs/MEGA/vid_R_101_C4_MEGA_1x.yaml")
    model = ref_shim.build_model(cfg)
    fe = model.roi_heads.box.feature_extractor
    fe_sd = {k[len(synth.FE):]: v for k, v in synth.make_state_dict(seed=4).items() if k.startswith(synth.FE)}
    fe.load_state_dict(fe_sd)
    x = torch.randn((23, 1024), generator=g)
    r = torch.randn((41, 1024), generator=g)
    with torch.no_grad():
        pe4 = fe.cal_position_embedding(bq, bk)
        out["att_x"], out["att_ref"] = x.numpy(), r.numpy()
        out["att_local1"] = fe.attention_module_multi_head(x, r, pe4, index=1, ver="local").numpy()
        out["att_global0"] = fe.attention_module_multi_head(x, r, None, index=0, ver="global").numpy()
    out["att_seed"] = np.int64(4)
    # RPN post-processor
    A, Hf, Wf = 12, 9, 13
    obj = torch.randn((1, A, Hf, Wf), generator=g) * 2
    reg = torch.randn((1, 4 * A, Hf, Wf), generator=g) * 0.5
    im_w, im_h = Wf * 16 - 8, Hf * 16 - 8
    anchors = [[BoxList(ag.grid_anchors([(Hf, Wf)])[0], (im_w, im_h), mode="xyxy")]]
    pp = RPNPostProcessor(pre_nms_top_n=600, post_nms_top_n=50, nms_thresh=0.7, min_size=0)
    pp.eval()
    res = pp(anchors, [obj], [reg])[0]
    out["rpn_obj"], out["rpn_reg"], out["rpn_imwh"] = obj.numpy(), reg.numpy(), np.array([im_w, im_h])
    out["rpn_boxes"], out["rpn_scores"] = res.bbox.numpy(), res.get_field("objectness").numpy()
    # box-head post-processor
    R = 60
    logits = torch.randn((R, 31), generator=g) * 1.5
    deltas = torch.randn((R, 124), generator=g) * 0.5
    props = BoxList(boxes[:R].clamp(min=0, max=250), (300, 260), mode="xyxy")
    post = PostProcessor(0.001, 0.5, 300, BoxCoder((10., 10., 5., 5.)))
    det = post((logits, deltas), [props])[0]
    out["post_logits"], out["post_deltas"], out["post_props"] = logits.numpy(), deltas.numpy(), props.bbox.numpy()
    out["post_boxes"], out["post_scores"] = det.bbox.numpy(), det.get_field("scores").numpy()
    out["post_labels"] = det.get_field("labels").numpy()
    # FGFA warp + aggregation (detector/generalized_rcnn_fgfa.py:45-76,:201-211), methods of the reference class
    from mega_core.modeling.detector.generalized_rcnn_fgfa import GeneralizedRCNNFGFA
    import torch.nn.functional as Fn
    fg = object.__new__(GeneralizedRCNNFGFA)     # the three methods used below touch no instance state
    T, Cf, Ce, Hh, Ww = 5, 16, 24, 9, 13
    feats_all = torch.randn((T, Cf + Ce, Hh, Ww), generator=g)
    flow = torch.randn((T, 2, Hh, Ww), generator=g) * 2.5
    flow[2] = 0
    warped = fg.resample(feats_all, flow)
    wf, emb = torch.split(warped, (Cf, Ce), dim=1)
    wts = Fn.softmax(fg.compute_weight(emb.contiguous(), emb[2:3]), dim=0)
    out["fgfa_feats"], out["fgfa_flow"], out["fgfa_key"] = feats_all.numpy(), flow.numpy(), np.int64(2)
    out["fgfa_out"], out["fgfa_weights"] = torch.sum(wts * wf, dim=0, keepdim=True).numpy(), wts.numpy()
    np.savez_compressed(os.path.join(HERE, "ref_ops.npz"), **out)
    print("reference op fixtures written: %d arrays" % len(out))


def golden_e2e():
    c = E2E
    cfg = ref_shim.make_cfg("configs/MEGA/vid_R_50_C4_MEGA_1x.yaml")
    model = ref_shim.build_model(cfg)
    sd = synth.make_state_dict(blocks=(3, 4, 6), reduce_channel=True, stage=3, global_res_stage=0, seed=c["seed_w"])
    model.load_state_dict(sd, strict=True)
    frames = synth.preprocess_cpu(synth.make_clip(c["T"], c["H"], c["W"], seed=c["seed_clip"]))
    rng = np.random.RandomState(c["global_seed"])
    shuffled = np.arange(c["T"]); rng.shuffle(shuffled)
    gs = cfg.MODEL.VID.MEGA.GLOBAL.SIZE

    import mega_core.modeling.detector.generalized_rcnn_mega as gm

    class _FakeImg(object):
        def __init__(self, i): self.i = i
        def convert(self, m): return self

    class _FakeImage(object):
        @staticmethod
        def open(path): return _FakeImg(int(path))
    gm.Image = _FakeImage
    # record intermediates with forward hooks
    trace = {}
    fe = model.roi_heads.box.feature_extractor
    pred = model.roi_heads.box.predictor
    pred.register_forward_hook(lambda m, i, o: trace.update(x=i[0].detach().clone(), logits=o[0].detach().clone(),
                                                            deltas=o[1].detach().clone()))
    out = {}
    for idx in range(c["nkey"]):
        gl = [int(shuffled[(idx + gs - i - 1) % c["T"]]) for i in range(gs if idx == 0 else 1)]
        images = {"cur": frames[idx], "ref_l": [frames[min(c["T"] - 1, idx + 12)]], "ref_g": [frames[g] for g in gl],
                  "frame_category": 0 if idx == 0 else 1, "seg_len": c["T"], "pattern": "%d", "img_dir": "%s",
                  "transforms": lambda im: frames[im.i]}
        with torch.no_grad():
            det = model(images)[0]
        out["boxes%d" % idx] = det.bbox.numpy()
        out["scores%d" % idx] = det.get_field("scores").numpy()
        out["labels%d" % idx] = det.get_field("labels").numpy()
        out["x%d" % idx] = trace["x"].numpy()[:48]          # first rows only: keeps the fixture small
        out["logits%d" % idx] = trace["logits"].numpy()
        out["deltas%d" % idx] = trace["deltas"].numpy()
        print("frame", idx, "dets", det.bbox.shape[0])
    for k, v in c.items():
        out["cfg_" + k] = np.int64(v)
    np.savez_compressed(os.path.join(HERE, "ref_e2e_r50.npz"), **out)


def golden_e2e_long():
    """ref_e2e_long_r50.npz: the unmodified reference stepped through 44 key frames of a 64-frame clip (R-50 MEGA
    config, 25 local / 10 global / 25 memory): every key frame's detections, predictor logits and the first rows of
    the box-head output; plus the state sizes the reference itself holds at each step (memory / global deque lengths)
    so that a test can assert that the fixture really covers eviction."""
    c = E2E_LONG
    cfg = ref_shim.make_cfg("configs/MEGA/vid_R_50_C4_MEGA_1x.yaml")
    model = ref_shim.build_model(cfg)
    sd = synth.make_state_dict(blocks=(3, 4, 6), reduce_channel=True, stage=3, global_res_stage=0, seed=c["seed_w"])
    model.load_state_dict(sd, strict=True)
    frames = synth.preprocess_cpu(synth.make_clip(c["T"], c["H"], c["W"], seed=c["seed_clip"]))
    rng = np.random.RandomState(c["global_seed"])
    shuffled = np.arange(c["T"]); rng.shuffle(shuffled)
    gs = cfg.MODEL.VID.MEGA.GLOBAL.SIZE
    import mega_core.modeling.detector.generalized_rcnn_mega as gm

    class _FakeImg(object):
        def __init__(self, i): self.i = i
        def convert(self, m): return self

    class _FakeImage(object):
        @staticmethod
        def open(path): return _FakeImg(int(path))
    gm.Image = _FakeImage
    trace = {}
    fe = model.roi_heads.box.feature_extractor
    model.roi_heads.box.predictor.register_forward_hook(
        lambda m, i, o: trace.update(x=i[0].detach().clone(), logits=o[0].detach().clone()))
    out = {"mem_len": [], "glob_len": [], "nprop": [], "mem_rows": []}
    for idx in range(c["nkey"]):
        gl = [int(shuffled[(idx + gs - i - 1) % c["T"]]) for i in range(gs if idx == 0 else 1)]
        images = {"cur": frames[idx], "ref_l": [frames[min(c["T"] - 1, idx + 12)]], "ref_g": [frames[g] for g in gl],
                  "frame_category": 0 if idx == 0 else 1, "seg_len": c["T"], "pattern": "%d", "img_dir": "%s",
                  "transforms": lambda im: frames[im.i]}
        with torch.no_grad():
            det = model(images)[0]
        out["boxes%d" % idx] = det.bbox.numpy()
        out["scores%d" % idx] = det.get_field("scores").numpy()
        out["labels%d" % idx] = det.get_field("labels").numpy()
        out["x%d" % idx] = trace["x"].numpy()[:16]
        out["logits%d" % idx] = trace["logits"].numpy()
        out["mem_len"].append([len(q["rois"]) for q in fe.mem_queue_list])
        out["mem_rows"].append([int(fe.mem[i]["rois"].shape[0]) for i in range(len(fe.mem))])
        out["glob_len"].append(len(fe.global_queue_list[0]["feats"]))
        out["nprop"].append(trace["logits"].shape[0])
        print("long frame", idx, "dets", det.bbox.shape[0], "props", trace["logits"].shape[0], "mem", out["mem_len"][-1])
    for k in ("mem_len", "glob_len", "nprop", "mem_rows"):
        out[k] = np.asarray(out[k], dtype=np.int64)
    for k, v in c.items():
        out["cfg_" + k] = np.int64(v)
    np.savez_compressed(os.path.join(HERE, "ref_e2e_long_r50.npz"), **out)


def golden_base():
    """ref_base_r50.npz: single-frame GeneralizedRCNN (configs/vid_R_50_C4_1x.yaml, BASELINE config 1)."""
    c = FGFA
    cfg = ref_shim.make_cfg("configs/vid_R_50_C4_1x.yaml")
    model = ref_shim.build_model(cfg)
    sd = {k: v for k, v in synth.make_fgfa_state_dict(seed=c["seed_w"]).items()
          if not k.startswith(("flownet.", "embednet."))}
    model.load_state_dict(sd, strict=True)
    frames = synth.preprocess_cpu(synth.make_clip(c["T"], c["H"], c["W"], seed=c["seed_clip"]))
    trace = {}
    model.roi_heads.box.predictor.register_forward_hook(
        lambda m, i, o: trace.update(logits=o[0].detach().clone(), deltas=o[1].detach().clone()))
    out = {}
    for idx in range(2):
        with torch.no_grad():
            det = model(frames[idx])[0]
        out["boxes%d" % idx] = det.bbox.numpy()
        out["scores%d" % idx] = det.get_field("scores").numpy()
        out["labels%d" % idx] = det.get_field("labels").numpy()
        out["logits%d" % idx] = trace["logits"].numpy()
        out["deltas%d" % idx] = trace["deltas"].numpy()
    for k, v in c.items():
        out["cfg_" + k] = np.int64(v)
    np.savez_compressed(os.path.join(HERE, "ref_base_r50.npz"), **out)


def golden_rdn():
    """ref_rdn_r50.npz: GeneralizedRCNNRDN (configs/RDN/vid_R_50_C4_RDN_base_1x.yaml + ADVANCED_STAGE 1, i.e. the
    full RDN head of vid_R_101_C4_RDN_1x.yaml on the R-50 body), 128x192 clip, 3 key frames."""
    c = RDN
    cfg = ref_shim.make_cfg("configs/RDN/vid_R_50_C4_RDN_base_1x.yaml",
                            ["MODEL.VID.ROI_BOX_HEAD.ATTENTION.ADVANCED_STAGE", 1])
    model = ref_shim.build_model(cfg)
    model.load_state_dict(synth.make_rdn_state_dict(advanced_stage=1, seed=c["seed_w"]), strict=True)
    frames = synth.preprocess_cpu(synth.make_clip(c["T"], c["H"], c["W"], seed=c["seed_clip"]))
    import mega_core.modeling.detector.generalized_rcnn_rdn as gm
    from mega_core.structures.image_list import to_image_list

    class _FakeImg(object):
        def __init__(self, i): self.i = i
        def convert(self, m): return self

    class _FakeImage(object):
        @staticmethod
        def open(path): return _FakeImg(int(path))
    gm.Image = _FakeImage
    trace = {}
    model.roi_heads.box.predictor.register_forward_hook(
        lambda m, i, o: trace.update(x=i[0].detach().clone(), logits=o[0].detach().clone(), deltas=o[1].detach().clone()))
    out = {}
    for idx in range(c["nkey"]):
        nxt = min(c["T"] - 1, idx + 18)
        images = {"cur": to_image_list(frames[idx]), "ref": [to_image_list(frames[nxt])],
                  "frame_category": 0 if idx == 0 else 1, "seg_len": c["T"], "pattern": "%d", "img_dir": "%s",
                  "transforms": lambda im: frames[im.i]}
        with torch.no_grad():
            det = model(images)[0]
        out["boxes%d" % idx] = det.bbox.numpy()
        out["scores%d" % idx] = det.get_field("scores").numpy()
        out["labels%d" % idx] = det.get_field("labels").numpy()
        out["x%d" % idx] = trace["x"].numpy()[:48]
        out["logits%d" % idx] = trace["logits"].numpy()
        out["deltas%d" % idx] = trace["deltas"].numpy()
        print("rdn frame", idx, "dets", det.bbox.shape[0])
    for k, v in c.items():
        out["cfg_" + k] = np.int64(v)
    np.savez_compressed(os.path.join(HERE, "ref_rdn_r50.npz"), **out)


def golden_dff():
    """ref_dff_r50.npz: GeneralizedRCNNDFF (configs/DFF/vid_R_50_C4_DFF_1x.yaml), key frame every 3rd frame."""
    c = FGFA
    cfg = ref_shim.make_cfg("configs/DFF/vid_R_50_C4_DFF_1x.yaml")
    model = ref_shim.build_model(cfg)
    model.load_state_dict(synth.make_dff_state_dict(seed=c["seed_w"]), strict=True)
    frames = synth.preprocess_cpu(synth.make_clip(c["T"], c["H"], c["W"], seed=c["seed_clip"]))
    trace = {}
    model.roi_heads.box.predictor.register_forward_hook(
        lambda m, i, o: trace.update(logits=o[0].detach().clone(), deltas=o[1].detach().clone()))
    model.rpn.register_forward_hook(lambda m, i, o: trace.update(feats=i[1][0].detach().clone()))
    out = {}
    for idx in range(4):
        with torch.no_grad():
            det = model({"cur": frames[idx], "is_key_frame": idx % 3 == 0})[0]
        out["boxes%d" % idx] = det.bbox.numpy()
        out["scores%d" % idx] = det.get_field("scores").numpy()
        out["labels%d" % idx] = det.get_field("labels").numpy()
        out["logits%d" % idx] = trace["logits"].numpy()
        out["feats%d" % idx] = trace["feats"].numpy()[0, ::16]
    for k, v in c.items():
        out["cfg_" + k] = np.int64(v)
    np.savez_compressed(os.path.join(HERE, "ref_dff_r50.npz"), **out)


def golden_checkpoint():
    """ref_checkpoint.json: (1) the reference's Caffe2 -> torch renaming (c2_model_loading._rename_weights_for_resnet)
    of a Detectron-style R-50 blob list, (2) the key alignment its load_state_dict computes when those weights (and,
    separately, a 'module.'-prefixed full checkpoint, and a FlowNet file) are loaded into its own MEGA / FGFA models."""
    import json
    cfg = ref_shim.make_cfg("configs/MEGA/vid_R_50_C4_MEGA_1x.yaml")
    from mega_core.utils import c2_model_loading as c2
    from mega_core.utils import model_serialization as ms
    blobs = ["conv1_w", "res_conv1_bn_s", "res_conv1_bn_b", "fc1000_w", "fc1000_b", "pred_w", "conv1_w_momentum",
             "conv_rpn_w", "conv_rpn_b", "rpn_cls_logits_w", "rpn_cls_logits_b", "rpn_bbox_pred_w", "rpn_bbox_pred_b",
             "cls_score_w", "cls_score_b", "bbox_pred_w", "bbox_pred_b"]
    for stage, nblk in (("res2", 3), ("res3", 4), ("res4", 6), ("res5", 3)):
        for b in range(nblk):
            for br in ("branch2a", "branch2b", "branch2c") + (("branch1",) if b == 0 else ()):
                blobs += ["%s_%d_%s_w" % (stage, b, br), "%s_%d_%s_bn_s" % (stage, b, br), "%s_%d_%s_bn_b" % (stage, b, br)]
    weights = {k: np.zeros((1,), np.float32) for k in blobs}
    renamed = c2._rename_weights_for_resnet(weights, c2._C2_STAGE_NAMES["R-50"])
    out = {"c2_blobs": sorted(blobs), "c2_renamed": list(renamed.keys())}

    def alignment(model, loaded_keys, flownet):
        class _Named(object):                                    # stands in for a tensor: the reference logs .shape
            shape = ()
            def __init__(self, name): self.name = name
        msd = {k: None for k in model.state_dict().keys()}       # matched entries are replaced by the loaded value
        ms.align_and_update_state_dicts(msd, {k: _Named(k) for k in loaded_keys}, flownet=flownet)
        return {k: (v.name if v is not None else None) for k, v in msd.items()}
    model = ref_shim.build_model(cfg)
    out["mega_from_c2"] = alignment(model, list(renamed.keys()), False)
    full = ["module." + k for k in model.state_dict().keys()]
    out["mega_from_module_prefixed"] = alignment(model, list(ms.strip_prefix_if_present({k: 0 for k in full}, "module.").keys()), False)
    fcfg = ref_shim.make_cfg("configs/FGFA/vid_R_50_C4_FGFA_1x.yaml")
    fmodel = ref_shim.build_model(fcfg)
    flow_keys = [k[len("flownet."):] for k in fmodel.state_dict().keys() if k.startswith("flownet.")]
    out["fgfa_flownet_file"] = alignment(fmodel, flow_keys, True)
    out["fgfa_from_c2"] = alignment(fmodel, list(renamed.keys()), False)
    with open(os.path.join(HERE, "ref_checkpoint.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)


def golden_configs():
    """ref_configs.json: for every reference config file this package mirrors, the values the REFERENCE's yacs tree
    holds (defaults.py merged with the yaml) for exactly the keys mega.pytorch_amd.config.get_cfg defines."""
    import json
    from mega.pytorch_amd import config as myconfig
    cases = {"mega_R-101": ("configs/MEGA/vid_R_101_C4_MEGA_1x.yaml", "R-101", "mega"),
             "mega_R-50": ("configs/MEGA/vid_R_50_C4_MEGA_1x.yaml", "R-50", "mega"),
             "rdn_R-101": ("configs/RDN/vid_R_101_C4_RDN_1x.yaml", "R-101", "rdn"),
             "rdn_base_R-101": ("configs/RDN/vid_R_101_C4_RDN_base_1x.yaml", "R-101", "rdn_base"),
             "rdn_base_R-50": ("configs/RDN/vid_R_50_C4_RDN_base_1x.yaml", "R-50", "rdn_base"),
             "fgfa_R-101": ("configs/FGFA/vid_R_101_C4_FGFA_1x.yaml", "R-101", "fgfa"),
             "fgfa_R-50": ("configs/FGFA/vid_R_50_C4_FGFA_1x.yaml", "R-50", "fgfa"),
             "dff_R-101": ("configs/DFF/vid_R_101_C4_DFF_1x.yaml", "R-101", "dff"),
             "dff_R-50": ("configs/DFF/vid_R_50_C4_DFF_1x.yaml", "R-50", "dff"),
             "base_R-101": ("configs/vid_R_101_C4_1x.yaml", "R-101", "base"),
             "base_R-50": ("configs/vid_R_50_C4_1x.yaml", "R-50", "base")}

    def flat(node, prefix=""):
        out = {}
        for k, v in node.items():
            if isinstance(v, dict):
                out.update(flat(v, prefix + k + "."))
            else:
                out[prefix + k] = v
        return out
    out = {}
    for name, (yaml_path, arch, method) in cases.items():
        ref = ref_shim.make_cfg(yaml_path)
        vals = {}
        for key in flat(myconfig.get_cfg(arch, method)):
            node = ref
            try:
                for part in key.split("."):
                    node = node[part] if isinstance(node, dict) else getattr(node, part)
            except (KeyError, AttributeError):
                continue            # a key this package adds (NMS_STRICT_GT): not in the reference
            vals[key] = list(node) if isinstance(node, (tuple, list)) else node
        out[name] = {"arch": arch, "method": method, "yaml": yaml_path, "values": vals}
    with open(os.path.join(HERE, "ref_configs.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)


def golden_feed():
    """ref_feed.npz: the reference's test-time transform chain (data/transforms/build.py:26-45 with the default
    INPUT.* of config/defaults.py) on VID-like frame sizes (Resize.get_size) and on one seeded frame (full chain).
    torchvision is absent here, so its three functional calls the chain makes are bound to the exact Pillow / torch
    calls torchvision implements them with (F.resize -> PIL resize BILINEAR, F.to_tensor -> HWC u8 / 255 -> CHW,
    F.normalize -> (x - mean) / std)."""
    import types
    from PIL import Image
    cfg = ref_shim.make_cfg("configs/MEGA/vid_R_101_C4_MEGA_1x.yaml")     # installs the import shims
    import mega_core.data.transforms.transforms as T
    Fm = types.SimpleNamespace(
        resize=lambda img, size: img.resize(size[::-1], Image.BILINEAR),
        to_tensor=lambda img: torch.from_numpy(np.asarray(img).copy()).permute(2, 0, 1).float().div(255),
        normalize=lambda t, mean, std: (t - torch.tensor(mean).view(-1, 1, 1)) / torch.tensor(std).view(-1, 1, 1))
    T.F = Fm
    chain = T.Compose([T.Resize(cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST), T.ToTensor(),
                       T.Normalize(mean=cfg.INPUT.PIXEL_MEAN, std=cfg.INPUT.PIXEL_STD, to_bgr255=cfg.INPUT.TO_BGR255)])
    sizes = [(1280, 720), (480, 360), (1920, 1080), (1000, 600), (375, 500), (211, 97), (1000, 333), (640, 480),
             (500, 500), (320, 240), (1280, 704), (718, 480)]
    rs = T.Resize(cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST)
    out = {"sizes_wh": np.array(sizes), "get_size_hw": np.array([rs.get_size(s) for s in sizes])}
    img = synth.make_clip(1, 360, 480, seed=11)[0].numpy()
    t, _ = chain(Image.fromarray(img), None)
    out["img"] = img
    out["transformed"] = t.numpy()
    np.savez_compressed(os.path.join(HERE, "ref_feed.npz"), **out)


def golden_fgfa():
    c = FGFA
    cfg = ref_shim.make_cfg("configs/FGFA/vid_R_50_C4_FGFA_1x.yaml")
    model = ref_shim.build_model(cfg)
    model.load_state_dict(synth.make_fgfa_state_dict(seed=c["seed_w"]), strict=True)
    frames = synth.preprocess_cpu(synth.make_clip(c["T"], c["H"], c["W"], seed=c["seed_clip"]))
    import mega_core.modeling.detector.generalized_rcnn_fgfa as gm

    class _FakeImg(object):
        def __init__(self, i): self.i = i
        def convert(self, m): return self

    class _FakeImage(object):
        @staticmethod
        def open(path): return _FakeImg(int(path))
    gm.Image = _FakeImage
    trace = {}
    model.flownet.register_forward_hook(lambda m, i, o: trace.update(flow=o.detach().clone()))
    model.rpn.register_forward_hook(lambda m, i, o: trace.update(feats=i[1][0].detach().clone()))
    model.roi_heads.box.predictor.register_forward_hook(
        lambda m, i, o: trace.update(logits=o[0].detach().clone(), deltas=o[1].detach().clone()))
    out = {}
    for idx in range(c["nkey"]):
        images = {"cur": frames[idx], "ref": [frames[min(c["T"] - 1, idx + 9)]], "frame_category": 0 if idx == 0 else 1,
                  "seg_len": c["T"], "pattern": "%d", "img_dir": "%s", "transforms": lambda im: frames[im.i]}
        with torch.no_grad():
            det = model(images)[0]
        out["boxes%d" % idx] = det.bbox.numpy()
        out["scores%d" % idx] = det.get_field("scores").numpy()
        out["labels%d" % idx] = det.get_field("labels").numpy()
        out["flow%d" % idx] = trace["flow"].numpy()
        out["feats%d" % idx] = trace["feats"].numpy()[0, ::16]     # every 16th channel: keeps the fixture small
        out["logits%d" % idx] = trace["logits"].numpy()
        out["deltas%d" % idx] = trace["deltas"].numpy()
        print("fgfa frame", idx, "dets", det.bbox.shape[0], "flow absmax", float(trace["flow"].abs().max()))
    for k, v in c.items():
        out["cfg_" + k] = np.int64(v)
    np.savez_compressed(os.path.join(HERE, "ref_fgfa_r50.npz"), **out)


if __name__ == "__main__":
    if not ref_shim.available():
        sys.exit("needs /root/reference")
    torch.set_num_threads(8)
    if len(sys.argv) > 1:                      # e.g. `python tests/golden/make_golden.py golden_e2e_long`
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    golden_from_reference_tests()
    golden_ops()
    golden_e2e()
    golden_e2e_long()
    golden_fgfa()
    golden_base()
    golden_feed()
    golden_rdn()
    golden_dff()
    golden_checkpoint()
    golden_configs()

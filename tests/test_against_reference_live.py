"""Value types and small host functions against the LIVE reference package (imported from /root/reference through
oracle/ref_shim.py; skipped where the reference tree is absent, e.g. on the GPU box)."""
import numpy as np
import pytest
import torch

import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import ref_shim  # noqa: E402
from mega.pytorch_amd import config, modeling, structures, synth

pytestmark = pytest.mark.skipif(not ref_shim.available(), reason="needs /root/reference")


@pytest.fixture(scope="module")
def ref():
    ref_shim.install()
    import mega_core.structures.bounding_box as bb
    import mega_core.structures.boxlist_ops as ops
    import mega_core.structures.image_list as il
    import mega_core.modeling.poolers as poolers
    return {"bb": bb, "ops": ops, "il": il, "poolers": poolers}


def _boxes(n, seed):
    g = torch.Generator().manual_seed(seed)
    c = torch.rand((n, 2), generator=g) * torch.tensor([300., 200.]) - 20
    wh = torch.rand((n, 2), generator=g) * 120 - 10          # some degenerate / out-of-image boxes on purpose
    return torch.cat([c, c + wh], dim=1)


def test_boxlist_semantics(ref):
    """bounding_box.py: clip_to_image (in place, remove_empty), indexing, fields, copy_with_fields, to()."""
    for seed in range(5):
        b = _boxes(40, seed)
        s = torch.rand(40)
        mine = structures.BoxList(b.clone(), (256, 160), "xyxy")
        theirs = ref["bb"].BoxList(b.clone(), (256, 160), "xyxy")
        for bl in (mine, theirs):
            bl.add_field("scores", s.clone())
            bl.add_field("labels", torch.arange(40))
        for remove_empty in (False, True):
            m2 = structures.BoxList(b.clone(), (256, 160), "xyxy"); m2.add_field("scores", s.clone())
            t2 = ref["bb"].BoxList(b.clone(), (256, 160), "xyxy"); t2.add_field("scores", s.clone())
            mo_, to_ = m2.clip_to_image(remove_empty), t2.clip_to_image(remove_empty)
            assert torch.equal(mo_.bbox, to_.bbox) and torch.equal(mo_.get_field("scores"), to_.get_field("scores"))
            assert torch.equal(m2.bbox, t2.bbox)                     # the clamp happened in place in both
        idx = torch.tensor([3, 1, 7])
        assert torch.equal(mine[idx].bbox, theirs[idx].bbox) and torch.equal(mine[idx].get_field("labels"), theirs[idx].get_field("labels"))
        keep = s > 0.5
        assert torch.equal(mine[keep].bbox, theirs[keep].bbox) and len(mine[keep]) == len(theirs[keep])
        assert mine.fields() == theirs.fields() and mine.size == theirs.size and mine.mode == theirs.mode
        c1, c2 = mine.copy_with_fields(["scores"]), theirs.copy_with_fields(["scores"])
        assert c1.fields() == c2.fields() and torch.equal(c1.bbox, c2.bbox)
        with pytest.raises(KeyError):
            mine.copy_with_fields(["nope"])
        assert repr(mine).startswith("BoxList(num_boxes=40")


def test_cat_boxlist_and_image_list(ref):
    a, b = _boxes(5, 1), _boxes(7, 2)
    def mk(mod, x, f):
        bl = mod.BoxList(x.clone(), (100, 80), "xyxy")
        bl.add_field("scores", f.clone())
        return bl
    fa, fb = torch.rand(5), torch.rand(7)
    m = structures.cat_boxlist([mk(structures, a, fa), mk(structures, b, fb)])
    t = ref["ops"].cat_boxlist([mk(ref["bb"], a, fa), mk(ref["bb"], b, fb)])
    assert torch.equal(m.bbox, t.bbox) and torch.equal(m.get_field("scores"), t.get_field("scores")) and m.size == t.size
    img = torch.rand(3, 37, 53)
    for inp in (img, [img], (img,)):
        mi, ti = structures.to_image_list(inp), ref["il"].to_image_list(inp)
        assert torch.equal(mi.tensors, ti.tensors) and [tuple(x) for x in mi.image_sizes] == [tuple(x) for x in ti.image_sizes]
    assert structures.to_image_list(mi) is mi


def test_convert_to_roi_format_and_anchor_buffers(ref):
    """poolers.py:78-89 (batch index column, concatenation order) and the anchor generator's cell anchors as they sit
    in the reference model's state_dict."""
    boxes = [structures.BoxList(_boxes(4, 3), (64, 64)), structures.BoxList(_boxes(2, 4), (64, 64))]
    theirs = [ref["bb"].BoxList(b.bbox.clone(), (64, 64), "xyxy") for b in boxes]
    pooler = ref["poolers"].Pooler((7, 7), (1.0 / 16,), 0)
    assert torch.equal(modeling.convert_to_roi_format(boxes), pooler.convert_to_roi_format(theirs))
    cfg = ref_shim.make_cfg("configs/MEGA/vid_R_50_C4_MEGA_1x.yaml")
    rmodel = ref_shim.build_model(cfg)
    mcfg = config.get_cfg("R-50")
    mcfg.MODEL.DEVICE = "cpu"
    mine = modeling.build_detection_model(mcfg)
    key = "rpn.anchor_generator.cell_anchors.0"
    assert torch.equal(mine.state_dict()[key], rmodel.state_dict()[key])
    assert torch.equal(synth._cell_anchors(16, (64, 128, 256, 512), (0.5, 1.0, 2.0)), rmodel.state_dict()[key])
    a, b = mine.state_dict(), rmodel.state_dict()
    assert list(a.keys()) == list(b.keys()) and all(a[k].shape == b[k].shape for k in a)

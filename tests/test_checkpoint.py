"""Weight loading against what the reference itself computes (tests/golden/ref_checkpoint.json,
produced by mega_core.utils.c2_model_loading / model_serialization in the build container)."""
import json
import os
import pickle

import numpy as np
import pytest
import torch

from mega.pytorch_amd import checkpoint, config, modeling, synth

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_checkpoint.json")))


def test_c2_blob_renaming_equals_reference():
    blobs = [b for b in GOLD["c2_blobs"] if "_momentum" not in b]
    got = [checkpoint.rename_c2_key(b, checkpoint._C2_STAGE_NAMES["R-50"]) for b in sorted(blobs)]
    # ('pred_w' and 'fc1000_w' are both in the list on purpose: they collapse to one entry, first occurrence's slot)
    assert list(dict.fromkeys(got)) == GOLD["c2_renamed"]
    assert "backbone" not in got[0] and "layer1.0.downsample.1.weight" in got and "rpn.head.cls_logits.bias" in got


@pytest.mark.parametrize("case,method,flownet", [("mega_from_c2", "mega", False), ("mega_from_module_prefixed", "mega", False),
                                                 ("fgfa_flownet_file", "fgfa", True), ("fgfa_from_c2", "fgfa", False)])
def test_key_alignment_equals_reference(case, method, flownet):
    model = modeling.build_detection_model(_cpu_cfg(method))
    want = GOLD[case]
    assert sorted(model.state_dict().keys()) == sorted(want.keys()), "state_dict layout differs from the reference's"
    loaded = sorted({v for v in want.values() if v is not None})
    if case == "mega_from_module_prefixed":
        loaded = sorted(checkpoint.strip_prefix_if_present({"module." + k: 0 for k in want.keys()}).keys())
    elif case == "fgfa_flownet_file":
        loaded = sorted(k[len("flownet."):] for k in want.keys() if k.startswith("flownet."))
    elif case.endswith("from_c2"):
        loaded = GOLD["c2_renamed"]
    got = checkpoint.match_keys(sorted(want.keys()), loaded, flownet=flownet)
    assert got == want


def _cpu_cfg(method="mega"):
    cfg = config.get_cfg("R-50", method)
    cfg.MODEL.DEVICE = "cpu"
    return cfg


def test_load_c2_pickle_and_pth_round_trip(tmp_path):
    cfg = _cpu_cfg()
    model = modeling.build_detection_model(cfg)
    sd = synth.make_state_dict(blocks=(3, 4, 6), reduce_channel=True, global_res_stage=0, seed=1)
    # a Detectron-style pickle of the R-50 trunk: invert the renaming for the backbone tensors
    blobs = {}
    inv = {checkpoint.rename_c2_key(b, checkpoint._C2_STAGE_NAMES["R-50"]): b for b in GOLD["c2_blobs"] if "_momentum" not in b}
    rng = np.random.RandomState(0)
    for new, old in inv.items():
        if not old.startswith(("conv1", "res")):       # an ImageNet trunk: no RPN / box-head blobs
            continue
        tgt = [k for k in sd if k.endswith(new) and "running" not in k]
        if tgt:
            blobs[old] = rng.randn(*sd[tgt[0]].shape).astype(np.float32)
    blobs["conv1_w_momentum"] = np.zeros((1,), np.float32)
    p = str(tmp_path / "R-50.pkl")
    with open(p, "wb") as f:
        pickle.dump({"blobs": blobs}, f)
    mapping = checkpoint.load_checkpoint(cfg, model, p)
    msd = model.state_dict()
    assert torch.equal(msd["backbone.body.stem.conv1.weight"], torch.from_numpy(blobs["conv1_w"]))
    assert torch.equal(msd["backbone.body.layer2.0.downsample.1.weight"], torch.from_numpy(blobs["res3_0_branch1_bn_s"]))
    assert torch.equal(msd["roi_heads.box.feature_extractor.head.layer4.2.conv3.weight"],
                       torch.from_numpy(blobs["res5_2_branch2c_w"]))
    assert mapping["roi_heads.box.feature_extractor.l_fcs.0.weight"] is None     # not in an ImageNet trunk: keeps init
    # full .pth with a DataParallel prefix and a bare state_dict
    p2 = str(tmp_path / "MEGA_R_50.pth")
    torch.save({"model": {"module." + k: v for k, v in sd.items()}, "iteration": 7}, p2)
    m2 = modeling.build_detection_model(cfg)
    mapping = checkpoint.load_checkpoint(cfg, m2, p2)
    assert all(v is not None for v in mapping.values())
    assert all(torch.equal(v, sd[k]) for k, v in m2.state_dict().items())
    torch.save(sd, str(tmp_path / "bare.pth"))
    m3 = modeling.build_detection_model(cfg)
    checkpoint.load_checkpoint(cfg, m3, str(tmp_path / "bare.pth"))
    assert torch.equal(m3.state_dict()["rpn.head.conv.weight"], sd["rpn.head.conv.weight"])
    # errors
    bad = dict(sd)
    bad["rpn.head.conv.weight"] = torch.zeros(3, 3)
    torch.save(bad, str(tmp_path / "bad.pth"))
    with pytest.raises(RuntimeError):
        checkpoint.load_checkpoint(cfg, modeling.build_detection_model(cfg), str(tmp_path / "bad.pth"))
    with pytest.raises(ValueError):
        checkpoint.load_file(cfg, "catalog://ImageNetPretrained/MSRA/R-50")


def _complex_model():
    """the nested model + flat state_dict of the reference's own loader test (tests/checkpoint.py:19-36)."""
    from collections import OrderedDict
    from torch import nn
    m = nn.Module()
    m.block1 = nn.Module()
    m.block1.layer1 = nn.Linear(2, 3)
    m.layer2 = nn.Linear(3, 2)
    m.res = nn.Module()
    m.res.layer2 = nn.Linear(3, 2)
    g = torch.Generator().manual_seed(0)
    sd = OrderedDict()
    sd["layer1.weight"] = torch.rand(3, 2, generator=g)
    sd["layer1.bias"] = torch.rand(3, generator=g)
    sd["layer2.weight"] = torch.rand(2, 3, generator=g)
    sd["layer2.bias"] = torch.rand(2, generator=g)
    sd["res.layer2.weight"] = torch.rand(2, 3, generator=g)
    sd["res.layer2.bias"] = torch.rand(2, generator=g)
    return m, sd


@pytest.mark.parametrize("data_parallel", [False, True])
def test_reference_complex_model_case(data_parallel):
    """tests/checkpoint.py:103-114 test_complex_model_loaded: keys that differ by a prefix, 'layer2' vs 'res.layer2'
    resolved by the longest suffix, with and without a DataParallel wrapper on the model side."""
    from torch import nn
    model, sd = _complex_model()
    if data_parallel:
        model = nn.DataParallel(model)
    checkpoint.load_state_dict(model, sd)
    for loaded, stored in zip(model.state_dict().values(), sd.values()):
        assert loaded.equal(stored)
    # and the loaded side wrapped instead (checkpoint saved from a DataParallel model)
    model2, _ = _complex_model()
    checkpoint.load_state_dict(model2, {"module." + k: v for k, v in sd.items()})
    for loaded, stored in zip(model2.state_dict().values(), sd.values()):
        assert loaded.equal(stored)

"""The reference's own "every registered builder runs and returns the documented shapes" tests
(tests/test_backbones.py, test_rpn_heads.py, test_predictors.py, test_feature_extractors.py, test_detectors.py),
restated for the registries of this package (same names, same constructor / call signatures), on the CPU twins."""
import pytest
import torch

from mega.pytorch_amd import config, modeling
import mega.pytorch_amd.fgfa  # noqa: F401  (registers ResNetConv52MLPFeatureExtractor and the FGFA / DFF / base detectors)
import mega.pytorch_amd.rdn  # noqa: F401
from mega.pytorch_amd.structures import BoxList
import cpu_ops


def _cfg(arch="R-50", method="mega"):
    cfg = config.get_cfg(arch, method)
    cfg.MODEL.DEVICE = "cpu"
    return cfg


def test_build_backbones(monkeypatch):
    """tests/test_backbones.py:24-51: out_channels attribute; out[i].shape[:2] == (N, out_channels)."""
    cpu_ops.install(monkeypatch)
    assert set(modeling.BACKBONES.keys()) >= {"R-50-C4", "R-101-C4"}
    for name, builder in modeling.BACKBONES.items():
        cfg = _cfg("R-101" if "101" in name else "R-50")
        backbone = builder(cfg)
        assert getattr(backbone, "out_channels", None) is not None, name
        if "101" in name:
            continue                                   # (23-block layer3 on the CPU twins: skipped for time)
        x = torch.rand([2, 3, 64, 96])
        with torch.no_grad():
            out = backbone(x)
        for cur in out:
            assert cur.shape[:2] == torch.Size([2, backbone.out_channels])
            assert cur.shape[2:] == torch.Size([4, 6])     # stride 16


def test_build_rpn_heads(monkeypatch):
    """tests/test_rpn_heads.py:21-59: builder(cfg, in_channels, num_anchors); forward(list of maps) ->
    (logits list, bbox_reg list) with A and 4A channels."""
    cpu_ops.install(monkeypatch)
    assert len(modeling.RPN_HEADS) > 0
    in_channels, num_anchors = 64, 10
    for name, builder in modeling.RPN_HEADS.items():
        rpn = builder(_cfg(), in_channels, num_anchors)
        x = torch.rand([2, in_channels, 24, 32])
        with torch.no_grad():
            out = rpn([x] * 3)
        assert len(out) == 2
        logits, bbox_reg = out
        for idx in range(3):
            assert logits[idx].shape == torch.Size([2, num_anchors, 24, 32])
            assert bbox_reg[idx].shape == torch.Size([2, num_anchors * 4, 24, 32])


def test_roi_box_predictors(monkeypatch):
    """tests/test_predictors.py:55-72: (scores [N, NUM_CLASSES], deltas [N, 4 NUM_CLASSES])."""
    cpu_ops.install(monkeypatch)
    assert len(modeling.ROI_BOX_PREDICTOR) > 0
    for name, builder in modeling.ROI_BOX_PREDICTOR.items():
        cfg = _cfg()
        pred = builder(cfg, 1024)
        x = torch.rand([2, 1024])
        with torch.no_grad():
            scores, deltas = pred(x)
        assert scores.shape == (2, cfg.MODEL.ROI_BOX_HEAD.NUM_CLASSES)
        assert deltas.shape == (2, 4 * cfg.MODEL.ROI_BOX_HEAD.NUM_CLASSES)


def test_roi_box_feature_extractors(monkeypatch):
    """tests/test_feature_extractors.py:26-60: out_channels; fe([maps], [boxes] * N) -> [N * len(boxes), out_channels]
    for the single-image extractor; the video extractors (MEGA / RDN) through their pre_calculate form, which has the
    same (features, proposals) signature (roi_box_feature_extractors.py:404-414,:885-896)."""
    cpu_ops.install(monkeypatch)
    assert set(modeling.ROI_BOX_FEATURE_EXTRACTORS.keys()) >= {"MEGAFeatureExtractor", "RDNFeatureExtractor",
                                                                "ResNetConv52MLPFeatureExtractor"}
    boxes = [[1, 1, 100, 100], [50, 50, 80, 80], [20, 20, 30, 40]]
    for name, builder in modeling.ROI_BOX_FEATURE_EXTRACTORS.items():
        method = {"MEGAFeatureExtractor": "mega", "RDNFeatureExtractor": "rdn"}.get(name, "base")
        fe = builder(_cfg("R-50", method), 1024).eval()      # (inference path only: the training forward raises)
        assert getattr(fe, "out_channels", None) is not None, name
        N = 1 if method != "base" else 2
        x = torch.rand([N, 1024, 12, 16])
        bl = [BoxList(torch.tensor(boxes, dtype=torch.float32), (256, 192), "xyxy") for _ in range(N)]
        with torch.no_grad():
            out = fe(x, bl, pre_calculate=True) if method != "base" else fe((x,), bl)
        assert out.shape[:2] == torch.Size([N * len(boxes), fe.out_channels])


@pytest.mark.parametrize("method,name", [("mega", "GeneralizedRCNNMEGA"), ("rdn", "GeneralizedRCNNRDN"),
                                         ("fgfa", "GeneralizedRCNNFGFA"), ("dff", "GeneralizedRCNNDFF"),
                                         ("base", "GeneralizedRCNN")])
def test_detectors_build_from_their_configs(method, name):
    """tests/test_detectors.py: every meta-architecture builds from its config and exposes backbone / rpn / roi_heads."""
    model = modeling.build_detection_model(_cfg("R-50", method))
    assert type(model).__name__ == name and name in modeling.DETECTION_META_ARCHITECTURES
    for part in ("backbone", "rpn", "roi_heads"):
        assert hasattr(model, part)
    assert not model.training
    with pytest.raises(ValueError):
        model(torch.zeros(3, 32, 32) if method == "base" else {"cur": torch.zeros(3, 32, 32), "frame_category": 0,
                                                                  "seg_len": 1, "is_key_frame": True}, targets=[None])
    with pytest.raises(AssertionError):
        modeling.DETECTION_META_ARCHITECTURES.register(name, object)      # utils/registry.py:4-6: names are unique


def _flat(node, prefix=""):
    out = {}
    for k, v in node.items():
        if isinstance(v, dict):
            out.update(_flat(v, prefix + k + "."))
        else:
            out[prefix + k] = v
    return out


def test_config_values_equal_the_reference_configs():
    """tests/test_configs.py loads every yaml; here: for the 11 config files this package mirrors, every key of
    config.get_cfg(arch, method) that exists in the reference holds the value the reference's defaults + yaml give
    (tests/golden/ref_configs.json, dumped from the reference's own yacs tree)."""
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_configs.json")))
    assert len(gold) == 11
    for name, case in gold.items():
        mine = _flat(config.get_cfg(case["arch"], case["method"]))
        assert len(case["values"]) >= 60, name
        for key, want in case["values"].items():
            got = mine[key]
            got = list(got) if isinstance(got, (tuple, list)) else got
            if key == "MODEL.DEVICE":
                continue
            assert got == want, "%s: %s = %r, reference has %r" % (name, key, got, want)

"""Weight loading: released MEGA / RDN / FGFA `.pth` checkpoints and Caffe2 `.pkl` backbones go
into the modules of this package exactly as they go into the reference's, because the parameter names are the same.

Mirror of  mega_core/utils/model_serialization.py:10-80   suffix alignment of checkpoint keys to model keys
           mega_core/utils/c2_model_loading.py:12-207     Caffe2 (Detectron) blob names -> torchvision-style names
           mega_core/utils/checkpoint.py:52-71,:103-148   load(): "model" entry, optional separate FlowNet file

Deliberately not mirrored: optimizer / scheduler state, `last_checkpoint` bookkeeping, catalog:// and http:// lookups
(training-side and network-side; there is no network in deployment images either -- pass a local path).
"""
import logging
import pickle
from collections import OrderedDict

import numpy as np
import torch

# c2_model_loading.py:12-60 + :96-107 + :110, as data: substring replacements applied IN THIS ORDER to every blob name.
_C2_RENAMES = [
    ("_", "."), (".w", ".weight"), (".bn", "_bn"), (".b", ".bias"), ("_bn.s", "_bn.scale"),
    (".biasranch", ".branch"), ("bbox.pred", "bbox_pred"), ("cls.score", "cls_score"), ("res.conv1_", "conv1_"),
    (".biasbox", ".bbox"), ("conv.rpn", "rpn.conv"), ("rpn.bbox.pred", "rpn.bbox_pred"),
    ("rpn.cls.logits", "rpn.cls_logits"),
    ("_bn.scale", "_bn.weight"),
    ("conv1_bn.", "bn1."),
    ("res2.", "layer1."), ("res3.", "layer2."), ("res4.", "layer3."), ("res5.", "layer4."),
    (".branch2a.", ".conv1."), (".branch2a_bn.", ".bn1."), (".branch2b.", ".conv2."), (".branch2b_bn.", ".bn2."),
    (".branch2c.", ".conv3."), (".branch2c_bn.", ".bn3."),
    (".branch1.", ".downsample.0."), (".branch1_bn.", ".downsample.1."),
    ("conv1.gn.s", "bn1.weight"), ("conv1.gn.bias", "bn1.bias"), ("conv2.gn.s", "bn2.weight"),
    ("conv2.gn.bias", "bn2.bias"), ("conv3.gn.s", "bn3.weight"), ("conv3.gn.bias", "bn3.bias"),
    ("downsample.0.gn.s", "downsample.1.weight"), ("downsample.0.gn.bias", "downsample.1.bias"),
]
# after the FPN renames (no-ops for the C4 bodies this package builds): mask / keypoint heads, then the RPN nesting
_C2_RENAMES_TAIL = [
    ("mask.fcn.logits", "mask_fcn_logits"), (".[mask].fcn", "mask_fcn"), ("conv5.mask", "conv5_mask"),
    ("kps.score.lowres", "kps_score_lowres"), ("kps.score", "kps_score"), ("conv.fcn", "conv_fcn"),
    ("rpn.", "rpn.head."),
]
_C2_STAGE_NAMES = {"R-50": ["1.2", "2.3", "3.5", "4.2"], "R-101": ["1.2", "2.3", "3.22", "4.2"],
                   "R-152": ["1.2", "2.7", "3.35", "4.2"]}


def rename_c2_key(name, stage_names=()):
    """One Caffe2 blob name -> the reference's parameter name (c2_model_loading.py:82-110)."""
    if name == "pred_b":
        name = "fc1000_b"
    elif name == "pred_w":
        name = "fc1000_w"
    for old, new in _C2_RENAMES:
        name = name.replace(old, new)
    for mapped_idx, stage in enumerate(stage_names, 1):                       # :62-79 (FPN lateral / output convs)
        suffix = ".lateral" if mapped_idx < 4 else ""
        name = name.replace("fpn.inner.layer%s.sum%s" % (stage, suffix), "fpn_inner%d" % mapped_idx)
        name = name.replace("fpn.layer%s.sum" % stage, "fpn_layer%d" % mapped_idx)
    for old, new in (("rpn.conv.fpn2", "rpn.conv"), ("rpn.bbox_pred.fpn2", "rpn.bbox_pred"),
                     ("rpn.cls_logits.fpn2", "rpn.cls_logits")):
        name = name.replace(old, new)
    for old, new in _C2_RENAMES_TAIL:
        name = name.replace(old, new)
    return name


def load_c2_format(cfg, path):
    """c2_model_loading.py:131-207: Detectron `.pkl` -> {"model": OrderedDict(name -> tensor)} (momentum blobs dropped)."""
    with open(path, "rb") as f:
        data = pickle.load(f, encoding="latin1")
    weights = data["blobs"] if "blobs" in data else data
    body = cfg.MODEL.BACKBONE.CONV_BODY
    arch = body.replace("-C4", "").replace("-C5", "").replace("-FPN", "").replace("-RETINANET", "")
    if arch not in _C2_STAGE_NAMES:
        raise KeyError("no Caffe2 loader for CONV_BODY %s" % body)
    out = OrderedDict()
    for k in sorted(weights.keys()):
        if "_momentum" in k:
            continue
        out[rename_c2_key(k, _C2_STAGE_NAMES[arch])] = torch.from_numpy(np.asarray(weights[k]))
    return {"model": out}


def strip_prefix_if_present(state_dict, prefix="module."):
    """model_serialization.py:59-67: drop a DataParallel prefix only if EVERY key carries it."""
    if not all(k.startswith(prefix) for k in state_dict.keys()):
        return state_dict
    return OrderedDict((k.replace(prefix, ""), v) for k, v in state_dict.items())


def match_keys(model_keys, loaded_keys, flownet=False):
    """model_serialization.py:10-46: for every model key the LONGEST loaded key that is a suffix of it, or None.
    flownet=False skips flownet./embednet. parameters, flownet=True touches only flownet ones, None = no filter."""
    loaded = set(loaded_keys)
    out = {}
    for i in model_keys:
        if flownet is None:
            ok = True
        elif not flownet:
            ok = ("flownet" not in i) and ("embednet" not in i)
        else:
            ok = "flownet" in i
        best = None
        if ok:
            for start in range(len(i)):          # suffixes of the model key, longest first: the first hit wins
                if i[start:] in loaded:
                    best = i[start:]
                    break
        out[i] = best
    return out


def load_state_dict(model, loaded_state_dict, flownet=False):
    """model_serialization.py:70-80: align by suffix, keep the model's own value where nothing matches, strict load."""
    logger = logging.getLogger("mega.pytorch_amd.checkpoint")
    model_sd = model.state_dict()
    loaded = strip_prefix_if_present(loaded_state_dict, "module.")
    mapping = match_keys(sorted(model_sd.keys()), sorted(loaded.keys()), flownet=flownet)
    for key, src in mapping.items():
        if src is None:
            continue
        if tuple(model_sd[key].shape) != tuple(loaded[src].shape):
            raise RuntimeError("size mismatch for %s: checkpoint %s has %s, model expects %s" % (
                key, src, tuple(loaded[src].shape), tuple(model_sd[key].shape)))
        model_sd[key] = loaded[src]
        logger.info("%s loaded from %s of shape %s", key, src, tuple(loaded[src].shape))
    model.load_state_dict(model_sd)
    return mapping


def load_file(cfg, path):
    """checkpoint.py:129-148 for local files: `.pkl` = Caffe2, anything else = torch.save'd dict (bare state_dicts
    are wrapped as {"model": ...})."""
    if path.startswith(("catalog://", "http://", "https://")):
        raise ValueError("remote / catalog weights are not resolved here: pass a local file (%s)" % path)
    if path.endswith(".pkl"):
        return load_c2_format(cfg, path)
    loaded = torch.load(path, map_location="cpu", weights_only=False)
    if "model" not in loaded:
        loaded = dict(model=loaded)
    return loaded


def load_checkpoint(cfg, model, path, flownet=False):
    """Checkpointer.load (checkpoint.py:52-71) minus optimizer / scheduler: returns the key mapping that was applied.
    FGFA / DFF load twice in the reference: the detector weights (flownet=False), then a FlowNet file (flownet=True)."""
    ckpt = load_file(cfg, path)
    return load_state_dict(model, ckpt.pop("model"), flownet=flownet)

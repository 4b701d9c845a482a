"""TEST INFRASTRUCTURE ONLY -- runs the UNMODIFIED python reference (/root/reference/mega_core)
on CPU inside this container, so that golden vectors can be generated from it and the python
restatement in ``oracle/mega_oracle.py`` can be pinned against it.

Nothing here is imported by the product package.  /root/reference does not exist on the GPU box;
``available()`` is False there and every caller must skip.

Shims installed *before* ``import mega_core`` (all are environment stand-ins, none changes
arithmetic):
  1. ``torch._six`` stub with ``PY3=True``                (mega_core/utils/imports.py:4)
  2. ``yacs.config.CfgNode`` stand-in                       (mega_core/config/defaults.py:4)
  3. ``apex.amp`` stub, ``float_function`` = identity       (mega_core/layers/nms.py:5, roi_align.py:10)
  4. ``numpy.float/int/bool`` aliases                       (mega_core/modeling/rpn/anchor_generator.py:229-238)
  5. empty ``cv2`` / ``pycocotools`` / ``torchvision`` / ``cityscapesscripts`` modules
  6. a no-op ``nvidia-smi`` on PATH                         (mega_core/utils/distributed.py:64-76)
  7. ``oracle/_ref`` (the reference's own csrc/cpu ops, see build_ref.py) as ``mega_core._C``
"""
import ast
import copy
import os
import stat
import sys
import tempfile
import types

REF_ROOT = "/root/reference"
_INSTALLED = False


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "mega_core"))


# ----------------------------------------------------------------------------- yacs stand-in
class CfgNode(dict):
    """Minimal yacs.config.CfgNode replacement: attribute dict + the merge API the reference uses."""

    def __init__(self, init_dict=None, key_list=None, new_allowed=False):
        super().__init__()
        init_dict = {} if init_dict is None else init_dict
        for k, v in init_dict.items():
            if isinstance(v, dict) and not isinstance(v, CfgNode):
                v = CfgNode(v)
            self[k] = v
        self.__dict__["_frozen"] = False

    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.__dict__.get("_frozen", False):
            raise AttributeError("Attempted to set {} on a frozen CfgNode".format(name))
        self[name] = value

    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def is_frozen(self):
        return self.__dict__.get("_frozen", False)

    def _set_frozen(self, flag):
        self.__dict__["_frozen"] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(flag)

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CfgNode()
        for k, v in self.items():
            dict.__setitem__(out, k, copy.deepcopy(v, memo))
        out.__dict__["_frozen"] = self.__dict__.get("_frozen", False)
        return out

    @staticmethod
    def _decode(v):
        if isinstance(v, str):
            try:
                return ast.literal_eval(v)
            except (ValueError, SyntaxError):
                return v
        return v

    def _merge_dict(self, d):
        for k, v in d.items():
            if isinstance(v, dict):
                if k not in self:
                    dict.__setitem__(self, k, CfgNode())
                self[k]._merge_dict(v)
            else:
                v = self._decode(v)
                if k in self and isinstance(self[k], tuple) and isinstance(v, list):
                    v = tuple(v)
                if k in self and isinstance(self[k], list) and isinstance(v, tuple):
                    v = list(v)
                dict.__setitem__(self, k, v)

    def merge_from_file(self, path):
        import yaml
        with open(path) as f:
            d = yaml.safe_load(f) or {}
        self._merge_dict(d)

    def merge_from_other_cfg(self, other):
        self._merge_dict(other)

    def merge_from_list(self, lst):
        assert len(lst) % 2 == 0
        for full_key, v in zip(lst[0::2], lst[1::2]):
            node = self
            keys = full_key.split(".")
            for k in keys[:-1]:
                node = node[k]
            v = self._decode(v)
            old = node.get(keys[-1])
            if isinstance(old, tuple) and isinstance(v, list):
                v = tuple(v)
            dict.__setitem__(node, keys[-1], v)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    """Install the shims and put /root/reference on sys.path.  Idempotent."""
    global _INSTALLED
    if _INSTALLED:
        return
    if not available():
        raise RuntimeError("reference tree not present at " + REF_ROOT)
    import numpy as np
    import torch

    # (1) torch._six
    if not hasattr(torch, "_six"):
        six = _stub("torch._six", PY3=True, string_classes=(str,), int_classes=(int,))
        torch._six = six
    # (2) yacs
    if "yacs" not in sys.modules:
        yacs = _stub("yacs")
        yacs.config = _stub("yacs.config", CfgNode=CfgNode)
    # (3) apex
    if "apex" not in sys.modules:
        amp = _stub("apex.amp", float_function=lambda f: f, half_function=lambda f: f,
                    init=lambda *a, **k: None, scale_loss=None)
        apex = _stub("apex", amp=amp)
        apex.amp = amp
    # (4) numpy aliases
    for alias, typ in (("float", float), ("int", int), ("bool", bool)):
        if not hasattr(np, alias):
            setattr(np, alias, typ)
    # (5) empty third-party modules only imported at package import time
    for name in ("cv2", "pycocotools", "pycocotools.mask", "pycocotools.coco", "pycocotools.cocoeval",
                 "torchvision", "torchvision.transforms", "torchvision.transforms.functional",
                 "torchvision.datasets", "torchvision.datasets.coco", "cityscapesscripts",
                 "cityscapesscripts.helpers", "cityscapesscripts.helpers.csHelpers",
                 "cityscapesscripts.evaluation", "cityscapesscripts.evaluation.instances2dict",
                 "cityscapesscripts.evaluation.evalInstanceLevelSemanticLabeling"):
        if name not in sys.modules:
            _stub(name)
    tv = sys.modules["torchvision"]
    tv.transforms = sys.modules["torchvision.transforms"]
    tv.transforms.functional = sys.modules["torchvision.transforms.functional"]
    tv.datasets = sys.modules["torchvision.datasets"]
    tv.datasets.coco = sys.modules["torchvision.datasets.coco"]
    tv.datasets.coco.CocoDetection = object
    tv.datasets.CocoDetection = object
    sys.modules["pycocotools"].mask = sys.modules["pycocotools.mask"]
    cs = sys.modules["cityscapesscripts"]
    cs.helpers = sys.modules["cityscapesscripts.helpers"]
    cs.helpers.csHelpers = sys.modules["cityscapesscripts.helpers.csHelpers"]
    # (6) fake nvidia-smi
    bindir = tempfile.mkdtemp(prefix="fake_nvsmi_")
    smi = os.path.join(bindir, "nvidia-smi")
    with open(smi, "w") as f:
        f.write("#!/bin/sh\nexit 0\n")
    os.chmod(smi, os.stat(smi).st_mode | stat.S_IEXEC)
    os.environ["PATH"] = bindir + os.pathsep + os.environ.get("PATH", "")
    # (7) the reference's own CPU native ops as mega_core._C
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import build_ref
    ref_c = build_ref.load()
    if ref_c is None:
        raise RuntimeError("oracle/_ref could not be built")
    sys.path.insert(0, REF_ROOT)
    sys.modules["mega_core._C"] = ref_c
    import mega_core
    mega_core._C = ref_c
    _INSTALLED = True


_PRISTINE_CFG = None


def make_cfg(config_file="configs/MEGA/vid_R_101_C4_MEGA_1x.yaml", opts=()):
    """cfg exactly as tools/test_net.py:75-79 builds it (BASE_RCNN_1gpu.yaml -> file -> opts)."""
    install()
    from mega_core.config import cfg as global_cfg
    cfg = global_cfg  # the reference reads the GLOBAL cfg inside some constructors
    cfg.defrost()
    # the global tree is merged IN PLACE: start every call from the pristine defaults, or values of a config merged
    # earlier in this process would leak into this one (e.g. METHOD "dff" into configs/vid_R_50_C4_1x.yaml)
    global _PRISTINE_CFG
    if _PRISTINE_CFG is None:
        _PRISTINE_CFG = copy.deepcopy(cfg)
    else:
        fresh = copy.deepcopy(_PRISTINE_CFG)
        for k in list(cfg.keys()):
            del cfg[k]
        for k, v in fresh.items():
            cfg[k] = v
    cfg.merge_from_file(os.path.join(REF_ROOT, "configs", "BASE_RCNN_1gpu.yaml"))
    cfg.merge_from_file(os.path.join(REF_ROOT, config_file))
    cfg.merge_from_list(["MODEL.DEVICE", "cpu"] + list(opts))
    return cfg


def build_model(cfg):
    install()
    from mega_core.modeling.detector import build_detection_model
    model = build_detection_model(cfg)
    model.eval()
    return model

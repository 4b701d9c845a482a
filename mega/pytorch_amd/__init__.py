"""mega.pytorch_amd -- MI355X (gfx950) native implementation of MEGA's per-key-frame inference hot path
(reference: Scalsol/mega.pytorch), behind the reference's own module / registry / _C operator API.

  build_detection_model(cfg) -> GeneralizedRCNNMEGA     (modeling.py; reference state_dict keys)
  engine.ClipEngine                                      (batched / frame-sharded clip driver)
  _C.nms, _C.roi_align_forward                           (drop-ins for mega_core._C)
  ops.*                                                  (tensor wrappers over include/mega_hip.h)

There is no CPU fallback: everything raises if libmega_hip.so is missing (build: __graft_entry__.build()).
"""
from .config import get_cfg  # noqa: F401
from .modeling import build_detection_model, GeneralizedRCNNMEGA  # noqa: F401
from .fgfa import GeneralizedRCNNFGFA, GeneralizedRCNNDFF, GeneralizedRCNN  # noqa: F401  (registers the FGFA / single-frame meta-architectures)
from .rdn import GeneralizedRCNNRDN  # noqa: F401  (registers the RDN meta-architecture + feature extractor)

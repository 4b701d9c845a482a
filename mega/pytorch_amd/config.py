"""The config keys MEGA's inference path reads, as an attribute tree with the SAME key
names as the reference's yacs config (mega_core/config/defaults.py + configs/BASE_RCNN_1gpu.yaml +
configs/MEGA/vid_R_{50,101}_C4_MEGA_1x.yaml).  Any object with these attributes (e.g. the reference's
own frozen yacs cfg) can be passed to the builders instead.
"""
import copy


class CfgNode(dict):
    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return copy.deepcopy(self)

    def merge_from_list(self, lst):
        assert len(lst) % 2 == 0
        for key, v in zip(lst[0::2], lst[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                node = node[p]
            if parts[-1] not in node:
                raise KeyError("unknown config key " + key)
            node[parts[-1]] = v
        return self


def get_cfg(arch="R-101", method="mega"):
    """Test-time defaults.  arch: 'R-101' | 'R-50'; method: 'mega' (configs/MEGA/vid_R_{101,50}_C4_MEGA_1x.yaml)
    or 'fgfa' (configs/FGFA/vid_R_{101,50}_C4_FGFA_1x.yaml: GeneralizedRCNNFGFA +
    ResNetConv52MLPFeatureExtractor, no relation attention) or 'base' (configs/vid_R_{50,101}_C4_1x.yaml: the
    single-frame GeneralizedRCNN, BASELINE config 1) or 'rdn' / 'rdn_base' (configs/RDN/vid_R_101_C4_RDN_1x.yaml with
    the advanced stage / vid_R_{50,101}_C4_RDN_base_1x.yaml without)."""
    r50 = arch in ("R-50", "R-50-C4")
    cfg = _mega_cfg(r50)
    if method != "mega":      # what only the MEGA yamls set goes back to config/defaults.py:409,:447
        cfg.MODEL.VID.ROI_BOX_HEAD.ATTENTION.STAGE = 2
        cfg.MODEL.VID.MEGA.GLOBAL.RES_STAGE = 1
    if method == "fgfa":
        cfg.MODEL.META_ARCHITECTURE = "GeneralizedRCNNFGFA"
        cfg.MODEL.VID.METHOD = "fgfa"
        cfg.MODEL.ROI_BOX_HEAD.FEATURE_EXTRACTOR = "ResNetConv52MLPFeatureExtractor"
        cfg.MODEL.VID.ROI_BOX_HEAD.ATTENTION.ENABLE = False
    elif method in ("rdn", "rdn_base"):
        cfg.MODEL.META_ARCHITECTURE = "GeneralizedRCNNRDN"
        cfg.MODEL.VID.METHOD = "rdn"
        cfg.MODEL.ROI_BOX_HEAD.FEATURE_EXTRACTOR = "RDNFeatureExtractor"
        cfg.MODEL.VID.ROI_BOX_HEAD.ATTENTION.STAGE = 2                    # defaults.py:409
        cfg.MODEL.VID.ROI_BOX_HEAD.ATTENTION.ADVANCED_STAGE = 1 if method == "rdn" else 0
    elif method == "dff":
        cfg.MODEL.META_ARCHITECTURE = "GeneralizedRCNNDFF"
        cfg.MODEL.VID.METHOD = "dff"
        cfg.MODEL.ROI_BOX_HEAD.FEATURE_EXTRACTOR = "ResNetConv52MLPFeatureExtractor"
        cfg.MODEL.VID.ROI_BOX_HEAD.ATTENTION.ENABLE = False
    elif method == "base":
        cfg.MODEL.META_ARCHITECTURE = "GeneralizedRCNN"
        cfg.MODEL.VID.METHOD = "base"
        cfg.MODEL.ROI_BOX_HEAD.FEATURE_EXTRACTOR = "ResNetConv52MLPFeatureExtractor"
        cfg.MODEL.VID.ROI_BOX_HEAD.ATTENTION.ENABLE = False
    elif method != "mega":
        raise ValueError("method must be 'mega', 'fgfa', 'dff', 'rdn', 'rdn_base' or 'base'")
    return cfg


def _mega_cfg(r50):
    return CfgNode({
        "DTYPE": "float32",                      # defaults.py:541; "bfloat16" selects the bf16 MFMA path, "float16" the same
                                                 # kernels on IEEE-half operands (11 significant bits, values < 65 504)
        "HEAD_DTYPE": "bfloat16",                # float16 mode only: operand type of the aggregation head, "bfloat16" | "float16" (modeling.head_dtype)
        # float16 mode only: "single" = one fp16 MFMA pass per product; "x2" = the two-pass form (modeling.conv_mode "h2"):
        # activations as float16 [hi | lo] planes against weights rounded to fp16 once, K x 2
        "F16_CONV": "single",
        # bf16 mode only: dtype of the aggregation head's activation stream (fc0 output -> x + attention -> stage FCs ->
        # predictor, roi_box_feature_extractors.py:806-829,:898-933).  "float32" (default): the stream is never rounded
        # to bf16 and the stage FCs / predictor run in exact-f32 MFMA (logits within ~1e-4 of the f32 path);
        # "bfloat16": every hand-off rounds (the round-3 behaviour, ~2 % faster, logit error median 1.7e-3)
        "HEAD_STREAM": "float32",
        # float32 mode only: "exact" = exact-f32 MFMA convolutions (157 TF/s roof); "bf16x3" = split-precision frame stage:
        # activations as [hi | lo] bf16 planes, every conv / fc0 as ONE bf16 matrix-core GEMM over K x 3
        # ([hi | lo | hi] . [Wh | Wh | Wl], f32 accumulation: x.W to ~2^-16) -- the parity mode at matrix-core rates
        "F32_CONV": "exact",
        # with F32_CONV "bf16x3": "auto" = the aggregation head's linear layers (Wq / Wk / Wv projections, stage FCs) run
        # in split precision too (attention core, position logits, predictor exact f32); "exact" = the whole head exact f32
        "F32_HEAD_LINEAR": "auto",
        # bfloat16 mode only: "bfloat16" = the residual trunk is a bf16 tensor (rounded at each of the 36 `out += identity`,
        # resnet.py:324-344); "planes" = the trunk is carried as [hi | lo] planes and added in f32 (modeling.conv_mode "wide")
        "RESIDUAL_STREAM": "bfloat16",
        "INPUT": {"MIN_SIZE_TEST": 600, "MAX_SIZE_TEST": 1000,
                  "PIXEL_MEAN": [102.9801, 115.9465, 122.7717], "PIXEL_STD": [1.0, 1.0, 1.0], "TO_BGR255": True},
        "MODEL": {
            "DEVICE": "cuda",
            "META_ARCHITECTURE": "GeneralizedRCNNMEGA",
            "BACKBONE": {"CONV_BODY": "R-50-C4" if r50 else "R-101-C4"},
            "RESNETS": {"NUM_GROUPS": 1, "WIDTH_PER_GROUP": 64, "STRIDE_IN_1X1": True,
                        "TRANS_FUNC": "BottleneckWithFixedBatchNorm", "STEM_FUNC": "StemWithFixedBatchNorm",
                        "RES5_DILATION": 2, "BACKBONE_OUT_CHANNELS": 256 * 4, "RES2_OUT_CHANNELS": 256,
                        "STEM_OUT_CHANNELS": 64},
            "RPN": {"RPN_HEAD": "SingleConvRPNHead", "ANCHOR_SIZES": (64, 128, 256, 512), "ANCHOR_STRIDE": (16,),
                    "ASPECT_RATIOS": (0.5, 1.0, 2.0), "STRADDLE_THRESH": 0, "PRE_NMS_TOP_N_TEST": 6000,
                    "POST_NMS_TOP_N_TEST": 300, "NMS_THRESH": 0.7, "MIN_SIZE": 0},
            "ROI_HEADS": {"SCORE_THRESH": 0.001, "NMS": 0.5, "DETECTIONS_PER_IMG": 300,
                          "BBOX_REG_WEIGHTS": (10.0, 10.0, 5.0, 5.0)},
            "ROI_BOX_HEAD": {"FEATURE_EXTRACTOR": "MEGAFeatureExtractor", "PREDICTOR": "FPNPredictor",
                             "POOLER_RESOLUTION": 7, "POOLER_SAMPLING_RATIO": 0, "POOLER_SCALES": (1.0 / 16,),
                             "NUM_CLASSES": 31, "MLP_HEAD_DIM": 1024},
            "VID": {
                "ENABLE": True, "METHOD": "mega",
                "RPN": {"REF_PRE_NMS_TOP_N": 6000, "REF_POST_NMS_TOP_N": 75},
                "ROI_BOX_HEAD": {"REDUCE_CHANNEL": bool(r50),
                                 "ATTENTION": {"ENABLE": True, "STAGE": 3, "ADVANCED_STAGE": 0, "GROUP": 16,
                                               "EMBED_DIM": 64}},
                "DFF": {"MIN_OFFSET": -9, "MAX_OFFSET": 0},
                "RDN": {"MIN_OFFSET": -18, "MAX_OFFSET": 18, "ALL_FRAME_INTERVAL": 37, "KEY_FRAME_LOCATION": 18,
                        "RATIO": 0.2},
                "FGFA": {"MIN_OFFSET": -9, "MAX_OFFSET": 9, "ALL_FRAME_INTERVAL": 19, "KEY_FRAME_LOCATION": 9},
                "MEGA": {"MIN_OFFSET": -12, "MAX_OFFSET": 12, "ALL_FRAME_INTERVAL": 25, "KEY_FRAME_LOCATION": 12,
                         "RATIO": 0.2,
                         "MEMORY": {"ENABLE": True, "SIZE": 25},
                         "GLOBAL": {"ENABLE": True, "SIZE": 10, "RES_STAGE": 0 if r50 else 1, "SHUFFLE": True}},
            },
        },
        # kernel-side switch: True = IoU > thr (the CUDA path the reference runs on GPU, csrc/cuda/nms.cu:60),
        # False = IoU >= thr (its CPU path, csrc/cpu/nms_cpu.cpp:60)
        "NMS_STRICT_GT": True,
    })

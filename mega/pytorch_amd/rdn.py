"""RDN (Relation Distillation Networks) inference path -- SURVEY.md 8f row 3, the second model family on the same
kernels:
  mega_core/modeling/detector/generalized_rcnn_rdn.py:108-190                      GeneralizedRCNNRDN._forward_test
  mega_core/modeling/roi_heads/box_head/roi_box_feature_extractors.py:178-238      AttentionExtractor (no `u` term)
  mega_core/modeling/roi_heads/box_head/roi_box_feature_extractors.py:253-454      RDNFeatureExtractor

A 37-frame window (key = slot 18), 75 reference proposals per frame; `base` stages attend the key frame's 300
proposals to all 2775 reference proposals, the optional `advanced` stage first distils the top-15 proposals of every
frame against the full set and then attends the key proposals to those.  The per-frame stage (backbone, RPN, res5,
ROIAlign, fcs[0]) is the same batched, graph-captured frame stage as MEGA's, so ClipEngine drives this detector too.
"""
from collections import deque

import torch
from torch import nn

from . import ops
from .modeling import (DETECTION_META_ARCHITECTURES, ROI_BOX_FEATURE_EXTRACTORS, GeneralizedRCNNMEGA, ResNetHead, _Packed,
                       _nhwc, _pack_conv, convert_to_roi_format)
from .relation import RelationWeights, relation_attention_forward
from .structures import BoxList


class RDNFeatureExtractor(_Packed):
    """roi_box_feature_extractors.py:253-454, test-time path (_forward_ref :404, _forward_test :416)."""

    def __init__(self, cfg, in_channels):
        super().__init__()
        rb, vid = cfg.MODEL.ROI_BOX_HEAD, cfg.MODEL.VID
        self.head = ResNetHead(cfg.MODEL.RESNETS.RES5_DILATION)
        self.pooled_c = 2048
        self.conv = None
        if vid.ROI_BOX_HEAD.REDUCE_CHANNEL:
            self.conv = nn.Conv2d(2048, 256, 1)
            self.pooled_c = 256
        self.resolution, self.scale, self.sampling_ratio = rb.POOLER_RESOLUTION, rb.POOLER_SCALES[0], rb.POOLER_SAMPLING_RATIO
        rep = rb.MLP_HEAD_DIM
        att = vid.ROI_BOX_HEAD.ATTENTION
        assert att.ENABLE and att.GROUP == 16 and att.EMBED_DIM == 64 and rep == 1024, "kernels are built for 16x64 heads"
        self.base_stage, self.advanced_stage = att.STAGE, att.ADVANCED_STAGE
        self.base_num = vid.RPN.REF_POST_NMS_TOP_N
        self.advanced_num = int(self.base_num * vid.RDN.RATIO)
        n_fc = self.base_stage + self.advanced_stage                                   # :311-319
        self.n_att = self.base_stage + self.advanced_stage + 1 if self.advanced_stage > 0 else self.base_stage
        in0 = self.pooled_c * self.resolution ** 2
        self.fcs = nn.ModuleList([nn.Linear(in0 if i == 0 else rep, rep) for i in range(n_fc)])
        self.Wgs = nn.ModuleList([nn.Conv2d(att.EMBED_DIM, att.GROUP, 1) for _ in range(self.n_att)])
        self.Wqs = nn.ModuleList([nn.Linear(rep, rep) for _ in range(self.n_att)])
        self.Wks = nn.ModuleList([nn.Linear(rep, rep) for _ in range(self.n_att)])
        self.Wvs = nn.ModuleList([nn.Conv2d(rep * att.GROUP, rep, 1, groups=att.GROUP) for _ in range(self.n_att)])
        self.out_channels = rep

    def _pack(self, dtype, device):
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        pk = {"att": [RelationWeights(sd, "", "", i, dtype, device, with_pos=True) for i in range(self.n_att)]}
        w0 = self.fcs[0].weight.detach()              # fc0 eats the bin-major [K,49,C] ROIAlign output
        r2 = self.resolution ** 2
        w0 = w0.view(w0.shape[0], self.pooled_c, r2).permute(0, 2, 1).reshape(w0.shape[0], -1)
        pk["fc_w"] = [w0.contiguous().to(dtype).to(device)] + [m.weight.detach().to(dtype).to(device).contiguous()
                                                                for m in list(self.fcs)[1:]]
        pk["fc_b"] = [m.bias.detach().float().to(device).contiguous() for m in self.fcs]
        if self.conv is not None:
            pk["rc_w"] = _pack_conv(self.conv, dtype).to(device)
            pk["rc_b"] = self.conv.bias.detach().float().to(device).contiguous()
        return pk

    def box_features(self, feat_nhwc, rois5):
        """res5 (+1x1 reduce) -> ROIAlign -> fcs[0] + ReLU (:404-414; for the key frame :421-433 with i = 0)."""
        return self.pooled_fc(self.res5_features(feat_nhwc), rois5)

    def res5_features(self, feat_nhwc):
        """the proposal-independent half of box_features (the engine runs it beside the RPN branch)"""
        pk = self._packed(feat_nhwc.dtype, feat_nhwc.device)
        x = self.head.run(feat_nhwc)
        if self.conv is not None:
            x = ops.conv2d_nhwc(x, pk["rc_w"], None, pk["rc_b"], relu=True)
        return x

    def pooled_fc(self, x5, rois5):
        pk = self._packed(x5.dtype, x5.device)
        pooled = ops.roi_align(x5, rois5, self.scale, (self.resolution, self.resolution), self.sampling_ratio)
        return ops.linear(pooled.view(pooled.shape[0], -1), pk["fc_w"][0], pk["fc_b"][0], relu=True)

    # the MEGA detector's reset hooks: RDN keeps no memory / global pools
    def init_memory(self):
        pass

    def init_global(self):
        pass

    def aggregate(self, x, rois_key, rois_ref, x_refs, adv_index=None):
        """x [nk,1024] = relu(fcs[0](pooled key proposals)); x_refs [Nr,1024], rois_ref [Nr,4] the whole window;
        adv_index [Na]: rows of the window forming the top-`advanced_num` set of every frame (:437-439)."""
        pk = self._packed(x.dtype, x.device)
        rois_key, rois_ref, x_refs = rois_key.contiguous(), rois_ref.contiguous(), x_refs.contiguous()
        for i in range(self.base_stage):                                                   # :431-436
            if i > 0:
                x = ops.linear(x, pk["fc_w"][i], pk["fc_b"][i], relu=True)
            x = relation_attention_forward(pk["att"][i], x.contiguous(), x_refs, rois_key, rois_ref, residual=True)
        if self.advanced_stage > 0:                                                        # :438-452
            x_adv = x_refs.index_select(0, adv_index)
            rois_adv = rois_ref.index_select(0, adv_index).contiguous()
            for i in range(self.advanced_stage):
                j = i + self.base_stage
                x_adv = relation_attention_forward(pk["att"][j], x_adv.contiguous(), x_refs, rois_adv, rois_ref, residual=True)
                x_adv = ops.linear(x_adv, pk["fc_w"][j], pk["fc_b"][j], relu=True)
            x = relation_attention_forward(pk["att"][self.n_att - 1], x.contiguous(), x_adv.contiguous(), rois_key,
                                           rois_adv, residual=True)
        return x

    def forward(self, x, proposals, pre_calculate=False, key_features=None):
        if self.training:
            raise NotImplementedError("inference path only")
        if pre_calculate:
            return self.box_features(_nhwc(x), convert_to_roi_format(proposals))
        props, proposals_ref, x_refs = proposals
        xk = key_features if key_features is not None else self.box_features(_nhwc(x), convert_to_roi_format(props))
        bn, an = self.base_num, self.advanced_num
        nref = proposals_ref.bbox.shape[0]
        adv = torch.cat([torch.arange(o, min(o + an, nref)) for o in range(0, nref, bn)]).to(x_refs.device)
        return self.aggregate(xk, props[0].bbox, proposals_ref.bbox, x_refs, adv)


ROI_BOX_FEATURE_EXTRACTORS.register("RDNFeatureExtractor", RDNFeatureExtractor)


class GeneralizedRCNNRDN(GeneralizedRCNNMEGA):
    """detector/generalized_rcnn_rdn.py:21-190, inference.  Re-uses the MEGA detector's frame stage, record window
    and reference call convention (images["ref"] instead of images["ref_l"]; extension "ref_init")."""
    _ref_key = "ref"

    def __init__(self, cfg):
        nn.Module.__init__(self)
        from .modeling import build_backbone, build_roi_heads, build_rpn
        self.cfg = cfg
        self.device = torch.device(cfg.MODEL.DEVICE)
        self.backbone = build_backbone(cfg)
        self.rpn = build_rpn(cfg, self.backbone.out_channels)
        self.roi_heads = build_roi_heads(cfg, self.backbone.out_channels)
        rdn = cfg.MODEL.VID.RDN
        self.memory_enable = self.global_enable = False
        self.base_num = cfg.MODEL.VID.RPN.REF_POST_NMS_TOP_N
        self.advanced_num = int(self.base_num * rdn.RATIO)
        self.all_frame_interval, self.key_frame_location = rdn.ALL_FRAME_INTERVAL, rdn.KEY_FRAME_LOCATION
        self.key_num = cfg.MODEL.RPN.POST_NMS_TOP_N_TEST
        assert cfg.MODEL.VID.RPN.REF_PRE_NMS_TOP_N == cfg.MODEL.RPN.PRE_NMS_TOP_N_TEST and self.base_num <= self.key_num
        self.eval()

    def _reset(self, seg_len):
        self.seg_len, self.end_id = seg_len, 0
        self.records = deque(maxlen=self.all_frame_interval)
        self._adv_cache = None

    @torch.no_grad()
    def step(self, new_local=None, new_globals=(), im_size=None, defer=False):
        fe = self.roi_heads.box.feature_extractor
        if new_local is not None:
            self.records.append(new_local)
        key = self.records[self.key_frame_location]
        bn, an = self.base_num, self.advanced_num
        ns = tuple(min(bn, r["boxes"].shape[0]) for r in self.records)
        rois = torch.cat([r["boxes"][:n] for r, n in zip(self.records, ns)], 0)          # :176-177
        feats = torch.cat([r["feats"][:n] for r, n in zip(self.records, ns)], 0)
        adv = None
        if fe.advanced_stage > 0:
            # the reference splits the concatenated window in chunks of base_num (:438): identical to per-frame
            # top-k only while every frame yields base_num proposals; reproduce the chunking literally
            if self._adv_cache is None or self._adv_cache[0] != sum(ns):
                nref = sum(ns)
                idx = torch.cat([torch.arange(o, min(o + an, nref)) for o in range(0, nref, bn)]).to(feats.device)
                self._adv_cache = (nref, idx)
            adv = self._adv_cache[1]
        x = fe.aggregate(key["feats"], key["boxes"], rois, feats, adv)
        logits, deltas = self.roi_heads.box.predictor(x)
        kb = BoxList(key["boxes"], im_size, "xyxy")
        kb.add_field("objectness", key["scores"])
        pp = self.roi_heads.box.post_processor
        if defer:
            return pp.run((logits, deltas), kb)
        return pp((logits, deltas), [kb])[0]


DETECTION_META_ARCHITECTURES.register("GeneralizedRCNNRDN", GeneralizedRCNNRDN)

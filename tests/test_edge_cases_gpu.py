"""-m gpu: the edge cases the domain has -- empty and degenerate inputs, all-suppressed / nothing-suppressed NMS,
no detections, more detections than the cap, ragged sizes -- each against the oracle."""
import numpy as np
import pytest
import torch

from oracle import mega_oracle as mo
from oracle import native

pytestmark = pytest.mark.gpu


def _ops():
    from mega.pytorch_amd import ops
    return ops


def test_empty_inputs(dev):
    ops = _ops()
    feat = torch.randn((1, 9, 11, 64), device=dev)
    assert ops.roi_align(feat, torch.zeros((0, 5), device=dev), 1 / 16, (7, 7), 0).shape == (0, 49, 64)
    assert ops.nms(torch.zeros((0, 4), device=dev), torch.zeros((0,), device=dev), 0.5).numel() == 0


@pytest.mark.parametrize("strict", [True, False])
def test_nms_degenerate_sets(dev, strict):
    """identical boxes (all but the best suppressed), disjoint boxes (none suppressed), IoU exactly at the threshold
    (the `>` vs `>=` switch decides), thresholds 0 and 1."""
    ops = _ops()
    same = torch.tensor([[10., 10., 50., 50.]]).repeat(70, 1)
    sc = torch.linspace(0.1, 0.9, 70)
    assert ops.nms(same.to(dev), sc.to(dev), 0.5, strict).cpu().tolist() == [69]
    grid = torch.tensor([[x * 20., y * 20., x * 20. + 9, y * 20. + 9] for x in range(12) for y in range(9)])
    sg = torch.rand(108, generator=torch.Generator().manual_seed(1))
    assert ops.nms(grid.to(dev), sg.to(dev), 0.0 if strict else 1e-9, strict).cpu().tolist() == list(range(108))
    # two boxes with IoU exactly 0.5 under the +1 convention: areas 100 and 50 -> inter 50, union 100
    pair = torch.tensor([[0., 0., 9., 9.], [0., 0., 9., 4.]])
    ps = torch.tensor([0.9, 0.8])
    want = native.nms(pair, ps, 0.5, strict).tolist()
    assert ops.nms(pair.to(dev), ps.to(dev), 0.5, strict).cpu().tolist() == want
    assert want == ([0, 1] if strict else [0])
    rnd = torch.rand((500, 4), generator=torch.Generator().manual_seed(2)) * 300
    rnd[:, 2:] += rnd[:, :2]
    rs = torch.rand(500, generator=torch.Generator().manual_seed(3))
    for thr in (0.0, 1.0):
        assert ops.nms(rnd.to(dev), rs.to(dev), thr, strict).cpu().tolist() == native.nms(rnd, rs, thr, strict).tolist()


def test_postprocess_no_detections_and_over_the_cap(dev):
    ops = _ops()
    cfg = mo.OracleCfg()
    R, NC = 300, 31
    g = torch.Generator().manual_seed(0)
    ctr = torch.rand((R, 2), generator=g) * torch.tensor([900., 500.])
    wh = torch.rand((R, 2), generator=g) * 200 + 10
    props = torch.cat([ctr - wh / 2, ctr + wh / 2], dim=1).clamp(min=0)
    props[:, 2].clamp_(max=999); props[:, 3].clamp_(max=599)
    deltas = torch.randn((R, NC * 4), generator=g) * 0.1
    # (a) background wins everywhere by a wide margin: every class score < SCORE_THRESH -> zero detections
    logits = torch.zeros((R, NC)); logits[:, 0] = 20.0
    wb, _, _ = mo.postprocess(logits, deltas, props, 1000, 600, cfg)
    ob, os_, ol, oc = ops.postprocess(logits.to(dev), deltas.to(dev), props.to(dev), None, cfg.bbox_reg_weights, 1000, 600,
                                      cfg.score_thresh, cfg.nms, cfg.detections_per_img, True)
    assert wb.shape[0] == 0 and int(oc.item()) == 0
    # (b) flat scores over disjoint-ish boxes: far more than DETECTIONS_PER_IMG survive NMS -> k-th value cut, ties kept
    logits = torch.randn((R, NC), generator=g) * 0.05
    wb, ws, wl = mo.postprocess(logits, deltas, props, 1000, 600, cfg)
    ob, os_, ol, oc = ops.postprocess(logits.to(dev), deltas.to(dev), props.to(dev), None, cfg.bbox_reg_weights, 1000, 600,
                                      cfg.score_thresh, cfg.nms, cfg.detections_per_img, True)
    n = int(oc.item())
    assert n == wb.shape[0] >= cfg.detections_per_img
    assert torch.equal(ol[:n].cpu(), wl) and (os_[:n].cpu() - ws).abs().max() < 1e-6


def test_rpn_select_fewer_anchors_than_topn_and_tiny_maps(dev):
    """feature maps so small that anchors < PRE_NMS_TOP_N and kept < POST_NMS_TOP_N (ragged proposal counts)."""
    from mega.pytorch_amd import synth
    ops = _ops()
    cell = synth._cell_anchors(16, (64, 128, 256, 512), (0.5, 1.0, 2.0))
    for (Hf, Wf) in [(1, 1), (2, 3), (5, 7)]:
        g = torch.Generator().manual_seed(Hf * 10 + Wf)
        A = 12
        obj = torch.randn((1, A, Hf, Wf), generator=g)
        reg = torch.randn((1, 4 * A, Hf, Wf), generator=g) * 0.3
        anchors = mo.grid_anchors(cell, Hf, Wf, 16)
        want, wsc = mo.rpn_select(obj[0], reg[0], anchors, Wf * 16, Hf * 16, 6000, 300, 0.7, 0, True)
        packed = torch.cat([obj, reg], dim=1).permute(0, 2, 3, 1).contiguous()      # [1,Hf,Wf,60] f32, as RPNHead emits
        props, scores, cnt = ops.rpn_select(packed.to(dev), cell.to(dev), Hf, Wf, 16, 6000, 300, 0.7, 0, Wf * 16, Hf * 16, True)
        n = int(cnt[0].item())
        assert n == want.shape[0] and n <= Hf * Wf * A
        assert (props[0, :n].cpu() - want).abs().max() < 1e-3


def test_relation_attention_single_query_and_ragged_keys(dev):
    ops = _ops()
    import cpu_ops
    g = torch.Generator().manual_seed(4)
    for (Nq, Nk) in [(1, 5), (3, 31), (33, 97)]:
        q = torch.randn((Nq, 1024), generator=g) * 0.3
        k = torch.randn((Nk, 1024), generator=g) * 0.3
        v = torch.randn((Nk, 1024), generator=g)
        ld = (Nk + 31) // 32 * 32
        vt = torch.zeros((1024, ld)); vt[:, :Nk] = v.t()
        ref = cpu_ops.relation_attention(q, k, vt, Nk)
        got = ops.relation_attention(q.to(dev), k.to(dev), vt.to(dev), Nk).cpu()
        assert (got - ref).abs().max() < 2e-4 * max(1.0, ref.abs().max().item())

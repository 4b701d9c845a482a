"""Host side of one relation-attention call (mirror of
MEGAFeatureExtractor.attention_module_multi_head, mega_core/modeling/roi_heads/box_head/
roi_box_feature_extractors.py:567-646) on the HIP kernels.

Algebraic re-arrangements (results identical up to f32 reassociation, verified against the literal
formula in tests/test_kernels_gpu.py::test_relation_attention):
  * aff_c = u . k (:619-624) is folded into the query:  (q + u) . k  ->  u is added to the Wq bias once;
  * the grouped 1x1 conv Wv (:642) is applied to V before the PV contraction (softmax rows sum to 1, so the
    conv bias passes through): V' = ref @ Wv^T is one [Nk,1024]x[1024,1024] GEMM instead of a
    [Nq*16,1024]x[1024,64] per-group contraction on a materialised [Nq,16,1024] tensor;
  * the [1,64,Nq,Nk] position embedding (:240-250) is never materialised (ops.position_logits).
"""
import torch

from . import ops


class RelationWeights(object):
    """Kernel-ready weights of one (kind, index) attention: kind 'l_' (local/memory, with position term),
    'g_' (global) or '' (RDN: position term, no u)."""

    def __init__(self, sd, pfx, kind, index, dtype, device, with_pos):
        def g(name):
            return sd["%s%s%s.%d%s" % (pfx, kind, name[0], index, name[1])].detach().float()
        ukey = "%s%sus.%d" % (pfx, kind, index)        # RDN's AttentionExtractor (kind '') has no u term
        u = sd[ukey].detach().float().reshape(-1) if ukey in sd else 0.0             # [16,1,64] -> [1024]
        self.wq = g(("Wqs", ".weight")).to(device=device, dtype=dtype).contiguous()
        self.bq = (g(("Wqs", ".bias")) + u).to(device).contiguous()                   # u folded into the bias
        self.wk = g(("Wks", ".weight")).to(device=device, dtype=dtype).contiguous()
        self.bk = g(("Wks", ".bias")).to(device).contiguous()
        self.wv = g(("Wvs", ".weight")).reshape(1024, 1024).to(device=device, dtype=dtype).contiguous()
        self.bv = g(("Wvs", ".bias")).to(device).contiguous()
        self.with_pos = with_pos
        if with_pos:
            wg = sd["%s%sWgs.%d.weight" % (pfx, kind, index)].detach().float().reshape(16, 64)
            self.wg_t = wg.t().contiguous().to(device)                                 # [64,16]
            self.bg = sd["%s%sWgs.%d.bias" % (pfx, kind, index)].detach().float().to(device).contiguous()
            feat_range = torch.arange(0, 64 / 8)
            self.dim_mat = torch.full((len(feat_range),), 1000.0).pow(8.0 / 64 * feat_range).to(device)


def relation_attention_forward(w, x, ref, rois_q=None, rois_k=None, residual=True, mem_kv=None, return_kv=False):
    """x [Nq,1024] queries, ref [Nr,1024] keys/values (both dtype of w), rois_* [N,4] f32 when w.with_pos.
    Returns x + attention(x, ref) (residual=True, as every call site of the reference does,
    roi_box_feature_extractors.py:697,:824) or the bare attention output.

    mem_kv = (k_mem [Nm,1024], vt_mem [1024,Nm]): already-projected keys / values of FURTHER reference rows that
    follow `ref` in key order (the memory pool: its features never change once pushed, so their Wk / Wv projections
    are computed once, when the rows were part of `ref`, instead of every step).  rois_k then covers Nr + Nm rows.
    return_kv: also return this call's (k [Nr,1024], vt [1024,ld]) of `ref` so the caller can keep slices of them."""
    Nr = ref.shape[0]
    k = ops.linear(ref, w.wk, w.bk)
    ldr = (Nr + 31) // 32 * 32
    vt = ops.linear_transposed(w.wv, ref, ldr)
    k_all, vt_all, Nk = k, vt, Nr
    if mem_kv is not None:
        k_mem, vt_mem = mem_kv
        Nk = Nr + k_mem.shape[0]
        ldv = (Nk + 31) // 32 * 32
        k_all = torch.cat([k, k_mem], dim=0)
        parts = [vt[:, :Nr], vt_mem]
        if ldv > Nk:
            parts.append(vt.new_zeros((vt.shape[0], ldv - Nk)))
        vt_all = torch.cat(parts, dim=1)
    q = ops.linear(x, w.wq, w.bq)
    pos = None
    if w.with_pos:
        fast = x.dtype != torch.float32      # bf16 mode: matrix-core kernel, bf16 logits in the attention's tile order
        pos = ops.position_logits(rois_q, rois_k, w.wg_t, w.bg, w.dim_mat, precise=not fast, tiled=fast)
    out = ops.relation_attention(q, k_all, vt_all, Nk, pos=pos, resid=x if residual else None, bias_v=w.bv)
    return (out, k, vt) if return_kv else out

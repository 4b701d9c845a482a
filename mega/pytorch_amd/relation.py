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

    def __init__(self, sd, pfx, kind, index, dtype, device, with_pos, split_v=False, x3=False):
        def g(name):
            return sd["%s%s%s.%d%s" % (pfx, kind, name[0], index, name[1])].detach().float()
        ukey = "%s%sus.%d" % (pfx, kind, index)        # RDN's AttentionExtractor (kind '') has no u term
        u = sd[ukey].detach().float().reshape(-1) if ukey in sd else 0.0             # [16,1,64] -> [1024]
        self.wq = g(("Wqs", ".weight")).to(device=device, dtype=dtype).contiguous()
        self.bq = (g(("Wqs", ".bias")) + u).to(device).contiguous()                   # u folded into the bias
        self.wk = g(("Wks", ".weight")).to(device=device, dtype=dtype).contiguous()
        self.bk = g(("Wks", ".bias")).to(device).contiguous()
        wv32 = g(("Wvs", ".weight")).reshape(1024, 1024)
        self.wv = wv32.to(device=device, dtype=dtype).contiguous()
        # split_v (bf16 mode with an f32 head stream): the part of Wv that bf16 drops, as a second bf16 matrix.  The
        # rounding error of Wv is the SAME for every key, and the values share a large common part (bias + mean of the
        # post-ReLU features), so its effect on sum_j p_j v_j does not average out over the keys the way the per-key
        # roundings of ref / V do: 3.3e-4 of the bf16 head's 6.5e-4 median logit error (tools/head_precision_cpu.py).
        self.wv_lo = None
        if split_v and dtype != torch.float32:
            self.wv_lo = (wv32 - self.wv.float().to(wv32.device)).to(device=device, dtype=dtype).contiguous()
        self.bv = g(("Wvs", ".bias")).to(device).contiguous()
        if x3:      # conv_mode "x3": the three projections as split-precision GEMMs (ops.X3Weight); operands stay f32
            assert dtype == torch.float32
            self.wq = ops.X3Weight(g(("Wqs", ".weight")), device)
            self.wk = ops.X3Weight(g(("Wks", ".weight")), device)
            self.wv = ops.X3Weight(wv32, device)
        self.with_pos = with_pos
        if with_pos:
            wg = sd["%s%sWgs.%d.weight" % (pfx, kind, index)].detach().float().reshape(16, 64)
            self.wg_t = wg.t().contiguous().to(device)                                 # [64,16]
            self.bg = sd["%s%sWgs.%d.bias" % (pfx, kind, index)].detach().float().to(device).contiguous()
            feat_range = torch.arange(0, 64 / 8)
            self.dim_mat = torch.full((len(feat_range),), 1000.0).pow(8.0 / 64 * feat_range).to(device)


def op_dtype(w, t):
    """t as a GEMM operand of w's projections: the bf16 mode with an f32 activation stream (cfg.HEAD_STREAM) hands the
    matrix cores a ROUNDED COPY of the stream (ops.cast_bf16) -- the stream itself, the attention's residual, stays f32."""
    return t if t.dtype == w.wq.dtype else ops.cast_half(t.contiguous(), w.wq.dtype)


def project_v(w, ref, ld):
    """V'^T = (ref @ Wv^T)^T [1024, ld] (key-contiguous, pad columns zero).  With split weights (w.wv_lo) two passes:
    the small term Wv_lo . ref first, then Wv_hi . ref with the first pass as the epilogue's residual -- the value
    that is finally rounded to bf16 is then Wv . ref to ~2^-16 instead of (Wv + dW) . ref with a 2^-9 dW common to all
    keys."""
    if getattr(w, "wv_lo", None) is None:
        return ops.linear_transposed(w.wv, ref, ld)
    return ops.linear_transposed(w.wv, ref, ld, residual=ops.linear_transposed(w.wv_lo, ref, ld))


def relation_attention_forward(w, x, ref, rois_q=None, rois_k=None, residual=True, mem_kv=None, return_kv=False):
    """x [Nq,1024] queries, ref [Nr,1024] keys/values (both dtype of w), rois_* [N,4] f32 when w.with_pos.
    Returns x + attention(x, ref) (residual=True, as every call site of the reference does,
    roi_box_feature_extractors.py:697,:824) or the bare attention output.

    mem_kv = (k_mem [Nm,1024], vt_mem [1024,Nm]): already-projected keys / values of FURTHER reference rows that
    follow `ref` in key order (the memory pool: its features never change once pushed, so their Wk / Wv projections
    are computed once, when the rows were part of `ref`, instead of every step).  rois_k then covers Nr + Nm rows.
    return_kv: also return this call's (k [Nr,1024], vt [1024,ld]) of `ref` so the caller can keep slices of them."""
    Nr = ref.shape[0]
    ref = op_dtype(w, ref)
    k = ops.linear(ref, w.wk, w.bk)
    ldr = (Nr + 31) // 32 * 32
    vt = project_v(w, ref, ldr)
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
    q = ops.linear(op_dtype(w, x), w.wq, w.bq)
    pos = None
    if w.with_pos:
        fast = w.wq.dtype != torch.float32   # bf16 mode: matrix-core kernel, bf16 logits in the attention's tile order
        pos = ops.position_logits(rois_q, rois_k, w.wg_t, w.bg, w.dim_mat, precise=not fast, tiled=w.wq.dtype if fast else False)
    out = ops.relation_attention(q, k_all, vt_all, Nk, pos=pos, resid=x if residual else None, bias_v=w.bv)
    return (out, k, vt) if return_kv else out


def cat_rows(ts):
    """torch.cat(ts, 0) for 2-D row blocks -- without a copy when the pieces already ARE consecutive row blocks of one
    buffer (slices of a batched GEMM / attention output handed back in order)."""
    v = _rows_view(ts)
    if v is not None:
        return v
    ts = [t for t in ts if t.shape[0] > 0] or list(ts[:1])
    return torch.cat(ts, dim=0)


def _rows_view(ts):
    """the free case of cat_rows: consecutive row blocks of one buffer -> the view, else None"""
    ts = [t for t in ts if t.shape[0] > 0] or list(ts[:1])
    if len(ts) == 1:
        return ts[0]
    t0 = ts[0]
    if t0.dim() == 2 and all(t.dim() == 2 and t.is_contiguous() and t.dtype == t0.dtype and t.device == t0.device
                             and t.shape[1] == t0.shape[1] for t in ts):
        base, ptr, es = t0.untyped_storage().data_ptr(), t0.data_ptr(), t0.element_size()
        for t in ts:
            if t.data_ptr() != ptr or t.untyped_storage().data_ptr() != base:
                return None
            ptr += t.numel() * es
        rows = sum(t.shape[0] for t in ts)
        return t0.as_strided((rows, t0.shape[1]), (t0.shape[1], 1), t0.storage_offset())
    return None


def cat_rows_many(lists):
    """cat_rows for several independent lists of row blocks: the lists that really need a copy are concatenated by ONE
    launch (ops.multi_cat) instead of one torch.cat each."""
    out = [_rows_view(ts) for ts in lists]
    todo = [i for i, v in enumerate(out) if v is None]
    if todo:
        res = ops.multi_cat([([t for t in lists[i] if t.shape[0] > 0], 0) for i in todo])
        for i, r in zip(todo, res):
            out[i] = r
    return out


def _flat(xs):
    """problems given as a tensor or as a tuple of row blocks -> (flat list of blocks, rows per problem)"""
    flat, rows = [], []
    for x in xs:
        parts = list(x) if isinstance(x, (tuple, list)) else [x]
        flat += parts
        rows.append(sum(p.shape[0] for p in parts))
    return flat, rows


_ZROWS = {}


def _zero_rows(like, n):
    """[n, cols of like] zeros (n < 32) cut from one cached block per (device, dtype, width)"""
    key = (like.device, like.dtype, like.shape[1])
    z = _ZROWS.get(key)
    if z is None:
        z = _ZROWS[key] = like.new_zeros((32, like.shape[1]))
    return z[:n]


def relation_project_batched(w, xs, refs, want_x=False, also_cat=(), pad_refs=False):
    """The Wq / Wk / Wv projections of several INDEPENDENT attention problems (the key frames of one engine batch) as
    ONE GEMM each over the concatenated rows: M = sum of the rows, so the 64x64-tile launches of the per-frame form
    become a few big-tile launches.  Every GEMM kernel is batch-invariant (an output row never depends on the rows it
    is batched with), so the slices have the same bits as relation_attention_forward's own projections.
    xs[i] [Nq_i,1024] queries, refs[i] [Nr_i,1024] keys/values -- each a tensor or a tuple of row blocks (concatenated
    here, once, together with everything else) -> (qs, ks, vts): qs[i] [Nq_i,1024], ks[i] [Nr_i,1024],
    vts[i] [1024,Nr_i] (views into the batched results; a caller that keeps a slice must copy it).
    want_x: also return the views xcat[i] [Nq_i,1024] of the concatenated queries (the attention's residual).
    also_cat: further lists of row blocks the caller wants concatenated (they ride in the same copy launch); their
    results are appended to the returned tuple as one list.
    pad_refs: every problem's key rows are followed by zero rows up to a multiple of 32 (block copies in the same
    launch), so that its V^T block starts at a 32-aligned column and ends in exact-zero pad columns (Wv . 0): vts[i] is
    then [1024, ceil32(Nr_i)] and can be handed to the attention kernel as it is -- no per-problem pad + concatenation
    (one fill and one 2-byte cat launch per key frame before).  ks[i] stays [Nr_i,1024]."""
    xf, nq = _flat(xs)
    if pad_refs:
        rf, nr, npad = [], [], []
        for r in refs:
            parts = list(r) if isinstance(r, (tuple, list)) else [r]
            n = sum(p.shape[0] for p in parts)
            pad = (-n) % 32
            rf += parts + ([_zero_rows(parts[0], pad)] if pad else [])
            nr.append(n)
            npad.append(n + pad)
    else:
        rf, nr = _flat(refs)
        npad = nr
    if rf[0].dtype != w.wq.dtype and _rows_view(rf) is None:
        # f32 activation stream: the key / value sources are only ever read as their bf16 copy -- concatenate AND round in one
        # launch, the f32 concatenation is never written (ops.cat_rows_cast_bf16)
        cats = cat_rows_many([xf] + [list(c) for c in also_cat])
        x_all, cats = cats[0], [None] + cats
        r_op = ops.cat_rows_cast_bf16(rf, w.wq.dtype)
    else:
        cats = cat_rows_many([rf, xf] + [list(c) for c in also_cat])
        x_all = cats[1]
        r_op = op_dtype(w, cats[0])       # (f32 activation stream: one rounded copy per concatenation, not per problem)
    k_all = ops.linear(r_op, w.wk, w.bk)
    vt_all = project_v(w, r_op, (r_op.shape[0] + 31) // 32 * 32)
    q_all = ops.linear(op_dtype(w, x_all), w.wq, w.bq)
    qs, ks, vts, xc = [], [], [], []
    oq = orr = 0
    for i in range(len(xs)):
        qs.append(q_all[oq:oq + nq[i]])
        xc.append(x_all[oq:oq + nq[i]])
        ks.append(k_all[orr:orr + nr[i]])
        vts.append(vt_all[:, orr:orr + npad[i]])
        oq += nq[i]
        orr += npad[i]
    res = (qs, ks, vts, xc) if want_x else (qs, ks, vts)
    return res + (cats[2:],) if also_cat else res


def relation_attend(w, x, q, k, vt, rois_q=None, rois_k=None, mem_kv=None, residual=True):
    """The attention core of one problem on already-projected operands (relation_project_batched): q [Nq,1024],
    k [Nr,1024], vt [1024,Nr] (may be a strided view), mem_kv as in relation_attention_forward.  -> x + attention."""
    Nk = k.shape[0]
    vparts = [vt]
    if mem_kv is not None:
        k_mem, vt_mem = mem_kv
        Nk += k_mem.shape[0]
        k = torch.cat([k, k_mem], dim=0)
        vparts.append(vt_mem)
    ldv = (Nk + 31) // 32 * 32
    if ldv > Nk:
        vparts.append(vt.new_zeros((vt.shape[0], ldv - Nk)))
    vv = torch.cat(vparts, dim=1) if len(vparts) > 1 else vt.contiguous()
    pos = None
    if w.with_pos:
        fast = w.wq.dtype != torch.float32
        pos = ops.position_logits(rois_q, rois_k, w.wg_t, w.bg, w.dim_mat, precise=not fast, tiled=w.wq.dtype if fast else False)
    return ops.relation_attention(q, k, vv, Nk, pos=pos, resid=x if residual else None, bias_v=w.bv)


def relation_attend_batched(w, jobs, residual=True, pos=None):
    """relation_attend for several problems of the SAME weights (the key frames of a step-batch at one stage) with the
    position logits and the attention core each as ONE launch: jobs = list of dict(x, q, k, vt, rois_q, rois_k, mem_kv),
    or, with the key sets already assembled by the caller, dict(x, q, k_all [Nk,1024], vt_all [1024,>=ceil32(Nk)]
    (unit column stride, any row stride), Nk, rois_q, rois_k), or in two segments that are read where they lie:
    dict(x, q, k [N1,1024], vt [1024,>=N1], k2 [Nk-N1,1024], vt2 [1024,>=Nk-N1], N1, Nk, rois_q, rois_k).
    Same bits per problem as relation_attend.
    The outputs are consecutive row blocks of one buffer (cat_rows() of them in order is free).
    pos: the problems' position logits if the caller already has them (position_logits_for)."""
    if not jobs:
        return []
    items, rq, rk = [], [], []
    for j in jobs:
        seg = None
        if "k2" in j:          # two key segments read in place: (k, vt) keys 0 .. N1-1, (k2, vt2) keys N1 .. Nk-1
            k, vv, Nk = j["k"], j["vt"], j["Nk"]
            seg = {"k2": j["k2"], "vt2": j["vt2"], "N1": j["N1"]}
        elif "k_all" in j:
            k, vv, Nk = j["k_all"], j["vt_all"], j["Nk"]
        else:
            k, vt = j["k"], j["vt"]
            Nk = k.shape[0]
            vparts = [vt]
            if j.get("mem_kv") is not None:
                k_mem, vt_mem = j["mem_kv"]
                Nk += k_mem.shape[0]
                k = torch.cat([k, k_mem], dim=0)
                vparts.append(vt_mem)
            ldv = (Nk + 31) // 32 * 32
            if ldv > Nk:
                vparts.append(vt.new_zeros((vt.shape[0], ldv - Nk)))
            vv = torch.cat(vparts, dim=1) if len(vparts) > 1 else vt.contiguous()
        items.append({"q": j["q"], "k": k, "vt": vv, "Nk": Nk, "resid": j["x"] if residual else None, "bias_v": w.bv})
        if seg is not None:
            items[-1].update(seg)
        rq.append(j.get("rois_q"))
        rk.append(j.get("rois_k"))
    if w.with_pos:
        if pos is None:      # (else: computed ahead of time by position_logits_for, same call, same bits)
            pos = position_logits_for(w, rq, rk)
        for it, p in zip(items, pos):
            it["pos"] = p
    return ops.relation_attention_batched(items)


def position_logits_for(w, rois_qs, rois_ks):
    """The position logits relation_attend_batched(w, jobs) computes for jobs with these query / key boxes (one launch per
    20 problems in bf16 mode).  They depend on boxes only, so a caller that knows the key sets' boxes of a stage before its
    features can compute them early, beside other work (MEGAFeatureExtractor.aggregate_batch does, on a side stream)."""
    fast = w.wq.dtype != torch.float32
    return ops.position_logits_batched(rois_qs, rois_ks, w.wg_t, w.bg, w.dim_mat, precise=not fast,
                                       tiled=w.wq.dtype if fast else False)

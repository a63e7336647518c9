"""IAOG seq2seq decoder blocks (reference mm_modeling.py:35-132 `Attention`, :558-666 decoder).

Reference quirks reproduced (SURVEY.md Appendix B + oracle/fcmf_oracle.py):
  * values are the projected KEYS (mm_modeling.py:129);
  * any 2-D `memory_len` means a tril(q_len, k_len) mask filled with -1e4, also on the
    decoder->encoder cross attention (:115-124);
  * the per-head weights are tiled batch-major while the inputs are tiled head-major (:79-85), so
    output slot s of batch element b is projected with head (s*B + b) % n_head;
  * the attention `dropout` constructor argument is never used.
The per-head projections are ONE GEMM against the [n_head*d, E] re-layout of w_kx / w_qx instead
of the reference's B-fold `repeat` + bmm.
"""
import math

import torch
import torch.nn as nn

from . import layers, ops


def _quirk_attn_fwd(qx, kx, heads, causal):
    """qx [G, R, heads*d], kx [G, T, heads*d] (row strides arbitrary, unit inner stride) -> out [G, R, heads*d] dense, lse"""
    from . import _hip as H
    G, R, HD = qx.shape
    out = torch.empty((G, R, HD), dtype=qx.dtype, device=qx.device)
    lse = torch.empty((G, heads, R), dtype=torch.float32, device=qx.device)
    a = ops._desc(qx, kx, kx, None, None, None, None, heads, 1, 1.0 / math.sqrt(HD // heads), 0.0, 0, causal, 1)
    H.check(H.lib().fcmf_attn_small_fwd(a, H.ptr(out), H.ptr(lse), H.stream()), "fcmf_attn_small_fwd")
    return out, lse


def _quirk_attn_bwd(qx, kx, out, lse, dout, heads, causal, dq_out, dk_out):
    """gradients per HEAD written into dq_out / dk_out ([G, R|T, heads*d] views with unit inner stride, any row stride): the
    attention backward produces them per output SLOT, fcmf_head_gather sums the slots that read each head
    (slot s of group g reads head (s*G + g) % heads, mm_modeling.py:79-85)"""
    from . import _hip as H
    G, R, HD = qx.shape
    T, d = kx.shape[1], HD // heads
    nch = max(1, (T + 127) // 128)
    dq_slot = torch.empty((nch, G, R, HD), dtype=qx.dtype, device=qx.device)
    dk_slot = torch.empty((G, T, HD), dtype=qx.dtype, device=qx.device)
    a = ops._desc(qx, kx, kx, None, None, None, None, heads, 1, 1.0 / math.sqrt(d), 0.0, 0, causal, 1)
    H.check(H.lib().fcmf_attn_small_bwd(a, H.ptr(out), H.ptr(dout.contiguous()), H.ptr(lse), H.ptr(dq_slot),
                                        H.ptr(dk_slot), 0, 0, 0, 0, H.stream()), "fcmf_attn_small_bwd")
    dq_slot = ops._sum_leading(dq_slot)
    L, st = H.lib(), H.stream()
    H.check(L.fcmf_head_gather(H.ptr(dq_slot), H.ptr(dq_out), dq_out.stride(1), G, R, heads, d, H.dt(qx), st), "fcmf_head_gather")
    H.check(L.fcmf_head_gather(H.ptr(dk_slot), H.ptr(dk_out), dk_out.stride(1), G, T, heads, d, H.dt(qx), st), "fcmf_head_gather")


class _QuirkAttentionFn(torch.autograd.Function):
    """attention over natural-head-order projections with the reference's slot->head pairing"""

    @staticmethod
    def forward(ctx, qx, kx, heads, causal):
        qx = qx if qx.stride(2) == 1 else qx.contiguous()
        kx = kx if kx.stride(2) == 1 else kx.contiguous()
        out, lse = _quirk_attn_fwd(qx, kx, heads, causal)
        ctx.save_for_backward(qx, kx, out, lse)
        ctx.cfg = (heads, causal)
        return out

    @staticmethod
    def backward(ctx, dout):
        qx, kx, out, lse = ctx.saved_tensors
        heads, causal = ctx.cfg
        dq = torch.empty(qx.shape, dtype=qx.dtype, device=qx.device)
        dk = torch.empty(kx.shape, dtype=kx.dtype, device=kx.device)
        _quirk_attn_bwd(qx, kx, out, lse, dout, heads, causal, dq, dk)
        return dq, dk, None, None


def _pair_layouts(wk, wq, dtype):
    """[2*n_head*d, E] (nn.Linear layout: keys' rows, then queries') and its transpose [E, 2*n_head*d] for the two per-head
    weight tensors of one decoder Attention, cached on the key parameter (both are stale after every optimizer step)"""
    nh, E, d = wk.shape

    def nk(_):
        return ops.cast(torch.cat((wk.detach().permute(0, 2, 1).reshape(nh * d, E), wq.detach().permute(0, 2, 1).reshape(nh * d, E)), 0), dtype)

    def kn(_):
        return ops.cast(torch.cat((wk.detach().permute(1, 0, 2).reshape(E, nh * d), wq.detach().permute(1, 0, 2).reshape(E, nh * d)), 1), dtype)
    return ops.shadows.derived(wk, ("pair_nk", dtype, wq.data_ptr()), nk), (lambda: ops.shadows.derived(wk, ("pair_kn", dtype, wq.data_ptr()), kn))


class _SelfQuirkAttentionFn(torch.autograd.Function):
    """decoder self attention `Attention(X, X, causal)` up to (not including) `proj`, as ONE node: the key and query projections
    of the same input are one GEMM against [w_kx | w_qx] (N = 2 * n_head * d), the attention reads the two halves of its output in
    place, and the backward gathers the per-slot gradients straight into the halves of one [rows, 2*n_head*d] buffer that feeds
    ONE dX and ONE dW GEMM (the reference: 2 x (repeat + bmm) forward, mm_modeling.py:79-92)."""

    @staticmethod
    def forward(ctx, x, wk, wq, heads, causal):
        G, T, E = x.shape
        x2 = ops._rows(x)
        HD = wk.shape[0] * wk.shape[2]
        wl = ops.shadows.head_nk([wk, wq]) if x2.dtype == torch.bfloat16 else _pair_layouts(wk, wq, x2.dtype)[0]
        kq = torch.empty((G * T, 2 * HD), dtype=x2.dtype, device=x2.device)
        ops.gemm(x2, wl, kq, G * T, 2 * HD, E, ops._ld(x2), E, 2 * HD, 0, 0)
        kq3 = kq.view(G, T, 2 * HD)
        out, lse = _quirk_attn_fwd(kq3[:, :, HD:], kq3[:, :, :HD], heads, causal)
        ctx.save_for_backward(x2, kq, out, lse, wk, wq)
        ctx.cfg = (heads, causal, x.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        x2, kq, out, lse, wk, wq = ctx.saved_tensors
        heads, causal, xshape = ctx.cfg
        G, T, E = xshape
        nh, _, d = wk.shape
        HD = nh * d
        kq3 = kq.view(G, T, 2 * HD)
        dkq = torch.empty_like(kq)
        d3 = dkq.view(G, T, 2 * HD)
        _quirk_attn_bwd(kq3[:, :, HD:], kq3[:, :, :HD], out, lse, dout, heads, causal, d3[:, :, HD:], d3[:, :, :HD])
        dx = torch.empty((G * T, E), dtype=x2.dtype, device=x2.device)
        if x2.dtype == torch.bfloat16:     # NN: dx = dkq W with W in the forward's [2*n_head*d, E] layout (no second re-layout)
            ops.gemm(dkq, ops.shadows.head_nk([wk, wq]), dx, G * T, E, 2 * HD, 2 * HD, E, E, 0, 1)
        else:
            _, kn = _pair_layouts(wk, wq, x2.dtype)
            ops.gemm(dkq, kn(), dx, G * T, E, 2 * HD, 2 * HD, 2 * HD, E, 0, 0)               # NT: both operands K-contiguous
        direct = ops.head_weight_grad(x2, dkq, [wk, wq])          # straight into the two parameters' adjacent [n_head, E, d] arena slices
        if direct is not None:
            return dx.view(xshape), direct[0], direct[1], None, None
        dwl = torch.empty((2 * HD, E), dtype=torch.float32, device=x2.device)
        ops.gemm(dkq, x2, dwl, 2 * HD, E, G * T, 2 * HD, ops._ld(x2), E, 1, 1)                # [2*n_head*d, E] = dkq^T x
        dw = dwl.view(2, nh, d, E).permute(0, 1, 3, 2)                                        # -> the parameters' [n_head, E, d]
        return dx.view(xshape), dw[0], dw[1], None, None


class _HoistedKeysFn(torch.autograd.Function):
    """the key projections of ALL decoder blocks' cross attention -- every block projects the same encoder output with its own
    w_kx (mm_modeling.py:601-605) -- as ONE GEMM (N = blocks * n_head * d) before the block loop; returns one [G, T, n_head*d]
    view per block.  Backward: the blocks' key gradients side by side -> ONE dX and ONE dW GEMM."""

    @staticmethod
    def forward(ctx, enc, *wks):
        G, T, E = enc.shape
        e2 = ops._rows(enc)
        nh, _, d = wks[0].shape
        HD, nb = nh * d, len(wks)

        def nk(_):
            return ops.cast(torch.cat([w.detach().permute(0, 2, 1).reshape(HD, E) for w in wks], 0), e2.dtype)
        wl = (ops.shadows.head_nk(list(wks)) if e2.dtype == torch.bfloat16
              else ops.shadows.derived(wks[0], ("hoist_nk", e2.dtype, nb, wks[-1].data_ptr()), nk))
        kx = torch.empty((G * T, nb * HD), dtype=e2.dtype, device=e2.device)
        ops.gemm(e2, wl, kx, G * T, nb * HD, E, ops._ld(e2), E, nb * HD, 0, 0)
        ctx.save_for_backward(e2, *wks)
        ctx.cfg = (enc.shape, nb, HD)
        k3 = kx.view(G, T, nb * HD)
        return tuple(k3[:, :, i * HD:(i + 1) * HD] for i in range(nb))

    @staticmethod
    def backward(ctx, *grads):
        e2, *wks = ctx.saved_tensors
        eshape, nb, HD = ctx.cfg
        G, T, E = eshape
        nh, _, d = wks[0].shape
        zero = None
        parts = []
        for g in grads:
            if g is None:
                zero = torch.zeros((G, T, HD), dtype=e2.dtype, device=e2.device) if zero is None else zero
                g = zero
            parts.append(g.reshape(G * T, HD))
        dk = torch.cat(parts, 1)                                                             # [G*T, blocks*HD]

        def kn(_):
            return ops.cast(torch.cat([w.detach().permute(1, 0, 2).reshape(E, HD) for w in wks], 1), e2.dtype)
        de = torch.empty((G * T, E), dtype=e2.dtype, device=e2.device)
        if e2.dtype == torch.bfloat16:
            ops.gemm(dk, ops.shadows.head_nk(list(wks)), de, G * T, E, nb * HD, nb * HD, E, E, 0, 1)       # NN
        else:
            wt = ops.shadows.derived(wks[0], ("hoist_kn", e2.dtype, nb, wks[-1].data_ptr()), kn)
            ops.gemm(dk, wt, de, G * T, E, nb * HD, nb * HD, nb * HD, E, 0, 0)
        direct = ops.head_weight_grad(e2, dk, list(wks))          # every block's w_kx: adjacent arena slices (dp.GradArena.for_model)
        if direct is not None:
            return (de.view(eshape),) + tuple(direct)
        dwl = torch.empty((nb * HD, E), dtype=torch.float32, device=e2.device)
        ops.gemm(dk, e2, dwl, nb * HD, E, G * T, nb * HD, ops._ld(e2), E, 1, 1)
        dw = dwl.view(nb, nh, d, E).permute(0, 1, 3, 2)
        return (de.view(eshape),) + tuple(dw[i] for i in range(nb))


_valid_lens_cache = {}


def _dec_valid_lens(B, T, device):
    """arange(1, T+1).repeat(B, 1) (mm_modeling.py:595-597): only its being 2-D matters (-> the tril rule); built once per shape"""
    key = (B, T, str(device))
    v = _valid_lens_cache.get(key)
    if v is None:
        v = _valid_lens_cache[key] = torch.arange(1, T + 1, device=device).repeat(B, 1)
    return v


class Attention(nn.Module):
    def __init__(self, embed_dim, hidden_dim=None, n_head=1, score_function='scaled_dot_product', dropout=0.1):
        super().__init__()
        if hidden_dim is None:
            hidden_dim = embed_dim // n_head
        if score_function != 'scaled_dot_product':
            # the reference raises RuntimeError('invalid score_function') for unknown names; 'mlp' and
            # 'bi_linear' exist there but are never used on the training path
            raise RuntimeError('invalid score_function' if score_function not in ('mlp', 'bi_linear')
                               else f"score_function '{score_function}' is not on the FCMF path")
        self.embed_dim, self.hidden_dim, self.n_head, self.score_function = embed_dim, hidden_dim, n_head, score_function
        self.w_kx = nn.Parameter(torch.empty(n_head, embed_dim, hidden_dim))
        self.w_qx = nn.Parameter(torch.empty(n_head, embed_dim, hidden_dim))
        self.proj = nn.Linear(n_head * hidden_dim, embed_dim)
        self.register_parameter('weight', None)
        nn.init.xavier_uniform_(self.w_kx)
        nn.init.xavier_uniform_(self.w_qx)
        self.attention_weights = None

    def forward(self, k, q, memory_len=None, kx=None):
        """NOTE the argument order: keys first (reference mm_modeling.py:66).  kx: the already projected keys (IAOGDecoder hoists
        the cross-attention key projections of all its blocks into one GEMM)."""
        same = k is q
        if k.dim() == 2:
            k = k.unsqueeze(1)
        if q.dim() == 2:
            q = q.unsqueeze(1)
        k, q = layers.to_compute(k), layers.to_compute(q)
        causal = False
        if memory_len is not None:
            if isinstance(memory_len, (list, tuple)):
                memory_len = torch.tensor(memory_len, device=k.device)
            if memory_len.dim() == 2:
                causal = True
            else:
                raise NotImplementedError("1-D memory_len (key-length fill mask) is only used by the disabled MDE")
        nh = self.n_head
        if same and kx is None and k.dim() == 3:
            out = _SelfQuirkAttentionFn.apply(q, self.w_kx, self.w_qx, nh, causal)       # one GEMM for [kx | qx]
        else:
            if kx is None:
                kx = ops.head_linear(k, self.w_kx)    # [.., nh*hd], natural head order
            qx = ops.head_linear(q, self.w_qx)
            out = _QuirkAttentionFn.apply(qx, kx, nh, causal)
        self.attention_weights = None  # probabilities are never materialised by the fused kernel
        return ops.linear(out, self.proj.weight, self.proj.bias), None


class PositionWiseFFN(nn.Module):
    def __init__(self, ffn_num_hiddens, ffn_num_outputs, hidden_size=None):
        super().__init__()
        from . import mm_modeling as mm
        H = hidden_size or mm.HIDDEN_SIZE
        self.dense1 = nn.Linear(H, ffn_num_hiddens)
        self.act = mm.ACT2FN[mm.HIDDEN_ACT]
        self.dense2 = nn.Linear(ffn_num_hiddens, ffn_num_outputs)

    def forward(self, x):
        return ops.ffn(layers.to_compute(x), self.dense1.weight, self.dense1.bias, self.dense2.weight, self.dense2.bias)


class AddNorm(nn.Module):
    """LN(dropout(Y) + X) (reference mm_modeling.py:566-573)"""

    def __init__(self, norm_shape, dropout):
        super().__init__()
        from .mm_modeling import FCMFLayerNorm
        self.dropout = nn.Dropout(dropout)
        self.ln = FCMFLayerNorm(norm_shape)

    def forward(self, X, Y):
        return ops.add_layer_norm(layers.to_compute(Y), layers.to_compute(X), self.ln.weight, self.ln.bias,
                                  self.ln.variance_epsilon, self.dropout.p, self.training)


class TransformerDecoderBlock(nn.Module):
    def __init__(self, i, hidden_size=None, num_heads=None):
        super().__init__()
        from . import mm_modeling as mm
        H, nh = hidden_size or mm.HIDDEN_SIZE, num_heads or mm.NUM_ATTENTION_HEADS
        p = mm.ATTENTION_PROBS_DROPOUT_PROB
        self.i = i
        self.attention1 = Attention(H, H // nh, nh, 'scaled_dot_product', p)
        self.addnorm1 = AddNorm(H, p)
        self.attention2 = Attention(H, H // nh, nh, 'scaled_dot_product', p)
        self.addnorm2 = AddNorm(H, p)
        self.ffn = PositionWiseFFN(H, H, H)
        self.add_norm3 = AddNorm(H, p)

    def forward(self, X, state, enc_attention_mask=None, is_train=True):
        enc_outputs, enc_valid_lens = state[0], state[1]
        if state[2][self.i] is not None:  # the reference concatenates a cache it never reads (:588-601)
            state[2][self.i] = torch.cat((state[2][self.i], X), dim=1)
        if is_train:
            B, T, _ = X.shape
            dec_valid_lens = _dec_valid_lens(B, T, X.device)
        else:
            dec_valid_lens = None
        X2, _ = self.attention1(X, X, dec_valid_lens)
        Y = self.addnorm1(X, X2)
        cross_mask = enc_attention_mask if enc_attention_mask is not None else enc_valid_lens
        Y2, _ = self.attention2(enc_outputs, Y, cross_mask, kx=getattr(self, "_hoisted_kx", None))
        Z = self.addnorm2(Y, Y2)
        return self.add_norm3(Z, self.ffn(Z)), state


class PositionalEncoding(nn.Module):
    def __init__(self, hidden_size=None):
        super().__init__()
        from . import mm_modeling as mm
        H = hidden_size or mm.HIDDEN_SIZE
        self.dropout = nn.Dropout(mm.ATTENTION_PROBS_DROPOUT_PROB)
        P = torch.zeros((1, mm.MAX_POSITION_EMBEDDINGS, H))
        X = torch.arange(mm.MAX_POSITION_EMBEDDINGS, dtype=torch.float32).reshape(-1, 1) / torch.pow(
            10000, torch.arange(0, H, 2, dtype=torch.float32) / H)
        P[:, :, 0::2] = torch.sin(X)
        P[:, :, 1::2] = torch.cos(X)
        self.register_buffer('P', P)

    def forward(self, X):
        pe = self.P[:, :X.size(1), :].to(device=X.device).type_as(X)
        return ops.dropout(X + pe, self.dropout.p, self.training)


class _ScaledEmbedding(torch.autograd.Function):
    """emb[ids] * sqrt(H) (+ P[:, :T]): `self.embedding(X) * math.sqrt(self.num_hiddens)` and PositionalEncoding's addition
    (reference mm_modeling.py:650, :633) in one kernel; the embedding gradient is scattered straight into the parameter's
    (arena) gradient slice -- no [V, H] zeros + index_add_ per step"""

    @staticmethod
    def forward(ctx, ids, weight, pos_table, scale, out_dtype):
        from . import _hip as H
        idc = ids.contiguous()
        n, Hd = idc.numel(), weight.shape[1]
        T = ids.shape[-1]
        out = torch.empty(tuple(ids.shape) + (Hd,), dtype=out_dtype, device=weight.device)
        P = None if pos_table is None else pos_table.reshape(-1, Hd)[:T].contiguous()
        H.check(H.lib().fcmf_embed_scale_fwd(H.ptr(idc), H.ptr(weight.detach()), H.ptr(P), H.ptr(out), n, Hd, T, weight.shape[0], float(scale),
                                             H.dt(out), H.stream()), "fcmf_embed_scale_fwd")
        ctx.save_for_backward(idc)
        ctx.scale = scale
        ctx.weight = weight if (weight.dtype == torch.float32 and weight.is_contiguous()) else None
        ctx.wshape = weight.shape
        return out

    @staticmethod
    def backward(ctx, dy):
        from . import _hip as H
        (ids,) = ctx.saved_tensors
        d = dy.contiguous()
        dw = ops.alloc_grad(ctx.weight, ctx.wshape) if ctx.weight is not None else \
            torch.zeros(ctx.wshape, dtype=torch.float32, device=dy.device)
        H.check(H.lib().fcmf_embed_scale_bwd(H.ptr(d), H.ptr(ids), H.ptr(dw), ids.numel(), ctx.wshape[1], ctx.wshape[0], None, float(ctx.scale),
                                             H.dt(d), H.stream()), "fcmf_embed_scale_bwd")      # (ids outside the table made the forward's rows NaN: the loss already says so)
        return None, dw, None, None, None


class IAOGDecoder(nn.Module):
    def __init__(self, vocab_size, hidden_size=None, num_layers=None, num_heads=None):
        super().__init__()
        from . import mm_modeling as mm
        self.num_hiddens = hidden_size or mm.HIDDEN_SIZE
        self.num_blks = num_layers or mm.NUM_HIDDEN_LAYERS
        self.embedding = nn.Embedding(vocab_size, self.num_hiddens)
        self.pos_encoding = PositionalEncoding(self.num_hiddens)
        self.blks = nn.Sequential()
        for i in range(self.num_blks):
            self.blks.add_module('block' + str(i), TransformerDecoderBlock(i, self.num_hiddens, num_heads))
        self.dense = nn.Linear(self.num_hiddens, vocab_size)
        self.dense.weight = self.embedding.weight

    def init_state(self, enc_outputs, enc_valid_lens):
        return [enc_outputs, enc_valid_lens, [None] * self.num_blks]

    def project_encoder(self, enc_outputs):
        """every block's cross-attention keys ( = values, mm_modeling.py:129) of an encoder output, one GEMM for all blocks;
        hand the result to forward(..., hoisted=) when the same encoder output is decoded again and again (decoding.py)"""
        return _HoistedKeysFn.apply(layers.to_compute(enc_outputs), *[blk.attention2.w_kx for blk in self.blks])

    def hidden_states(self, X, state, enc_attention_mask=None, is_train=True, hoisted=None):
        """the decoder stack up to (not including) the vocabulary projection: [B, Ld, H]"""
        # embedding * sqrt(H) + P in one kernel, then PositionalEncoding's dropout (mm_modeling.py:650, :633)
        X = _ScaledEmbedding.apply(X, self.embedding.weight, self.pos_encoding.P, math.sqrt(self.num_hiddens), ops.compute_dtype())
        X = ops.dropout(X, self.pos_encoding.dropout.p, self.training)
        self._attention_weights = [[None] * len(self.blks) for _ in range(2)]
        # every block's cross attention projects the SAME encoder output with its own w_kx: one GEMM for all of them
        enc = layers.to_compute(state[0])
        if hoisted is None:
            hoisted = _HoistedKeysFn.apply(enc, *[blk.attention2.w_kx for blk in self.blks]) if enc.dim() == 3 else [None] * len(self.blks)
        for i, blk in enumerate(self.blks):
            blk._hoisted_kx = hoisted[i]
            try:
                X, state = blk(X, state, enc_attention_mask=enc_attention_mask, is_train=is_train)
            finally:
                blk._hoisted_kx = None
            self._attention_weights[0][i] = blk.attention1.attention_weights
            self._attention_weights[1][i] = blk.attention2.attention_weights
        return X

    def forward(self, X, state, enc_attention_mask=None, is_train=True, hoisted=None):
        X = self.hidden_states(X, state, enc_attention_mask, is_train, hoisted)
        return ops.vocab_linear(X, self.dense.weight, self.dense.bias)

    def loss(self, X, state, labels, enc_attention_mask=None, ignore_index=-100):
        """CrossEntropyLoss(ignore_index)(forward(X).permute(0, 2, 1), labels) (run_pretraining_fcmf.py:322-324) with
        the vocabulary projection and the loss fused: the [B, Ld, V] logits are never handed out"""
        X = self.hidden_states(X, state, enc_attention_mask, True)
        return ops.vocab_cross_entropy(X, self.dense.weight, self.dense.bias, labels, ignore_index)

    @property
    def attention_weights(self):
        return self._attention_weights

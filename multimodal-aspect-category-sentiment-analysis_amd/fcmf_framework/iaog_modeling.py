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


class _QuirkAttentionFn(torch.autograd.Function):
    """attention over natural-head-order projections with the reference's slot->head pairing"""

    @staticmethod
    def forward(ctx, qx, kx, heads, causal):
        from . import _hip as H
        qx, kx = qx.contiguous(), kx.contiguous()
        G, R, HD = qx.shape
        out = torch.empty_like(qx)
        lse = torch.empty((G, heads, R), dtype=torch.float32, device=qx.device)
        a = ops._desc(qx, kx, kx, None, None, None, None, heads, 1, 1.0 / math.sqrt(HD // heads), 0.0, 0, causal, 1)
        H.check(H.lib().fcmf_attn_small_fwd(a, H.ptr(out), H.ptr(lse), H.stream()), "fcmf_attn_small_fwd")
        ctx.save_for_backward(qx, kx, out, lse)
        ctx.cfg = (heads, causal)
        return out

    @staticmethod
    def backward(ctx, dout):
        from . import _hip as H
        qx, kx, out, lse = ctx.saved_tensors
        heads, causal = ctx.cfg
        G, R, HD = qx.shape
        d = HD // heads
        nch = max(1, (kx.shape[1] + 127) // 128)
        dq_slot = torch.empty((nch,) + tuple(qx.shape), dtype=qx.dtype, device=qx.device)
        dk_slot = torch.empty_like(kx)
        a = ops._desc(qx, kx, kx, None, None, None, None, heads, 1, 1.0 / math.sqrt(d), 0.0, 0, causal, 1)
        H.check(H.lib().fcmf_attn_small_bwd(a, H.ptr(out), H.ptr(dout.contiguous()), H.ptr(lse), H.ptr(dq_slot),
                                            H.ptr(dk_slot), 0, 0, 0, 0, H.stream()), "fcmf_attn_small_bwd")
        dq_slot = ops._sum_leading(dq_slot)
        # slot s of group g read head (s*G + g) % heads: scatter-add slot gradients back to heads
        slot = torch.arange(heads, device=qx.device).view(1, heads)
        hh = (slot * G + torch.arange(G, device=qx.device).view(G, 1)) % heads          # [G, heads]
        def to_heads(gs):
            T = gs.shape[1]
            g4 = gs.view(G, T, heads, d).float()
            idx = hh.view(G, 1, heads, 1).expand(G, T, heads, d)
            return torch.zeros_like(g4).scatter_add_(2, idx, g4).view(G, T, HD).to(gs.dtype)
        return to_heads(dq_slot), to_heads(dk_slot), None, None


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

    def forward(self, k, q, memory_len=None):
        """NOTE the argument order: keys first (reference mm_modeling.py:66)."""
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
        kx = ops.head_linear(k, self.w_kx)        # [.., nh*hd], natural head order
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
            dec_valid_lens = torch.arange(1, T + 1, device=X.device).repeat(B, 1)
        else:
            dec_valid_lens = None
        X2, _ = self.attention1(X, X, dec_valid_lens)
        Y = self.addnorm1(X, X2)
        cross_mask = enc_attention_mask if enc_attention_mask is not None else enc_valid_lens
        Y2, _ = self.attention2(enc_outputs, Y, cross_mask)
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
    """emb[ids] * sqrt(H) (reference mm_modeling.py:650) through the embedding kernels"""

    @staticmethod
    def forward(ctx, ids, weight, scale, out_dtype):
        e = weight.detach()[ids] * scale        # gather glue; tiny ([B, Ld, H])
        ctx.save_for_backward(ids)
        ctx.scale, ctx.wshape = scale, weight.shape
        return e.to(out_dtype)

    @staticmethod
    def backward(ctx, dy):
        (ids,) = ctx.saved_tensors
        dw = torch.zeros(ctx.wshape, dtype=torch.float32, device=dy.device)
        dw.index_add_(0, ids.reshape(-1), dy.reshape(-1, dy.shape[-1]).float() * ctx.scale)
        return None, dw, None, None


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

    def hidden_states(self, X, state, enc_attention_mask=None, is_train=True):
        """the decoder stack up to (not including) the vocabulary projection: [B, Ld, H]"""
        X = _ScaledEmbedding.apply(X, self.embedding.weight, math.sqrt(self.num_hiddens), ops.compute_dtype())
        X = self.pos_encoding(X)
        self._attention_weights = [[None] * len(self.blks) for _ in range(2)]
        for i, blk in enumerate(self.blks):
            X, state = blk(X, state, enc_attention_mask=enc_attention_mask, is_train=is_train)
            self._attention_weights[0][i] = blk.attention1.attention_weights
            self._attention_weights[1][i] = blk.attention2.attention_weights
        return X

    def forward(self, X, state, enc_attention_mask=None, is_train=True):
        X = self.hidden_states(X, state, enc_attention_mask, is_train)
        return ops.vocab_linear(X, self.dense.weight, self.dense.bias)

    def loss(self, X, state, labels, enc_attention_mask=None, ignore_index=-100):
        """CrossEntropyLoss(ignore_index)(forward(X).permute(0, 2, 1), labels) (run_pretraining_fcmf.py:322-324) with
        the vocabulary projection and the loss fused: the [B, Ld, V] logits are never handed out"""
        X = self.hidden_states(X, state, enc_attention_mask, True)
        return ops.vocab_cross_entropy(X, self.dense.weight, self.dense.bias, labels, ignore_index)

    @property
    def attention_weights(self):
        return self._attention_weights

"""Transformer building blocks of FCMF on the MI355X kernels.

Same public names, constructor defaults, attribute trees and state-dict keys as the reference's
fcmf_framework/mm_modeling.py (class list at SURVEY.md section 2); every forward runs on
libfcmf_hip.so through `ops` -- there is no torch fallback.  Unlike the reference, layer sizes can
be passed to the constructors (defaulting to the module constants below) so that a `large` text
encoder needs no source edit.
"""
import copy
import math

import torch
import torch.nn as nn

from . import layers, ops
from .fused import QKVStorageMixin
from .roberta import RobertaModel

# module constants kept for `from .mm_modeling import *` users (reference mm_modeling.py:21-32)
HIDDEN_SIZE = 768
NUM_HIDDEN_LAYERS = 12
NUM_ATTENTION_HEADS = 12
INTERMEDIATE_SIZE = 3072
HIDDEN_ACT = "gelu"
HIDDEN_DROPOUT_PROB = 0.1
ATTENTION_PROBS_DROPOUT_PROB = 0.1
MAX_POSITION_EMBEDDINGS = 512
TYPE_VOCAB_SIZE = 2
INITIALIZER_RANGE = 0.02


def gelu(x):
    """exact-erf GELU (reference mm_modeling.py:10-15).  Plain tensor expression for API
    completeness; on the hot path GELU is the epilogue of the FFN GEMM (ops.ffn)."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def swish(x):
    return x * torch.sigmoid(x)


ACT2FN = {"gelu": gelu, "relu": torch.nn.functional.relu, "swish": swish}


def _dims(hidden_size, num_heads, intermediate_size=None):
    g = globals()
    return (hidden_size or g["HIDDEN_SIZE"], num_heads or g["NUM_ATTENTION_HEADS"],
            intermediate_size or g["INTERMEDIATE_SIZE"])


class FCMFLayerNorm(nn.Module):
    """TF-style LayerNorm, eps inside the sqrt (reference mm_modeling.py:158-171)."""

    def __init__(self, hidden_size, eps=1e-12):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.bias = nn.Parameter(torch.zeros(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x):
        return ops.add_layer_norm(layers.to_compute(x), None, self.weight, self.bias, self.variance_epsilon)


class BertSelfAttention(QKVStorageMixin, nn.Module):
    """reference mm_modeling.py:174-219: returns the merged-head context [B,T,H]"""

    def __init__(self, hidden_size=None, num_heads=None):
        super().__init__()
        H, nh, _ = _dims(hidden_size, num_heads)
        self.num_attention_heads = nh
        self.attention_head_size = H // nh
        self.all_head_size = H
        self.query = nn.Linear(H, H)
        self.key = nn.Linear(H, H)
        self.value = nn.Linear(H, H)
        self.dropout = nn.Dropout(ATTENTION_PROBS_DROPOUT_PROB)
        self._fuse_qkv_storage()

    def forward(self, hidden_states, attention_mask):
        return _mha(self, hidden_states, hidden_states, attention_mask)


class BertCoAttention(BertSelfAttention):
    """reference mm_modeling.py:221-266: queries from s1, keys/values from s2"""

    def forward(self, s1_hidden_states, s2_hidden_states, s2_attention_mask):
        return _mha(self, s1_hidden_states, s2_hidden_states, s2_attention_mask)


def _flat_mask(mask, B, Tk):
    """accept the reference's extended additive mask [B,1,1,Tk] (or [B,Tk]) -> float32 [B,Tk]"""
    if mask is None:
        return None
    m = mask.to(torch.float32).reshape(mask.shape[0], -1)
    if m.shape[1] != Tk:
        raise ValueError(f"additive attention mask must broadcast over heads and queries: got {tuple(mask.shape)}")
    return m.expand(B, Tk).contiguous()


def _mha(mod, xq, xkv, mask):
    xq, xkv = layers.to_compute(xq), layers.to_compute(xkv)
    q = ops.linear(xq, mod.query.weight, mod.query.bias)
    k = ops.linear(xkv, mod.key.weight, mod.key.bias)
    v = ops.linear(xkv, mod.value.weight, mod.value.bias)
    return ops.attention(q, k1=k, v1=v, mask=_flat_mask(mask, xq.shape[0], xkv.shape[1]),
                         heads=mod.num_attention_heads, p=mod.dropout.p, training=mod.training)


class BertSelfOutput(nn.Module):
    """dense -> dropout -> LayerNorm(x + input) (reference mm_modeling.py:269-280)"""

    def __init__(self, hidden_size=None):
        super().__init__()
        H, _, _ = _dims(hidden_size, None)
        self.dense = nn.Linear(H, H)
        self.LayerNorm = FCMFLayerNorm(H, eps=1e-12)
        self.dropout = nn.Dropout(HIDDEN_DROPOUT_PROB)

    def forward(self, hidden_states, input_tensor):
        h = ops.linear(layers.to_compute(hidden_states), self.dense.weight, self.dense.bias)
        return ops.add_layer_norm(h, layers.to_compute(input_tensor), self.LayerNorm.weight, self.LayerNorm.bias,
                                  self.LayerNorm.variance_epsilon, self.dropout.p, self.training)


class BertAttention(nn.Module):
    def __init__(self, hidden_size=None, num_heads=None):
        super().__init__()
        self.self = BertSelfAttention(hidden_size, num_heads)
        self.output = BertSelfOutput(hidden_size)

    def forward(self, input_tensor, attention_mask):
        return self.output(self.self(input_tensor, attention_mask), input_tensor)


class BertCrossAttention(nn.Module):
    def __init__(self, hidden_size=None, num_heads=None):
        super().__init__()
        self.self = BertCoAttention(hidden_size, num_heads)
        self.output = BertSelfOutput(hidden_size)

    def forward(self, s1_input_tensor, s2_input_tensor, s2_attention_mask):
        return self.output(self.self(s1_input_tensor, s2_input_tensor, s2_attention_mask), s1_input_tensor)


class BertIntermediate(nn.Module):
    """dense + erf-GELU (reference mm_modeling.py:305-314); GELU is the GEMM epilogue"""

    def __init__(self, hidden_size=None, intermediate_size=None):
        super().__init__()
        H, _, I = _dims(hidden_size, None, intermediate_size)
        self.dense = nn.Linear(H, I)
        self.intermediate_act_fn = ACT2FN[HIDDEN_ACT]

    def forward(self, hidden_states):
        x = layers.to_compute(hidden_states)
        return _GeluLinear.apply(x, self.dense.weight, self.dense.bias)


class _GeluLinear(torch.autograd.Function):
    """standalone Linear+GELU for the module API (the fused layers use ops.ffn instead)"""

    @staticmethod
    def forward(ctx, x, w, b):
        from . import _hip as H
        x2 = ops._rows(x)
        c = ops.as_compute(w, x2.dtype)
        u = torch.empty((x2.shape[0], w.shape[0]), dtype=x2.dtype, device=x2.device)
        a = ops._linear_fwd(x2, c, b.detach(), H.EPI_GELU, aux=u)
        ctx.save_for_backward(x2, w, u)
        ctx.xshape = x.shape
        return a.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        from . import _hip as H
        x2, w, u = ctx.saved_tensors
        d = dy.reshape(-1, dy.shape[-1]).contiguous()
        du = torch.empty_like(d)
        H.check(H.lib().fcmf_act_bwd(H.ptr(d), H.ptr(u), H.ptr(du), d.numel(), 1, H.dt(d), H.stream()), "act_bwd")
        dx, dw, db = ops._linear_bwd(x2, ops.as_compute(w, x2.dtype), du)
        return dx.view(ctx.xshape), dw, db


class BertOutput(nn.Module):
    """reference mm_modeling.py:317-328"""

    def __init__(self, hidden_size=None, intermediate_size=None):
        super().__init__()
        H, _, I = _dims(hidden_size, None, intermediate_size)
        self.dense = nn.Linear(I, H)
        self.LayerNorm = FCMFLayerNorm(H, eps=1e-12)
        self.dropout = nn.Dropout(HIDDEN_DROPOUT_PROB)

    def forward(self, hidden_states, input_tensor):
        h = ops.linear(layers.to_compute(hidden_states), self.dense.weight, self.dense.bias)
        return ops.add_layer_norm(h, layers.to_compute(input_tensor), self.LayerNorm.weight, self.LayerNorm.bias,
                                  self.LayerNorm.variance_epsilon, self.dropout.p, self.training)


class BertLayer(nn.Module):
    """reference mm_modeling.py:331-342; fused: QKV GEMMs -> attention -> out GEMM -> add+LN -> FFN -> add+LN"""

    def __init__(self, hidden_size=None, num_heads=None, intermediate_size=None):
        super().__init__()
        self.attention = BertAttention(hidden_size, num_heads)
        self.intermediate = BertIntermediate(hidden_size, intermediate_size)
        self.output = BertOutput(hidden_size, intermediate_size)

    def forward(self, hidden_states, attention_mask):
        x = layers.to_compute(hidden_states)
        sa = self.attention.self
        m = _flat_mask(attention_mask, x.shape[0], x.shape[1])
        return layers.transformer_layer(self, x, x, m, sa.num_attention_heads,
                                        self.output.LayerNorm.variance_epsilon, self.output.dropout.p,
                                        sa.dropout.p, self.training)


class BertCrossAttentionLayer(nn.Module):
    """reference mm_modeling.py:344-355"""

    def __init__(self, hidden_size=None, num_heads=None, intermediate_size=None):
        super().__init__()
        self.attention = BertCrossAttention(hidden_size, num_heads)
        self.intermediate = BertIntermediate(hidden_size, intermediate_size)
        self.output = BertOutput(hidden_size, intermediate_size)

    def forward(self, s1_hidden_states, s2_hidden_states, s2_attention_mask):
        x1, x2 = layers.to_compute(s1_hidden_states), layers.to_compute(s2_hidden_states)
        sa = self.attention.self
        m = _flat_mask(s2_attention_mask, x1.shape[0], x2.shape[1])
        return layers.transformer_layer(self, x1, x2, m, sa.num_attention_heads,
                                        self.output.LayerNorm.variance_epsilon, self.output.dropout.p,
                                        sa.dropout.p, self.training)


class MultimodalEncoder(nn.Module):
    """one shared BertLayer (reference mm_modeling.py:373-387)"""

    def __init__(self, hidden_size=None, num_heads=None, intermediate_size=None):
        super().__init__()
        layer = BertLayer(hidden_size, num_heads, intermediate_size)
        self.layer = nn.ModuleList([copy.deepcopy(layer) for _ in range(1)])

    def forward(self, hidden_states, attention_mask, output_all_encoded_layers=True):
        outs = []
        for layer_module in self.layer:
            hidden_states = layer_module(hidden_states, attention_mask)
            if output_all_encoded_layers:
                outs.append(hidden_states)
        if not output_all_encoded_layers:
            outs.append(hidden_states)
        return outs


class BertCrossEncoder(nn.Module):
    """one BertCrossAttentionLayer (reference mm_modeling.py:389-403)"""

    def __init__(self, hidden_size=None, num_heads=None, intermediate_size=None):
        super().__init__()
        layer = BertCrossAttentionLayer(hidden_size, num_heads, intermediate_size)
        self.layer = nn.ModuleList([copy.deepcopy(layer) for _ in range(1)])

    def forward(self, s1_hidden_states, s2_hidden_states, s2_attention_mask, output_all_encoded_layers=True):
        outs = []
        for layer_module in self.layer:
            s1_hidden_states = layer_module(s1_hidden_states, s2_hidden_states, s2_attention_mask)
            if output_all_encoded_layers:
                outs.append(s1_hidden_states)
        if not output_all_encoded_layers:
            outs.append(s1_hidden_states)
        return outs


class BertPooler(nn.Module):
    """tanh(W h[:,0] + b): row gather through the GEMM's A row stride, tanh as its epilogue
    (reference mm_modeling.py:419-431)"""
    token_index = 0

    def __init__(self, hidden_size=None):
        super().__init__()
        H, _, _ = _dims(hidden_size, None)
        self.dense = nn.Linear(H, H)
        self.activation = nn.Tanh()

    def forward(self, hidden_states):
        x = layers.to_compute(hidden_states)
        first = x[:, self.token_index] if x.dim() == 3 else x
        return ops.linear(first, self.dense.weight, self.dense.bias, act="tanh")


class BertText1Pooler(BertPooler):
    """pools token 1 instead of token 0 (reference mm_modeling.py:405-417)"""
    token_index = 1


class AttentionPooler(nn.Module):
    """tanh(W h + b) on every position (reference mm_modeling.py:148-157)"""

    def __init__(self, hidden_size):
        super().__init__()
        self.dense = nn.Linear(hidden_size, hidden_size)
        self.activation = nn.Tanh()

    def forward(self, hidden_states):
        return ops.linear(layers.to_compute(hidden_states), self.dense.weight, self.dense.bias, act="tanh")


class FeatureExtractor(nn.Module):
    """wraps the text encoder as `.cell` (reference mm_modeling.py:433-446)"""

    def __init__(self, pretrained_path):
        super().__init__()
        self.cell = RobertaModel.from_pretrained(pretrained_path)

    def forward(self, input_ids, token_type_ids, attention_mask):
        seq_out, pooled_out, enc_attentions = self.cell(input_ids=input_ids, token_type_ids=token_type_ids,
                                                        attention_mask=attention_mask, output_attentions=True)[:3]
        return seq_out, pooled_out, enc_attentions


class MultimodalDenoisingEncoder(nn.Module):
    """MDE is disabled in the reference (its instantiation is commented out at
    fcmf_pretraining.py:35) and is outside the hot path."""

    def __init__(self, alpha=0.7):
        super().__init__()
        self.alpha = alpha

    def forward(self, *a, **k):
        raise NotImplementedError("MultimodalDenoisingEncoder is unused by the reference training path")


# ---- IAOG decoder blocks: see iaog_modeling.py (imported at the bottom to keep this file focused) ----
from .iaog_modeling import (Attention, AddNorm, IAOGDecoder, PositionalEncoding, PositionWiseFFN,  # noqa: E402,F401
                            TransformerDecoderBlock)

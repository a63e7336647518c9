"""Geometry-aware multi-head attention over ROIs ("Relation Networks") on the HIP kernels.

Same surface as the reference's fcmf_framework/roi_modeling.py (clones, box_attention,
BoxMultiHeadedAttention with .linears[0..3], .WGs[0..7], BoxRelationalEmbedding).  The pairwise
geometry -> sin/cos embedding -> 8 x Linear(64,1) -> ReLU -> log-clamp chain is ONE kernel
(fcmf_box_bias_fwd) whose output enters the attention kernel as an additive bias; the
[B,N,N,64] embedding is never written to HBM on the training path.
"""
import copy
import math

import torch
import torch.nn as nn

from . import layers, ops


def clones(module, N):
    "Produce N identical layers."
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])


def box_attention(query, key, value, box_relation_embds_matrix, mask=None, dropout=None):
    """softmax(log(clamp(w_g, 1e-6)) + QK^T/sqrt(d_k)) V   (reference roi_modeling.py:14-47)
    query/key/value: [B, h, N, d_k]; box_relation_embds_matrix: [B, h, N, N] (>= 0)."""
    if mask is not None:
        raise NotImplementedError("box_attention is only ever called with mask=None (fcmf_pretraining.py:106-111)")
    B, h, N, dk = query.shape

    def merge(x):
        return layers.to_compute(x).transpose(1, 2).reshape(B, N, h * dk)
    bias = torch.log(torch.clamp(box_relation_embds_matrix.float(), min=1e-6))
    p = dropout.p if (dropout is not None and dropout.training) else 0.0
    out = ops.attention(merge(query), k1=merge(key), v1=merge(value), bias=bias, heads=h,
                        scale=1.0 / math.sqrt(dk), p=p, training=p > 0)
    return out.view(B, N, h, dk).transpose(1, 2), None


class BoxMultiHeadedAttention(nn.Module):
    def __init__(self, h, d_model, trignometric_embedding=True, legacy_extra_skip=False, dropout=0.1):
        super().__init__()
        assert d_model % h == 0
        if not trignometric_embedding:
            raise NotImplementedError("only the trigonometric (dim_g=64) embedding is used by FCMF")
        self.trignometric_embedding = trignometric_embedding
        self.legacy_extra_skip = legacy_extra_skip
        self.h = h
        self.d_k = d_model // h
        self.dim_g = 64
        self.linears = clones(nn.Linear(d_model, d_model), 4)
        self.WGs = clones(nn.Linear(self.dim_g, 1, bias=True), 8)
        self.attn = None
        self.box_attn = None
        self.dropout = nn.Dropout(p=dropout)

    def BoxRelationalEmbedding(self, f_g, dim_g=64, wave_len=1000, trignometric_embedding=True):
        """[B,N,4] (x_min,x_max,y_min,y_max) -> [B,N,N,64] float32 (reference roi_modeling.py:79-138;
        evaluated in f_g's dtype, returned in float32 as the reference casts it at :150)"""
        if dim_g != 64 or wave_len != 1000 or not trignometric_embedding:
            raise NotImplementedError("BoxRelationalEmbedding: only dim_g=64, wave_len=1000, trigonometric")
        return ops.box_embedding(f_g)

    def geometry_bias(self, input_box):
        """log(clamp(relu(WG_h(emb)), 1e-6)) for the 8 heads: [B,8,N,N] float32"""
        wg_w = torch.cat([l.weight for l in self.WGs], dim=0)            # [8,64]
        wg_b = torch.cat([l.bias for l in self.WGs], dim=0)              # [8]
        return ops.box_bias(input_box, wg_w, wg_b)

    def forward(self, input_query, input_key, input_value, input_box, mask=None):
        if mask is not None:
            raise NotImplementedError("BoxMultiHeadedAttention is only called with mask=None")
        q_in, k_in, v_in = (layers.to_compute(t) for t in (input_query, input_key, input_value))
        bias = self.geometry_bias(input_box)
        if input_query is input_key and input_key is input_value:      # (the only way FCMF calls it, fcmf_pretraining.py:108)
            q, k, v = ops.linear_multi(q_in, self.linears[0].weight, self.linears[0].bias, self.linears[1].weight, self.linears[1].bias,
                                       self.linears[2].weight, self.linears[2].bias)
        else:
            q = ops.linear(q_in, self.linears[0].weight, self.linears[0].bias)
            k = ops.linear(k_in, self.linears[1].weight, self.linears[1].bias)
            v = ops.linear(v_in, self.linears[2].weight, self.linears[2].bias)
        x = ops.attention(q, k1=k, v1=v, bias=bias, heads=self.h, scale=1.0 / math.sqrt(self.d_k),
                          p=self.dropout.p, training=self.training)
        if self.legacy_extra_skip:
            x = v_in + x
        return ops.linear(x, self.linears[3].weight, self.linears[3].bias)

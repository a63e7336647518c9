"""Functional building blocks shared by the RoBERTa text encoder and the FCMF fusion layers.

A "layer" here is the post-LN transformer layer that appears three times in the reference:
HF RobertaLayer, BertLayer (mm_modeling.py:331-342) and BertCrossAttentionLayer (:344-355).
All three share parameter sub-names (attention.self.{query,key,value}, attention.output.{dense,
LayerNorm}, intermediate.dense, output.{dense,LayerNorm}), so one functional form serves them.
"""
import torch

from . import ops


def additive_mask(mask01, length, value=-10000.0):
    """(1 - mask[:, :length]) * value as float32 [G, length]  (fcmf_pretraining.py:53-56,97-100,133-136;
    value = finfo(float32).min for the HF text encoder)"""
    m = mask01[:, :length]
    if m.dtype.is_floating_point:
        return (1.0 - m.to(torch.float32)) * value
    if m.is_cuda and m.dtype == torch.int64 and m.dim() == 2 and m.stride(1) == 1:
        return ops.additive_mask(m, float(value))                           # one launch, no torch arithmetic
    return (m - 1) * (-float(value))        # integer 0 / 1 masks: the same values ((1 - m) * value exactly)


def attn_sublayer_params(mod):
    """(Wq,bq,Wk,bk,Wv,bv) of a *.attention.self module"""
    return (mod.query.weight, mod.query.bias, mod.key.weight, mod.key.bias, mod.value.weight, mod.value.bias)


def post_attention(layer, ctx, residual, eps, p, training):
    """attention.output.dense -> dropout -> +residual -> LN -> FFN -> dropout -> +res -> LN
    (BertSelfOutput mm_modeling.py:276-280, BertIntermediate :311-314, BertOutput :324-328) as ONE
    autograd node (fused.PostAttentionFn)"""
    from .fused import PostAttentionFn
    ao = layer.attention.output
    p = float(p) if training else 0.0
    s0, s1 = (ops.next_seed(), ops.next_seed()) if p > 0 else (0, 0)
    return PostAttentionFn.apply(ctx, residual, ao.dense.weight, ao.dense.bias, ao.LayerNorm.weight, ao.LayerNorm.bias,
                                 layer.intermediate.dense.weight, layer.intermediate.dense.bias,
                                 layer.output.dense.weight, layer.output.dense.bias,
                                 layer.output.LayerNorm.weight, layer.output.LayerNorm.bias, float(eps), p, s0, s1)


def transformer_layer(layer, xq, xkv, add_mask, heads, eps, p_hidden, p_attn, training):
    """Full (unpruned) post-LN layer: xq [B,Tq,H] attends to xkv [B,Tk,H]; add_mask [B,Tk] float32.
    Self-attention (xq is xkv) runs as one fused node (fused.SelfLayerFn)."""
    sa = layer.attention.self
    if xq is xkv and xq.dim() == 3 and xq.shape[1] <= 512 and xq.shape[2] // heads <= 128:
        from .fused import SelfLayerFn
        ao = layer.attention.output
        ph = float(p_hidden) if training else 0.0
        pa = float(p_attn) if training else 0.0
        sa_seed = ops.next_seed() if pa > 0 else 0
        s0, s1 = (ops.next_seed(), ops.next_seed()) if ph > 0 else (0, 0)
        return SelfLayerFn.apply(xq, add_mask, sa.query.weight, sa.query.bias, sa.key.weight, sa.key.bias,
                                 sa.value.weight, sa.value.bias, ao.dense.weight, ao.dense.bias,
                                 ao.LayerNorm.weight, ao.LayerNorm.bias, layer.intermediate.dense.weight,
                                 layer.intermediate.dense.bias, layer.output.dense.weight, layer.output.dense.bias,
                                 layer.output.LayerNorm.weight, layer.output.LayerNorm.bias, heads, float(eps), ph, pa,
                                 sa_seed, s0, s1)
    q = ops.linear(xq, sa.query.weight, sa.query.bias)
    k = ops.linear(xkv, sa.key.weight, sa.key.bias)
    v = ops.linear(xkv, sa.value.weight, sa.value.bias)
    ctx = ops.attention(q, k1=k, v1=v, mask=add_mask, heads=heads, p=p_attn, training=training)
    return post_attention(layer, ctx, xq, eps, p_hidden, training)


def to_compute(x):
    """floating inputs enter the hot path in the configured activation dtype"""
    dt = ops.compute_dtype()
    if x.dtype in (torch.float32, torch.bfloat16, torch.float64, torch.float16) and x.dtype != dt:
        if x.dtype in (torch.float64, torch.float16):
            x = x.float()
        return ops.cast_ad(x, dt)
    return x

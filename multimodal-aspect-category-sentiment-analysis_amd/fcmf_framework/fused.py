"""Layer-level fused autograd Functions of the post-LN transformer layer (HF RobertaLayer,
BertLayer mm_modeling.py:331-342, BertCrossAttentionLayer :344-355).

One autograd node per layer instead of ~10, so that the backward can chain the kernels'
fused epilogues: the residual gradients are added by the dX GEMMs (FCMF_EPI_ADD), the bias gradients
of the two dense layers fall out of the LayerNorm backward (dxsum) and of the gelu' epilogue (colsum),
query/key/value run as ONE GEMM against the [3H,H] weight block the three nn.Linear parameters are
views of, and all weight-gradient buffers of a layer come from a single zero fill.
"""
import math

import os

import torch

from . import _hip as H
from . import ops


# --------------------------------------------------------------------------------------
# q/k/v parameters as views of one [3H, H] block
# --------------------------------------------------------------------------------------
class QKVStorageMixin:
    """nn.Module mixin: keeps self.query/key/value (nn.Linear) weights adjacent in one buffer so that
    the three projections are one GEMM.  State-dict keys, Parameter identities and shapes are unchanged."""

    def _fuse_qkv_storage(self):
        q, k, v = self.query, self.key, self.value
        Hh = q.weight.shape[0]
        W = torch.empty((3 * Hh, q.weight.shape[1]), dtype=q.weight.dtype, device=q.weight.device)
        b = torch.empty((3 * Hh,), dtype=q.bias.dtype, device=q.bias.device)
        for i, lin in enumerate((q, k, v)):
            W[i * Hh:(i + 1) * Hh].copy_(lin.weight.data)
            b[i * Hh:(i + 1) * Hh].copy_(lin.bias.data)
            lin.weight.data = W[i * Hh:(i + 1) * Hh]
            lin.bias.data = b[i * Hh:(i + 1) * Hh]
        self._qkv_w, self._qkv_b = W, b

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        self._fuse_qkv_storage()      # .to()/.cuda()/.float() re-allocate every parameter separately
        return out

    def qkv_params(self):
        """(W [3H,H], b [3H]) if the storage is still fused, else None"""
        q, k, v = self.query.weight, self.key.weight, self.value.weight
        n = q.numel() * q.element_size()
        if (k.data_ptr() == q.data_ptr() + n and v.data_ptr() == k.data_ptr() + n and q.is_contiguous()
                and self.key.bias.data_ptr() == self.query.bias.data_ptr() + self.query.bias.numel() * 4
                and self.value.bias.data_ptr() == self.key.bias.data_ptr() + self.key.bias.numel() * 4):
            return self._qkv_w, self._qkv_b
        return None


def _fused_weight(ws, dtype):
    """[sum(N_i), K] compute-dtype weight for a list of adjacent (or not) float32 parameters"""
    w0 = ws[0]
    n = w0.numel() * 4
    adjacent = all(ws[i + 1].data_ptr() == ws[i].data_ptr() + n for i in range(len(ws) - 1)) and w0.is_contiguous()
    if adjacent:
        flat = torch.as_strided(w0.detach(), (len(ws) * w0.shape[0],) + tuple(w0.shape[1:]),
                                (w0.stride(0),) + tuple(w0.stride()[1:])) if w0.dim() == 2 else \
            torch.as_strided(w0.detach(), (len(ws) * w0.shape[0],), (1,))
        if dtype == torch.float32:
            return flat
        # the shadow cache keys on (ptr, shape); `flat` shares w0's version counter, and FusedAdamW marks
        # every shadow it did not refresh itself as stale
        return ops.shadows.get(flat)
    cat = torch.cat([w.detach() for w in ws], 0)
    return cat if dtype == torch.float32 else ops.cast(cat, dtype)


# --------------------------------------------------------------------------------------
# weight-gradient GEMMs on a side HIP stream
# --------------------------------------------------------------------------------------
# dW = dY^T X depends only on tensors that already exist when it is issued and nothing downstream in the layer's
# backward reads it, so it CAN run on a second stream (FCMF_SIDE_STREAM=1); the main stream re-joins at the end of
# the layer's backward, before autograd sees the gradients.  It paid while the GEMM left LDS to spare; the 160 KiB
# persistent GEMM owns its CU, nothing overlaps any more (60.0 vs 59.8 ms/step measured), so the default is one stream.
USE_SIDE_STREAM = os.environ.get("FCMF_SIDE_STREAM", "0") == "1"
_side = {}


def _side_stream(device):
    st = _side.get(device)
    if st is None:
        st = _side[device] = torch.cuda.Stream(device=device)
    return st


class _SideGemms:
    """context for one layer backward: launch() runs a GEMM on the side stream after everything issued
    so far on the main stream; join() makes the main stream wait for all of them"""

    def __init__(self, device):
        self.dev = device
        self.on = USE_SIDE_STREAM and ops._gemm_trace is None   # event timing assumes one stream
        self.used = False

    def launch(self, tensors, *a, **kw):
        if not self.on:
            return ops.gemm(*a, **kw)
        main, side = torch.cuda.current_stream(self.dev), _side_stream(self.dev)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            ops.gemm(*a, **kw)
        for t in tensors:
            t.record_stream(side)     # keep the caching allocator from recycling operands early
        self.used = True

    def join(self):
        if self.on and self.used:
            torch.cuda.current_stream(self.dev).wait_stream(_side_stream(self.dev))


# --------------------------------------------------------------------------------------
# self-attention over a fused [G,T,3H] q|k|v buffer
# --------------------------------------------------------------------------------------
def _use_mfma(dtype, Hd, heads, T):
    return dtype == torch.bfloat16 and Hd // heads == 64 and T <= 256 and ops.USE_MFMA_ATTENTION


def _qkv_desc(qkv, G, T, Hd, heads, mask, scale, p, seed):
    es = qkv.element_size()
    a = H.AttnDesc()
    a.dtype, a.G, a.heads, a.d, a.R, a.T1, a.T2, a.group_div = H.dt(qkv), G, heads, Hd // heads, T, T, 0, 1
    a.q_sg, a.q_sr, a.k1_sg, a.k1_st = T * 3 * Hd, 3 * Hd, T * 3 * Hd, 3 * Hd
    a.o_sg, a.o_sr = T * Hd, Hd
    base = qkv.data_ptr()
    a.q, a.k1, a.v1 = base, base + Hd * es, base + 2 * Hd * es
    a.mask, a.scale, a.dropout_p, a.seed = H.ptr(mask), scale, p, seed
    return a


def self_attention_fwd(qkv, mask, G, T, Hd, heads, p, seed):
    scale = 1.0 / math.sqrt(Hd // heads)
    out = torch.empty((G * T, Hd), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((G, heads, T), dtype=torch.float32, device=qkv.device)
    es, base, L = qkv.element_size(), qkv.data_ptr(), H.lib()
    if _use_mfma(qkv.dtype, Hd, heads, T):
        H.check(L.fcmf_attn_mfma_fwd(base, base + Hd * es, base + 2 * Hd * es, H.ptr(mask), H.ptr(out), H.ptr(lse), G, heads,
                                     T, T, 3 * Hd, 3 * Hd, Hd, scale, p, seed, H.stream()), "fcmf_attn_mfma_fwd")
    else:
        a = _qkv_desc(qkv, G, T, Hd, heads, mask, scale, p, seed)
        H.check(L.fcmf_attn_small_fwd(a, H.ptr(out), H.ptr(lse), H.stream()), "fcmf_attn_small_fwd")
    return out, lse


def self_attention_bwd(qkv, mask, out, lse, dout, G, T, Hd, heads, p, seed, bias_grad=None):
    """-> dqkv [G*T, 3H].  bias_grad (float32 [3H], accumulated into): the column sums of dqkv = the gradient of the fused
    q|k|v bias; the MFMA kernel produces them per sequence from its f32 accumulators (no extra pass over dqkv)."""
    scale = 1.0 / math.sqrt(Hd // heads)
    es, base, L = qkv.element_size(), qkv.data_ptr(), H.lib()
    if _use_mfma(qkv.dtype, Hd, heads, T):
        dqkv = torch.empty_like(qkv)
        db = dqkv.data_ptr()
        part = None if bias_grad is None else torch.empty((G, 3 * Hd), dtype=torch.float32, device=qkv.device)
        H.check(L.fcmf_attn_mfma_bwd(base, base + Hd * es, base + 2 * Hd * es, H.ptr(mask), H.ptr(out), H.ptr(dout), H.ptr(lse),
                                     db, db + Hd * es, db + 2 * Hd * es, G, heads, T, T, 3 * Hd, 3 * Hd, Hd, scale, p, seed,
                                     H.ptr(part), H.stream()), "fcmf_attn_mfma_bwd")
        if part is not None:
            H.check(L.fcmf_colsum(H.ptr(part), H.ptr(bias_grad), G, 3 * Hd, 3 * Hd, H.dt(part), 1, H.stream()), "fcmf_colsum")
        return dqkv
    a = _qkv_desc(qkv, G, T, Hd, heads, mask, scale, p, seed)
    nch = max(1, (T + 127) // 128)
    dq = torch.empty((nch, G, T, Hd), dtype=qkv.dtype, device=qkv.device)
    dk = torch.empty((G, T, Hd), dtype=qkv.dtype, device=qkv.device)
    dv = torch.empty((G, T, Hd), dtype=qkv.dtype, device=qkv.device)
    H.check(L.fcmf_attn_small_bwd(a, H.ptr(out), H.ptr(dout), H.ptr(lse), H.ptr(dq), H.ptr(dk), H.ptr(dv), 0, 0, 0, H.stream()),
            "fcmf_attn_small_bwd")
    dqkv = torch.cat((ops._sum_leading(dq), dk, dv), dim=-1).view(G * T, 3 * Hd)
    if bias_grad is not None:
        H.check(L.fcmf_colsum(H.ptr(dqkv), H.ptr(bias_grad), G * T, 3 * Hd, 3 * Hd, H.dt(dqkv), 1, H.stream()), "fcmf_colsum")
    return dqkv


# --------------------------------------------------------------------------------------
# shared tail: out-proj -> add+LN -> FFN -> add+LN
# --------------------------------------------------------------------------------------
def _want_q(rows, Hd, dtype):
    """the fp8 mode quantises the LayerNorm output inside the LayerNorm kernel when an fp8 GEMM will consume it"""
    return ops.fp8_enabled() and dtype == torch.bfloat16 and Hd % 128 == 0 and rows >= 256


def _q_buffers(rows, Hd, device):
    return torch.empty((rows, Hd), dtype=torch.uint8, device=device), torch.empty(rows, dtype=torch.float32, device=device)


def _ln_fwd(x, res, res_ld, g, b, eps, p, seed, quant=False):
    """-> y, z, mean, rstd, yq (= (e4m3 y, row scales) when `quant`, else None)"""
    rows, Hd = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    # z (the pre-LN sum) overwrites x in place
    if quant:
        q, sc = _q_buffers(rows, Hd, x.device)
        H.check(H.lib().fcmf_add_ln_fwd_fp8(H.ptr(x), H.ptr(res), res_ld, H.ptr(g), H.ptr(b), H.ptr(y), H.ptr(x), H.ptr(mean),
                                            H.ptr(rstd), rows, Hd, eps, p, seed, H.dt(x), H.ptr(q), H.ptr(sc), H.stream()), "fcmf_add_ln_fwd_fp8")
        return y, x, mean, rstd, (q, sc)
    H.check(H.lib().fcmf_add_ln_fwd(H.ptr(x), H.ptr(res), res_ld, H.ptr(g), H.ptr(b), H.ptr(y), H.ptr(x), H.ptr(mean),
                                    H.ptr(rstd), rows, Hd, eps, p, seed, H.dt(x), H.stream()), "fcmf_add_ln_fwd")
    return y, x, mean, rstd, None


def _ln_bwd(dy, z, g, mean, rstd, p, seed, dg, db, dxsum, quant=False):
    """-> dz, dx (the gradient into the producing Linear), dxq (its e4m3 copy + row scales when `quant`)"""
    dz = torch.empty_like(z)
    dx = torch.empty_like(z) if p > 0 else None
    ws = ops.ln_workspace(z.shape[0], z.shape[1], z.device)
    if quant:
        q, sc = _q_buffers(z.shape[0], z.shape[1], z.device)
        H.check(H.lib().fcmf_add_ln_bwd_fp8(H.ptr(dy), H.ptr(z), H.ptr(g), H.ptr(mean), H.ptr(rstd), H.ptr(dz), H.ptr(dx), H.ptr(dg),
                                            H.ptr(db), H.ptr(dxsum), H.ptr(ws), z.shape[0], z.shape[1], p, seed, H.dt(z), H.ptr(q),
                                            H.ptr(sc), H.stream()), "fcmf_add_ln_bwd_fp8")
        return dz, (dx if dx is not None else dz), (q, sc)
    H.check(H.lib().fcmf_add_ln_bwd(H.ptr(dy), H.ptr(z), H.ptr(g), H.ptr(mean), H.ptr(rstd), H.ptr(dz), H.ptr(dx), H.ptr(dg),
                                    H.ptr(db), H.ptr(dxsum), H.ptr(ws), z.shape[0], z.shape[1], p, seed, H.dt(z), H.stream()),
            "fcmf_add_ln_bwd")
    return dz, (dx if dx is not None else dz), None


def _post_fwd(c, res, res_ld, wo, bo, g1, be1, w1, b1, w2, b2, g2, be2, eps, p, seeds):
    """c [M,H] attention context; res rows at stride res_ld.  Returns y and the tensors the backward needs."""
    dt = c.dtype
    M, Hd = c.shape
    I = w1.shape[0]
    co, c1, c2 = ops.as_compute(wo, dt), ops.as_compute(w1, dt), ops.as_compute(w2, dt)
    h = torch.empty((M, Hd), dtype=dt, device=c.device)
    ops.gemm_nt(c, wo, co, h, M, Hd, Hd, Hd, bias=bo.detach())
    h1, z1, m1, r1, h1q = _ln_fwd(h, res, res_ld, g1, be1, eps, p, seeds[0], quant=_want_q(M, Hd, dt))
    u = torch.empty((M, I), dtype=dt, device=c.device)
    a = torch.empty((M, I), dtype=dt, device=c.device)
    ops.gemm_nt(h1, w1, c1, a, M, I, Hd, Hd, bias=b1.detach(), aux=u, epi=H.EPI_GELU, xq=h1q)
    f = torch.empty((M, Hd), dtype=dt, device=c.device)
    ops.gemm_nt(a, w2, c2, f, M, Hd, I, I, bias=b2.detach())
    y, z2, m2, r2, _ = _ln_fwd(f, h1, Hd, g2, be2, eps, p, seeds[1])
    return y, (z1, m1, r1, h1, u, a, z2, m2, r2)


def _post_bwd(dy, c, saved, wo, w1, w2, g1, g2, p, seeds, G, side):
    """-> dc [M,H], dres (= dz1) [M,H]; weight / bias / LN gradients are accumulated into the views in G"""
    z1, m1, r1, h1, u, a, z2, m2, r2 = saved
    dt = c.dtype
    M, Hd = c.shape
    I = w1.shape[0]
    co, c1, c2 = ops.as_compute(wo, dt), ops.as_compute(w1, dt), ops.as_compute(w2, dt)
    dz2, df, dfq = _ln_bwd(dy, z2, g2, m2, r2, p, seeds[1], G["g2"], G["be2"], G["b2"], quant=_want_q(M, Hd, dt))
    du = torch.empty((M, I), dtype=dt, device=c.device)
    ops.gemm_dx(df, w2, c2, du, M, I, Hd, aux=u, epi=H.EPI_DGELU, colsum=G["b1"], dyq=dfq)      # (df W2) * gelu'(u); db1
    side.launch((df, a), df, a, G["w2"], Hd, I, M, Hd, I, I, 1, 1, acc=True)                   # dW2 = df^T a
    dh1 = torch.empty((M, Hd), dtype=dt, device=c.device)
    ops.gemm_dx(du, w1, c1, dh1, M, Hd, I, aux=dz2, epi=H.EPI_ADD)                             # du W1 + dz2 (residual)
    side.launch((du, h1), du, h1, G["w1"], I, Hd, M, I, Hd, Hd, 1, 1, acc=True)                # dW1 = du^T h1
    dz1, dh, dhq = _ln_bwd(dh1, z1, g1, m1, r1, p, seeds[0], G["g1"], G["be1"], G["bo"], quant=_want_q(M, Hd, dt))
    dc = torch.empty((M, Hd), dtype=dt, device=c.device)
    ops.gemm_dx(dh, wo, co, dc, M, Hd, Hd, dyq=dhq)
    side.launch((dh, c), dh, c, G["wo"], Hd, Hd, M, Hd, Hd, Hd, 1, 1, acc=True)
    return dc, dz1


def _grad_arena(device, Hd, I, with_qkv, params=None):
    """all float32 gradient buffers of one layer: the parameters' slices of the step's flat gradient arena when one is
    active (dp.GradArena: zeroed once per step), else views of ONE zero fill.  params: name -> Parameter (or the
    [q, k, v] list for "wqkv" / "bqkv")."""
    sizes = [("wo", (Hd, Hd)), ("w1", (I, Hd)), ("w2", (Hd, I)), ("bo", (Hd,)), ("b1", (I,)), ("b2", (Hd,)),
             ("g1", (Hd,)), ("be1", (Hd,)), ("g2", (Hd,)), ("be2", (Hd,))]
    if with_qkv:
        sizes = [("wqkv", (3 * Hd, Hd)), ("bqkv", (3 * Hd,))] + sizes
    arena = ops.grad_arena() if params is not None else None
    out = {"_inplace": set()}       # names whose buffer is the slice an EARLIER use of the same (shared) parameter returned: autograd gets None
    if arena is not None:
        for n, s in sizes:
            q = params.get(n)
            v = None
            if isinstance(q, (list, tuple)):
                v = arena.take_block(q)
                if v is None:
                    v = arena.retake_block(q)
                    if v is not None:
                        out["_inplace"].add(n)
            elif q is not None:
                v = arena.take(q)
                if v is None:
                    v = arena.retake(q)
                    if v is not None:
                        out["_inplace"].add(n)
            if v is not None:
                out[n] = v.view(s)
        sizes = [(n, s) for n, s in sizes if n not in out]
        if not sizes:
            return out
    total = sum(math.prod(s) for _, s in sizes)
    flat = torch.zeros(total, dtype=torch.float32, device=device)
    off = 0
    for n, s in sizes:
        k = math.prod(s)
        out[n] = flat[off:off + k].view(s)
        off += k
    return out


class PostAttentionFn(torch.autograd.Function):
    """attention.output.dense -> dropout -> +res -> LN -> FFN(GELU) -> dropout -> +res -> LN"""

    @staticmethod
    def forward(ctx, c, res, wo, bo, g1, be1, w1, b1, w2, b2, g2, be2, eps, p, seed0, seed1):
        c2 = ops._rows(c).contiguous()
        r2 = ops._rows(res)
        if r2.stride(1) != 1:
            r2 = r2.contiguous()
        y, saved = _post_fwd(c2, r2, ops._ld(r2), wo, bo, g1, be1, w1, b1, w2, b2, g2, be2, eps, p, (seed0, seed1))
        ctx.save_for_backward(c2, wo, w1, w2, g1, g2, bo, be1, b1, b2, be2, *saved)
        ctx.cfg = (p, seed0, seed1, c.shape, res.shape)
        return y.view(c.shape)

    @staticmethod
    def backward(ctx, dy):
        c2, wo, w1, w2, g1, g2, bo, be1, b1, b2, be2, *saved = ctx.saved_tensors
        p, s0, s1, cshape, rshape = ctx.cfg
        G = _grad_arena(c2.device, c2.shape[1], w1.shape[0], False,
                        dict(wo=wo, w1=w1, w2=w2, bo=bo, b1=b1, b2=b2, g1=g1, be1=be1, g2=g2, be2=be2))
        side = _SideGemms(c2.device)
        dc, dres = _post_bwd(dy.reshape(c2.shape).contiguous(), c2, saved, wo, w1, w2, g1, g2, p, (s0, s1), G, side)
        side.join()
        r = lambda n: None if n in G["_inplace"] else G[n]      # (accumulated in place: see dp.GradArena.retake)
        return (dc.view(cshape), dres.view(rshape), r("wo"), r("bo"), r("g1"), r("be1"), r("w1"), r("b1"), r("w2"), r("b2"),
                r("g2"), r("be2"), None, None, None, None)


class SelfLayerFn(torch.autograd.Function):
    """the whole self-attention layer: fused QKV GEMM -> attention -> PostAttention tail"""

    @staticmethod
    def forward(ctx, x, mask, wq, bq, wk, bk, wv, bv, wo, bo, g1, be1, w1, b1, w2, b2, g2, be2, heads, eps, p_h, p_a,
                seed_a, seed0, seed1):
        G_, T, Hd = x.shape
        M = G_ * T
        x2 = x.reshape(M, Hd)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        dt = x2.dtype
        wqkv = _fused_weight([wq, wk, wv], dt)
        bqkv = _fused_weight([bq, bk, bv], torch.float32)
        qkv = torch.empty((M, 3 * Hd), dtype=dt, device=x.device)
        wm = _fused_weight([wq, wk, wv], torch.float32)             # the [3H, H] float32 view when the storage is fused
        ops.gemm_nt(x2, wm if wm.data_ptr() == wq.data_ptr() else None, wqkv, qkv, M, 3 * Hd, Hd, Hd, bias=bqkv, owner=wq)
        mk = None if mask is None else mask.contiguous().float()
        c, lse = self_attention_fwd(qkv, mk, G_, T, Hd, heads, p_a, seed_a)
        y, saved = _post_fwd(c, x2, Hd, wo, bo, g1, be1, w1, b1, w2, b2, g2, be2, eps, p_h, (seed0, seed1))
        ctx.save_for_backward(x2, mk, qkv, c, lse, wq, wk, wv, wo, w1, w2, g1, g2, bq, bk, bv, bo, be1, b1, b2, be2, *saved)
        ctx.cfg = (heads, p_h, p_a, seed_a, seed0, seed1, x.shape)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, mk, qkv, c, lse, wq, wk, wv, wo, w1, w2, g1, g2, bq, bk, bv, bo, be1, b1, b2, be2, *saved = ctx.saved_tensors
        heads, p_h, p_a, seed_a, s0, s1, xshape = ctx.cfg
        G_, T, Hd = xshape
        M, dt = G_ * T, x2.dtype
        G = _grad_arena(x2.device, Hd, w1.shape[0], True,
                        dict(wqkv=[wq, wk, wv], bqkv=[bq, bk, bv], wo=wo, w1=w1, w2=w2, bo=bo, b1=b1, b2=b2, g1=g1, be1=be1,
                             g2=g2, be2=be2))
        side = _SideGemms(x2.device)
        dc, dz1 = _post_bwd(dy.reshape(M, Hd).contiguous(), c, saved, wo, w1, w2, g1, g2, p_h, (s0, s1), G, side)
        dqkv = self_attention_bwd(qkv, mk, c, lse, dc, G_, T, Hd, heads, p_a, seed_a, bias_grad=G["bqkv"])
        wqkv = _fused_weight([wq, wk, wv], dt)
        dx = torch.empty((M, Hd), dtype=dt, device=x2.device)
        wm = _fused_weight([wq, wk, wv], torch.float32)
        ops.gemm_dx(dqkv, wm if wm.data_ptr() == wq.data_ptr() else None, wqkv, dx, M, Hd, 3 * Hd, aux=dz1, epi=H.EPI_ADD, owner=wq)   # + residual gradient
        side.launch((dqkv, x2), dqkv, x2, G["wqkv"], 3 * Hd, Hd, M, 3 * Hd, Hd, Hd, 1, 1, acc=True)
        side.join()
        W, b = G["wqkv"], G["bqkv"]
        r = lambda n: None if n in G["_inplace"] else G[n]      # (accumulated in place: see dp.GradArena.retake)
        Ws = (None, None, None) if "wqkv" in G["_inplace"] else (W[:Hd], W[Hd:2 * Hd], W[2 * Hd:])
        bs = (None, None, None) if "bqkv" in G["_inplace"] else (b[:Hd], b[Hd:2 * Hd], b[2 * Hd:])
        return (dx.view(xshape), None, Ws[0], bs[0], Ws[1], bs[1], Ws[2], bs[2], r("wo"), r("bo"),
                r("g1"), r("be1"), r("w1"), r("b1"), r("w2"), r("b2"), r("g2"), r("be2"),
                None, None, None, None, None, None, None)

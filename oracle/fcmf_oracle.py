"""CPU ORACLE for the FCMF training hot path -- TEST INFRASTRUCTURE ONLY.

This file is a from-scratch, functional (no nn.Module) torch-CPU restatement of the
arithmetic the reference performs on the path named by BASELINE.json `north_star`:
FCMF fine-tune step = forward (6 aspects) + backward + clip + AdamW.

It is NOT part of the product: only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it.  The product path
(`multimodal-aspect-category-sentiment-analysis_amd/fcmf_framework`) never imports
anything from `oracle/` and fails loudly if the HIP library is missing.

Parity status: PINNED.  `oracle/make_golden.py` imports the reference package from
/root/reference in the build container, runs it on seeded inputs/weights produced by
`synthetic_data.py`, checks this restatement against it (<=1e-5) and commits the
reference's outputs as fixtures under tests/golden/.  `tests/test_oracle_golden.py`
re-checks this file against those fixtures on every run.

Every function cites the reference file:line (relative to /root/reference) it follows.
Parameters are passed as a flat dict {state_dict_key: tensor} using the reference's
own state-dict key names.
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# elementary pieces
# --------------------------------------------------------------------------------------
def gelu_erf(x):
    """mm_modeling.py:10-15  x * 0.5 * (1 + erf(x / sqrt(2)))"""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def tf_layer_norm(x, w, b, eps):
    """mm_modeling.py:167-171 (FCMFLayerNorm: biased variance, eps inside the sqrt).
    HF nn.LayerNorm(eps=cfg.layer_norm_eps) is the same formula with eps=1e-5
    (transformers modeling_roberta.py RobertaEmbeddings/RobertaSelfOutput/RobertaOutput)."""
    u = x.mean(-1, keepdim=True)
    s = (x - u).pow(2).mean(-1, keepdim=True)
    return w * ((x - u) / torch.sqrt(s + eps)) + b


def linear(x, P, name):
    return F.linear(x, P[name + ".weight"], P[name + ".bias"])


def dropout(x, p, training):
    return F.dropout(x, p=p, training=training) if (training and p > 0) else x


def split_heads(x, nh):
    """mm_modeling.py:188-191 transpose_for_scores: [B,T,H] -> [B,nh,T,H/nh]"""
    B, T, H = x.shape
    return x.view(B, T, nh, H // nh).permute(0, 2, 1, 3)


def merge_heads(x):
    """mm_modeling.py:216-218"""
    B, nh, T, d = x.shape
    return x.permute(0, 2, 1, 3).contiguous().view(B, T, nh * d)


def mha(P, prefix, xq, xkv, add_mask, nh, p_drop, training):
    """BertSelfAttention / BertCoAttention mm_modeling.py:193-219, :240-266 and HF
    eager_attention_forward (modeling_roberta.py:158-183): softmax(QK^T/sqrt(d)+mask) V
    with dropout on the probabilities.  `add_mask` broadcasts against [B,nh,Tq,Tk]."""
    q = split_heads(linear(xq, P, prefix + ".query"), nh)
    k = split_heads(linear(xkv, P, prefix + ".key"), nh)
    v = split_heads(linear(xkv, P, prefix + ".value"), nh)
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(q.shape[-1])
    if add_mask is not None:
        s = s + add_mask
    pr = dropout(torch.softmax(s, dim=-1), p_drop, training)
    return merge_heads(torch.matmul(pr, v))


def post_ln_layer(P, prefix, xq, xkv, add_mask, nh, eps, p_drop, training, hf_names=False):
    """One post-LN transformer layer.

    BertLayer mm_modeling.py:331-342 / BertCrossAttentionLayer :344-355 with
    BertSelfOutput :269-280, BertIntermediate :305-314, BertOutput :317-328.
    HF RobertaLayer has the identical graph (modeling_roberta.py RobertaAttention,
    RobertaIntermediate, RobertaOutput) and identical parameter sub-names."""
    ctx = mha(P, prefix + ".attention.self", xq, xkv, add_mask, nh, p_drop, training)
    h = dropout(linear(ctx, P, prefix + ".attention.output.dense"), p_drop, training)
    h1 = tf_layer_norm(h + xq, P[prefix + ".attention.output.LayerNorm.weight"],
                       P[prefix + ".attention.output.LayerNorm.bias"], eps)
    a = gelu_erf(linear(h1, P, prefix + ".intermediate.dense"))
    o = dropout(linear(a, P, prefix + ".output.dense"), p_drop, training)
    return tf_layer_norm(o + h1, P[prefix + ".output.LayerNorm.weight"],
                         P[prefix + ".output.LayerNorm.bias"], eps)


def pooler(P, prefix, x):
    """BertPooler mm_modeling.py:425-431: tanh(W x[:,0] + b)"""
    return torch.tanh(linear(x[:, 0], P, prefix + ".dense"))


# --------------------------------------------------------------------------------------
# HF RoBERTa text encoder (third party: transformers==4.40.0 pinned in requirements.txt:11,
# restated from the installed transformers 5.15 modeling_roberta.py; call site
# mm_modeling.py:436-446)
# --------------------------------------------------------------------------------------
def roberta_position_ids(input_ids, pad_id):
    """modeling_roberta.py create_position_ids_from_input_ids:
    cumsum(ids != pad) * (ids != pad) + pad"""
    m = input_ids.ne(pad_id).int()
    return (torch.cumsum(m, dim=1).type_as(m) * m).long() + pad_id


def roberta_forward(P, prefix, cfg, input_ids, token_type_ids, attention_mask, training=False):
    """RobertaModel forward, eager attention, post-LN, eps=cfg['layer_norm_eps'];
    padding keys get an additive finfo.min mask.  Returns sequence_output [B,S,H]
    (the pooled output and the attention probabilities are dead in training:
    fcmf_pretraining.py:41, fcmf_multimodal.py:42-45)."""
    p_drop = cfg.get("hidden_dropout_prob", 0.1)
    e = prefix + ".embeddings"
    pos = roberta_position_ids(input_ids, cfg["pad_token_id"])
    if token_type_ids is None:
        token_type_ids = torch.zeros_like(input_ids)
    # nn.Embedding(padding_idx=pad) for word AND position tables (modeling_roberta.py:61,72-74):
    # the pad row receives no gradient.
    pad = cfg["pad_token_id"]
    x = (F.embedding(input_ids, P[e + ".word_embeddings.weight"], padding_idx=pad)
         + P[e + ".token_type_embeddings.weight"][token_type_ids]
         + F.embedding(pos, P[e + ".position_embeddings.weight"], padding_idx=pad))
    x = tf_layer_norm(x, P[e + ".LayerNorm.weight"], P[e + ".LayerNorm.bias"], cfg["layer_norm_eps"])
    x = dropout(x, p_drop, training)
    ext = (1.0 - attention_mask[:, None, None, :].to(x.dtype)) * torch.finfo(x.dtype).min
    for l in range(cfg["num_hidden_layers"]):
        x = post_ln_layer(P, f"{prefix}.encoder.layer.{l}", x, x, ext, cfg["num_attention_heads"],
                          cfg["layer_norm_eps"], p_drop, training)
    return x


# --------------------------------------------------------------------------------------
# geometry-aware ROI attention (roi_modeling.py)
# --------------------------------------------------------------------------------------
def box_relational_embedding(f_g, dim_g=64, wave_len=1000):
    """roi_modeling.py:79-138.  f_g [B,N,4] = (x_min, x_max, y_min, y_max), computed in
    f_g's dtype (float64 in the fine-tune loop, vimacsa_dataset.py:189,199).
    Output [B,N,N,64] = [sin(100*delta*w_k) (4x8) | cos(...) (4x8)]."""
    B = f_g.size(0)
    x_min, x_max, y_min, y_max = torch.chunk(f_g, 4, dim=-1)
    cx = (x_min + x_max) * 0.5
    cy = (y_min + y_max) * 0.5
    w = (x_max - x_min) + 1.0
    h = (y_max - y_min) + 1.0
    dx = torch.log(torch.clamp(torch.abs((cx - cx.view(B, 1, -1)) / w), min=1e-3))
    dy = torch.log(torch.clamp(torch.abs((cy - cy.view(B, 1, -1)) / h), min=1e-3))
    dw = torch.log(w / w.view(B, 1, -1))
    dh = torch.log(h / h.view(B, 1, -1))
    pos = torch.stack((dx, dy, dw, dh), dim=-1)                       # [B,N,N,4]
    feat_range = torch.arange(dim_g / 8)                              # float32, :123
    dim_mat = 1.0 / torch.pow(wave_len, feat_range / (dim_g / 8))     # [8] float32
    mul = (100.0 * pos).unsqueeze(-1) * dim_mat.view(1, 1, 1, 1, -1)  # [B,N,N,4,8]
    mul = mul.reshape(B, pos.shape[1], pos.shape[2], -1)              # [B,N,N,32]
    return torch.cat((torch.sin(mul), torch.cos(mul)), dim=-1)


def box_head(P, prefix, x, boxes, p_drop, training, h=8):
    """BoxMultiHeadedAttention.forward roi_modeling.py:140-180 with box_attention :14-47
    (query = key = value = x, mask None as at the only call site fcmf_pretraining.py:106-111)."""
    B, N, H = x.shape
    d_k = H // h
    emb = box_relational_embedding(boxes).to(x.dtype)                 # :148-150
    q, k, v = [F.linear(x, P[f"{prefix}.linears.{i}.weight"], P[f"{prefix}.linears.{i}.bias"])
               .view(B, N, h, d_k).transpose(1, 2) for i in range(3)]
    wg = torch.cat([F.linear(emb, P[f"{prefix}.WGs.{i}.weight"], P[f"{prefix}.WGs.{i}.bias"])
                    .view(B, 1, N, N) for i in range(h)], dim=1)      # :161-162
    wg = F.relu(wg)                                                   # :163
    s = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(d_k)         # :29-30
    s = torch.log(torch.clamp(wg, min=1e-6)) + s                      # :40
    pr = dropout(torch.softmax(s, dim=-1), p_drop, training)          # :41-43
    o = torch.matmul(pr, v).transpose(1, 2).contiguous().view(B, N, H)
    return F.linear(o, P[f"{prefix}.linears.3.weight"], P[f"{prefix}.linears.3.bias"])


# --------------------------------------------------------------------------------------
# FCMF encoder / classifier / loss
# --------------------------------------------------------------------------------------
def fcmf_encoder_forward(P, cfg, input_ids, visual_embeds_att, roi_embeds_att, roi_coors,
                         token_type_ids, attention_mask, added_attention_mask,
                         num_imgs, num_roi, training=False, prefix="encoder", collect=None):
    """FCMFEncoder.forward fcmf_pretraining.py:39-141 (dense, as written: no pruning,
    no hoisting).  Returns fusion output [B, 1+2*num_imgs, H]."""
    nh = cfg["num_attention_heads"]
    pd = cfg.get("fcmf_dropout", 0.1)
    eps = 1e-12
    seq = roberta_forward(P, prefix + ".bert.cell", cfg, input_ids, token_type_ids, attention_mask, training)
    S = seq.size(1)
    if collect is not None:
        collect["sequence_output"] = seq
    hs, rs = [], []
    for i in range(num_imgs):
        img = linear(visual_embeds_att[:, i], P, prefix + ".vismap2text")               # :49-50
        m_img = (1.0 - added_attention_mask[:, :49][:, None, None, :].to(img.dtype)) * -10000.0  # :53-56
        t2i = post_ln_layer(P, prefix + ".text2img_attention.layer.0", seq, img, m_img, nh, eps, pd, training)
        hs.append(pooler(P, prefix + ".text2img_pooler", t2i).unsqueeze(1))             # :89-93
        m_roi = added_attention_mask[:, :S + num_roi][:, None, None, :]                 # :97-100
        m_roi = (1.0 - m_roi) * -10000.0
        roi = linear(roi_embeds_att[:, i], P, prefix + ".roimap2text")                  # :102-103
        rel = box_head(P, prefix + ".box_head", roi, roi_coors[:, i], 0.1, training)    # :106-111
        tr = torch.cat((seq, rel), dim=1)                                               # :114
        mm = post_ln_layer(P, prefix + ".mm_attention.layer.0", tr, tr, m_roi.to(tr.dtype), nh, eps, pd, training)
        rs.append(pooler(P, prefix + ".text2roi_pooler", mm).unsqueeze(1))              # :121-124
        if collect is not None and i == 0:
            collect["img0_proj"] = img
            collect["t2i0"] = t2i
            collect["rel0"] = rel
            collect["mm0"] = mm
    fusion = torch.cat([seq[:, 0:1]] + hs + rs, dim=1)                                  # :127-131
    m_f = added_attention_mask[:, :1 + 2 * num_imgs][:, None, None, :]                  # :133-136
    m_f = ((1.0 - m_f) * -10000.0).to(fusion.dtype)
    out = post_ln_layer(P, prefix + ".mm_attention.layer.0", fusion, fusion, m_f, nh, eps, pd, training)  # :139
    if collect is not None:
        collect["fusion_in"] = fusion
        collect["fusion_out"] = out
    return out


def fcmf_forward(P, cfg, input_ids, visual_embeds_att, roi_embeds_att, roi_coors,
                 token_type_ids, attention_mask, added_attention_mask, num_imgs, num_roi,
                 training=False, collect=None):
    """FCMF.forward fcmf_multimodal.py:39-51 -> logits [B, num_labels]"""
    enc = fcmf_encoder_forward(P, cfg, input_ids, visual_embeds_att, roi_embeds_att, roi_coors,
                               token_type_ids, attention_mask, added_attention_mask,
                               num_imgs, num_roi, training, collect=collect)
    pooled = dropout(pooler(P, "text_pooler", enc), cfg.get("fcmf_dropout", 0.1), training)
    return linear(pooled, P, "classifier")


def fcmf_step_loss(P, cfg, batch, num_imgs, num_roi, training=False, accum=1):
    """Loop over aspects run_multimodal_fcmf.py:463-478: sum over aspects of the
    batch-mean CE, divided by gradient_accumulation_steps.  Returns (loss, logits[B,A,C])."""
    A = batch["input_ids"].shape[1]
    total = 0.0
    logits_all = []
    for a in range(A):
        lg = fcmf_forward(P, cfg, batch["input_ids"][:, a], batch["visual_embeds_att"],
                          batch["roi_embeds_att"], batch["roi_coors"],
                          batch["token_type_ids"][:, a], batch["attention_mask"][:, a],
                          batch["added_attention_mask"][:, a], num_imgs, num_roi, training)
        total = total + F.cross_entropy(lg, batch["labels"][:, a])
        logits_all.append(lg)
    if accum > 1:
        total = total / accum
    return total, torch.stack(logits_all, dim=1)


# --------------------------------------------------------------------------------------
# optimizer side
# --------------------------------------------------------------------------------------
NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight")
HEAD_NAMES = ("classifier", "text_pooler")


def fcmf_param_groups(names, lr_enc=7e-5, lr_head=7e-4):
    """run_multimodal_fcmf.py:249-287: 4 groups by substring match on the names."""
    groups = [dict(names=[], weight_decay=0.01, lr=lr_enc), dict(names=[], weight_decay=0.0, lr=lr_enc),
              dict(names=[], weight_decay=0.01, lr=lr_head), dict(names=[], weight_decay=0.0, lr=lr_head)]
    for n in names:
        head = any(h in n for h in HEAD_NAMES)
        nd = any(x in n for x in NO_DECAY)
        groups[(2 if head else 0) + (1 if nd else 0)]["names"].append(n)
    return groups


def clip_grad_norm(grads, max_norm=1.0):
    """torch.nn.utils.clip_grad_norm_ (run_multimodal_fcmf.py:485): global L2 norm,
    coef = max_norm / (norm + 1e-6) clamped to 1."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return {k: g * coef for k, g in grads.items()}, total


def linear_schedule_factor(step, warmup, total):
    """transformers get_linear_schedule_with_warmup (run_multimodal_fcmf.py:310-314)"""
    if step < warmup:
        return float(step) / float(max(1, warmup))
    return max(0.0, float(total - step) / float(max(1, total - warmup)))


def adamw_update(p, g, m, v, step, lr, wd, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.AdamW single-tensor semantics (run_multimodal_fcmf.py:289):
    decoupled wd, bias-corrected, eps outside sqrt(v_hat).  `step` is 1-based."""
    p = p * (1.0 - lr * wd)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


def bertadam_update(p, g, m, v, step0, lr, wd, t_total=-1, warmup=-1, schedule="warmup_linear",
                    b1=0.9, b2=0.999, e=1e-6, max_grad_norm=1.0):
    """BertAdam.step optimization.py:94-162: per-parameter clip (:127-128), no bias
    correction, wd added to the update (:143-144), lr from the warmup schedule evaluated at
    the pre-increment step count (:146-150).  `step0` is the 0-based state['step']."""
    if max_grad_norm > 0:
        n = g.norm(2)
        g = g * torch.clamp(max_grad_norm / (n + 1e-6), max=1.0)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    upd = m / (v.sqrt() + e)
    if wd > 0:
        upd = upd + wd * p
    if t_total != -1:
        x = step0 / t_total
        if schedule == "warmup_linear":
            f = x / warmup if x < warmup else 1.0 - x
        elif schedule == "warmup_constant":
            f = x / warmup if x < warmup else 1.0
        else:
            f = x / warmup if x < warmup else 0.5 * (1.0 + math.cos(math.pi * x))
        lr = lr * f
    return p - lr * upd, m, v


# --------------------------------------------------------------------------------------
# IAOG seq2seq decoder (fcmf_pretraining.py:143-207, mm_modeling.py:35-132,558-666)
# --------------------------------------------------------------------------------------
def iaog_attention(P, prefix, k_in, q_in, causal, nh):
    """Attention.forward mm_modeling.py:66-132: per-head weights, VALUES = projected KEYS
    (:129); any 2-D memory_len => tril(q_len,k_len) mask filled with -1e4 (:115-124)."""
    B, Tk, H = k_in.shape
    Tq = q_in.shape[1]
    wk, wq = P[prefix + ".w_kx"], P[prefix + ".w_qx"]               # [nh,H,d]
    # Reference quirk (mm_modeling.py:79-85): inputs are tiled head-major (row i = slot*B + b,
    # `k.repeat(n_head,1,1)`) but the weights are tiled batch-major (`w_kx.repeat(mb,1,1)`,
    # row i = r*n_head + h), so output slot s of batch element b is projected with head
    # (s*B + b) % n_head -- the pairing depends on the batch size.  Reproduced exactly.
    slot = torch.arange(nh).view(nh, 1)
    hh = (slot * B + torch.arange(B).view(1, B)) % nh                # [nh(slot), B]
    kx = torch.einsum("bth,sbhd->sbtd", k_in, wk[hh])
    qx = torch.einsum("bth,sbhd->sbtd", q_in, wq[hh])
    s = torch.matmul(qx, kx.transpose(-1, -2)) / math.sqrt(wk.shape[-1])
    if causal:
        tri = torch.tril(torch.ones(Tq, Tk))
        s = s.masked_fill(tri == 0, -1e4)
    pr = torch.softmax(s, dim=-1)
    o = torch.matmul(pr, kx)                                        # [nh,B,Tq,d]
    o = o.permute(1, 2, 0, 3).reshape(B, Tq, -1)                    # cat over heads :130
    return F.linear(o, P[prefix + ".proj.weight"], P[prefix + ".proj.bias"])


def positional_table(max_len, H):
    """PositionalEncoding mm_modeling.py:621-626"""
    Pm = torch.zeros((max_len, H))
    X = torch.arange(max_len, dtype=torch.float32).reshape(-1, 1) / torch.pow(
        10000, torch.arange(0, H, 2, dtype=torch.float32) / H)
    Pm[:, 0::2] = torch.sin(X)
    Pm[:, 1::2] = torch.cos(X)
    return Pm


def iaog_decoder_forward(P, cfg, dec_X, enc_out, training=False, prefix="decoder",
                         emb_key="decoder.embedding.weight", out_w_key="decoder.dense.weight", is_train=True):
    """IAOGDecoder.forward mm_modeling.py:649-662 in teacher-forced training mode
    (is_train=True): causal self-attention, and -- because the encoder mask handed down is
    2-D (fcmf_pretraining.py:184-199) -- the SAME tril rule on the cross attention.
    is_train=False (the decode call of the commented-out beam search, fcmf_pretraining.py:470): dec_valid_lens = None
    (mm_modeling.py:595-596) and no encoder mask is handed down (cross_mask = enc_valid_lens = None, :606), so neither
    attention is masked; the per-block cache state[2][i] stays None (:584-588 assign it only when it already is a tensor)."""
    H = cfg["hidden_size"]
    nh = cfg["num_attention_heads"]
    causal = bool(is_train)
    x = P[emb_key][dec_X] * math.sqrt(H) + positional_table(512, H)[: dec_X.shape[1]].to(enc_out.dtype)
    x = dropout(x, 0.1, training)
    for i in range(cfg["num_hidden_layers"]):
        b = f"{prefix}.blks.block{i}"
        x2 = iaog_attention(P, b + ".attention1", x, x, causal, nh)                 # :601
        y = tf_layer_norm(dropout(x2, 0.1, training) + x, P[b + ".addnorm1.ln.weight"], P[b + ".addnorm1.ln.bias"], 1e-12)
        y2 = iaog_attention(P, b + ".attention2", enc_out, y, causal, nh)           # :610
        z = tf_layer_norm(dropout(y2, 0.1, training) + y, P[b + ".addnorm2.ln.weight"], P[b + ".addnorm2.ln.bias"], 1e-12)
        f = F.linear(gelu_erf(F.linear(z, P[b + ".ffn.dense1.weight"], P[b + ".ffn.dense1.bias"])),
                     P[b + ".ffn.dense2.weight"], P[b + ".ffn.dense2.bias"])
        x = tf_layer_norm(dropout(f, 0.1, training) + z, P[b + ".add_norm3.ln.weight"], P[b + ".add_norm3.ln.bias"], 1e-12)
    return F.linear(x, P[out_w_key], P[prefix + ".dense.bias"])                     # :662


def beam_search_ids(step_logprobs, start_id, sep_id, beam_size=3, max_len=20):
    """The beam search the reference keeps (commented out) in fcmf_pretraining.py:383-517, on token ids.
    `step_logprobs(seq)` = log_softmax of the decoder's logits for the LAST token of `seq` (a list of ids) -- the
    reference feeds ONLY that token ([1, 1], :452) to a decoder whose per-block cache is never filled, so the step sees
    the last token at position 0 and the encoder output, nothing else.
      beams start as [(0.0, [start])] (:430-436); per step every live beam is expanded by its top `beam_size` tokens
      (:476-486); a beam whose last token is SEP moves to the finished list instead (:444-447); the candidates are
      sorted by score, descending and stable (:493), the best `beam_size` survive (:497); all survivors finished ->
      they are added to the finished list and the loop ends (:500-502); nothing finished by max_len -> the live beams
      are the finished list (:505-506); the best finished sequence wins, first in list order on ties (:509).
    Returns (ids of the best sequence incl. the start token, its score, the finished list [(score, ids)])."""
    beams = [(0.0, [int(start_id)])]
    final = []
    for _ in range(max_len):
        cands = []
        for score, seq in beams:
            if seq[-1] == sep_id:
                final.append((score, seq))
                continue
            lp = step_logprobs(seq)
            top_s, top_i = torch.topk(lp, beam_size)
            for k in range(beam_size):
                cands.append((score + top_s[k].item(), seq + [int(top_i[k])]))
        if not cands:
            break
        beams = sorted(cands, key=lambda c: c[0], reverse=True)[:beam_size]
        if all(seq[-1] == sep_id for _, seq in beams):
            final.extend(beams)
            break
    if not final:
        final = list(beams)
    best_score, best_seq = sorted(final, key=lambda c: c[0], reverse=True)[0]
    return best_seq, best_score, final

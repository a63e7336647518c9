"""FCMFEncoder (the fine-grained cross-modal fusion graph) and FCMFSeq2Seq (IAOG pre-training)
on the MI355X kernels.  Same constructor / forward signatures, attribute tree and state-dict keys as
the reference's fcmf_framework/fcmf_pretraining.py:14-221.

MI355X-first restructuring of FCMFEncoder.forward (fcmf_pretraining.py:39-141) -- results are
mathematically identical, proven against the oracle in tests/:
  * aspect batching: the 6 per-aspect forwards of one step (run_multimodal_fcmf.py:463-475) run
    as ONE batch of B*A sequences (`encode_aspects`), so every GEMM sees M = B*A*S rows;
  * image batching: the per-image Python loop (:47) becomes batched GEMMs over B*NI images;
  * hoisting: vismap2text / roimap2text / box_head and the key/value projections of image
    patches and ROIs do not depend on the aspect and run once per step, not 6x;
  * dead-row pruning: only row 0 of text2img_attention / per-image mm_attention outputs is ever
    consumed (BertPooler, mm_modeling.py:428), so queries, output projection, LayerNorms and FFN
    run on that row only; keys/values still cover all rows.  The text rows' key/value
    projection in mm_attention is shared by the 7 images (same sequence_output, same weights).
"""
import torch
import torch.nn as nn

from . import layers, ops
from .mm_modeling import *  # noqa: F401,F403  (reference does the same; exposes the constants)
from .mm_modeling import BertCrossEncoder, BertPooler, FeatureExtractor, IAOGDecoder, MultimodalEncoder
from .decoding import beam_search  # noqa: F401  (the reference keeps it in this module, :383-517, commented out)
from .roi_modeling import *  # noqa: F401,F403
from .roi_modeling import BoxMultiHeadedAttention


class FCMFEncoder(nn.Module):
    def __init__(self, pretrained_hf_path, num_imgs=7, num_roi=4, alpha=0.7):
        super().__init__()
        self.num_imgs = num_imgs
        self.num_roi = num_roi
        self.alpha = alpha
        self.bert = FeatureExtractor(pretrained_hf_path)
        cfg = self.bert.cell.config
        H, nh, I = cfg.hidden_size, cfg.num_attention_heads, cfg.intermediate_size
        self.vismap2text = nn.Linear(2048, H)
        self.roimap2text = nn.Linear(2048, H)
        self.box_head = BoxMultiHeadedAttention(8, H)
        self.text2img_attention = BertCrossEncoder(H, nh, I)
        self.text2img_pooler = BertPooler(H)
        self.text2roi_pooler = BertPooler(H)
        self.mm_attention = MultimodalEncoder(H, nh, I)

    # ------------------------------------------------------------------------------------
    def encode_aspects(self, input_ids, visual_embeds_att, roi_embeds_att, roi_coors, token_type_ids,
                       attention_mask, added_attention_mask):
        """input_ids / token_type_ids / attention_mask [B,A,S]; added_attention_mask [B,A,>=S+num_roi];
        visual_embeds_att [B,NI,49,F]; roi_embeds_att [B,NI,NR,F]; roi_coors [B,NI,NR,4].
        Returns the fusion-layer output [B*A, 1+2*NI, H] (row b*A+a = sample b, aspect a)."""
        B, A, S = input_ids.shape
        Bt, NI, NR = B * A, self.num_imgs, self.num_roi
        tr = self.training
        cell = self.bert.cell
        H = cell.config.hidden_size
        nh = cell.config.num_attention_heads
        P = visual_embeds_att.shape[2]
        ids = input_ids.reshape(Bt, S)
        tt = None if token_type_ids is None else token_type_ids.reshape(Bt, S)
        am = None if attention_mask is None else attention_mask.reshape(Bt, S)
        added = added_attention_mask.reshape(Bt, -1)

        # 1. text encoder on all B*A sequences at once                      (fcmf_pretraining.py:41)
        seq = cell.encode(ids, tt, am)                                                   # [Bt,S,H]

        cross = self.text2img_attention.layer[0]
        mm = self.mm_attention.layer[0]
        csa, msa = cross.attention.self, mm.attention.self
        # the sequence's three consumers -- text keys / values of the mm layer (:97-124) and the [CLS] row -- as one autograd node
        Kt, Vt, cls = ops.seq_fan(seq, msa.key.weight, msa.key.bias, msa.value.weight, msa.value.bias)   # [Bt,S,H] x2, [Bt,H]
        eps = mm.output.LayerNorm.variance_epsilon
        p_h, p_a = mm.output.dropout.p, msa.dropout.p

        # 2. aspect-independent image side, hoisted out of the aspect loop     (:49-50, :102-111)
        vis = layers.to_compute(visual_embeds_att[:, :NI]).reshape(B * NI * P, -1)
        img = ops.linear(vis, self.vismap2text.weight, self.vismap2text.bias)            # [B*NI*P,H]
        Kc, Vc = (t.view(B, NI, P, H) for t in ops.linear_multi(img, csa.key.weight, csa.key.bias, csa.value.weight, csa.value.bias))
        roi = layers.to_compute(roi_embeds_att[:, :NI]).reshape(B * NI * NR, -1)
        roi_p = ops.linear(roi, self.roimap2text.weight, self.roimap2text.bias).view(B * NI, NR, H)
        rel = self.box_head(roi_p, roi_p, roi_p, roi_coors[:, :NI].reshape(B * NI, NR, 4))  # [B*NI,NR,H]
        Kr, Vr = (t.view(B, NI, NR, H) for t in ops.linear_multi(rel, msa.key.weight, msa.key.bias, msa.value.weight, msa.value.bias))

        cls_rep = cls.unsqueeze(1).expand(Bt, NI, H)

        # 3. text -> image-patch cross attention, live row 0 only              (:84-93)
        qc = ops.linear(cls, csa.query.weight, csa.query.bias)                           # [Bt,H]
        m_img = layers.additive_mask(added, P)                                           # (:53-56)
        ctx_c = ops.attention(qc.unsqueeze(1).expand(Bt, NI, H), k2=Kc, v2=Vc, mask=m_img, heads=nh,
                              group_div=A, p=csa.dropout.p, training=tr)                 # [Bt,NI,H]
        t2i = layers.post_attention(cross, ctx_c, cls_rep, eps, p_h, tr)
        pl = self.text2img_pooler
        h_feat = ops.linear(t2i, pl.dense.weight, pl.dense.bias, act="tanh")             # [Bt,NI,H]

        # 4. text+ROI multimodal layer, live row 0 only                        (:97-124)
        qm = ops.linear(cls, msa.query.weight, msa.query.bias)
        m_roi = layers.additive_mask(added, S + NR)                                      # (:97-100)
        ctx_m = ops.attention(qm.unsqueeze(1).expand(Bt, NI, H), k1=Kt, v1=Vt, k2=Kr, v2=Vr, mask=m_roi,
                              heads=nh, group_div=A, p=p_a, training=tr)                 # [Bt,NI,H]
        mmo = layers.post_attention(mm, ctx_m, cls_rep, eps, p_h, tr)
        pr = self.text2roi_pooler
        r_feat = ops.linear(mmo, pr.dense.weight, pr.dense.bias, act="tanh")             # [Bt,NI,H]

        # 5. fusion: [CLS] + image features + ROI features through the SAME mm layer (:127-140)
        fusion = torch.cat((cls.unsqueeze(1), h_feat, r_feat), dim=1)                    # [Bt,1+2NI,H]
        m_f = layers.additive_mask(added, 1 + 2 * NI)
        return layers.transformer_layer(mm, fusion, fusion, m_f, nh, eps, p_h, p_a, tr)

    def forward(self, input_ids, visual_embeds_att, roi_embeds_att, roi_coors=None, token_type_ids=None,
                attention_mask=None, added_attention_mask=None):
        out = self.encode_aspects(input_ids.unsqueeze(1), visual_embeds_att, roi_embeds_att, roi_coors,
                                  None if token_type_ids is None else token_type_ids.unsqueeze(1),
                                  None if attention_mask is None else attention_mask.unsqueeze(1),
                                  added_attention_mask.unsqueeze(1))
        # the reference also returns the text encoder's attention probabilities; the fused
        # kernels never materialise them (they are unused in training)
        return out, ()


class FCMFSeq2Seq(nn.Module):
    def __init__(self, vocab_size, max_len_decoder, pretrained_hf_path, num_imgs, num_roi, alpha):
        super().__init__()
        self.encoder = FCMFEncoder(pretrained_hf_path, num_imgs=num_imgs, num_roi=num_roi, alpha=alpha)
        cfg = self.encoder.bert.cell.config
        self.decoder = IAOGDecoder(vocab_size=vocab_size, hidden_size=cfg.hidden_size,
                                   num_layers=cfg.num_hidden_layers, num_heads=cfg.num_attention_heads)
        self.num_imgs = num_imgs
        # N(0, 0.02) re-initialisation of Linear / Embedding weights (reference :150-155)
        self.decoder.apply(self._init_weights)
        self.encoder.vismap2text.apply(self._init_weights)
        self.encoder.roimap2text.apply(self._init_weights)
        self.encoder.box_head.apply(self._init_weights)
        self.encoder.text2img_attention.apply(self._init_weights)
        self.encoder.mm_attention.apply(self._init_weights)
        if hasattr(self.encoder.bert.cell, 'resize_token_embeddings'):
            self.encoder.bert.cell.resize_token_embeddings(vocab_size)
        # weight tying (reference :163-166)
        if hasattr(self.encoder.bert.cell, 'embeddings'):
            self.decoder.embedding.weight = self.encoder.bert.cell.embeddings.word_embeddings.weight
        self.decoder.dense.weight = self.decoder.embedding.weight

    def _decoder_state(self, enc_X, visual_embeds_att, roi_embeds_att, roi_coors, token_type_ids, attention_mask,
                       added_attention_mask):
        enc_output, enc_attentions = self.encoder(enc_X, visual_embeds_att, roi_embeds_att, roi_coors,
                                                  token_type_ids, attention_mask, added_attention_mask)
        num_visual_tokens = self.num_imgs * 2
        current_text_len = enc_output.size(1) - num_visual_tokens
        text_mask = attention_mask[:, :current_text_len]
        vis_mask = torch.ones((text_mask.size(0), num_visual_tokens), device=text_mask.device, dtype=text_mask.dtype)
        combined_mask = torch.cat((text_mask, vis_mask), dim=1)     # 2-D => tril rule in the decoder (:184-199)
        return [enc_output, combined_mask, [None] * self.decoder.num_blks], enc_attentions

    def forward(self, enc_X, dec_X, visual_embeds_att, roi_embeds_att, roi_coors=None, token_type_ids=None,
                attention_mask=None, added_attention_mask=None, source_valid_len=None, is_train=True):
        dec_state, enc_attentions = self._decoder_state(enc_X, visual_embeds_att, roi_embeds_att, roi_coors,
                                                        token_type_ids, attention_mask, added_attention_mask)
        logits = self.decoder(dec_X, dec_state, is_train=is_train)
        if not is_train:
            return logits, enc_attentions
        return logits

    def forward_loss(self, enc_X, dec_X, labels, visual_embeds_att, roi_embeds_att, roi_coors=None, token_type_ids=None,
                     attention_mask=None, added_attention_mask=None, ignore_index=-100):
        """the training step's `CrossEntropyLoss(ignore_index=-100)(forward(...).permute(0, 2, 1), labels)`
        (run_pretraining_fcmf.py:309-324) as one call: same value, the 64001-wide logits stay inside the fused
        projection + loss node (ops.VocabCrossEntropyFn)"""
        dec_state, _ = self._decoder_state(enc_X, visual_embeds_att, roi_embeds_att, roi_coors, token_type_ids,
                                           attention_mask, added_attention_mask)
        return self.decoder.loss(dec_X, dec_state, labels, ignore_index=ignore_index)

    def _init_weights(self, module):
        if isinstance(module, nn.Linear):
            module.weight.data.normal_(mean=0.0, std=0.02)
            if module.bias is not None:
                module.bias.data.zero_()
        elif isinstance(module, nn.LayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)
        elif isinstance(module, nn.Embedding):
            module.weight.data.normal_(mean=0.0, std=0.02)
            if module.padding_idx is not None:
                module.weight.data[module.padding_idx].zero_()

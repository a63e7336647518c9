"""IAOG decode: the beam search the reference keeps beside FCMFSeq2Seq (fcmf_pretraining.py:383-517, commented out there,
as are its call sites run_pretraining_fcmf.py:405-414,523-533) with the same signature, semantics and return value.

What the reference's loop does, and what this module does with it on the GPU:
  * the encoder runs ONCE per sample (:404-418); here also the per-block cross-attention key projections of its output
    (`w_kx` of all decoder blocks, one GEMM -- `IAOGDecoder.project_encoder`) run once per sample instead of once per beam
    and step;
  * every step feeds ONLY the last token of a beam to the decoder ([1, 1], :452,470) and the decoder's per-block cache
    is never filled (mm_modeling.py:584-588 write state[2][i] only when it already holds a tensor; the reference's
    deep copies :461-465 copy Nones), so a step sees the last token at position 0 and the encoder output -- nothing else.
    The step's log-probabilities are therefore a function of (sample, last token): they are computed once per distinct
    token and kept (a beam search of width k and length L costs at most 1 + k (L - 1) decoder steps, usually far fewer),
    top-k on the device, one host read per new token;
  * scores are Python floats summed in the reference's order; candidates are sorted by score, descending and stable (:493).
Batch size 1 per decoder call, as in the reference: the decoder `Attention`'s slot -> head pairing depends on the batch
size (mm_modeling.py:79-85), so stacking beams into one call would change the arithmetic.
"""
import torch
import torch.nn.functional as F

__all__ = ["beam_search", "beam_search_ids"]


@torch.no_grad()
def beam_search_ids(model, start_id, sep_id, enc_ids, enc_mask, enc_type, add_mask, vis_embeds, roi_embeds, roi_coors,
                    beam_size=3, max_len=20):
    """-> (token ids of the best sequence incl. the start token, its log-likelihood, [(score, ids)] of the finished beams)"""
    model.eval()
    if enc_ids.dim() == 1:                                    # one sample without a batch axis (:395-402)
        enc_ids, enc_mask, enc_type, add_mask = (t.unsqueeze(0) for t in (enc_ids, enc_mask, enc_type, add_mask))
        vis_embeds, roi_embeds, roi_coors = (t.unsqueeze(0) for t in (vis_embeds, roi_embeds, roi_coors))
    enc = model.encoder(enc_ids, vis_embeds, roi_embeds, roi_coors, enc_type, enc_mask, add_mask)
    enc = enc[0] if isinstance(enc, tuple) else enc
    dec = model.decoder
    keys = dec.project_encoder(enc)                           # the blocks' cross-attention keys of this sample, once
    device = enc.device
    memo = {}

    def step(tok):
        """(top-k log-probabilities, top-k ids) after `tok`, as Python lists"""
        if tok not in memo:
            state = dec.init_state(enc, None)                 # (:436; valid lens None: no mask on either attention)
            logits = dec(torch.tensor([[tok]], device=device, dtype=torch.long), state, is_train=False, hoisted=keys)
            lp = F.log_softmax(logits[0, -1, :].float(), dim=-1)
            s, i = torch.topk(lp, beam_size)
            memo[tok] = (s.tolist(), i.tolist())
        return memo[tok]

    beams = [(0.0, [int(start_id)])]
    final = []
    for _ in range(max_len):
        cands = []
        for score, seq in beams:
            if seq[-1] == sep_id:                             # finished beams leave the search (:444-447)
                final.append((score, seq))
                continue
            top_s, top_i = step(seq[-1])
            for k in range(beam_size):
                cands.append((score + top_s[k], seq + [top_i[k]]))
        if not cands:
            break
        beams = sorted(cands, key=lambda c: c[0], reverse=True)[:beam_size]
        if all(seq[-1] == sep_id for _, seq in beams):        # (:500-502)
            final.extend(beams)
            break
    if not final:                                             # nothing finished within max_len (:505-506)
        final = list(beams)
    best_score, best_seq = sorted(final, key=lambda c: c[0], reverse=True)[0]
    return best_seq, best_score, final


def beam_search(model, tokenizer, enc_ids, enc_mask, enc_type, add_mask, vis_embeds, roi_embeds, roi_coors,
                beam_size=3, num_preds=1, max_len=20, device='cuda'):
    """the reference's signature and result: [decoded text of the best sequence] (fcmf_pretraining.py:383-517)"""
    start = tokenizer.bos_token_id if tokenizer.bos_token_id is not None else tokenizer.cls_token_id
    ids, _, _ = beam_search_ids(model, start, tokenizer.sep_token_id, enc_ids, enc_mask, enc_type, add_mask, vis_embeds,
                                roi_embeds, roi_coors, beam_size=beam_size, max_len=max_len)
    return [tokenizer.decode(torch.tensor(ids), skip_special_tokens=True).strip()]

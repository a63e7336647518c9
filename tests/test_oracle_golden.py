"""The CPU oracle against the committed golden fixtures (outputs of the REFERENCE import,
oracle/make_golden.py).  This is the pin that lets the oracle stand in for the reference on the GPU
box, where /root/reference does not exist."""
import math
import os

import numpy as np
import pytest
import torch

from conftest import GOLD
from oracle import fcmf_oracle as O
import synthetic_data as synth


def _fixture(tag, cfg):
    z = np.load(os.path.join(GOLD, f"fcmf_{tag}.npz"))
    B, S, NI, NR = int(z["B"]), int(z["S"]), int(z["NI"]), int(z["NR"])
    P = synth.synth_params(synth.fcmf_param_shapes(cfg))
    batch = synth.synth_batch(B, cfg, S=S, num_imgs=NI, num_roi=NR, seed=42)
    return z, P, batch, NI, NR


def test_oracle_tiny_forward_backward_step():
    cfg = synth.TINY_CFG
    z, P, batch, NI, NR = _fixture("tiny", cfg)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    loss, logits = O.fcmf_step_loss(Pg, cfg, batch, NI, NR)
    assert (logits.detach() - torch.from_numpy(z["logits"])).abs().max() < 1e-5
    assert abs(loss.item() - float(z["loss"])) < 1e-5
    loss.backward()
    grads = {k: v.grad for k, v in Pg.items() if v.grad is not None}
    for n, ref in zip(z["grad_names"], z["grad_norms"]):
        n = str(n)
        if n.endswith(".key.bias") or n.endswith("box_head.linears.1.bias"):
            continue
        assert abs(grads[n].norm().item() - ref) < 1e-4 * max(ref, 1e-3), n
    assert all(Pg[str(n)].grad is None for n in z["nograd_names"])
    # clip + AdamW (4 groups) + schedule factor 1/10, as the reference loop
    clipped, total = O.clip_grad_norm(grads, 1.0)
    assert abs(total.item() - float(z["total_grad_norm"])) < 1e-4 * float(z["total_grad_norm"])
    f = O.linear_schedule_factor(1, 10, 100)
    groups = O.fcmf_param_groups(list(grads))
    for g in groups:
        for n in g["names"]:
            p1, _, _ = O.adamw_update(P[n], clipped[n], torch.zeros_like(P[n]), torch.zeros_like(P[n]), 1,
                                      g["lr"] * f, g["weight_decay"])
            key = "d_" + n
            if key in z.files and not n.endswith(".key.bias"):
                d = (p1 - P[n]).flatten()
                if ("gidx_" + n) in z.files:
                    d = d[torch.from_numpy(z["gidx_" + n])]
                ref = torch.from_numpy(z[key])
                gref = torch.from_numpy(z["g_" + n]).abs()
                ok = gref > 1e-6 * gref.max()
                assert (d[ok] - ref[ok]).abs().max() < 2e-3 * ref.abs().max(), n


def test_oracle_base_logits():
    """FCMF-base geometry, forward only (the backward pin ran inside oracle/make_golden.py)"""
    cfg = synth.BASE_CFG
    z, P, batch, NI, NR = _fixture("base", cfg)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        loss, logits = O.fcmf_step_loss(P, cfg, batch, NI, NR)
    assert (logits - torch.from_numpy(z["logits"])).abs().max() < 1e-5
    assert abs(loss.item() - float(z["loss"])) < 1e-5


def test_oracle_box_embedding_known_answers():
    z = np.load(os.path.join(GOLD, "box_embedding.npz"))
    c = torch.from_numpy(z["coords"])
    assert (O.box_relational_embedding(c) - torch.from_numpy(z["emb64"])).abs().max() < 1e-9
    assert (O.box_relational_embedding(c.float()) - torch.from_numpy(z["emb32"])).abs().max() < 1e-5
    assert torch.isfinite(torch.from_numpy(z["emb64"])).all()      # zero-padded boxes stay finite


def test_oracle_bertadam_and_schedules():
    z = np.load(os.path.join(GOLD, "bertadam.npz"))
    p = torch.from_numpy(z["p0"])
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for i, g in enumerate(z["grads"]):
        p, m, v = O.bertadam_update(p, torch.from_numpy(g), m, v, i, 1e-2, 0.01, t_total=20, warmup=0.1)
        assert (p - torch.from_numpy(z["traj"][i])).abs().max() < 1e-6
    import fcmf_framework.optimization as opt
    for x, wl, wc in zip(z["xs"], z["warmup_linear"], z["warmup_constant"]):
        assert abs(opt.warmup_linear(float(x)) - wl) < 1e-12
        assert abs(opt.warmup_constant(float(x)) - wc) < 1e-12


@pytest.mark.parametrize("B", [3, 4])
def test_oracle_iaog_tiny(B):
    """IAOG logits, loss AND gradients of the oracle against the reference fixture, at two batch sizes (the decoder's
    slot->head pairing depends on B mod n_head, mm_modeling.py:79-85)"""
    z = np.load(os.path.join(GOLD, "iaog_tiny.npz"))
    t = f"b{B}_"
    cfg = synth.TINY_CFG
    V, NI, NR, S = cfg["vocab_size"], 2, 5, 16
    shapes = {k: v for k, v in synth.fcmf_param_shapes(cfg).items() if k.startswith("encoder.")}
    shapes.update(synth.iaog_decoder_param_shapes(cfg, V))
    P = {k: v.clone().requires_grad_(True) for k, v in synth.synth_params(shapes).items()}
    P["decoder.dense.weight"] = P["encoder.bert.cell.embeddings.word_embeddings.weight"]
    batch = synth.synth_batch(B, cfg, S=S, num_imgs=NI, num_roi=NR, seed=5, coord_dtype=torch.float32)
    enc = O.fcmf_encoder_forward(P, cfg, batch["input_ids"][:, 0], batch["visual_embeds_att"],
                                 batch["roi_embeds_att"], batch["roi_coors"], batch["token_type_ids"][:, 0],
                                 batch["attention_mask"][:, 0], batch["added_attention_mask"][:, 0], NI, NR)
    logits = O.iaog_decoder_forward(P, cfg, torch.from_numpy(z[t + "dec"]), enc)
    assert (logits.detach()[:, :, ::8] - torch.from_numpy(z[t + "logits"])).abs().max() < 1e-5
    loss = torch.nn.functional.cross_entropy(logits.permute(0, 2, 1), torch.from_numpy(z[t + "labels"]), ignore_index=-100)
    assert abs(loss.item() - float(z[t + "loss"])) < 1e-5
    loss.backward()
    for n, ref in zip(z[t + "grad_names"], z[t + "grad_norms"]):
        n = str(n)
        if n.endswith(".key.bias") or n.endswith("box_head.linears.1.bias"):
            continue
        assert abs(P[n].grad.norm().item() - ref) < 1e-4 * max(ref, 1e-4), n
    for key in z.files:
        if key.startswith(t + "g_"):
            n = key[len(t) + 2:]
            g = P[n].grad.flatten()
            if (t + "gidx_" + n) in z.files:
                g = g[torch.from_numpy(z[t + "gidx_" + n])]
            ref = torch.from_numpy(z[key])
            assert (g - ref).abs().max().item() < 1e-4 * max(ref.abs().max().item(), 1e-6), n


def test_oracle_iaog_base_geometry_forward():
    """the oracle at IAOG's REAL geometry (H 768, 12 heads, V 64001, seq 128, Ld 12, 12 decoder blocks, B = 5: slot -> head pairing
    (s * 5 + b) mod 12) against the reference fixture iaog_base.npz: logits and loss (forward only here -- make_golden.py
    checked the gradients against the reference import when it wrote the fixture; the GPU test checks the product's)"""
    z = np.load(os.path.join(GOLD, "iaog_base.npz"))
    B, t = 5, "b5_"
    cfg = synth.BASE_CFG
    V = cfg["vocab_size"]
    NI, NR, S, Ld, step = (int(x) for x in z["geometry"])
    shapes = {k: v for k, v in synth.fcmf_param_shapes(cfg).items() if k.startswith("encoder.")}
    shapes.update(synth.iaog_decoder_param_shapes(cfg, V))
    P = synth.synth_params(shapes)
    P["decoder.dense.weight"] = P["encoder.bert.cell.embeddings.word_embeddings.weight"]
    batch = synth.synth_batch(B, cfg, S=S, num_imgs=NI, num_roi=NR, seed=5, coord_dtype=torch.float32)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        enc = O.fcmf_encoder_forward(P, cfg, batch["input_ids"][:, 0], batch["visual_embeds_att"],
                                     batch["roi_embeds_att"], batch["roi_coors"], batch["token_type_ids"][:, 0],
                                     batch["attention_mask"][:, 0], batch["added_attention_mask"][:, 0], NI, NR)
        logits = O.iaog_decoder_forward(P, cfg, torch.from_numpy(z[t + "dec"]), enc)
        assert tuple(logits.shape) == (B, Ld, V)
        assert (logits[:, :, ::step] - torch.from_numpy(z[t + "logits"])).abs().max() < 1e-4
        loss = torch.nn.functional.cross_entropy(logits.permute(0, 2, 1), torch.from_numpy(z[t + "labels"]), ignore_index=-100)
    assert abs(loss.item() - float(z[t + "loss"])) < 1e-4


def test_oracle_iaog_decode_matches_reference_fixture():
    """the oracle's decode path -- encoder, the decoder's is_train=False step (no mask on either attention, one token at
    position 0), the restated beam search (fcmf_pretraining.py:383-517) -- against iaog_decode.npz: step logits of the
    REFERENCE import and the beam search run over the reference's decoder step (oracle/make_golden.py decode_fixture)"""
    z = np.load(os.path.join(GOLD, "iaog_decode.npz"))
    NI, NR, S, max_len = (int(v) for v in z["geometry"])
    cfg = synth.TINY_CFG
    V = cfg["vocab_size"]
    shapes = {k: v for k, v in synth.fcmf_param_shapes(cfg).items() if k.startswith("encoder.")}
    shapes.update(synth.iaog_decoder_param_shapes(cfg, V))
    P = dict(synth.synth_params(shapes))
    P["decoder.dense.weight"] = P["encoder.bert.cell.embeddings.word_embeddings.weight"]
    batch = synth.synth_batch(2, cfg, S=S, num_imgs=NI, num_roi=NR, seed=5, coord_dtype=torch.float32)
    for b in range(2):
        sl = slice(b, b + 1)
        with torch.no_grad():
            enc = O.fcmf_encoder_forward(P, cfg, batch["input_ids"][sl, 0], batch["visual_embeds_att"][sl], batch["roi_embeds_att"][sl],
                                         batch["roi_coors"][sl], batch["token_type_ids"][sl, 0], batch["attention_mask"][sl, 0],
                                         batch["added_attention_mask"][sl, 0], NI, NR)

        def logits(tok):
            with torch.no_grad():
                return O.iaog_decoder_forward(P, cfg, torch.tensor([[tok]]), enc, is_train=False)[0, -1]

        for tok in (0, 7, 123):
            assert (logits(tok)[::4] - torch.from_numpy(z[f"s{b}_logits_tok{tok}"])).abs().max() < 1e-5
        step = lambda seq: torch.log_softmax(logits(seq[-1]), dim=-1)
        ids, score, fin = O.beam_search_ids(step, 0, 2, beam_size=2, max_len=max_len)
        assert ids == z[f"s{b}_a_ids"].tolist() and abs(score - float(z[f"s{b}_a_score"])) < 1e-4
        assert np.allclose([f[0] for f in fin], z[f"s{b}_a_final_scores"], atol=1e-4)
        ids, score, fin = O.beam_search_ids(step, 0, int(z[f"s{b}_b_sep"]), beam_size=3, max_len=max_len)
        assert ids == z[f"s{b}_b_ids"].tolist() and abs(score - float(z[f"s{b}_b_score"])) < 1e-4
        assert [len(f[1]) for f in fin] == z[f"s{b}_b_final_lens"].tolist()
        assert ids[-1] == int(z[f"s{b}_b_sep"]) and len(ids) < max_len + 1            # the finishing path was taken


def test_oracle_beam_search_loop_rules():
    """the loop's rules on a hand-made step function: stable descending sort, finished beams leave the search, all
    survivors finished -> stop, nothing finished -> the live beams compete (fcmf_pretraining.py:444-447,493-509)"""
    table = {0: [0.5, 0.3, 0.2, 0.0], 1: [0.1, 0.1, 0.1, 0.7], 2: [0.25, 0.25, 0.25, 0.25], 3: [0.0, 0.0, 0.0, 1.0]}
    step = lambda seq: torch.log(torch.tensor(table[seq[-1]]) + 1e-30)
    ids, score, fin = O.beam_search_ids(step, 0, 3, beam_size=2, max_len=6)
    assert ids == [0, 1, 3] and abs(score - (math.log(0.3) + math.log(0.7))) < 1e-6
    assert all(f[1][-1] == 3 for f in fin)
    ids, score, fin = O.beam_search_ids(step, 0, 9, beam_size=2, max_len=3)          # SEP never produced
    assert ids == [0, 1, 3, 3] and len(fin) == 2 and abs(score - (math.log(0.3) + math.log(0.7))) < 1e-6
    assert fin[1][1] == [0, 0, 0, 0] and abs(fin[1][0] - 3 * math.log(0.5)) < 1e-6

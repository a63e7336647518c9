"""IAOG pre-training step ("next" #2) on synthetic batches: FCMF-base encoder + 12-block
decoder + tied 64001-wide vocabulary GEMM + CE + clip + AdamW, bf16.  Prints one JSON line (not the headline metric:
bench.py is the contract; this is the measurement of the next row).
    python tools/iaog_bench.py [--batch 16] [--steps 6] [--warmup 2]"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "multimodal-aspect-category-sentiment-analysis_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)       # run_pretraining_fcmf.py default
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dec_len", type=int, default=12)
    ap.add_argument("--num_rois", type=int, default=4)     # run_pretraining_fcmf.py default
    args = ap.parse_args()
    import synthetic_data as synth
    from fcmf_framework import ops
    from fcmf_framework.fcmf_pretraining import FCMFSeq2Seq
    from fcmf_framework.optimization import FusedAdamW
    from fcmf_framework.roberta import RobertaConfig, RobertaModel
    dev = torch.device("cuda:0")
    cfg = synth.BASE_CFG
    d = tempfile.mkdtemp()
    RobertaModel(RobertaConfig(**cfg)).save_pretrained(d)
    V, NI, NR, B, S = cfg["vocab_size"], 7, args.num_rois, args.batch, 128
    ops.set_compute_dtype(torch.bfloat16)
    model = FCMFSeq2Seq(V, 20, d, NI, NR, 1.0)
    model.decoder.embedding = torch.nn.Embedding(V, model.decoder.num_hiddens)
    model = model.to(dev).train()
    no_decay = ['bias', 'LayerNorm.bias', 'LayerNorm.weight']
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    opt = FusedAdamW([{'params': [p for n, p in named if not any(nd in n for nd in no_decay)], 'weight_decay': 1e-5},
                      {'params': [p for n, p in named if any(nd in n for nd in no_decay)], 'weight_decay': 0.0}], lr=3e-5)
    b = synth.synth_batch(B, cfg, S=S, num_imgs=NI, num_roi=NR, num_aspects=1, seed=3, coord_dtype=torch.float32)
    b = {k: v.to(dev) for k, v in b.items()}
    dec = torch.randint(3, V, (B, args.dec_len), generator=torch.Generator().manual_seed(1)).to(dev)
    lab = torch.roll(dec, -1, dims=1)
    lab[:, -1] = -100

    def step():
        logits = model(enc_X=b["input_ids"][:, 0], dec_X=dec, visual_embeds_att=b["visual_embeds_att"],
                       roi_embeds_att=b["roi_embeds_att"], roi_coors=b["roi_coors"], token_type_ids=b["token_type_ids"][:, 0],
                       attention_mask=b["attention_mask"][:, 0], added_attention_mask=b["added_attention_mask"][:, 0],
                       source_valid_len=None, is_train=True)
        loss = ops.cross_entropy(logits, lab, ignore_index=-100)
        loss.backward()
        opt.step(max_grad_norm=1.0)
        opt.zero_grad(set_to_none=True)
        return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"metric": "IAOG pre-train samples/sec (fwd+bwd+step), FCMF-base seq128, %d ROIs, dec len %d" % (NR, args.dec_len),
                      "value": round(B * args.steps / dt, 2), "unit": "samples/s", "ms_per_step": round(dt / args.steps * 1e3, 2),
                      "batch": B, "dtype": "bf16", "loss": round(float(loss.item()), 4)}))


if __name__ == "__main__":
    main()

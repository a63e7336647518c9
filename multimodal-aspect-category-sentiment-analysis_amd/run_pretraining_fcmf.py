#!/usr/bin/env python3
"""IAOG seq2seq pre-training driver on MI355X -- drop-in for the reference's run_pretraining_fcmf.py.

Accepts every flag the reference declares (run_pretraining_fcmf.py:45-84; tests/test_surface.py feeds it the published
command line of Pretraining-Notebook.ipynb:6927-6945), same step (:284-337: FCMFSeq2Seq forward,
CE(ignore_index=-100) over [B,V,Ld], clip 1.0, AdamW (wd 1e-5 / 0, eps=--adam_epsilon), linear
schedule), same per-epoch checkpoint dict (:27-42,455-460).  Reference behaviours kept on purpose:
`model.decoder.embedding` is re-created after construction (:189), which un-ties it from the
encoder's word embeddings while `decoder.dense.weight` stays tied to them.
--fine_tune_cnn as in the reference (:203-207): the parameters of BOTH ResNet-152 extractors join the two AdamW groups
(and the gradient arena / the data-parallel exchange), so the trunks really train; the extractors are checkpointed as
`seed_{seed}_resimg_model_last.pth` / `..._resroi_model_last.pth` (:457-459) and restored on resume through the
reference's `iaog_model` -> `resimg_model` / `resroi_model` path rewrite (:244-255).
Extra flags: --bf16, --synthetic_steps N (seeded synthetic batches, precomputed features), --synthetic_pixels SIZE
(those batches carry pixel crops and the HIP ResNet-152 trunks run inside the step).
With real data the driver imports the user's `iaog_dataset.IAOGDataset` (host-side producer,
SURVEY.md section 8(f) "next") and torchvision, as the reference does.
"""
import argparse
import logging
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

from fcmf_framework import ops  # noqa: E402
from fcmf_framework.dp import GradArena, GradReducer  # noqa: E402
from fcmf_framework.fcmf_pretraining import FCMFSeq2Seq  # noqa: E402
from fcmf_framework.optimization import FusedAdamW, get_linear_schedule_with_warmup  # noqa: E402
from device_prefetch import DevicePrefetcher  # noqa: E402


def save_model(path, model, optimizer, scheduler, epoch, best_score=0.0):
    m = model.module if hasattr(model, 'module') else model
    torch.save({'epoch': epoch, 'best_score': best_score, 'model_state_dict': m.state_dict(),
                'optimizer_state_dict': optimizer.state_dict(), 'scheduler_state_dict': scheduler.state_dict()}, path)


def build_parser():
    p = argparse.ArgumentParser()
    # (reference :47-48 marks both `required`; here they are only read by the real-data branch, so --synthetic_steps runs without)
    p.add_argument("--data_dir", default='../vimacsa', type=str)
    p.add_argument("--pretrained_data_dir", default='../iaog-pretraining', type=str)
    p.add_argument("--output_dir", default=None, type=str, required=True)
    p.add_argument('--image_dir', default='../vimacsa/image')
    p.add_argument("--pretrained_hf_model", default=None, type=str, required=True)
    p.add_argument("--resume_from_checkpoint", default=None, type=str)
    # accepted exactly as the reference declares them (run_pretraining_fcmf.py:53,57,60,66,82); --bert_score_model / --beam_size /
    # --resnet_label_path feed only the reference's commented-out evaluation half (:376-632, dead code there, not built here)
    p.add_argument('--bert_score_model', default='uitnlp/visobert', type=str,
                   help="HuggingFace model name or local path for BERTScore (reference: used by its disabled evaluation only)")
    p.add_argument('--resnet_label_path', default='/kaggle/input/resnet-output')
    p.add_argument("--max_seq_length", default=170, type=int, help="encoder prompt length (the reference's dataset hard-codes 170)")
    p.add_argument("--beam_size", default=2, type=int)
    p.add_argument("--list_aspect", nargs='+', default=[],
                   help="aspect categories that produce IAOG samples (empty = the six ViMACSA categories, as the reference hard-codes)")
    p.add_argument("--num_imgs", default=7, type=int)
    p.add_argument("--num_rois", default=4, type=int)
    p.add_argument("--max_len_decoder", default=20, type=int)
    p.add_argument("--do_train", action='store_true')
    p.add_argument("--do_eval", action='store_true')
    p.add_argument("--train_batch_size", default=16, type=int)
    p.add_argument("--eval_batch_size", default=16, type=int)
    p.add_argument("--learning_rate", default=3e-5, type=float)
    p.add_argument("--adam_epsilon", default=1e-8, type=float)
    p.add_argument("--num_train_epochs", default=8.0, type=float)
    p.add_argument("--warmup_proportion", default=0.1, type=float)
    p.add_argument('--gradient_accumulation_steps', type=int, default=1)
    p.add_argument('--seed', type=int, default=42)
    p.add_argument('--fp16', action='store_true')
    p.add_argument('--alpha', type=float, default=1)
    p.add_argument('--fine_tune_cnn', action='store_true')
    p.add_argument("--no_cuda", action='store_true')
    p.add_argument("--ddp", action='store_true')
    p.add_argument("--local_rank", type=int, default=-1)
    p.add_argument('--bf16', action='store_true')
    p.add_argument('--synthetic_steps', type=int, default=0)
    p.add_argument('--synthetic_dec_len', type=int, default=12)
    p.add_argument('--vocab_size', type=int, default=0, help="decoder vocabulary (len(tokenizer) with real data)")
    p.add_argument('--feature_cache_dir', default=None, type=str,
                   help="precomputed ResNet-152 features of the training reviews (feature_cache.build); else pixels + the HIP trunk")
    p.add_argument('--resnet_checkpoint', default=None, type=str, help="torchvision resnet152 state dict for the HIP trunk")
    p.add_argument('--synthetic_pixels', type=int, default=0,
                   help="with --synthetic_steps: batches carry SIZE x SIZE pixel crops and the ResNet-152 trunks run inside the step")
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.no_cuda or not torch.cuda.is_available():
        raise SystemExit("run_pretraining_fcmf.py (MI355X build) has no CPU path: a ROCm GPU is required")
    if args.ddp:
        rank, local_rank, world = int(os.environ['RANK']), int(os.environ['LOCAL_RANK']), int(os.environ['WORLD_SIZE'])
    else:
        rank, local_rank, world = 0, 0, 1
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    master = rank == 0
    random.seed(args.seed); np.random.seed(args.seed); torch.manual_seed(args.seed); ops.manual_seed(args.seed + rank)
    if world > 1:
        torch.distributed.init_process_group(backend='nccl', device_id=device)
    logger = logging.getLogger("iaog")
    if master:
        os.makedirs(args.output_dir, exist_ok=True)
        logger.setLevel(logging.INFO)
        for h in (logging.FileHandler(f'{args.output_dir}/pretraining_iaog.log'), logging.StreamHandler()):
            logger.addHandler(h)
    ops.set_compute_dtype(torch.bfloat16 if (args.bf16 or args.fp16) else torch.float32)

    tokenizer = None
    if args.synthetic_steps <= 0:
        from transformers import AutoTokenizer
        tokenizer = AutoTokenizer.from_pretrained(args.pretrained_hf_model)
    vocab = args.vocab_size or (len(tokenizer) if tokenizer is not None else 0)
    if vocab <= 0:
        from fcmf_framework.roberta import RobertaConfig
        vocab = RobertaConfig.from_pretrained(args.pretrained_hf_model).vocab_size
    model = FCMFSeq2Seq(vocab, args.max_len_decoder, args.pretrained_hf_model, args.num_imgs, args.num_rois, args.alpha)
    model.decoder.embedding = torch.nn.Embedding(vocab, model.decoder.num_hiddens)       # reference :189
    model = model.to(device)

    # the two ResNet-152 extractors (reference :191-194) whenever pixels enter the step: real data without a feature cache,
    # or --synthetic_pixels
    r_img = r_roi = None
    if (args.synthetic_steps > 0 and args.synthetic_pixels > 0) or (args.synthetic_steps <= 0 and not args.feature_cache_dir):
        from fcmf_framework.resnet import resnet152
        from fcmf_framework.resnet_utils import myResNetImg, myResNetRoI
        sd = torch.load(args.resnet_checkpoint, map_location='cpu', weights_only=True) if args.resnet_checkpoint else None
        r_img = myResNetImg(resnet152(weights=sd).to(device), args.fine_tune_cnn, device).train()
        r_roi = myResNetRoI(resnet152(weights=sd).to(device), args.fine_tune_cnn, device).train()

    no_decay = ['bias', 'LayerNorm.bias', 'LayerNorm.weight']
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    cnn_params = []
    if args.fine_tune_cnn and r_img is not None:                     # reference :203-207 (names: 'resnet.conv1.weight', ...)
        cnn_params = list(r_img.named_parameters()) + list(r_roi.named_parameters())
        named += cnn_params
    groups = [{'params': [p for n, p in named if not any(nd in n for nd in no_decay)], 'weight_decay': 1e-5},
              {'params': [p for n, p in named if any(nd in n for nd in no_decay)], 'weight_decay': 0.0}]
    optimizer = FusedAdamW(groups, lr=args.learning_rate, eps=args.adam_epsilon)

    if args.synthetic_steps > 0:
        import synthetic_data as synth
        cfg = model.encoder.bert.cell.config
        cfgd = dict(vocab_size=cfg.vocab_size, pad_token_id=cfg.pad_token_id)

        def batches():
            for i in range(args.synthetic_steps):
                b = synth.synth_batch(args.train_batch_size, cfgd, S=min(args.max_seq_length, 128, cfg.max_position_embeddings - 2), num_imgs=args.num_imgs, num_roi=args.num_rois,
                                      num_aspects=1, seed=args.seed + 1000 * rank + i, coord_dtype=torch.float32)
                g = torch.Generator().manual_seed(args.seed + i)
                dec = torch.randint(3, vocab, (args.train_batch_size, args.synthetic_dec_len), generator=g)
                lab = torch.roll(dec, -1, dims=1)
                lab[:, -1] = -100                                                           # iaog_dataset.py:93-96
                if args.synthetic_pixels:             # pixel crops in the IAOG dataset's float32 layout (iaog_dataset.py:148)
                    S_, n_ = args.synthetic_pixels, args.train_batch_size * args.num_imgs
                    b["visual_embeds_att"] = synth.synth_crops(n_, S_, seed=args.seed + i).view(args.train_batch_size, args.num_imgs, 3, S_, S_)
                    b["roi_embeds_att"] = synth.synth_crops(n_ * args.num_rois, S_, seed=args.seed + i + 7919).view(
                        args.train_batch_size, args.num_imgs, args.num_rois, 3, S_, S_)
                yield (b["visual_embeds_att"], b["roi_embeds_att"], b["roi_coors"], b["input_ids"][:, 0],
                       b["token_type_ids"][:, 0], b["attention_mask"][:, 0], b["added_attention_mask"][:, 0], dec, lab)
        steps_per_epoch = args.synthetic_steps
        make_loader = batches
    else:
        # real data (reference :130-183): reviews with `iaog_labels`, one sample per (review, aspect); photos through the
        # HIP ResNet-152 trunk inside the step, or a precomputed feature cache.  (The reference also runs underthesea's
        # Vietnamese text normalisation over the comments; that host-side text cleaning is not part of this build --
        # feed already-normalised JSON.)
        import json
        import pandas as pd
        from torch.utils.data import DataLoader, DistributedSampler, RandomSampler
        from iaog_dataset import IAOGDataset
        train_data = pd.read_json(f'{args.pretrained_data_dir}/train_with_iaog.json')
        if 'iaog_labels' not in train_data.columns:
            raise ValueError("'iaog_labels' column not found in data. Check JSON file structure.")
        roi_df = pd.read_csv(f"{args.data_dir}/roi_data.csv")
        roi_df['file_name'] = roi_df['file_name'] + '.png'
        with open(f'{args.data_dir}/resnet152_image_label.json') as f:
            dict_image_aspect = json.load(f)
        with open(f'{args.data_dir}/resnet152_roi_label.json') as f:
            dict_roi_aspect = json.load(f)
        cache = None
        if args.feature_cache_dir:
            from feature_cache import FeatureCache
            cache = FeatureCache(args.feature_cache_dir)
        train_ds = IAOGDataset(train_data, tokenizer, args.image_dir, roi_df, dict_image_aspect, dict_roi_aspect,
                               args.num_imgs, args.num_rois, args.max_len_decoder, feature_cache=cache,
                               max_seq_length=args.max_seq_length, list_aspect=args.list_aspect or None)
        if len(train_ds) == 0:
            raise SystemExit("train_dataset is empty: no 'sentiment_word#Aspect' labels in iaog_labels")
        sampler = DistributedSampler(train_ds) if world > 1 else RandomSampler(train_ds)      # shard once
        loader = DataLoader(train_ds, sampler=sampler, batch_size=args.train_batch_size, pin_memory=True)

        def batches():
            for t_img, roi_img, coors, labels, dec, enc_ids, enc_type, enc_mask, added, _, _ in loader:
                yield (t_img, roi_img, coors.float(), enc_ids, enc_type, enc_mask, added, dec, labels)
        steps_per_epoch = len(loader)
        make_loader = batches

    num_train_steps = int(steps_per_epoch / args.gradient_accumulation_steps * args.num_train_epochs)
    scheduler = get_linear_schedule_with_warmup(optimizer, int(num_train_steps * args.warmup_proportion), num_train_steps)
    # the trunks' gradients are produced LAST in backward (the extractors run first in the step): behind the model's in the arena
    arena = GradArena.for_model(model, extra=[p for _, p in cnn_params])
    reducer = None
    if world > 1:
        reducer = GradReducer(arena)
        reducer.broadcast_parameters(0)
    start_epoch = 0
    if args.resume_from_checkpoint and os.path.isfile(args.resume_from_checkpoint):
        ck = torch.load(args.resume_from_checkpoint, map_location=device, weights_only=True)
        model.load_state_dict(ck['model_state_dict'])
        for net, tag in ((r_img, "resimg_model"), (r_roi, "resroi_model")):           # reference :244-255
            d, f = os.path.split(args.resume_from_checkpoint)
            q = os.path.join(d, f.replace("iaog_model", tag))
            if net is not None and os.path.exists(q):
                if master:
                    logger.info("    Loading ResNet: %s", q)
                net.load_state_dict(torch.load(q, map_location=device, weights_only=True)['model_state_dict'])
        optimizer.load_state_dict(ck['optimizer_state_dict'])
        scheduler.load_state_dict(ck['scheduler_state_dict'])
        start_epoch = ck['epoch'] + 1
        ops.shadows.clear()

    def features(vis, roi):
        """pixels -> ResNet-152 features (reference :305-317) unless the batch already holds features"""
        if r_img is None:
            return vis, roi
        from fcmf_framework.resnet_utils import extract_features
        return extract_features(r_img, r_roi, vis, roi.float())

    if args.do_train:
        for epoch in range(start_epoch, int(args.num_train_epochs)):
            model.train()
            if r_img is not None:
                r_img.train(); r_roi.train()
            arena.zero()
            for step, batch in enumerate(DevicePrefetcher(make_loader(), device)):
                vis, roi, coors, enc_X, tt, am, added, dec_X, labels = batch
                vis, roi = features(vis, roi)
                # model(...) -> logits -> CrossEntropyLoss(ignore_index=-100) (reference :309-324) as one fused call
                loss = model.forward_loss(enc_X, dec_X, labels, vis, roi, coors, tt, am, added, ignore_index=-100)
                if args.gradient_accumulation_steps > 1:
                    loss = loss / args.gradient_accumulation_steps
                boundary = (step + 1) % args.gradient_accumulation_steps == 0
                if reducer is not None:
                    reducer.enabled = boundary
                loss.backward()
                if boundary:
                    if reducer is not None:
                        reducer.finish()
                    optimizer.step(max_grad_norm=1.0)
                    scheduler.step()
                    arena.zero()
                if master and step % 10 == 0:
                    logger.info("epoch %d step %d loss %.4f", epoch, step, loss.item() * args.gradient_accumulation_steps)
            if world > 1:
                torch.distributed.barrier()
            if master:
                save_model(f'{args.output_dir}/seed_{args.seed}_iaog_model_last.pth', model, optimizer, scheduler, epoch)
                if r_img is not None:                                                   # reference :458-459
                    save_model(f'{args.output_dir}/seed_{args.seed}_resimg_model_last.pth', r_img, optimizer, scheduler, epoch)
                    save_model(f'{args.output_dir}/seed_{args.seed}_resroi_model_last.pth', r_roi, optimizer, scheduler, epoch)
    arena.deactivate()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

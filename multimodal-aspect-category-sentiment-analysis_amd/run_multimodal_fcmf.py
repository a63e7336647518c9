#!/usr/bin/env python3
"""MACSA fine-tuning driver on MI355X -- drop-in for the reference's run_multimodal_fcmf.py.

Same command-line flags (reference run_multimodal_fcmf.py:65-118), same step order (:427-489), same
checkpoint dictionary and file names (:40-58,557-563), same torchrun environment contract
(RANK / LOCAL_RANK / WORLD_SIZE).  What changes is how a step is executed:
  * the 6 aspect forwards of a batch run as ONE `FCMF.forward_aspects` call on the HIP kernels;
  * clip_grad_norm_(1.0) + AdamW are the two fused multi-tensor kernels of `FusedAdamW`;
  * under --ddp the gradient mean is a bucketed RCCL all-reduce overlapped with backward
    (fcmf_framework.dp.GradReducer); the data is sharded ONCE (the reference shards it twice,
    :208-210 and :421);
  * --bf16 selects the MFMA path (bf16 activations, fp32 master weights, no loss scaler);
    --fp16 is accepted for compatibility and maps to --bf16.
The ResNet-152 extractors are checkpointed next to the model under the reference's names
(`seed_{seed}_resimg_model_{best,last}.pth`, `..._resroi_model_...`, :557-563) and restored on resume / for the test
evaluation through the reference's path rewrites (:334-346, :588-598); after training, --do_eval runs the TEST-set
evaluation of :567-694 and writes `test_results_fcmf.txt` + `test_predictions_formatted.txt` in the reference's format.
Batches reach the GPU through `device_prefetch.DevicePrefetcher` (pinned host memory, copy stream, one batch ahead).
Extra flags (not in the reference): --bf16, --synthetic_steps N (seeded synthetic batches with
precomputed features: no dataset / tokenizer / torchvision needed), --precomputed_features.
The host-side batch producer (vimacsa_dataset.MACSADataset, image decoding, ResNet-152 feature
extraction) is the "next" row of SURVEY.md section 8(f); with real data this driver imports the
user's `vimacsa_dataset` module and torchvision exactly as the reference does.
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
from fcmf_framework.fcmf_multimodal import FCMF  # noqa: E402
from fcmf_framework.optimization import FusedAdamW, get_linear_schedule_with_warmup  # noqa: E402
from device_prefetch import DevicePrefetcher  # noqa: E402


POLARITY_MAP = {0: 'None', 1: 'Negative', 2: 'Neutral', 3: 'Positive'}      # reference :28


def macro_f1(y_true, y_pred):
    from sklearn.metrics import precision_recall_fscore_support
    p, r, f, _ = precision_recall_fscore_support(y_true, y_pred, average='macro', zero_division=0)
    return p, r, f


def save_model(path, model, optimizer, scheduler, epoch, best_score=0.0, scaler=None):
    """checkpoint dict of the reference (run_multimodal_fcmf.py:40-58)"""
    m = model.module if hasattr(model, 'module') else model
    ck = {'epoch': epoch, 'best_score': best_score, 'model_state_dict': m.state_dict(),
          'optimizer_state_dict': optimizer.state_dict(), 'scheduler_state_dict': scheduler.state_dict()}
    if scaler is not None:
        ck['scaler_state_dict'] = scaler.state_dict()
    torch.save(ck, path)


def companion_path(path, old, new):
    """the reference's checkpoint-path rewrite (`checkpoint_path.replace("fcmf_model", "resimg_model")`, :334-335;
    `best_path.replace("fcmf", "resimg")`, :588,594) applied to the FILE NAME only -- a directory called e.g.
    `runs/fcmf/` must not be rewritten with it"""
    d, f = os.path.split(path)
    return os.path.join(d, f.replace(old, new))


def load_resnets(path, resnet_img, resnet_roi, device, logger=None, old="fcmf_model", strict=True):
    """restore the two extractors saved beside the model checkpoint `path` (reference :334-346 / :588-598)"""
    loaded = []
    for net, tag in ((resnet_img, "resimg"), (resnet_roi, "resroi")):
        if net is None:
            continue
        q = companion_path(path, old, old.replace("fcmf", tag))
        if os.path.exists(q):
            ck = torch.load(q, map_location=device, weights_only=True)
            net.load_state_dict(ck['model_state_dict'], strict=strict)
            loaded.append(q)
            if logger is not None:
                logger.info("    Loading ResNet %s from: %s", tag, q)
    if loaded:
        ops.shadows.clear()                  # cached bf16 weight matrices of the trunk are stale
    return loaded


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("--data_dir", default='../vimacsa', type=str)
    parser.add_argument("--output_dir", default=None, type=str, required=True)
    parser.add_argument('--image_dir', default='../vimacsa/image')
    parser.add_argument('--resnet_label_path', default='/kaggle/input/resnet-output')
    parser.add_argument("--pretrained_hf_model", default=None, type=str, required=True)
    parser.add_argument("--pretrained_iaog_path", default=None, type=str)
    parser.add_argument("--resume_from_checkpoint", default=None, type=str)
    parser.add_argument("--model_checkpoint", default='checkpoint_path', type=str)
    parser.add_argument("--list_aspect", default=['Location', 'Food', 'Room', 'Facilities', 'Service', 'Public_area'], nargs='+')
    parser.add_argument("--num_polarity", default=4, type=int)
    parser.add_argument("--num_imgs", default=7, type=int)
    parser.add_argument("--num_rois", default=7, type=int)
    parser.add_argument("--max_seq_length", default=170, type=int)
    parser.add_argument("--do_train", action='store_true')
    parser.add_argument("--do_eval", action='store_true')
    parser.add_argument("--freeze_encoder", action='store_true')
    parser.add_argument("--train_batch_size", default=4, type=int)
    parser.add_argument("--eval_batch_size", default=4, type=int)
    parser.add_argument("--encoder_learning_rate", default=7e-5, type=float)
    parser.add_argument("--classifier_head_learning_rate", default=7e-4, type=float)
    parser.add_argument("--num_train_epochs", default=8.0, type=float)
    parser.add_argument("--warmup_proportion", default=0.1, type=float)
    parser.add_argument('--gradient_accumulation_steps', type=int, default=2)
    parser.add_argument('--seed', type=int, default=42)
    parser.add_argument('--fp16', action='store_true')
    parser.add_argument('--alpha', type=float, default=1)
    parser.add_argument('--fine_tune_cnn', action='store_true')
    parser.add_argument("--no_cuda", action='store_true')
    parser.add_argument("--ddp", action='store_true')
    parser.add_argument("--local_rank", type=int, default=-1)
    # ---- MI355X additions ----
    parser.add_argument('--bf16', action='store_true', help="bf16 activations on the MFMA kernels")
    parser.add_argument('--synthetic_steps', type=int, default=0,
                        help="train on N seeded synthetic batches per epoch (precomputed features); no dataset needed")
    parser.add_argument('--precomputed_features', action='store_true',
                        help="the dataset yields ResNet-152 features instead of pixels (BASELINE.json configs)")
    parser.add_argument('--synthetic_pixels', type=int, default=0,
                        help="with --synthetic_steps: batches carry SIZE x SIZE pixel crops and the ResNet-152 trunk runs inside "
                             "the step (reference :449-460); 0 = precomputed features")
    parser.add_argument('--resnet_checkpoint', default=None, type=str,
                        help="torchvision resnet152 state dict (.pth, loaded with weights_only=True) for the HIP trunk")
    return parser


def param_groups(model, args):
    """4 groups by substring match on the parameter names (reference :249-287)"""
    no_decay = ['bias', 'LayerNorm.bias', 'LayerNorm.weight']
    head_names = ['classifier', 'text_pooler']
    enc, head = [], []
    for n, p in model.named_parameters():
        if not p.requires_grad:
            continue
        (head if any(nd in n for nd in head_names) else enc).append((n, p))
    return [
        {'params': [p for n, p in enc if not any(nd in n for nd in no_decay)], 'weight_decay': 0.01, 'lr': args.encoder_learning_rate},
        {'params': [p for n, p in enc if any(nd in n for nd in no_decay)], 'weight_decay': 0.0, 'lr': args.encoder_learning_rate},
        {'params': [p for n, p in head if not any(nd in n for nd in no_decay)], 'weight_decay': 0.01, 'lr': args.classifier_head_learning_rate},
        {'params': [p for n, p in head if any(nd in n for nd in no_decay)], 'weight_decay': 0.0, 'lr': args.classifier_head_learning_rate},
    ]


class SyntheticBatches:
    """seeded stand-in for DataLoader(MACSADataset): the reference's 9-tuple layout
    (vimacsa_dataset.py:202) with precomputed features in place of pixel tensors"""

    def __init__(self, cfg, steps, batch, S, num_imgs, num_rois, num_aspects, seed, pixels=0):
        self.cfg, self.steps, self.batch, self.S = cfg, steps, batch, S
        self.ni, self.nr, self.na, self.seed = num_imgs, num_rois, num_aspects, seed
        self.pixels = pixels

    def __len__(self):
        return self.steps

    def __iter__(self):
        import synthetic_data as synth
        for i in range(self.steps):
            b = synth.synth_batch(self.batch, self.cfg, S=self.S, num_imgs=self.ni, num_roi=self.nr,
                                  num_aspects=self.na, seed=self.seed + i)
            vis, roi = b["visual_embeds_att"], b["roi_embeds_att"]
            if self.pixels:          # the reference's pixel layout: float32 images, float64 ROI crops (vimacsa_dataset.py:175-199)
                n = self.batch * self.ni
                vis = synth.synth_crops(n, self.pixels, seed=self.seed + i).view(self.batch, self.ni, 3, self.pixels, self.pixels)
                roi = synth.synth_crops(n * self.nr, self.pixels, seed=self.seed + i + 7919).view(
                    self.batch, self.ni, self.nr, 3, self.pixels, self.pixels).double()
            texts = [f"synthetic review {self.seed + i}:{j}" for j in range(self.batch)]
            yield (vis, roi, b["roi_coors"], b["input_ids"], b["token_type_ids"],
                   b["attention_mask"], b["added_attention_mask"], b["labels"], texts)


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.no_cuda or not torch.cuda.is_available():
        raise SystemExit("run_multimodal_fcmf.py (MI355X build) has no CPU path: a ROCm GPU is required")
    if args.ddp:
        rank, local_rank, world = int(os.environ['RANK']), int(os.environ['LOCAL_RANK']), int(os.environ['WORLD_SIZE'])
    else:
        rank, local_rank, world = 0, 0, 1
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    master = rank == 0
    if args.gradient_accumulation_steps < 1:
        raise ValueError("Invalid gradient_accumulation_steps parameter: {}, should be >= 1".format(args.gradient_accumulation_steps))
    args.train_batch_size = int(args.train_batch_size / args.gradient_accumulation_steps)
    random.seed(args.seed); np.random.seed(args.seed); torch.manual_seed(args.seed); ops.manual_seed(args.seed + rank)
    if world > 1:
        torch.distributed.init_process_group(backend='nccl', device_id=device)   # RCCL over xGMI

    logger = logging.getLogger("fcmf")
    if master:
        os.makedirs(args.output_dir, exist_ok=True)
        logger.setLevel(logging.INFO)
        fmt = logging.Formatter('%(asctime)s - %(levelname)s - %(name)s - %(message)s', datefmt='%m/%d/%Y %H:%M:%S')
        for h in (logging.FileHandler(f'{args.output_dir}/training_fcmf.log'), logging.StreamHandler()):
            h.setFormatter(fmt); logger.addHandler(h)
        logger.info("device: %s n_gpu: %d, distributed training: %s, bf16: %s", device, world, bool(args.ddp), args.bf16 or args.fp16)
    ops.set_compute_dtype(torch.bfloat16 if (args.bf16 or args.fp16) else torch.float32)

    ASPECT = args.list_aspect
    model = FCMF(pretrained_path=args.pretrained_hf_model, num_labels=args.num_polarity, num_imgs=args.num_imgs,
                 num_roi=args.num_rois, alpha=args.alpha)
    cfg = model.encoder.bert.cell.config
    train_loader = dev_loader = test_loader = None
    resnet_img = resnet_roi = None
    if args.synthetic_steps > 0:
        cfgd = dict(vocab_size=cfg.vocab_size, pad_token_id=cfg.pad_token_id)
        train_loader = SyntheticBatches(cfgd, args.synthetic_steps, args.train_batch_size,
                                        min(args.max_seq_length, 128, cfg.max_position_embeddings - 2),
                                        args.num_imgs, args.num_rois, len(ASPECT), args.seed + 1000 * rank,
                                        pixels=args.synthetic_pixels)
        if args.do_eval:
            dev_loader = SyntheticBatches(cfgd, max(1, args.synthetic_steps // 2), args.eval_batch_size,
                                          min(args.max_seq_length, 128, cfg.max_position_embeddings - 2),
                                          args.num_imgs, args.num_rois, len(ASPECT), args.seed + 77, pixels=args.synthetic_pixels)
            test_loader = SyntheticBatches(cfgd, max(1, args.synthetic_steps // 2), args.eval_batch_size,
                                           min(args.max_seq_length, 128, cfg.max_position_embeddings - 2),
                                           args.num_imgs, args.num_rois, len(ASPECT), args.seed + 99, pixels=args.synthetic_pixels)
    elif args.do_train or args.do_eval:
        # real data: the reference's host-side producer (tokenizer, pandas, MACSADataset, torchvision ResNet-152)
        from transformers import AutoTokenizer
        import pandas as pd
        from torch.utils.data import DataLoader, DistributedSampler, RandomSampler, SequentialSampler
        from vimacsa_dataset import MACSADataset   # user-provided, same tuple layout as the reference
        tokenizer = AutoTokenizer.from_pretrained(args.pretrained_hf_model)
        model.encoder.bert.cell.resize_token_embeddings(len(tokenizer))
        roi_df = pd.read_csv(f"{args.data_dir}/roi_data.csv")
        roi_df['file_name'] = roi_df['file_name'] + '.png'
        import json
        with open(f'{args.data_dir}/resnet152_image_label.json') as f:
            dict_image_aspect = json.load(f)
        with open(f'{args.data_dir}/resnet152_roi_label.json') as f:
            dict_roi_aspect = json.load(f)
        mk = lambda df: MACSADataset(df, tokenizer, args.image_dir, roi_df, dict_image_aspect, dict_roi_aspect, args.num_imgs, args.num_rois)
        train_ds, dev_ds = mk(pd.read_json(f'{args.data_dir}/train.json')), mk(pd.read_json(f'{args.data_dir}/dev.json'))
        sampler = DistributedSampler(train_ds) if world > 1 else RandomSampler(train_ds)     # shard ONCE
        train_loader = DataLoader(train_ds, sampler=sampler, batch_size=args.train_batch_size, pin_memory=True)
        dev_loader = DataLoader(dev_ds, sampler=SequentialSampler(dev_ds), batch_size=args.eval_batch_size, pin_memory=True)
        if args.do_eval and os.path.exists(f'{args.data_dir}/test.json'):                     # reference :570-573
            test_ds = mk(pd.read_json(f'{args.data_dir}/test.json'))
            test_loader = DataLoader(test_ds, sampler=SequentialSampler(test_ds), batch_size=args.eval_batch_size, pin_memory=True)
    pixels = (args.synthetic_steps > 0 and args.synthetic_pixels > 0) or \
             (args.synthetic_steps <= 0 and (args.do_train or args.do_eval) and not args.precomputed_features)
    if pixels:
        # the ResNet-152 feature extractor of the step (reference :224-227) on the HIP trunk.  Weights: a local
        # torchvision checkpoint (--resnet_checkpoint), else torchvision's IMAGENET1K_V2 if torchvision is importable
        # and has them cached, else random initialisation (logged): nothing is ever downloaded.
        from fcmf_framework.resnet import resnet152
        from fcmf_framework.resnet_utils import myResNetImg, myResNetRoI

        def trunk():
            if args.resnet_checkpoint:
                return resnet152(weights=torch.load(args.resnet_checkpoint, map_location='cpu', weights_only=True))
            try:
                from torchvision.models import resnet152 as tv_resnet152, ResNet152_Weights
                return tv_resnet152(weights=ResNet152_Weights.IMAGENET1K_V2)       # adopted by the HIP trunk (from_module)
            except Exception as e:                                              # not installed / weights not cached
                if master:
                    logger.info("ResNet-152: no checkpoint given and torchvision weights unavailable (%s): random init", type(e).__name__)
                return resnet152()
        resnet_img = myResNetImg(trunk().to(device), args.fine_tune_cnn, device)
        resnet_roi = myResNetRoI(trunk().to(device), args.fine_tune_cnn, device)

    model = model.to(device)
    if args.freeze_encoder:
        for p in model.encoder.parameters():
            p.requires_grad = False
    optimizer = FusedAdamW(param_groups(model, args), lr=args.classifier_head_learning_rate)
    steps_per_epoch = len(train_loader) if train_loader is not None else 0
    num_train_steps = int(steps_per_epoch / args.gradient_accumulation_steps * args.num_train_epochs)
    scheduler = get_linear_schedule_with_warmup(optimizer, int(num_train_steps * args.warmup_proportion), num_train_steps)
    # gradients live in one flat arena (one memset per step; buckets of it are all-reduced in place under --ddp)
    arena = GradArena.for_model(model)
    reducer = None
    if world > 1:
        reducer = GradReducer(arena)
        reducer.broadcast_parameters(0)

    start_epoch, max_f1 = 0, 0.0
    if args.resume_from_checkpoint and os.path.isfile(args.resume_from_checkpoint):
        ck = torch.load(args.resume_from_checkpoint, map_location=device, weights_only=True)
        model.load_state_dict(ck['model_state_dict'])
        optimizer.load_state_dict(ck['optimizer_state_dict'])
        if 'scheduler_state_dict' in ck:
            scheduler.load_state_dict(ck['scheduler_state_dict'])
        start_epoch, max_f1 = ck['epoch'] + 1, ck.get('best_score', 0.0)
        ops.shadows.clear()
        load_resnets(args.resume_from_checkpoint, resnet_img, resnet_roi, device, logger if master else None)   # reference :334-346
    elif args.pretrained_iaog_path and os.path.isfile(args.pretrained_iaog_path):
        sd = torch.load(args.pretrained_iaog_path, map_location='cpu', weights_only=True)['model_state_dict']
        model.load_state_dict({k: v for k, v in sd.items() if k.startswith('encoder.')}, strict=False)   # reference :385-391
        ops.shadows.clear()

    def features(t_img, roi_img):
        """pixels -> ResNet-152 features (reference :449-460) unless the batch already holds features: the reference's
        num_imgs + num_imgs * num_rois trunk calls as two batched passes with per-call BatchNorm statistics"""
        if resnet_img is None:
            return t_img, roi_img
        from fcmf_framework.resnet_utils import extract_features
        return extract_features(resnet_img, resnet_roi, t_img, roi_img)

    if args.do_train:
        for epoch in range(start_epoch, int(args.num_train_epochs)):
            if world > 1 and hasattr(train_loader, 'sampler') and hasattr(train_loader.sampler, 'set_epoch'):
                train_loader.sampler.set_epoch(epoch)
            model.train()
            if resnet_img is not None:
                resnet_img.train(); resnet_roi.train()                # reference :431 (BatchNorm in batch-statistics mode)
            arena.zero()
            for step, batch in enumerate(DevicePrefetcher(train_loader, device, float32_fields=(1,))):     # (1 = the float64 ROI crops) next batch: pinned, on the copy stream
                t_img, roi_img, roi_coors, ids, tts, ams, added, labels, _ = batch
                vis, roi = features(t_img, roi_img)
                logits = model.forward_aspects(input_ids=ids, token_type_ids=tts, attention_mask=ams, added_attention_mask=added,
                                               visual_embeds_att=vis, roi_embeds_att=roi, roi_coors=roi_coors)
                loss = model.loss_aspects(logits, labels)            # sum over aspects of the batch-mean CE
                if args.gradient_accumulation_steps > 1:
                    loss = loss / args.gradient_accumulation_steps
                boundary = (step + 1) % args.gradient_accumulation_steps == 0
                if reducer is not None:
                    reducer.enabled = boundary                        # all-reduce only the accumulated gradients
                loss.backward()
                if boundary:
                    if reducer is not None:
                        reducer.finish()
                    optimizer.step(max_grad_norm=1.0)                 # clip_grad_norm_(1.0) fused into AdamW
                    scheduler.step()
                    arena.zero()
                if master and step % 10 == 0:
                    logger.info("epoch %d step %d loss %.4f", epoch, step, loss.item() * args.gradient_accumulation_steps)
            if master:
                logger.info("--> Epoch %d Completed. Encoder LR %.2e Head LR %.2e", epoch,
                            optimizer.param_groups[0]['lr'], optimizer.param_groups[2]['lr'])
            f1 = 0.0
            if dev_loader is not None and master:
                if resnet_img is not None:
                    resnet_img.eval(); resnet_roi.eval()              # reference :502
                f1 = evaluate(model, dev_loader, device, features, len(ASPECT), logger)
            if world > 1:
                torch.distributed.barrier()
            if master:
                tags = ['last'] + (['best'] if f1 > max_f1 else [])
                max_f1 = max(max_f1, f1)
                for tag in tags:                                      # reference :555-563: the model and BOTH extractors
                    save_model(f'{args.output_dir}/seed_{args.seed}_fcmf_model_{tag}.pth', model, optimizer, scheduler, epoch, max_f1)
                    if resnet_img is not None:
                        save_model(f'{args.output_dir}/seed_{args.seed}_resimg_model_{tag}.pth', resnet_img, optimizer, scheduler, epoch)
                        save_model(f'{args.output_dir}/seed_{args.seed}_resroi_model_{tag}.pth', resnet_roi, optimizer, scheduler, epoch)
    # ---- 7. TEST EVALUATION (reference :567-694) ---------------------------------------------------------------------
    if args.do_eval and master and test_loader is not None:
        logger.info("===================== STARTING TEST EVALUATION =====================")
        best_path = args.model_checkpoint if os.path.exists(args.model_checkpoint) else \
            f'{args.output_dir}/seed_{args.seed}_fcmf_model_best.pth'
        if os.path.exists(best_path):
            logger.info("Loading Best Checkpoint from: %s", best_path)
            ck = torch.load(best_path, map_location=device, weights_only=True)
            model.load_state_dict(ck['model_state_dict'], strict=False)
            ops.shadows.clear()
            load_resnets(best_path, resnet_img, resnet_roi, device, logger, old="fcmf", strict=False)      # :588-598
        else:
            logger.warning("No best model found! Using current weights.")
        if resnet_img is not None:
            resnet_img.eval(); resnet_roi.eval()
        test_evaluate(model, test_loader, device, features, ASPECT, args.output_dir, logger)
    arena.deactivate()
    if world > 1:
        torch.distributed.destroy_process_group()


@torch.no_grad()
def test_evaluate(model, loader, device, features, aspects, output_dir, logger):
    """test-set pass of the reference (:600-694): per-aspect precision / recall / macro-F1 into `test_results_fcmf.txt`, and
    one block per review with the predicted and gold polarity of every aspect into `test_predictions_formatted.txt`"""
    model.eval()
    true = {a: [] for a in aspects}
    pred = {a: [] for a in aspects}
    formatted = []
    for batch in DevicePrefetcher(loader, device, float32_fields=(1,)):
        t_img, roi_img, roi_coors, ids, tts, ams, added, labels, texts = batch
        vis, roi = features(t_img, roi_img)
        logits = model.forward_aspects(input_ids=ids, token_type_ids=tts, attention_mask=ams, added_attention_mask=added,
                                       visual_embeds_att=vis, roi_embeds_att=roi, roi_coors=roi_coors)
        p = logits.argmax(-1).cpu().numpy()
        y = labels.cpu().numpy()
        logs = [{"text": t, "aspects": {}} for t in (texts if texts is not None else [""] * len(p))]
        for i, a in enumerate(aspects):
            true[a].append(y[:, i]); pred[a].append(p[:, i])
            for j, (pp, ll) in enumerate(zip(p[:, i], y[:, i])):
                logs[j]["aspects"][a] = {"predict": POLARITY_MAP.get(int(pp), "Unknown"), "label": POLARITY_MAP.get(int(ll), "Unknown")}
        formatted.extend(logs)
    with open(os.path.join(output_dir, "test_results_fcmf.txt"), "w") as w:
        w.write("***** Test results *****\n")
        all_f1 = 0.0
        for a in aspects:
            precision, recall, f1 = macro_f1(np.concatenate(true[a]), np.concatenate(pred[a]))
            all_f1 += f1
            w.write(f"{a} - P: {precision:.4f}, R: {recall:.4f}, F1: {f1:.4f}\n")
            logger.info("%s - F1: %.4f", a, f1)
        avg_f1 = all_f1 / len(aspects)
        w.write(f"Average F1: {avg_f1:.4f}\n")
        logger.info("Average F1: %.4f", avg_f1)
    log_path = f"{output_dir}/test_predictions_formatted.txt"
    with open(log_path, "w", encoding="utf-8") as f:
        f.write("TEST DETAILED PREDICTIONS\n")
        f.write(f"Average Macro F1: {avg_f1:.4f}\n")
        f.write("=" * 50 + "\n\n")
        for i, sample in enumerate(formatted):
            f.write("{\n")
            f.write(f"Sentence {i}: {sample['text']}\n")
            for a in aspects:
                res = sample['aspects'].get(a, {'predict': 'N/A', 'label': 'N/A'})
                f.write(f"{a}:\n")
                f.write(f"   predict: {res['predict']}\n")
                f.write(f"   label:   {res['label']}\n")
            f.write("}\n")
    logger.info("Formatted predictions saved to %s", log_path)
    return avg_f1


@torch.no_grad()
def evaluate(model, loader, device, features, num_aspects, logger):
    """dev-set macro-F1 averaged over aspects (reference :500-552)"""
    model.eval()
    true, pred = [[] for _ in range(num_aspects)], [[] for _ in range(num_aspects)]
    for batch in DevicePrefetcher(loader, device, float32_fields=(1,)):
        t_img, roi_img, roi_coors, ids, tts, ams, added, labels, _ = batch
        vis, roi = features(t_img, roi_img)
        logits = model.forward_aspects(input_ids=ids, token_type_ids=tts, attention_mask=ams, added_attention_mask=added,
                                       visual_embeds_att=vis, roi_embeds_att=roi, roi_coors=roi_coors)
        p = logits.argmax(-1).cpu().numpy()
        for a in range(num_aspects):
            true[a] += labels[:, a].cpu().tolist()
            pred[a] += p[:, a].tolist()
    f1s = [macro_f1(true[a], pred[a])[2] for a in range(num_aspects)]
    logger.info("Dev macro-F1 per aspect: %s  mean %.4f", ["%.4f" % f for f in f1s], float(np.mean(f1s)))
    return float(np.mean(f1s))


if __name__ == "__main__":
    main()

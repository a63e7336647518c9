#!/usr/bin/env python3
"""FCMF training throughput on MI355X: train samples/sec (fwd + bwd + clip + AdamW step).

  python bench.py --gpus 1 --steps 10 --warmup 3
  python bench.py --gpus N --steps K --warmup W          (starts the N ranks itself: torch.distributed.run as a child process)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Default workload (`--workload fcmf`, the headline metric) = BASELINE.json configs[1] per GPU: FCMF-base (PhoBERT-base
geometry: H768 L12 heads12 I3072 vocab 64001) bf16, batch 64 reviews x 6 aspects, seq 128, 7 images x (49 patches +
36 ROIs), precomputed ResNet-152 features, dropout on (p=0.1), 4-group AdamW + clip 1.0 + linear schedule.
1 sample = 1 review = 6 aspect forwards (what the reference's tqdm counts).  Weak scaling: every rank processes its
own 64-review shard; one gradient all-reduce (RCCL) per step.  Synthetic data, random-init weights (no dataset /
checkpoint is reachable offline); inputs are resident in HBM before the timed region.

Other workloads print their OWN line (never mixed into the headline metric):
  --workload iaog    IAOG seq2seq pre-training step (BASELINE configs[3] geometry per GPU: B=64, seq 128, Ld=12,
                     vocab 64001, 4 ROIs as run_pretraining_fcmf.py defaults), fused vocabulary projection + loss
  --workload resnet  the ResNet-152 feature extractor of the step: crops/s of the batched trunk
  --workload fcmf-large  BASELINE configs[4] geometry (XLM-R-large, seq 256, 100 ROIs); --dtype fp8 = its e4m3 MFMA path, bf16 for comparison

One JSON line on stdout (rank 0).  `roofline` describes the dominant kernel (the bf16 MFMA GEMM family):
achieved = executed GEMM FLOPs / summed launch durations measured with HIP events on the launch stream during the
last timed step.  `cpu_baseline` times the CPU oracle (a port, not the product) on config C0 (B=4) on this host.
"""
import argparse
import glob
import hashlib
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "multimodal-aspect-category-sentiment-analysis_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA peak of MI355X (MI355X_MICROARCH.md, chip-level parameters)
PEAK_FP8_TFLOPS = 5000.0    # dense fp8 MFMA peak (block-scaled form, same table)
PEAK_HBM_GBS = 8000.0       # HBM3E spec (same table); ~6.3 TB/s is what a streaming copy achieves

BASE_CFG = dict(vocab_size=64001, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                intermediate_size=3072, max_position_embeddings=258, type_vocab_size=1, pad_token_id=1,
                layer_norm_eps=1e-5, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight")
HEAD = ("classifier", "text_pooler")


def param_groups(model, lr_enc=7e-5, lr_head=7e-4):
    """4 groups by substring match on the names (run_multimodal_fcmf.py:249-287)"""
    g = [dict(params=[], weight_decay=0.01, lr=lr_enc), dict(params=[], weight_decay=0.0, lr=lr_enc),
         dict(params=[], weight_decay=0.01, lr=lr_head), dict(params=[], weight_decay=0.0, lr=lr_head)]
    for n, p in model.named_parameters():
        if not p.requires_grad:
            continue
        g[(2 if any(h in n for h in HEAD) else 0) + (1 if any(x in n for x in NO_DECAY) else 0)]["params"].append(p)
    return g


def algorithmic_flops_per_sample(cfg, S, NI, P, N, A):
    """SURVEY.md section 8(d) formula (forward, per aspect) -> fwd+bwd (x3) per sample (x A)."""
    H, I, Lyr = cfg["hidden_size"], cfg["intermediate_size"], cfg["num_hidden_layers"]

    def L(Tq, Tk):
        return 2 * Tq * H * H + 4 * Tk * H * H + 4 * Tq * Tk * H + 2 * Tq * H * H + 4 * Tq * H * I
    dense = (Lyr * L(S, S) + NI * (2 * P * 2048 * H + L(S, P) + 2 * N * 2048 * H + 8 * N * H * H + 16 * N * N * 64
                                   + 4 * N * N * H + L(S + N, S + N) + 4 * H * H) + L(1 + 2 * NI, 1 + 2 * NI)
             + 2 * H * H + 8 * H)
    return 3 * A * dense


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline():
    """the CPU oracle (torch-CPU restatement, `port`) on config C0: B=4, fp32, dropout on, clip + AdamW:
    1 warm-up step + 3 timed steps"""
    import synthetic_data as synth
    from oracle import fcmf_oracle as O
    # 16 threads = the CPU share of a one-GPU box on this pool and the fastest setting measured on it
    # (tests/oracle_thread_scan.py: 16 -> 0.53, 32 -> 0.35, 64 -> 0.16, 128 -> 0.04 samples/s fwd+bwd)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    cfg, NI, NR, B, TIMED = synth.BASE_CFG, 7, 36, 4, 3
    P = {k: v.requires_grad_(True) for k, v in synth.synth_params(synth.fcmf_param_shapes(cfg)).items()}
    groups = O.fcmf_param_groups(list(P))
    opt = torch.optim.AdamW([dict(params=[P[n] for n in g["names"]], weight_decay=g["weight_decay"], lr=g["lr"])
                             for g in groups], lr=7e-4)

    def step(seed):
        batch = synth.synth_batch(B, cfg, S=128, num_imgs=NI, num_roi=NR, seed=seed)
        opt.zero_grad(set_to_none=True)
        loss, _ = O.fcmf_step_loss(P, cfg, batch, NI, NR, training=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_([p for p in P.values() if p.grad is not None], 1.0)
        opt.step()
    step(1)                          # untimed warm-up (allocator, thread pool)
    times = []
    for i in range(TIMED):
        t0 = time.perf_counter()
        step(2 + i)
        times.append(time.perf_counter() - t0)
    mean = sum(times) / len(times)
    return dict(value=round(B / mean, 4), unit="samples/s", cores=torch.get_num_threads(), kind="port",
                cpu=cpu_model(), host_cpu_count=os.cpu_count(), torch_threads=torch.get_num_threads(),
                step_seconds=[round(t, 2) for t in times],
                sample=f"config C0 (B=4 reviews x 6 aspects, seq128, 7x(49+36), fp32, dropout on, clip+AdamW): "
                       f"1 warm-up + {TIMED} timed steps, mean {mean:.1f} s/step")


def committed_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 counter passes (tools/pmc_traffic.py) --
    PMC counters cannot be read from inside this process.  The file carries the sha256 of the gemm.hip it was
    measured with: a stale measurement (kernel source changed since) is reported as null, never as a number."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            doc = json.load(f)
        with open(os.path.join(PKG, "csrc", "gemm.hip"), "rb") as f:
            sha = hashlib.sha256(f.read()).hexdigest()
        if doc.get("gemm_hip_sha256") != sha:
            return None, os.path.basename(files[-1]) + " (stale: gemm.hip changed since it was measured)"
        return doc["kernels"].get(kernel, {}).get("hbm_bytes_per_launch"), os.path.basename(files[-1])
    except (OSError, ValueError, KeyError):
        return None, None


def gemm_roofline(trace):
    per = {}
    for name, flops, ms in trace:
        e = per.setdefault(name, [0, 0.0, 0.0])
        e[0] += 1; e[1] += flops; e[2] += ms
    mf = {k: v for k, v in per.items() if k.startswith(("gemm_bf16_", "gemm_fp8_"))}   # the MFMA kernels (tile256 / tile192 / 128x128; e4m3)
    if not mf:
        return None
    dom = max(mf, key=lambda k: mf[k][2])
    n, fl, ms = mf[dom]
    tot_fl, tot_ms = sum(v[1] for v in mf.values()), sum(v[2] for v in mf.values())
    ach = fl / (ms * 1e-3) / 1e12
    traffic, src = committed_traffic(dom)
    peak = PEAK_FP8_TFLOPS if dom.startswith("gemm_fp8_") else PEAK_BF16_TFLOPS
    return dict(bound="mfma", kernel=dom, achieved=round(ach, 1), peak=peak, unit="TFLOP/s",
                frac=round(ach / peak, 4), traffic=traffic, traffic_source=src, launches=n,
                avg_launch_ms=round(ms / n, 4), flops_per_launch=fl / n,
                all_bf16_gemms=dict(launches=sum(v[0] for v in mf.values()), achieved=round(tot_fl / (tot_ms * 1e-3) / 1e12, 1),
                                    ms_per_step=round(tot_ms, 3)),
                per_kernel={k: dict(launches=v[0], tflops=round(v[1] / (v[2] * 1e-3) / 1e12, 1), ms=round(v[2], 3))
                            for k, v in sorted(per.items(), key=lambda kv: -kv[1][2])})


def timed_loop(step, args, world, dev, trace=True):
    """W warm-up steps, then EXACTLY K steps between barrier + synchronize pairs; MAX over ranks"""
    from fcmf_framework import ops
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = None
    for i in range(args.steps):
        if trace and i == args.steps - 1:
            ops.gemm_trace_begin()       # HIP events around every GEMM launch of the last timed step
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    tr = ops.gemm_trace_end() if trace else []
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    return dt, tr, out


# ---------------------------------------------------------------------------------------
LARGE_CFG = dict(vocab_size=250002, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16,
                 intermediate_size=4096, max_position_embeddings=514, type_vocab_size=1, pad_token_id=1,
                 layer_norm_eps=1e-5, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)


def run_fcmf(args, rank, world, dev, large=False):
    import synthetic_data as synth
    from fcmf_framework import ops
    from fcmf_framework.dp import GradArena, GradReducer
    from fcmf_framework.fcmf_multimodal import FCMF
    from fcmf_framework.optimization import FusedAdamW, get_linear_schedule_with_warmup
    from fcmf_framework.roberta import RobertaConfig, RobertaModel

    CFG = LARGE_CFG if large else BASE_CFG
    S, NI, NR, A, B = (256, 7, 100, 6, args.batch) if large else (128, 7, 36, 6, args.batch)
    torch.manual_seed(42)
    hf = tempfile.mkdtemp(prefix="hf_")
    RobertaModel(RobertaConfig(**CFG)).save_pretrained(hf)
    model = FCMF(hf, num_labels=4, num_imgs=NI, num_roi=NR).to(dev)
    model.train(not args.no_dropout)
    ops.manual_seed(42 + rank)
    ops.set_compute_dtype(torch.float32 if args.dtype == "fp32" else torch.bfloat16)
    ops.set_fp8(args.dtype == "fp8")     # forward + dX GEMMs on e4m3 operands (BASELINE configs[4]); everything else as in bf16
    opt = FusedAdamW(param_groups(model), lr=7e-4)
    sched = get_linear_schedule_with_warmup(opt, int(0.1 * 1000), 1000)
    red = arena = None
    if world > 1 or args.dp_one_rank or not args.no_arena:
        arena = GradArena.for_model(model)     # as run_multimodal_fcmf.py does (leaves out bert.cell.pooler: it never gets a gradient)
        if world > 1 or args.dp_one_rank:
            red = GradReducer(arena, exchange=args.dp_exchange, native=args.dp_native, group_mb=args.dp_group_mb,
                              single_rank=args.dp_one_rank)
            red.broadcast_parameters(0)
    host = synth.synth_batch(B, CFG, S=S, num_imgs=NI, num_roi=NR, num_aspects=A, seed=42 + rank)
    batch = {k: v.to(dev) for k, v in host.items()}

    def step(batch=batch):
        if arena is not None:
            arena.zero()
        else:
            opt.zero_grad(set_to_none=True)
        logits = model.forward_aspects(batch["input_ids"], batch["visual_embeds_att"], batch["roi_embeds_att"],
                                       batch["roi_coors"], batch["token_type_ids"], batch["attention_mask"],
                                       batch["added_attention_mask"])
        loss = model.loss_aspects(logits, batch["labels"])
        loss.backward()
        if red is not None:
            red.finish()
        opt.step(max_grad_norm=1.0)
        sched.step()
        return loss

    from fcmf_framework.ops import deferred_dw
    dw0 = (deferred_dw.batched_launches, deferred_dw.batched_matrices)
    dt, trace, loss = timed_loop(step, args, world, dev)
    comm = red.stats() if red is not None else None
    if comm is not None:
        n = max(1, args.steps)
        comm["dw_batched_launches_per_step"] = round((deferred_dw.batched_launches - dw0[0]) / n, 1)
        comm["dw_matrices_per_batched_launch"] = round((deferred_dw.batched_matrices - dw0[1]) / max(1, deferred_dw.batched_launches - dw0[0]), 2)
    if rank != 0:
        return None
    # host -> HBM: (1) the bare copy of one batch from pinned memory; (2) the step loop fed through the drivers' DevicePrefetcher --
    # every step consumes a FRESH host batch (pinned producer buffers, copy stream, one batch ahead): `value_with_h2d` is that
    # loop's rate.  Never part of `value` (inputs are resident before ITS timed region).
    # (single-GPU runs only: with more ranks every step is a collective, and the other ranks have left by now -- the 2-rank
    #  rehearsal of round 3 caught rank 0 waiting for them here)
    h2d_ms = ms_step_pf = None
    n_pf = max(3, min(args.steps, 8))
    if world == 1:
        from device_prefetch import DevicePrefetcher
        pinned = {k: v.pin_memory() for k, v in host.items()}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            tmp = {k: v.to(dev, non_blocking=True) for k, v in pinned.items()}
        torch.cuda.synchronize()
        h2d_ms = (time.perf_counter() - t0) / 3 * 1e3
        del tmp
        # PAGEABLE host batches, as a DataLoader's collate produces them (built before the timed region): the prefetcher's worker
        # thread page-locks each one (a host memcpy) while the previous step runs -- round-3 advisor finding: feeding the same
        # already-pinned dict made that stage a no-op
        # (a few distinct pageable batches, cycled: each pass through the prefetcher page-locks its batch again)
        fresh = [{k: v.clone() for k, v in host.items()} for _ in range(3)]
        pre = DevicePrefetcher(None, dev)
        n_fill = pre._ring_len + 1                       # pipeline fill + the one-time allocation of the pinned staging ring (each
        pre.loader = iter([fresh[i % 3] for i in range(n_fill + n_pf)])     # new ring buffer page-locks 300 MB: ~60 ms, once)
        pf = iter(pre)
        for _ in range(n_fill):
            step(next(pf))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in pf:
            step(b)
        torch.cuda.synchronize()
        ms_step_pf = (time.perf_counter() - t0) / n_pf * 1e3
    ms_step = dt / args.steps * 1e3
    out = {
        "metric": f"train samples/sec (fwd+bwd+step) FCMF-large seq256x100ROI ({args.dtype}; BASELINE configs[4])"
        if large else "train samples/sec (fwd+bwd+step) FCMF seq128x36ROI",
        "value": round(world * B * args.steps / dt, 2), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_step, 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic (seeded batch, random-init weights)",
        "config": {"workload": (f"FCMF-large (XLM-R-large geometry H1024 L24 heads16 I4096 vocab 250002) fine-tune step, BASELINE configs[4] "
                                f"geometry in {args.dtype}: batch {B} reviews x 6 aspects per GPU, seq 256, 7 images x (49 patches + 100 ROIs), "
                                "precomputed features, dropout " if large else
                                "FCMF-base fine-tune step, BASELINE configs[1]: batch 64 reviews x 6 aspects per GPU, "
                                "seq 128, 7 images x (49 patches + 36 ROIs), precomputed ResNet-152 features, dropout ")
                               + ("off" if args.no_dropout else "0.1") + ", clip 1.0 + 4-group AdamW + linear schedule",
                   "global_batch": world * B, "per_gpu_batch": B, "seq_len": S, "parallelism": f"dp{world}",
                   "sample_unit": "1 review = 6 aspect forwards",
                   "dense_flops_per_sample_fwd_bwd": algorithmic_flops_per_sample(CFG, S, NI, 49, NR, A)},
        "loss": round(float(loss.item()), 4),
        "h2d_ms_per_batch": None if h2d_ms is None else round(h2d_ms, 2),
        "value_with_h2d": None if ms_step_pf is None else round(world * B / (ms_step_pf * 1e-3), 2),
        "ms_per_step_with_h2d": None if ms_step_pf is None else round(ms_step_pf, 2),
        "h2d": (f"{n_pf} steps, each on a PAGEABLE host batch through device_prefetch.DevicePrefetcher (worker thread copies it into the pinned staging ring, copy stream, one batch ahead; timed after the ring is allocated)"
                if world == 1 else "measured on single-GPU runs only"),
        "roofline": gemm_roofline(trace),
    }
    if comm is not None:
        rl = out["roofline"] or {}
        dwk = (rl.get("per_kernel") or {}).get("gemm_bf16_dw_batched_kernel")
        comm["dw_batched_tflops"] = dwk["tflops"] if dwk else None      # the batched weight-gradient kernel keeps running under DP
        out["dp"] = comm
    if world == 1 and not args.no_cpu_baseline and not large:
        out["cpu_baseline"] = cpu_baseline()
    return out


def run_iaog(args, rank, world, dev):
    """IAOG pre-training step (run_pretraining_fcmf.py:295-337): encoder (1 'aspect') + 12-block decoder + tied 64001-wide
    vocabulary projection + CE(ignore_index) + clip + 2-group AdamW"""
    import synthetic_data as synth
    from fcmf_framework import ops
    from fcmf_framework.dp import GradArena, GradReducer
    from fcmf_framework.fcmf_pretraining import FCMFSeq2Seq
    from fcmf_framework.optimization import FusedAdamW
    from fcmf_framework.roberta import RobertaConfig, RobertaModel
    cfg = BASE_CFG
    V, NI, NR, B, S, Ld = cfg["vocab_size"], 7, 4, args.batch, 128, args.dec_len
    hf = tempfile.mkdtemp(prefix="hf_")
    torch.manual_seed(42)
    RobertaModel(RobertaConfig(**cfg)).save_pretrained(hf)
    ops.set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    ops.manual_seed(42 + rank)
    model = FCMFSeq2Seq(V, 20, hf, NI, NR, 1.0)
    model.decoder.embedding = torch.nn.Embedding(V, model.decoder.num_hiddens)       # run_pretraining_fcmf.py:189
    model = model.to(dev).train(not args.no_dropout)
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    opt = FusedAdamW([{'params': [p for n, p in named if not any(nd in n for nd in NO_DECAY)], 'weight_decay': 1e-5},
                      {'params': [p for n, p in named if any(nd in n for nd in NO_DECAY)], 'weight_decay': 0.0}], lr=3e-5)
    arena = GradArena.for_model(model)      # one memset per step instead of a zero fill per weight gradient
    red = None
    if world > 1 or args.dp_one_rank:
        red = GradReducer(arena, exchange=args.dp_exchange, native=args.dp_native, group_mb=args.dp_group_mb,
                          single_rank=args.dp_one_rank)
        red.broadcast_parameters(0)
    b = synth.synth_batch(B, cfg, S=S, num_imgs=NI, num_roi=NR, num_aspects=1, seed=3 + rank, coord_dtype=torch.float32)
    b = {k: v.to(dev) for k, v in b.items()}
    dec = torch.randint(3, V, (B, Ld), generator=torch.Generator().manual_seed(1 + rank)).to(dev)
    lab = torch.roll(dec, -1, dims=1)
    lab[:, -1] = -100

    def step():
        if arena is not None:
            arena.zero()
        else:
            opt.zero_grad(set_to_none=True)
        loss = model.forward_loss(b["input_ids"][:, 0], dec, lab, b["visual_embeds_att"], b["roi_embeds_att"], b["roi_coors"],
                                  b["token_type_ids"][:, 0], b["attention_mask"][:, 0], b["added_attention_mask"][:, 0])
        loss.backward()
        if red is not None:
            red.finish()
        opt.step(max_grad_norm=1.0)
        return loss

    dt, trace, loss = timed_loop(step, args, world, dev)
    comm = red.stats() if red is not None else None
    if rank != 0:
        return None
    return {
        "dp": comm,
        "metric": "IAOG pre-train samples/sec (fwd+bwd+step) FCMF-base seq128", "value": round(world * B * args.steps / dt, 2),
        "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic (seeded batch, random-init weights)",
        "config": {"workload": f"IAOG seq2seq pre-training step (BASELINE configs[3] geometry per GPU): batch {B}, seq 128, "
                               f"7 images x (49 patches + {NR} ROIs), decoder length {Ld}, vocabulary {V}, fused vocabulary "
                               "projection + CE(ignore_index=-100), clip 1.0 + 2-group AdamW(wd 1e-5)",
                   "global_batch": world * B, "per_gpu_batch": B, "seq_len": S, "parallelism": f"dp{world}"},
        "loss": round(float(loss.item()), 4), "roofline": gemm_roofline(trace),
    }


def run_resnet(args, rank, world, dev):
    """ResNet-152 feature extractor (run_multimodal_fcmf.py:449-460): crops/s of the batched, grouped-BatchNorm trunk in
    train() mode; one 'step' = the image pass of one batch (NI call groups of B crops, 224 x 224)"""
    import synthetic_data as synth
    from fcmf_framework import ops
    from fcmf_framework.resnet import resnet152
    from fcmf_framework.resnet_utils import myResNetImg
    B, NI = args.batch, 7
    ops.set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    torch.manual_seed(42)
    net = myResNetImg(resnet152().to(dev), False, dev).train()
    x = synth.synth_crops(NI * B, 224, seed=rank).to(dev)

    def step():
        return net.forward_groups(x, NI, 7, tokens=True)

    dt, trace, y = timed_loop(step, args, world, dev)
    if rank != 0:
        return None
    crops = NI * B
    gflop_per_crop = 2 * 11.51         # ResNet-152 forward at 224x224: 11.5 GMACs (He et al. 2016 table 1: 11.3e9 FLOPs = MACs)
    ms = dt / args.steps * 1e3
    return {
        "metric": "ResNet-152 trunk crops/sec (forward, train-mode grouped BatchNorm)", "value": round(world * crops * args.steps / dt, 1),
        "unit": "crops/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic crops, random-init weights",
        "config": {"workload": f"ResNet-152 trunk forward of one image pass: {NI} call groups x {B} crops of 224x224, NHWC implicit-GEMM "
                               "convolutions (3x3, strided 1x1, the 7x7 stem over RGB0 runs), per-group batch statistics out of the "
                               "convolution epilogues, -> [B*NI, 49, 2048]",
                   "crops_per_step": crops, "parallelism": f"dp{world}"},
        "achieved_tflops_conv": round(crops * gflop_per_crop / ms, 1), "finite": bool(torch.isfinite(y).all().item()),
        "roofline": gemm_roofline(trace),
    }


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start one rank per GPU under torch.distributed.run as a CHILD process
    (this process has not touched the GPU and never will), relay rank 0's JSON line, return the child's exit code (non-zero
    when any rank failed).  reference: torchrun + init_process_group('nccl') at run_multimodal_fcmf.py:126-131,237-240"""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL between processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout:
        if ln.startswith("{") and '"metric"' in ln:
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    if rc == 0 and line is None:
        sys.stderr.write("bench.py: the ranks exited cleanly but rank 0 printed no result line\n")
        rc = 1
    return rc


def launch_check(args, rank, world):
    """--launch-check: the rendezvous and one collective of the N-rank launch path, WITHOUT touching a GPU (the CPU test of
    the bare `python bench.py --gpus N` command: tests/test_dp_gloo.py)"""
    dist.init_process_group(args.backend)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    ok = t.item() == world * (world + 1) / 2
    if rank == 0:
        print(json.dumps({"metric": "launch check (no GPU work)", "value": None, "n_gpus": world, "launch_check": bool(ok),
                          "dp_backend": dist.get_backend(), "dp_ranks_seen": dist.get_world_size()}), flush=True)
    dist.destroy_process_group()
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="fcmf", choices=["fcmf", "fcmf-large", "iaog", "resnet"])
    ap.add_argument("--batch", type=int, default=None, help="reviews per GPU (default 64; fcmf-large: 16)")
    ap.add_argument("--dec_len", type=int, default=12, help="IAOG decoder length")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "fp8"],
                    help="fp8: e4m3 forward / dX GEMMs (v_mfma_scale_f32_16x16x128_f8f6f4) with bf16 weight-gradient GEMMs: --workload fcmf-large")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dropout", action="store_true")
    ap.add_argument("--no-arena", dest="no_arena", action="store_true",
                    help="single GPU: per-weight gradient tensors instead of the flat gradient arena the drivers use (always on for --gpus > 1)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL)")
    ap.add_argument("--dp-one-rank", action="store_true",
                    help="rehearsal on ONE GPU: run the whole data-parallel machinery (arena, hooks, launch groups, one RCCL collective per "
                         "bucket on the side stream, finish()) in a process group of one rank; the line carries the `dp` object")
    ap.add_argument("--dp-exchange", dest="dp_exchange", default="fp32", choices=["fp32", "bf16"],
                    help="gradient exchange: float32 all-reduce in place (default, DDP-comparable) or bf16 on the links with float32 accumulation")
    ap.add_argument("--dp-native", dest="dp_native", action="store_true",
                    help="all-reduce through the library's own RCCL binding (fcmf_dp_allreduce_bucket) instead of torch.distributed")
    ap.add_argument("--dp-group-mb", dest="dp_group_mb", type=float, default=160.0,
                    help="launch granularity of the gradient exchange: consecutive ready buckets go out (and their queued weight "
                         "gradients are multiplied together) in groups of at least this many MB")
    ap.add_argument("--launch-check", dest="launch_check", action="store_true",
                    help="only the N-rank rendezvous + one collective, no GPU work (CPU test of the launch path)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: one process per GPU, started here (before anything touches the GPU)
        sys.exit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if args.gpus != world:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}")
    if args.launch_check:
        sys.exit(launch_check(args, rank, world))
    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a five-line version banner to stdout when the
    # first communicator is created -- seen in the one-rank rehearsal): from here on file descriptor 1 of this process is its stderr,
    # and the JSON line goes to the ORIGINAL stdout kept in `json_out`.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    local = int(os.environ.get("LOCAL_RANK", 0))
    if os.environ.get("FCMF_BENCH_SINGLE_DEVICE"):   # rehearsal: several ranks share cuda:0 (use with --backend gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or args.dp_one_rank:
        if world == 1:                                          # --dp-one-rank outside a launcher: a group of this process alone
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)      # RCCL over xGMI
        else:
            dist.init_process_group(args.backend)
    if args.batch is None:
        args.batch = 16 if args.workload == "fcmf-large" else 64
    runner = {"fcmf": run_fcmf, "fcmf-large": lambda *a: run_fcmf(*a, large=True), "iaog": run_iaog, "resnet": run_resnet}
    out = runner[args.workload](args, rank, world, dev)
    if rank == 0:
        if dist.is_initialized():
            out["dp_backend"] = dist.get_backend()
            out["dp_ranks_seen"] = dist.get_world_size()
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Data-parallel step on the GPU: two ranks (sharing the box's one MI355X, gloo between them -- RCCL refuses two ranks on one device)
run the product's DP path end to end -- gradient arena, deferred + batched weight gradients flushed by launch group, in-place
accumulation of the shared fusion layer's gradients, bucketed all-reduce overlapped with backward, fused clip + AdamW -- and must
reproduce the SINGLE-process step on the concatenated batch (what torch DDP guarantees the reference, run_multimodal_fcmf.py:237-240,
421, 463-489): every gradient and every parameter after the update."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT

pytestmark = pytest.mark.gpu
NI, NR, B, S = 2, 5, 4, 16


def _build(dev, dtype):
    sys.path.insert(0, PKG); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import synthetic_data as synth
    from helpers import build_fcmf
    from fcmf_framework import ops
    ops.set_compute_dtype(dtype)
    model, _ = build_fcmf(synth.TINY_CFG, NI, NR, dev)
    model.eval()                                       # dropout off: the two ranks' masks would differ from the single process's
    batch = synth.synth_batch(B, synth.TINY_CFG, S=S, num_imgs=NI, num_roi=NR, seed=11)
    return model, batch


def _step(model, b, arena, red, opt):
    arena.zero()
    logits = model.forward_aspects(b["input_ids"], b["visual_embeds_att"], b["roi_embeds_att"], b["roi_coors"],
                                   b["token_type_ids"], b["attention_mask"], b["added_attention_mask"])
    model.loss_aspects(logits, b["labels"]).backward()
    if red is not None:
        red.finish()
    # (numpy: pickled by value -- CPU tensors travel through a multiprocessing queue as shared-memory handles that die with the worker)
    grads = {n: p.grad.detach().float().cpu().numpy().copy() for n, p in model.named_parameters() if p.grad is not None}
    opt.step(max_grad_norm=1.0)
    torch.cuda.synchronize()
    return grads, {n: p.detach().float().cpu().numpy().copy() for n, p in model.named_parameters()}


def _worker(rank, world, port, dtype_name, q, backend="gloo", exchange="fp32", native=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)      # as bench.py does
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model, batch = _build(dev, getattr(torch, dtype_name))
        from fcmf_framework import ops
        from fcmf_framework.dp import GradArena, GradReducer
        from fcmf_framework.optimization import FusedAdamW
        arena = GradArena.for_model(model)
        red = GradReducer(arena, bucket_mb=0.05, group_mb=0.12, exchange=exchange, native=native,      # tiny model: several buckets, several launch groups
                          single_rank=world == 1)
        red.broadcast_parameters(0)
        opt = FusedAdamW([p for p in model.parameters()], lr=1e-3)
        lo = rank * (B // world)
        shard = {k: v[lo:lo + B // world].to(dev) for k, v in batch.items()}
        for _ in range(2):            # step 1 fixes the dead set; step 2 runs with early launches
            grads, params = _step(model, shard, arena, red, opt)
        q.put((rank, grads, params, list(red.launch_log), len(red.buckets), len(red.group_log), ops.deferred_dw.batched_matrices))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dtype_name,world,backend,exchange,native", [
    ("float32", 2, "gloo", "fp32", False), ("bfloat16", 2, "gloo", "fp32", False),
    ("bfloat16", 1, "nccl", "fp32", False), ("float32", 1, "nccl", "bf16", False), ("bfloat16", 1, "nccl", "fp32", True)])
def test_dp_step_equals_the_single_process_step(dev, dtype_name, world, backend, exchange, native):
    """(1, "nccl"): the same path over RCCL -- torch's ProcessGroupNCCL bound to the device, the buckets' collectives issued on
    the reducer's side stream (all-reduce; the bf16 exchange's all-to-all + all-gather; the library's own communicator) -- as
    far as one GPU allows: a one-rank group (the mean over one rank is the identity, up to the bf16 exchange's rounding)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, dtype_name, q, backend, exchange, native)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # ---- the single-process reference: the same two steps on the whole batch ------------------------------------------
    from fcmf_framework import ops
    from fcmf_framework.dp import GradArena
    from fcmf_framework.optimization import FusedAdamW
    old = ops.grad_arena()
    try:
        model, batch = _build(dev, getattr(torch, dtype_name))
        arena = GradArena.for_model(model)
        opt = FusedAdamW([p for p in model.parameters()], lr=1e-3)
        full = {k: v.to(dev) for k, v in batch.items()}
        for _ in range(2):
            ref_g, ref_p = _step(model, full, arena, None, opt)
    finally:
        if "arena" in locals():
            arena.deactivate()
        ops.set_grad_arena(old)
        ops.set_compute_dtype(torch.float32)
        ops.shadows.clear()
    tol = 2e-4 if dtype_name == "float32" else 6e-2      # bf16: each rank rounds its own half-batch activations
    if exchange == "bf16":
        tol = 8e-3                                       # gradients rounded to bf16 on the links (2^-8 per element)
    for rank, grads, params, log, nb, ngroups, nbatched in res:
        assert nb >= 4 and log == list(range(nb)) and 2 <= ngroups <= nb, (nb, log, ngroups)      # every bucket once, in arena order, in groups
        assert grads.keys() == ref_g.keys()
        worst = ("", 0.0)
        for n, g in ref_g.items():
            g = torch.from_numpy(g)
            if n.endswith((".key.bias", "box_head.linears.1.bias")) or g.norm().item() < 1e-7:
                continue                                                  # analytically zero: rounding noise
            e = (torch.from_numpy(grads[n]) - g).norm().item() / g.norm().item()
            worst = max(worst, (n, e), key=lambda t: t[1])
        assert worst[1] < tol, (rank, worst)
        # the update: both ranks hold the same parameters, and they are the single process's
        for n, w in ref_p.items():
            assert (params[n] == res[0][2][n]).all(), ("ranks diverged", n)
            d = float(abs(params[n] - w).max())
            assert d < (2e-5 if dtype_name == "float32" and exchange == "fp32" else 2.1e-3), (n, d)       # (lr 1e-3: one Adam step moves a weight by <= 1e-3)
        if dtype_name == "bfloat16":
            assert nbatched > 0          # the batched weight-gradient entry point ran under data parallelism

"""world_size-2 gloo test (CPU) of the data-parallel gradient exchange: the bucketed, hook-driven
all-reduce must equal the gradient mean of a single process that sees both shards."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "multimodal-aspect-category-sentiment-analysis_amd")


def _worker(rank, world, port, q):
    sys.path.insert(0, PKG)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fcmf_framework.dp import GradReducer
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Tanh(), torch.nn.Linear(32, 8), torch.nn.Linear(8, 3))
    dead = torch.nn.Linear(4, 4)                      # never used: like the text encoder's pooler
    params = list(net.parameters()) + list(dead.parameters())
    red = GradReducer(params, bucket_mb=0.001)        # tiny buckets -> several collectives
    if rank == 1:
        with torch.no_grad():
            for p in params:
                p.add_(1.0)                           # diverge, then broadcast must repair it
    red.broadcast_parameters(0)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(8, 16, generator=g)
    y = torch.randn(8, 3, generator=g)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
    for step in range(2):                             # two steps: the reducer must re-arm itself
        red.arena.zero()
        ((net(xs) - ys) ** 2).mean().backward()
        red.finish()
    grads = [p.grad.clone() for p in net.parameters()]
    # single-process reference on the full batch (mean of the two shard means)
    for p in net.parameters():
        p.grad = None
    ((net(x) - y) ** 2).mean().backward()
    err = max((a - p.grad).abs().max().item() for a, p in zip(grads, net.parameters()))
    q.put((rank, err, len(red.buckets), all(p.grad is None for p in dead.parameters())))
    dist.destroy_process_group()


def _accum_worker(rank, world, port, q):
    """gradient accumulation: the exchange happens only on the boundary micro-step (`enabled`), on gradients that
    live in ONE flat arena and are all-reduced in place"""
    sys.path.insert(0, PKG)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fcmf_framework.dp import GradArena, GradReducer
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Tanh(), torch.nn.Linear(32, 3))
    params = list(net.parameters())
    arena = GradArena(params)
    red = GradReducer(arena, bucket_mb=0.0005)
    red.broadcast_parameters(0)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(16, 16, generator=g)
    y = torch.randn(16, 3, generator=g)
    launches = []
    orig = red._launch
    red._launch = lambda bi: (None if red._launched[bi] else launches.append((micro[0], bi)), orig(bi))[1]
    micro = [0]
    ok_views = True
    for step in range(2):
        arena.zero()
        for m in range(2):                            # two micro-steps of 4 samples per rank
            micro[0] = m
            lo = rank * 8 + m * 4
            red.enabled = (m == 1)
            (((net(x[lo:lo + 4]) - y[lo:lo + 4]) ** 2).mean() / 2).backward()
        red.finish()
        ok_views = ok_views and all(p.grad.data_ptr() == arena.view[id(p)].data_ptr() for p in params)
    grads = [p.grad.clone() for p in params]
    for p in params:
        p.grad = None
    ((net(x) - y) ** 2).mean().backward()
    err = max((a - p.grad).abs().max().item() for a, p in zip(grads, params))
    q.put((rank, err, all(m == 1 for m, _ in launches), len(launches), len(red.buckets), ok_views))
    dist.destroy_process_group()


def test_grad_reducer_accumulation_boundary_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_accum_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, boundary_only, nlaunch, nb, ok_views in res:
        assert err < 1e-6, (rank, err)
        assert boundary_only and nlaunch == 2 * nb and nb > 1      # one in-place collective per bucket per optimizer step
        assert ok_views                                            # p.grad IS the arena slice after the exchange


def test_grad_reducer_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, nb, dead_ok in res:
        assert err < 1e-6, (rank, err)
        assert nb > 1 and dead_ok

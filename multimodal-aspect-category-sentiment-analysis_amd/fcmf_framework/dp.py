"""Single-node data parallelism for the FCMF step: one process per GPU, replicated parameters, the minibatch
sharded across ranks, ONE exchange per optimizer step -- the gradient mean.

The reference wraps the model in torch DDP with find_unused_parameters=True over NCCL
(run_multimodal_fcmf.py:237-240).  Here the exchange is explicit and sized for xGMI:

* `GradArena`: every parameter's gradient is a view of ONE flat float32 buffer, laid out in the order backward
  produces gradients (reverse registration order; q/k/v weights of a fused attention block adjacent so that their
  [3H, H] weight-gradient GEMM writes one slice).  The backward kernels write straight into the slices
  (`ops.alloc_grad`): one memset per step replaces the per-tensor zero fills, autograd adopts the slices as
  `p.grad` without copying, and a bucket of gradients is a contiguous range of the buffer.
* `GradReducer`: all-reduces (RCCL, `torch.distributed` backend "nccl" on ROCm; gloo in the CPU tests) each bucket
  IN PLACE -- no concatenation, no copy-back -- on a side stream as soon as the last gradient of the bucket has been
  accumulated, so the collective overlaps the rest of backward; divides by the world size on that stream; reports how
  long the main stream actually waited for communication (`stats()`).  Parameters that never receive a gradient (the
  text encoder's pooler, dead at fcmf_pretraining.py:41) are left out of the arena instead of searched for every step
  (DDP's find_unused_parameters).
"""
import torch
import torch.distributed as dist

ALIGN = 64      # elements: every slice starts on a 256-byte boundary (16-byte vector accesses, clean bucket edges)


class GradArena:
    def __init__(self, params, blocks=()):
        """params: the parameters that receive gradients, in registration order.  blocks: lists of parameters that
        must sit next to each other, in the given order (fused q|k|v weight / bias gradients)."""
        params = [p for p in params if p.requires_grad]
        seen, uniq = set(), []
        for p in params:                      # tied parameters (decoder.dense.weight) appear once
            if id(p) not in seen:
                seen.add(id(p))
                uniq.append(p)
        self.params = uniq
        in_block = {id(p): bi for bi, b in enumerate(blocks) for p in b}
        order, done = [], set()
        for p in reversed(self.params):       # the order backward produces them
            if id(p) in done:
                continue
            group = blocks[in_block[id(p)]] if id(p) in in_block else [p]
            for q in group:
                if id(q) in seen and id(q) not in done:
                    done.add(id(q))
                    order.append(q)
        self.order = order
        self.offset, off = {}, 0
        for p in order:
            adj = id(p) in in_block and blocks[in_block[id(p)]][0] is not p      # packed tight behind its block head
            if not adj:
                off = (off + ALIGN - 1) // ALIGN * ALIGN
            self.offset[id(p)] = off
            off += p.numel()
        self.total = (off + ALIGN - 1) // ALIGN * ALIGN
        dev = self.params[0].device
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.view = {id(p): self.flat[self.offset[id(p)]:self.offset[id(p)] + p.numel()].view_as(p) for p in order}
        self._by_ptr = {p.data_ptr(): p for p in order}
        self._taken = set()
        self.on_zero = []                     # callbacks (the reducer resets its bucket state here)
        self.activate()

    @classmethod
    def for_model(cls, model, skip=lambda name: "bert.cell.pooler" in name):
        """arena over the model's live parameters; q|k|v parameters of every fused attention block adjacent"""
        from .fused import QKVStorageMixin
        blocks = []
        for m in model.modules():
            if isinstance(m, QKVStorageMixin):
                blocks.append([m.query.weight, m.key.weight, m.value.weight])
                blocks.append([m.query.bias, m.key.bias, m.value.bias])
        return cls([p for n, p in model.named_parameters() if not skip(n)], blocks)

    # ---- step protocol -------------------------------------------------------------------------
    def activate(self):
        from . import ops
        ops.set_grad_arena(self)

    def deactivate(self):
        from . import ops
        if ops.grad_arena() is self:
            ops.set_grad_arena(None)

    def zero(self):
        """start of an optimizer step: ONE memset; gradients are detached so that the first producer's slice is
        adopted by autograd as p.grad without a copy.  (With gradient accumulation call this once per optimizer
        step, not per micro-step.)"""
        self.flat.zero_()
        self._taken.clear()
        for p in self.order:
            p.grad = None
        for cb in self.on_zero:
            cb()

    def take(self, param):
        """the slice of `param` if nothing has claimed it in this step (else None: the caller uses a temporary and
        autograd accumulates it into the slice in place)"""
        p = self._by_ptr.get(param.data_ptr())
        if p is None or id(p) in self._taken or p.grad is not None:
            return None
        self._taken.add(id(p))
        return self.view[id(p)]

    def take_block(self, params):
        """one tensor covering the adjacent slices of `params` (or None)"""
        ps = [self._by_ptr.get(q.data_ptr()) for q in params]
        if any(p is None or id(p) in self._taken or p.grad is not None for p in ps):
            return None
        off = self.offset[id(ps[0])]
        for a, b in zip(ps, ps[1:]):
            if self.offset[id(b)] != self.offset[id(a)] + a.numel():
                return None
        for p in ps:
            self._taken.add(id(p))
        return self.flat[off:off + sum(p.numel() for p in ps)]

    def adopt(self, p):
        """make p.grad the arena slice (copying a gradient that was produced elsewhere into it)"""
        v = self.view.get(id(p))
        if v is None or p.grad is None or p.grad.data_ptr() == v.data_ptr():
            return
        v.copy_(p.grad)
        p.grad = v

    def adopt_all(self):
        for p in self.order:
            self.adopt(p)


class GradReducer:
    def __init__(self, arena, bucket_mb=128, process_group=None, overlap=True):
        if not isinstance(arena, GradArena):                      # list of parameters (round-1 signature)
            arena = GradArena(list(arena))
        self.arena = arena
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.params = arena.order
        self.overlap = overlap
        cap = int(bucket_mb * 1024 * 1024 / 4)
        self.buckets = []                     # (lo, hi, [params]) contiguous ranges of arena.flat, backward order
        cur, lo = [], 0
        for p in arena.order:
            cur.append(p)
            hi = arena.offset[id(p)] + p.numel()
            if hi - lo >= cap:
                self.buckets.append((lo, (hi + ALIGN - 1) // ALIGN * ALIGN, cur))
                cur, lo = [], (hi + ALIGN - 1) // ALIGN * ALIGN
        if cur:
            self.buckets.append((lo, arena.total, cur))
        self._bucket_of = {id(p): bi for bi, (_, _, ps) in enumerate(self.buckets) for p in ps}
        self._hooks = []
        self._stream = None
        self.enabled = True        # set False on non-boundary micro-steps of gradient accumulation
        self._exposed_ms, self._steps, self._events = 0.0, 0, []
        self._dead = None
        if self.world > 1:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        arena.on_zero.append(self.reset)
        self.reset()

    def reset(self):
        self._ready = [set() for _ in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._work = []

    # -- called by autograd right after p.grad has been accumulated ------------------------------
    def _on_grad(self, p):
        if not self.enabled:
            return
        self.arena.adopt(p)
        bi = self._bucket_of[id(p)]
        self._ready[bi].add(id(p))
        if self.overlap and len(self._ready[bi]) == len(self.buckets[bi][2]):
            self._launch(bi)

    def _launch(self, bi):
        if self._launched[bi]:
            return
        self._launched[bi] = True
        lo, hi, ps = self.buckets[bi]
        for p in ps:                          # gradients that arrived while disabled (accumulation) or not at all
            self.arena.adopt(p)
        buf = self.arena.flat[lo:hi]
        if buf.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=buf.device)
            self._stream.wait_stream(torch.cuda.current_stream(buf.device))
            with torch.cuda.stream(self._stream):
                w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._work.append((w, buf))

    def finish(self):
        """after backward: flush buckets whose gradients never all arrived, wait for the collectives and leave the
        MEAN in the arena (p.grad of every parameter that received a gradient on ANY rank is its arena slice)."""
        if self.world == 1:
            return
        for bi in range(len(self.buckets)):
            self._launch(bi)
        cuda = self.arena.flat.is_cuda
        if cuda:
            main = torch.cuda.current_stream(self.arena.flat.device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main)
        for w, buf in self._work:
            if cuda:
                with torch.cuda.stream(self._stream):
                    w.wait()                  # (RCCL: makes the SIDE stream wait for the collective, not the host)
                    buf.div_(self.world)
            else:
                w.wait()
                buf.div_(self.world)
        if cuda:
            main.wait_stream(self._stream)
            e1.record(main)
            self._events.append((e0, e1))
        if self._dead is None:
            # which parameters received a gradient on NO rank (statically dead ones that were not left out of the
            # arena): decided once, on the first step -- they keep p.grad = None so that the optimizer skips them
            # exactly as it does in a single process (DDP's find_unused_parameters, made static)
            have = torch.tensor([0.0 if p.grad is None else 1.0 for p in self.params], device=self.arena.flat.device)
            dist.all_reduce(have, op=dist.ReduceOp.SUM, group=self.group)
            self._dead = {id(p) for p, h in zip(self.params, have.tolist()) if h == 0.0}
        for p in self.params:                 # a parameter without a LOCAL gradient still takes part in the mean
            if p.grad is None and id(p) not in self._dead:
                p.grad = self.arena.view[id(p)]
        self._steps += 1
        self._work = []

    def stats(self):
        """exposed (not overlapped) communication: how long the main stream sat in finish() per step"""
        ms = [a.elapsed_time(b) for a, b in self._events]
        self._events = []
        return dict(world=self.world, buckets=len(self.buckets), bucket_mb=[round((hi - lo) * 4 / 2 ** 20, 1) for lo, hi, _ in self.buckets],
                    arena_mb=round(self.arena.total * 4 / 2 ** 20, 1),
                    exposed_comm_ms_per_step=round(sum(ms) / max(1, len(ms)), 3) if ms else None, steps=self._steps)

    def broadcast_parameters(self, src=0):
        """replicate rank `src`'s parameters (what DDP does at wrap time)"""
        if self.world == 1:
            return
        for p in self.params:
            dist.broadcast(p.data, src=src, group=self.group)

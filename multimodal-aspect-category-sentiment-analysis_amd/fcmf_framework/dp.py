"""Single-node data parallelism for the FCMF step: one process per GPU, replicated parameters,
the minibatch sharded across ranks, ONE exchange per optimizer step -- the gradient mean.

The reference wraps the model in torch DDP with find_unused_parameters=True over NCCL
(run_multimodal_fcmf.py:237-240).  Here the exchange is explicit and sized for xGMI: gradients are
packed into a few large flat buckets in reverse parameter order (the order backward produces
them), each bucket is all-reduced over RCCL (`torch.distributed`, backend "nccl" on ROCm) on a
side stream as soon as its last gradient has been accumulated, so the collective overlaps the
rest of backward; parameters that never receive a gradient (the text encoder's pooler, dead at
fcmf_pretraining.py:41) are excluded statically instead of searched for every step.
"""
import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, params, bucket_mb=128, process_group=None, overlap=True):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.params = [p for p in params if p.requires_grad]
        self.overlap = overlap
        self.buckets = []          # list of lists of params, reverse registration order
        cap = int(bucket_mb * 1024 * 1024 / 4)
        cur, n = [], 0
        for p in reversed(self.params):
            cur.append(p)
            n += p.numel()
            if n >= cap:
                self.buckets.append(cur)
                cur, n = [], 0
        if cur:
            self.buckets.append(cur)
        self._bucket_of = {id(p): bi for bi, b in enumerate(self.buckets) for p in b}
        self._pending = [0] * len(self.buckets)
        self._flat = [None] * len(self.buckets)
        self._work = []
        self._hooks = []
        self._stream = None
        self.enabled = True        # set False on non-boundary micro-steps of gradient accumulation
        if self.world > 1:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self.reset()

    def reset(self):
        self._pending = [len(b) for b in self.buckets]
        self._ready = [set() for _ in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._work = []

    # -- called by autograd right after p.grad has been accumulated ------------------------------
    def _on_grad(self, p):
        if not self.enabled:
            return
        bi = self._bucket_of[id(p)]
        if id(p) in self._ready[bi]:
            return
        self._ready[bi].add(id(p))
        if self.overlap and len(self._ready[bi]) == len(self.buckets[bi]):
            self._launch(bi)

    def _launch(self, bi):
        if self._launched[bi]:
            return
        self._launched[bi] = True
        ps = [p for p in self.buckets[bi] if p.grad is not None]
        if not ps:
            return
        flat = torch.cat([p.grad.reshape(-1) for p in ps])
        if flat.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=flat.device)
            self._stream.wait_stream(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(self._stream):
                w = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            flat.record_stream(self._stream)
        else:
            w = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._work.append((w, flat, ps))

    def finish(self):
        """after backward: flush buckets whose gradients never all arrived (parameters without a
        gradient this step), wait for the collectives, write the MEAN back into p.grad."""
        if self.world == 1:
            return
        for bi in range(len(self.buckets)):
            self._launch(bi)
        for w, flat, ps in self._work:
            w.wait()
            if flat.is_cuda:
                torch.cuda.current_stream(flat.device).wait_stream(self._stream)
            flat.div_(self.world)
            off = 0
            for p in ps:
                n = p.numel()
                p.grad.copy_(flat[off:off + n].view_as(p.grad))
                off += n
        self.reset()

    def broadcast_parameters(self, src=0):
        """replicate rank `src`'s parameters (what DDP does at wrap time)"""
        if self.world == 1:
            return
        for p in self.params:
            dist.broadcast(p.data, src=src, group=self.group)

"""Optimizers of the FCMF training step on the MI355X kernels.

* `BertAdam`, `SCHEDULES`, `warmup_*`: API parity with the reference's
  fcmf_framework/optimization.py:23-162 (imported by the driver, never instantiated there).
* `FusedAdamW`: what the drivers actually use -- torch.optim.AdamW semantics
  (run_multimodal_fcmf.py:289) fused with `clip_grad_norm_` (:485) as two multi-tensor kernels:
  one pass computes the global L2 norm, one pass applies clip + decoupled weight decay + Adam and
  refreshes the bf16 weight copies.  Per-group lr / weight_decay as in torch; works with
  torch.optim.lr_scheduler.LambdaLR (`get_linear_schedule_with_warmup` below).
"""
import ctypes
import math

import torch
from torch.optim import Optimizer
from torch.optim.optimizer import required

from . import _hip as H
from . import ops


def warmup_cosine(x, warmup=0.002):
    if x < warmup:
        return x / warmup
    return 0.5 * (1.0 + math.cos(math.pi * x))


def warmup_constant(x, warmup=0.002):
    if x < warmup:
        return x / warmup
    return 1.0


def warmup_linear(x, warmup=0.002):
    if x < warmup:
        return x / warmup
    return 1.0 - x


SCHEDULES = {
    'warmup_cosine': warmup_cosine,
    'warmup_constant': warmup_constant,
    'warmup_linear': warmup_linear,
}


def get_linear_schedule_with_warmup(optimizer, num_warmup_steps, num_training_steps, last_epoch=-1):
    """transformers.get_linear_schedule_with_warmup (run_multimodal_fcmf.py:310-314)"""
    def lr_lambda(step):
        if step < num_warmup_steps:
            return float(step) / float(max(1, num_warmup_steps))
        return max(0.0, float(num_training_steps - step) / float(max(1, num_training_steps - num_warmup_steps)))
    return torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda, last_epoch)


class BertAdam(Optimizer):
    """BERT Adam: per-parameter gradient clipping, no bias correction, weight decay added to the
    update, built-in warmup schedule (reference optimization.py:45-162)."""

    def __init__(self, params, lr=required, warmup=-1, t_total=-1, schedule='warmup_linear',
                 b1=0.9, b2=0.999, e=1e-6, weight_decay=0.01, max_grad_norm=1.0):
        if lr is not required and lr < 0.0:
            raise ValueError("Invalid learning rate: {} - should be >= 0.0".format(lr))
        if schedule not in SCHEDULES:
            raise ValueError("Invalid schedule parameter: {}".format(schedule))
        if not 0.0 <= warmup < 1.0 and not warmup == -1:
            raise ValueError("Invalid warmup: {} - should be in [0.0, 1.0[ or -1".format(warmup))
        if not 0.0 <= b1 < 1.0:
            raise ValueError("Invalid b1 parameter: {} - should be in [0.0, 1.0[".format(b1))
        if not 0.0 <= b2 < 1.0:
            raise ValueError("Invalid b2 parameter: {} - should be in [0.0, 1.0[".format(b2))
        if not e >= 0.0:
            raise ValueError("Invalid epsilon value: {} - should be >= 0.0".format(e))
        defaults = dict(lr=lr, schedule=schedule, warmup=warmup, t_total=t_total, b1=b1, b2=b2, e=e,
                        weight_decay=weight_decay, max_grad_norm=max_grad_norm)
        super().__init__(params, defaults)
        self._scratch = None

    def _lr(self, group, state):
        if group['t_total'] != -1:
            return group['lr'] * SCHEDULES[group['schedule']](state['step'] / group['t_total'], group['warmup'])
        return group['lr']

    def get_lr(self):
        lr = []
        for group in self.param_groups:
            for p in group['params']:
                state = self.state[p]
                if len(state) == 0:
                    return [0]
                lr.append(self._lr(group, state))
        return lr

    def step(self, closure=None):
        loss = closure() if closure is not None else None
        L = H.lib()
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is None:
                    continue
                grad = p.grad.data
                if grad.is_sparse:
                    raise RuntimeError('Adam does not support sparse gradients, please consider SparseAdam instead')
                H.require_cuda(p)
                state = self.state[p]
                if len(state) == 0:
                    state['step'] = 0
                    state['next_m'] = torch.zeros_like(p.data)
                    state['next_v'] = torch.zeros_like(p.data)
                if self._scratch is None or self._scratch.device != p.device:
                    self._scratch = torch.zeros(1, dtype=torch.float64, device=p.device)
                g = grad.contiguous().float()
                H.check(L.fcmf_bertadam(H.ptr(p.data), H.ptr(g), H.ptr(state['next_m']), H.ptr(state['next_v']),
                                        p.numel(), float(self._lr(group, state)), group['b1'], group['b2'], group['e'],
                                        group['weight_decay'], group['max_grad_norm'], H.ptr(self._scratch),
                                        H.stream()), "fcmf_bertadam")
                state['step'] += 1
        ops.shadows.mark_all_stale()
        return loss


class FusedAdamW(Optimizer):
    """torch.optim.AdamW semantics, multi-tensor, with the global-norm clip fused in.

    `step(max_grad_norm=1.0)` == `clip_grad_norm_(params, 1.0); AdamW.step()` of the reference
    loop (run_multimodal_fcmf.py:485-487) without materialising the clipped gradients."""
    CHUNK = 65536

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        if lr < 0.0 or eps < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) > 8:
            raise ValueError("FusedAdamW supports at most 8 parameter groups")
        self._tables = None
        self._sumsq = None
        self.last_grad_norm = None

    def _build(self, plist, device):
        sizes = [p.numel() for p, _ in plist]
        ct, co = [], []
        for t, n in enumerate(sizes):
            for off in range(0, n, self.CHUNK):
                ct.append(t)
                co.append(off)
        for p, _ in plist:
            st = self.state[p]
            if len(st) == 0:
                st['step'] = 0
                st['exp_avg'] = torch.zeros_like(p.data)
                st['exp_avg_sq'] = torch.zeros_like(p.data)
        i64 = lambda xs: torch.tensor(xs, dtype=torch.int64, device=device)
        self._tables = dict(
            key=tuple(id(p) for p, _ in plist),
            p=i64([p.data_ptr() for p, _ in plist]),
            m=i64([self.state[p]['exp_avg'].data_ptr() for p, _ in plist]),
            v=i64([self.state[p]['exp_avg_sq'].data_ptr() for p, _ in plist]),
            sizes=i64(sizes),
            group=torch.tensor([g for _, g in plist], dtype=torch.int32, device=device),
            ct=torch.tensor(ct, dtype=torch.int32, device=device), co=i64(co), nchunks=len(ct),
            pptr=[p.data_ptr() for p, _ in plist],
            mvptr=[(self.state[p]['exp_avg'].data_ptr(), self.state[p]['exp_avg_sq'].data_ptr()) for p, _ in plist])

    def load_state_dict(self, state_dict):
        """accepts a FusedAdamW checkpoint or one written by the reference's torch.optim.AdamW
        (run_multimodal_fcmf.py:289,389-391): same per-parameter keys; torch stores `step` as a tensor"""
        super().load_state_dict(state_dict)
        for st in self.state.values():
            if 'step' in st and torch.is_tensor(st['step']):
                st['step'] = int(st['step'].item())
        self._tables = None        # the moment tensors were replaced: the cached device pointer tables are stale

    @torch.no_grad()
    def step(self, closure=None, max_grad_norm=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        plist = [(p, gi) for gi, g in enumerate(self.param_groups) for p in g['params'] if p.grad is not None]
        if not plist:
            return loss
        device = plist[0][0].device
        for p, _ in plist:
            H.require_cuda(p)
            if p.dtype != torch.float32 or p.grad.dtype != torch.float32:
                raise H.HipLibraryError("FusedAdamW expects float32 master parameters and gradients")
            if not p.is_contiguous() or not p.grad.is_contiguous():
                raise H.HipLibraryError("FusedAdamW expects contiguous parameters and gradients")
        T = self._tables
        if (T is None or T['key'] != tuple(id(p) for p, _ in plist) or T['pptr'] != [p.data_ptr() for p, _ in plist]
                or T['mvptr'] != [(self.state[p]['exp_avg'].data_ptr(), self.state[p]['exp_avg_sq'].data_ptr())
                                  if len(self.state[p]) else None for p, _ in plist]):
            self._build(plist, device)
            T = self._tables
        # gradient / shadow addresses: with the gradient arena they are the same every step -- uploaded again only when they change
        sh = [ops.shadows.peek(p) for p, _ in plist]
        gl, sl = [p.grad.data_ptr() for p, _ in plist], [0 if s is None else s.data_ptr() for s in sh]
        if T.get('gl') != gl:
            T['gl'], T['g_ptrs'] = gl, torch.tensor(gl, dtype=torch.int64).to(device, non_blocking=True)
        if T.get('sl') != sl:
            T['sl'], T['sh_ptrs'] = sl, torch.tensor(sl, dtype=torch.int64).to(device, non_blocking=True)
        g_ptrs, sh_ptrs = T['g_ptrs'], T['sh_ptrs']
        L = H.lib()
        st = H.stream()
        if self._sumsq is None or self._sumsq.device != device:
            self._sumsq = torch.zeros(1, dtype=torch.float64, device=device)
        mg = -1.0
        if max_grad_norm is not None and max_grad_norm > 0:
            self._sumsq.zero_()
            H.check(L.fcmf_multi_sumsq(H.ptr(g_ptrs), H.ptr(T['sizes']), H.ptr(T['ct']), H.ptr(T['co']), T['nchunks'],
                                       self.CHUNK, H.ptr(self._sumsq), 0, st), "fcmf_multi_sumsq")
            mg = float(max_grad_norm)
            self.last_grad_norm = self._sumsq  # sqrt taken lazily by grad_norm()
        steps = {int(self.state[p]['step']) for p, _ in plist}
        if len(steps) != 1:
            raise H.HipLibraryError("FusedAdamW: parameters with different step counts are not supported")
        step = steps.pop() + 1
        ng = len(self.param_groups)
        lr = (ctypes.c_float * 8)(*[float(g['lr']) for g in self.param_groups] + [0.0] * (8 - ng))
        wd = (ctypes.c_float * 8)(*[float(g['weight_decay']) for g in self.param_groups] + [0.0] * (8 - ng))
        b1, b2 = self.param_groups[0]['betas']
        eps = self.param_groups[0]['eps']
        H.check(L.fcmf_multi_adamw(H.ptr(T['p']), H.ptr(g_ptrs), H.ptr(T['m']), H.ptr(T['v']), H.ptr(sh_ptrs),
                                   H.ptr(T['sizes']), H.ptr(T['group']), H.ptr(T['ct']), H.ptr(T['co']), T['nchunks'],
                                   self.CHUNK, lr, wd, ng, b1, b2, eps, step, H.ptr(self._sumsq), mg, st),
                "fcmf_multi_adamw")
        ops.shadows.mark_all_stale()          # e.g. fused [3H,H] q|k|v shadows are re-cast lazily
        ops.shadows.refresh_transposed()      # ... the transposed copies (dX GEMM operands) all at once, one launch
        for (p, _), s in zip(plist, sh):
            self.state[p]['step'] = step
            if s is not None:
                ops.shadows.mark_fresh(p)     # refreshed by the kernel itself
        return loss

    def grad_norm(self):
        """global gradient L2 norm measured by the last clipped step (device tensor)"""
        return None if self.last_grad_norm is None else self.last_grad_norm.sqrt().float()

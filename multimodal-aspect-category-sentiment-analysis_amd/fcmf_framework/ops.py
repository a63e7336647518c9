"""Autograd operators of the FCMF hot path, each a thin torch.autograd.Function over the C ABI
of libfcmf_hip.so.  PyTorch supplies device memory, the stream and the autograd tape; every
FLOP and every byte of the step runs in the hand-written gfx950 kernels.

Activations are float32 (parity mode) or bfloat16 (throughput mode); master parameters and
their gradients are always float32.  bf16 copies of the weights ("shadows") are cached and
refreshed when the parameter changes.
"""
import ctypes
import math
import os
import weakref

import torch

from . import _hip as H

# --------------------------------------------------------------------------------------
# global state: dropout seeds and bf16 weight shadows
# --------------------------------------------------------------------------------------
_compute_dtype = torch.float32


def set_compute_dtype(dtype):
    """Activation storage type of the hot path: torch.float32 (parity) or torch.bfloat16 (MFMA)."""
    global _compute_dtype
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("compute dtype must be torch.float32 or torch.bfloat16")
    _compute_dtype = dtype


def compute_dtype():
    return _compute_dtype


_fp8 = False


def set_fp8(on):
    """BASELINE configs[4]: forward and dX GEMMs of the bf16 mode on e4m3 operands (per-row scales, float32 accumulation,
    v_mfma_scale_f32_16x16x128_f8f6f4); weight gradients, attention, LayerNorm and the optimizer are unchanged"""
    global _fp8
    _fp8 = bool(on)


def fp8_enabled():
    return _fp8 and _compute_dtype == torch.bfloat16


_seed_base = 0x5DEECE66D
_seed_ctr = 0


def manual_seed(seed):
    """Seed the counter-based dropout generator (independent of torch's generator)."""
    global _seed_base, _seed_ctr
    _seed_base = (int(seed) * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019) & 0xFFFFFFFFFFFFFFFF
    _seed_ctr = 0


def next_seed():
    global _seed_ctr
    _seed_ctr += 1
    x = (_seed_base + _seed_ctr * 0xD1342543DE82EF95) & 0xFFFFFFFFFFFFFFFF
    x ^= x >> 29
    return (x * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF


VOCAB_PAD = 32   # row padding of ragged 2-D weights (one MFMA k-tile)


class _Shadows:
    """bf16 copies of float32 master parameters, keyed by storage address."""

    def __init__(self):
        self.map = {}
        self.pad = {}
        self.mapT = {}
        self.mapD = {}
        self.mapQ = {}
        self.mapQT = {}
        self.mapH = {}
        self._mt_tables = None

    def get_t(self, w, owner=None):
        """bf16 TRANSPOSE [K, N] of a float32 [N, K] weight: dX = dY W then reads W^T as a K-contiguous operand
        (ds_read_b128 instead of transposed LDS reads: the NT kernels run 10-25 % faster than the NN ones).
        Rebuilt lazily after every optimizer step (one small kernel per weight).
        owner: the Parameter whose storage `w` is (a view of); defaults to `w`.  The entry only holds a WEAK reference to
        it: `refresh_transposed` re-reads the source by raw address, which is only legal while that tensor is alive."""
        key = (w.data_ptr(), tuple(w.shape))
        ent = self.mapT.get(key)
        owner = w if owner is None else owner
        if ent is not None and ent[1] == w._version and not ent[2] and ent[4]() is owner:
            return ent[0]
        sh = ent[0] if ent is not None else torch.empty((w.shape[1], w.shape[0]), dtype=torch.bfloat16, device=w.device)
        src = w.detach()
        if not src.is_contiguous():
            src = src.contiguous()
        H.check(H.lib().fcmf_cast_transpose(H.ptr(src), H.ptr(sh), w.shape[0], w.shape[1], H.stream()), "fcmf_cast_transpose")
        self.mapT[key] = [sh, w._version, False, w.is_contiguous(), weakref.ref(owner), w.data_ptr() - owner.data_ptr()]
        return sh

    def _prune_transposed(self):
        """drop the transposed copies whose source tensor is gone (a freed model: its storage may have been returned to the
        driver, or recycled for something else) or has moved (`.to()`, re-fused q|k|v storage)"""
        dead = []
        for k, ent in self.mapT.items():
            o = ent[4]()
            if o is None or o.data_ptr() + ent[5] != k[0]:
                dead.append(k)
        for k in dead:
            del self.mapT[k]
        if dead:
            self._mt_tables = None

    def refresh_transposed(self):
        """rebuild EVERY cached transposed copy of a LIVE weight in one launch (called by the fused optimizers right after
        their update: all of them are stale at that point, and rebuilding them lazily costs one small launch per weight)"""
        if not self.mapT:
            return
        self._prune_transposed()
        items = [(k, ent) for k, ent in self.mapT.items() if ent[3]]      # (sources read in place must be dense [R, C])
        if not items:
            return
        sig = tuple((k[0], ent[0].data_ptr()) for k, ent in items)
        if self._mt_tables is None or self._mt_tables[0] != sig:
            dev = items[0][1][0].device
            desc = []
            for t, (k, ent) in enumerate(items):
                R, C = k[1]
                desc += [(t, r, c) for r in range((R + 63) // 64) for c in range((C + 63) // 64)]
            self._mt_tables = (sig,
                               torch.tensor([k[0] for k, _ in items], dtype=torch.int64, device=dev),
                               torch.tensor([ent[0].data_ptr() for _, ent in items], dtype=torch.int64, device=dev),
                               torch.tensor([list(k[1]) for k, _ in items], dtype=torch.int32, device=dev),
                               torch.tensor(desc, dtype=torch.int32, device=dev), len(desc))
        _, src, dst, dims, desc, n = self._mt_tables
        H.check(H.lib().fcmf_multi_cast_transpose(H.ptr(src), H.ptr(dst), H.ptr(dims), H.ptr(desc), n, H.stream()),
                "fcmf_multi_cast_transpose")
        for _, ent in items:
            ent[2] = False
            ent[1] = ent[4]()._version       # fresh as of the owner's current version

    def head_nk(self, ws):
        """bf16 [len(ws) * n_head * d, E] re-layout ("nn.Linear layout", natural head order) of the per-head weights `ws` (float32
        Parameters [n_head, E, d] of the IAOG decoder's Attention): row (i * n_head + h) * d + j = ws[i][h, :, j].  Every [d, E]
        row block is the transpose of the dense [E, d] slice ws[i][h] and is REGISTERED as one of the transposed copies, so
        `refresh_transposed` rebuilds all of them -- every block of the decoder -- in the optimizer's one launch (the round-2 form
        rebuilt each layout with permute + reshape copies, a cat and a cast per step and block: 130 launches).  With another
        optimizer the first use after an update refreshes everything, also in one launch."""
        key = tuple(w.data_ptr() for w in ws) + (tuple(ws[0].shape),)
        ent = self.mapH.get(key)
        if ent is None or any(r() is None for r in ent[2]):
            nh, E, d = ws[0].shape
            buf = torch.empty((len(ws) * nh * d, E), dtype=torch.bfloat16, device=ws[0].device)
            pieces = []
            for i, w in enumerate(ws):
                if w.dtype != torch.float32 or not w.is_contiguous() or tuple(w.shape) != (nh, E, d):
                    raise H.HipLibraryError("head_nk: dense float32 [n_head, E, d] parameters expected")
                for h in range(nh):
                    src, dst = w.detach()[h], buf[(i * nh + h) * d:(i * nh + h + 1) * d]
                    # (the GROUP is part of the key: the same parameter may sit in two groupings -- [w_kx] alone and [w_kx, w_qx] --
                    #  and each grouping's buffer must keep its own registered pieces, or the loser would serve stale weights)
                    k2 = (src.data_ptr(), (E, d), key)
                    self.mapT[k2] = [dst, -1, True, True, weakref.ref(w), src.data_ptr() - w.data_ptr()]
                    pieces.append(k2)
            self._mt_tables = None
            ent = self.mapH[key] = (buf, pieces, [weakref.ref(w) for w in ws])
        buf, pieces, owners = ent
        for k2 in pieces:
            e = self.mapT.get(k2)
            if e is None or e[2] or e[1] != e[4]()._version:
                if e is None:              # (pruned: the parameter moved) -- start over
                    del self.mapH[key]
                    return self.head_nk(ws)
                self.refresh_transposed()
                break
        return buf

    def get_fp8(self, w, owner=None):
        """(q [N, K] e4m3 bytes, scale [N] float32) of a float32 [N, K] weight, quantised per output row; rebuilt lazily
        after the parameter changed (FusedAdamW marks every shadow stale)"""
        key = (w.data_ptr(), tuple(w.shape))
        ent = self.mapQ.get(key)
        owner = w if owner is None else owner
        if ent is not None and ent[1] == w._version and not ent[2] and ent[3]() is owner:
            return ent[0]
        src = self.get(w) if w.is_contiguous() else w.detach().contiguous()       # the bf16 shadow: half the bytes to read
        q, sc = quant_fp8_rows(src, w.shape[0], w.shape[1], w.shape[1], out=None if ent is None else ent[0])
        self.mapQ[key] = [(q, sc), w._version, False, weakref.ref(owner)]
        return q, sc

    def get_fp8_t(self, w, owner=None):
        """(q [K, N] e4m3, scale [K]) of the TRANSPOSE of a float32 [N, K] weight, quantised per input row: the B operand
        of dX = dY W on the fp8 kernel (contraction over N)"""
        key = (w.data_ptr(), tuple(w.shape))
        ent = self.mapQT.get(key)
        owner = w if owner is None else owner
        if ent is not None and ent[1] == w._version and not ent[2] and ent[3]() is owner:
            return ent[0]
        wt = self.get_t(w, owner)                                                   # [K, N] bf16, fresh
        q, sc = quant_fp8_rows(wt, wt.shape[0], wt.shape[1], wt.shape[1], out=None if ent is None else ent[0])
        self.mapQT[key] = [(q, sc), w._version, False, weakref.ref(owner)]
        return q, sc

    def padded(self, w):
        """the fresh bf16 copy of a 2-D weight including its zero rows up to a multiple of 32"""
        sh = self.get(w)
        return self.pad.get((w.data_ptr(), tuple(w.shape)), sh)

    def get(self, w):
        key = (w.data_ptr(), tuple(w.shape))
        ent = self.map.get(key)
        if ent is not None and ent[1] == w._version and not ent[2]:
            return ent[0]
        if ent is not None:
            sh = ent[0]
        elif w.dim() == 2 and w.shape[0] % VOCAB_PAD != 0:
            # 2-D weights with a ragged row count (the 64001-row tied vocabulary matrix) get ZERO rows up to the next
            # multiple of 32 behind the copy: `padded(w)` hands the MFMA kernels a regular [rows32, K] operand
            rows = (w.shape[0] + VOCAB_PAD - 1) // VOCAB_PAD * VOCAB_PAD
            full = torch.zeros((rows, w.shape[1]), dtype=torch.bfloat16, device=w.device)
            sh = full[:w.shape[0]]
            self.pad[key] = full
        else:
            sh = torch.empty(w.shape, dtype=torch.bfloat16, device=w.device)
        src = w.detach()
        if not src.is_contiguous():
            src = src.contiguous()
        H.check(H.lib().fcmf_cast(H.ptr(src), H.ptr(sh), src.numel(), H.F32, H.BF16, H.stream()), "fcmf_cast")
        self.map[key] = [sh, w._version, False]
        return sh

    def derived(self, w, tag, build):
        """a tensor derived from the PARAMETER `w` (a re-layout and / or cast), cached per (storage, shape, tag) and
        rebuilt by `build(w.detach())` when the parameter has changed.  Only ever key this on persistent leaf
        parameters: a temporary's address is recycled by the caching allocator (and its version is always 0)."""
        key = (w.data_ptr(), tuple(w.shape), tag)
        ent = self.mapD.get(key)
        if ent is not None and ent[1] == w._version and not ent[2]:
            return ent[0]
        t = build(w.detach())
        self.mapD[key] = [t, w._version, False]
        return t

    def peek(self, w):
        ent = self.map.get((w.data_ptr(), tuple(w.shape)))
        return None if ent is None else ent[0]

    def mark_fresh(self, w):
        ent = self.map.get((w.data_ptr(), tuple(w.shape)))
        if ent is not None:
            ent[1] = w._version
            ent[2] = False

    def mark_all_stale(self):
        for ent in self.map.values():
            ent[2] = True
        for ent in self.mapT.values():
            ent[2] = True
        for ent in self.mapD.values():
            ent[2] = True
        for m in (self.mapQ, self.mapQT):
            for k in [k for k, ent in m.items() if ent[3]() is None]:      # quantised copies of freed weights
                del m[k]
            for ent in m.values():
                ent[2] = True

    def clear(self):
        self.map.clear()
        self.pad.clear()
        self.mapT.clear()
        self.mapD.clear()
        self.mapQ.clear()
        self.mapQT.clear()
        self.mapH.clear()
        self._mt_tables = None


shadows = _Shadows()

# --------------------------------------------------------------------------------------
# flat gradient arena (dp.GradArena): weight-gradient buffers come out of the parameter's slice of ONE zeroed
# buffer when an arena is active, so that autograd adopts the slice as p.grad (no copy) and a step needs one memset
# --------------------------------------------------------------------------------------
_grad_arena = None


def set_grad_arena(arena):
    global _grad_arena
    _grad_arena = arena
    if arena is not None and deferred_dw.reset not in arena.on_zero:
        arena.on_zero.append(deferred_dw.reset)      # a step that died inside backward() must not leave its queue to the next one


def grad_arena():
    return _grad_arena


def alloc_grad(param, shape=None):
    """zero-initialised float32 buffer for the gradient of `param` (viewed as `shape`): the parameter's arena
    slice if an arena is active and the slice is still unclaimed in this step, else a fresh zero tensor"""
    a = _grad_arena
    if a is not None and param is not None:
        v = a.take(param)
        if v is not None:
            return v if shape is None else v.view(shape)
    return torch.zeros(tuple(param.shape) if shape is None else shape, dtype=torch.float32, device=param.device)


def alloc_grad_ex(param, shape=None):
    """-> (buffer to ACCUMULATE the gradient of `param` into, what to hand autograd for it).  First producer of the pass: the
    parameter's zeroed arena slice, returned to autograd; a later producer of the same pass (shared weights): the same slice,
    accumulated in place, and `None` for autograd (dp.GradArena.retake); no arena: a fresh zero tensor."""
    a = _grad_arena
    if a is not None and param is not None:
        v = a.take(param)
        if v is not None:
            v = v if shape is None else v.view(shape)
            return v, v
        v = a.retake(param)
        if v is not None:
            return (v if shape is None else v.view(shape)), None
    z = torch.zeros(tuple(param.shape) if shape is None else shape, dtype=torch.float32, device=param.device)
    return z, z


def alloc_grad_block(params, shape):
    """one zero-initialised buffer covering the gradients of `params` back to back (fused q|k|v)"""
    a = _grad_arena
    if a is not None:
        v = a.take_block(params)
        if v is not None:
            return v.view(shape)
    return torch.zeros(shape, dtype=torch.float32, device=params[0].device)


def as_compute(w, dtype):
    """parameter as seen by a kernel computing in `dtype`"""
    if dtype == torch.float32:
        return w.detach()
    return shadows.get(w)


def cast(x, dtype):
    """dtype conversion through fcmf_cast (float32 <-> bfloat16)"""
    if x.dtype == dtype:
        return x
    H.require_cuda(x)
    x = x.contiguous()
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    H.check(H.lib().fcmf_cast(H.ptr(x), H.ptr(y), x.numel(), H.dt(x), H.dt(y), H.stream()), "fcmf_cast")
    return y


class _CastFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        return cast(x, dtype)

    @staticmethod
    def backward(ctx, dy):
        return cast(dy.contiguous(), ctx.src), None


def cast_ad(x, dtype):
    return x if x.dtype == dtype else _CastFn.apply(x, dtype)


# --------------------------------------------------------------------------------------
# raw kernel wrappers
# --------------------------------------------------------------------------------------
def _rows(x):
    """2-D view [rows, features] with unit inner stride (row stride may be arbitrary)"""
    if x.dim() != 2:
        x = x.reshape(-1, x.shape[-1])
    if x.stride(1) != 1 or (x.shape[0] > 1 and x.stride(0) < x.shape[1]):
        x = x.contiguous()
    return x


_gemm_trace = None


def gemm_trace_begin():
    """start recording a HIP event pair (on the launch stream) around every GEMM launch"""
    global _gemm_trace
    _gemm_trace = []


def gemm_trace_end():
    """-> [(kernel name, flops, milliseconds)] for the launches since gemm_trace_begin()"""
    global _gemm_trace
    tr, _gemm_trace = _gemm_trace, None
    if not tr:
        return []
    torch.cuda.synchronize()
    return [(name, fl, e0.elapsed_time(e1)) for name, fl, e0, e1 in tr]


class trace_launch:
    """`with trace_launch(flops):` around a direct library call that multiplies on the GEMM context of the current stream (the trunk's
    implicit-GEMM convolutions): recorded like `gemm` when a trace is open, free otherwise"""

    def __init__(self, flops):
        self.flops = flops

    def __enter__(self):
        if _gemm_trace is not None:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if _gemm_trace is not None and exc[0] is None:
            self.e1.record()
            _gemm_trace.append((H.lib().fcmf_gemm_ctx_last_kernel(H.gemm_ctx()).decode(), float(self.flops), self.e0, self.e1))
        return False


# ---- deferred weight gradients -------------------------------------------------------------------------------------------
# dW = dY^T X of an nn.Linear is needed only once the backward pass is over (grad clipping, the optimizer, the gradient
# exchange of its bucket), and one layer's dW is a handful of 256 x 256 tiles: alone it fills the chip only through a deep
# split of K plus a reduce pass per matrix.  During loss.backward() the weight-gradient GEMMs whose destination is a slice of the
# step's gradient arena are therefore QUEUED (operands kept alive) and multiplied together, all same-shape matrices of a group in
# one fcmf_gemm_dw_batched launch: at the end of the backward pass (an autograd-engine callback), before a data-parallel bucket that
# contains queued gradients is sent (dp.GradReducer), before anything else reads the arena (flush_deferred_dw), or when the queue
# holds more than DEFER_DW_MAX_BYTES of operands.  Only parameters with exactly ONE forward use in the step qualify (the arena
# counts them in gemm_nt): a weight applied twice has two gradient producers whose results autograd adds the moment each
# Function returns.  The destination is remembered by address, never by tensor: autograd must stay the only holder of the
# returned slice, or AccumulateGrad clones it -- unwritten -- instead of adopting it as p.grad.
DEFER_DW = os.environ.get("FCMF_DEFER_DW", "1") == "1"
DEFER_DW_MAX_BYTES = 48 << 30


class _DeferredDW:
    def __init__(self):
        self.q, self.bytes, self.armed = [], 0, False
        self.batched_launches = self.batched_matrices = 0

    def reset(self):
        self.q, self.bytes, self.armed = [], 0, False

    def wanted(self, A, B, C, ta, tb, bias, aux, epi, colsum):
        if not (DEFER_DW and ta and tb and bias is None and aux is None and colsum is None and epi == H.EPI_NONE
                and C.dtype == torch.float32 and A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16):
            return False
        arena = _grad_arena
        if not (arena is not None and C.device == arena.flat.device
                and C.untyped_storage().data_ptr() == arena.flat.untyped_storage().data_ptr()):
            return False
        # A weight applied SEVERAL times (the mm layer's key / value weights: text keys, ROI keys, the fusion layer) has several
        # producers, but since round 4 every one of them accumulates in place into the slice the first one claimed and hands
        # autograd None (alloc_grad_ex / GradArena.retake): nothing is added by the engine, so they may all wait for the flush and
        # batch with the same-shape gradients of the text encoder.  (A destination outside the arena -- the temporaries of a
        # gradient-accumulation micro-step -- never gets here.)
        n = arena.uses(C.data_ptr())
        return n == 1 or (n > 1 and DEFER_SHARED_DW)

    def push(self, A, B, C, M, N, K, lda, ldb, ldc, acc):
        if not self.armed:
            try:      # (only legal while the autograd engine runs: outside of backward() the GEMM runs right away)
                torch.autograd.Variable._execution_engine.queue_callback(self.flush)
            except RuntimeError:
                return False
            self.armed = True
        # (the destination is remembered by ADDRESS: the arena owns that memory, and a second reference to the tensor object would
        #  make autograd's AccumulateGrad clone the -- not yet written -- slice instead of adopting it as p.grad)
        self.q.append((A, B, C.data_ptr(), M, N, K, lda, ldb, ldc, bool(acc)))
        self.bytes += A.numel() * A.element_size() + B.numel() * B.element_size()
        if self.bytes > DEFER_DW_MAX_BYTES:
            self.flush(final=False)
        return True

    def flush(self, final=True, lo=None, hi=None):
        """multiply the queued gradients; lo / hi: only those whose destination address lies in [lo, hi) (the arena range a
        data-parallel bucket group is about to send) -- the rest stays queued and keeps batching"""
        if lo is not None:
            q = [e for e in self.q if lo <= e[2] < hi]
            if not q:
                return
            self.q = [e for e in self.q if not (lo <= e[2] < hi)]
            self.bytes = sum(e[0].numel() * e[0].element_size() + e[1].numel() * e[1].element_size() for e in self.q)
        else:
            q, self.q, self.bytes = self.q, [], 0
        if final:
            self.armed = False
        groups = {}
        for e in q:
            groups.setdefault((e[0].device,) + e[3:], []).append(e)
        for key, es in groups.items():
            _, M, N, K, lda, ldb, ldc, acc = key
            H.require_cuda(es[0][0])
            with torch.cuda.device(key[0]):
                ctx = H.gemm_ctx(workspace=True)
                arr = lambda i: (ctypes.c_void_p * len(es))(*[e[i] if i == 2 else e[i].data_ptr() for e in es])
                if _gemm_trace is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                H.check(H.lib().fcmf_gemm_dw_batched(ctx, len(es), arr(0), arr(1), arr(2), M, N, K, lda, ldb, ldc, int(acc), H.stream()),
                        "fcmf_gemm_dw_batched")
                if _gemm_trace is not None:
                    e1.record()
                    _gemm_trace.append((H.lib().fcmf_gemm_ctx_last_kernel(ctx).decode(), 2.0 * M * N * K * len(es), e0, e1))
            self.batched_launches += 1
            self.batched_matrices += len(es)


deferred_dw = _DeferredDW()


def flush_deferred_dw(lo=None, hi=None):
    """multiply the queued weight gradients now (anything that reads the gradient arena before backward() has returned);
    lo / hi (device addresses): only the ones whose destination lies in that range of the arena"""
    if deferred_dw.q:
        deferred_dw.flush(final=False, lo=lo, hi=hi)


def gemm(A, B, C, M, N, K, lda, ldb, ldc, ta, tb, bias=None, aux=None, epi=H.EPI_NONE, acc=False, colsum=None):
    H.require_cuda(A, B, C)
    if deferred_dw.wanted(A, B, C, ta, tb, bias, aux, epi, colsum) and deferred_dw.push(A, B, C, M, N, K, lda, ldb, ldc, acc):
        return
    ctx = H.gemm_ctx(workspace=acc or C.dtype == torch.float32)   # (weight-gradient GEMMs: the context owns the split-K scratch of this stream)
    if _gemm_trace is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    H.check(H.lib().fcmf_gemm(ctx, H.ptr(A), H.ptr(B), H.ptr(C), H.ptr(bias), H.ptr(aux), H.ptr(colsum), M, N, K, lda, ldb, ldc,
                              int(ta), int(tb), H.dt(A), H.dt(C), epi, int(acc), H.stream()), "fcmf_gemm")
    if _gemm_trace is not None:
        e1.record()
        _gemm_trace.append((H.lib().fcmf_gemm_ctx_last_kernel(ctx).decode(), 2.0 * M * N * K, e0, e1))


# (see _DeferredDW.wanted; measured on one box, 20 steps, twice each: 36.76 / 36.85 ms off, 36.70 / 36.67 ms on -- 168 instead of 176 GEMM
#  launches, 26.43 instead of 26.70 ms of GEMM time -- but the batched kernel then also runs the small shared-weight batches and its
#  per-launch average drops from 1223 to 1194 TFLOP/s.  Off by default: the two configurations are within the box-to-box spread)
DEFER_SHARED_DW = os.environ.get("FCMF_DEFER_SHARED_DW", "0") == "1"
HEAD_WGRAD_DIRECT = os.environ.get("FCMF_HEAD_WGRAD_DIRECT", "1") == "1"      # (A/B switch)


def head_weight_grad(x2, dy2, params):
    """gradient of the per-head projection weights `params` (float32 Parameters [n_head, E, d], side by side in dy2's columns) of the
    IAOG decoder's Attention, written DIRECTLY in the parameters' layout into their adjacent arena slices: dW^T [E, len * n_head * d]
    = x2^T dy2 through fcmf_gemm_colblocks (column block = d, block stride = E * d, row stride = d).  -> the tensors to hand autograd
    (the arena views), or None when it does not apply (no arena, slices not adjacent / already claimed, shape the blocked path
    refuses): the caller then multiplies into a plain [n * d, E] buffer and lets autograd permute-copy it (round 3: 96 launches
    per IAOG step)."""
    a = _grad_arena
    if a is None or dy2.dtype != torch.bfloat16 or x2.dtype != torch.bfloat16 or not HEAD_WGRAD_DIRECT:
        return None
    nh, E, d = params[0].shape
    if any(p.dtype != torch.float32 or tuple(p.shape) != (nh, E, d) for p in params) or d % 4:
        return None
    M, N = x2.shape[0], len(params) * nh * d
    if E < 256 or N < 256 or dy2.shape[1] != N or not dy2.is_contiguous():
        return None
    flat = a.take_block(list(params)) if len(params) > 1 else a.take(params[0])
    if flat is None:
        return None
    ctx = H.gemm_ctx(workspace=True)
    if _gemm_trace is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = H.lib().fcmf_gemm_colblocks(ctx, H.ptr(x2), H.ptr(dy2), H.ptr(flat), E, N, M, _ld(x2), N, d, 1, 1, d, E * d, 0, H.stream())
    if rc == H.ERR_UNSUPPORTED:
        a.untake(params)
        return None
    H.check(rc, "fcmf_gemm_colblocks")
    if _gemm_trace is not None:
        e1.record()
        _gemm_trace.append((H.lib().fcmf_gemm_ctx_last_kernel(ctx).decode(), 2.0 * E * N * M, e0, e1))
    # (FRESH aliases: autograd adopts a gradient as p.grad without a copy only when it is the sole holder of the tensor object --
    #  the arena's own view objects would be cloned)
    return [a.view[id(a._by_ptr[p.data_ptr()])].view(nh, E, d) for p in params]


def quant_fp8_rows(x, rows, K, ldx, out=None):
    """x [rows, K] (bf16 / f32 rows at stride ldx) -> (q [rows, K] uint8 e4m3, scale [rows] float32) through fcmf_quant_fp8_rows"""
    H.require_cuda(x)
    if out is None:
        q = torch.empty((rows, K), dtype=torch.uint8, device=x.device)
        sc = torch.empty(rows, dtype=torch.float32, device=x.device)
    else:
        q, sc = out
    H.check(H.lib().fcmf_quant_fp8_rows(H.ptr(x), ldx, H.ptr(q), K, H.ptr(sc), rows, K, H.dt(x), H.stream()), "fcmf_quant_fp8_rows")
    return q, sc


def _fp8_ok(M, N, K, ldc, *tensors):
    return (fp8_enabled() and K % 128 == 0 and N % 8 == 0 and ldc % 8 == 0 and M >= 256 and N >= 256
            and all(t is None or t.dtype == torch.bfloat16 for t in tensors))


def gemm_fp8(xq, sx, wq, sw, C, M, N, K, bias=None, aux=None, epi=H.EPI_NONE, colsum=None):
    ctx = H.gemm_ctx()
    if _gemm_trace is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    H.check(H.lib().fcmf_gemm_fp8(ctx, H.ptr(xq), H.ptr(sx), H.ptr(wq), H.ptr(sw), H.ptr(C), H.ptr(bias), H.ptr(aux), H.ptr(colsum),
                                  M, N, K, K, K, N, epi, H.stream()), "fcmf_gemm_fp8")
    if _gemm_trace is not None:
        e1.record()
        _gemm_trace.append((H.lib().fcmf_gemm_ctx_last_kernel(ctx).decode(), 2.0 * M * N * K, e0, e1))


def gemm_nt(x, weight, w_compute, y, M, N, K, ldx, bias=None, aux=None, epi=H.EPI_NONE, colsum=None, owner=None, xq=None):
    """y [M, N] = epilogue(x [M, K] W^T + bias), W [N, K]: the forward GEMM of nn.Linear.  `weight` = the float32 master (or a
    view of it; `owner` = its Parameter) when there is one: the fp8 mode then multiplies e4m3 copies (x quantised per row here,
    W per output row, cached); otherwise the bf16 / f32 kernel on `w_compute`."""
    if _grad_arena is not None and (owner is not None or weight is not None):
        _grad_arena.note_forward((owner if owner is not None else weight).data_ptr())     # (see deferred_dw.wanted)
    if weight is not None and weight.dtype == torch.float32 and weight.dim() == 2 and _fp8_ok(M, N, K, N, x, y, aux):
        xq, sx = quant_fp8_rows(x, M, K, ldx) if xq is None else xq       # (xq: already quantised by the producing LayerNorm)
        wq, sw = shadows.get_fp8(weight, owner)
        gemm_fp8(xq, sx, wq, sw, y, M, N, K, bias=bias, aux=aux, epi=epi, colsum=colsum)
    else:
        gemm(x, w_compute, y, M, N, K, ldx, K, N, 0, 0, bias=bias, aux=aux, epi=epi, colsum=colsum)


def colsum(X, M, N, ldx):
    out = torch.empty(N, dtype=torch.float32, device=X.device)
    H.check(H.lib().fcmf_colsum(H.ptr(X), H.ptr(out), M, N, ldx, H.dt(X), 0, H.stream()), "fcmf_colsum")
    return out


def _ld(x):
    return x.stride(0) if x.shape[0] > 1 else x.shape[1]


def gemm_dx(dy, weight, w_compute, dx, M, K_in, N_out, aux=None, epi=H.EPI_NONE, colsum=None, owner=None, dyq=None):
    """dx [M,K_in] = dy [M,N_out] @ W [N_out,K_in] (+ epilogue).  bf16 mode multiplies by the transposed bf16 copy of the
    float32 master `weight` (an NT GEMM); f32 mode, or a weight without a master, uses `w_compute` as it lies (NN)."""
    if (weight is not None and weight.dtype == torch.float32 and weight.dim() == 2 and dy.is_contiguous()
            and _fp8_ok(M, K_in, N_out, K_in, dy, dx, aux)):
        dyq, sdy = quant_fp8_rows(dy, M, N_out, N_out) if dyq is None else dyq
        wq, sw = shadows.get_fp8_t(weight, owner)                    # [K_in, N_out] e4m3, scales per input row
        gemm_fp8(dyq, sdy, wq, sw, dx, M, K_in, N_out, aux=aux, epi=epi, colsum=colsum)
    elif dy.dtype == torch.bfloat16 and weight is not None and weight.dtype == torch.float32 and weight.dim() == 2:
        wt = shadows.get_t(weight, owner)                            # [K_in, N_out]; owner: the Parameter behind a temporary view
        gemm(dy, wt, dx, M, K_in, N_out, N_out, N_out, K_in, 0, 0, aux=aux, epi=epi, colsum=colsum)
    else:
        gemm(dy, w_compute, dx, M, K_in, N_out, N_out, K_in, K_in, 0, 1, aux=aux, epi=epi, colsum=colsum)


def gemm_dx_long_k(dy, w, M, K_in, N_out, ldw):
    """dx [M,K_in] = dy [M,N_out] @ w [N_out,K_in] for a LONG contraction and a small output (the vocabulary projection:
    N_out = 64032, M x K_in = 768 x 768): a bf16 output has 9 tiles of 256x256 = 9 busy CUs; accumulating into float32 lets
    the kernel split the contraction 28 ways over the chip (workspace + reduce pass), then one cast"""
    if dy.dtype != torch.bfloat16 or N_out < 8192:
        dx = torch.empty((M, K_in), dtype=dy.dtype, device=dy.device)
        gemm(dy, w, dx, M, K_in, N_out, N_out, ldw, K_in, 0, 1)
        return dx
    dx32 = torch.zeros((M, K_in), dtype=torch.float32, device=dy.device)
    gemm(dy, w, dx32, M, K_in, N_out, N_out, ldw, K_in, 0, 1, acc=True)
    return cast(dx32, torch.bfloat16)


def _linear_fwd(x, w, bias, epi=H.EPI_NONE, aux=None, master=None):
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty((M, N), dtype=x.dtype, device=x.device)
    if epi == H.EPI_TANH or master is None:
        gemm(x, w, y, M, N, K, _ld(x), K, N, 0, 0, bias=bias, aux=aux, epi=epi)
    else:
        gemm_nt(x, master, w, y, M, N, K, _ld(x), bias=bias, aux=aux, epi=epi)
    return y


def _linear_bwd(x, w, dy, need_dx=True, need_dw=True, need_db=True, dx_epi=H.EPI_NONE, dx_aux=None, master=None, bias_param=None):
    """x [M,K], w [N,K] (compute dtype), dy [M,N] -> dx [M,K], dW [N,K] f32, db [N] f32
    (master = the float32 parameter behind w, if any: bf16 mode then uses its transposed copy for dx)"""
    M, K = x.shape
    N = w.shape[0]
    dx = dw = db = None
    if need_dx:
        dx = torch.empty((M, K), dtype=dy.dtype, device=dy.device)
        gemm_dx(dy, master, w, dx, M, K, N, aux=dx_aux, epi=dx_epi)
    if need_dw:
        if master is not None and master.dtype == torch.float32 and master.is_contiguous() and tuple(master.shape) == (N, K):
            dwbuf, dw = alloc_grad_ex(master, (N, K))         # (dw = None: accumulated in place into the slice an earlier use returned)
        else:
            dwbuf = dw = torch.zeros((N, K), dtype=torch.float32, device=dy.device)
        gemm(dy, x, dwbuf, N, K, M, N, _ld(x), K, 1, 1, acc=True)
    if need_db:
        if bias_param is not None and bias_param.dtype == torch.float32 and tuple(bias_param.shape) == (N,):
            # straight into the parameter's (already zero) arena slice: no output tensor, no memset inside fcmf_colsum
            dbbuf, db = alloc_grad_ex(bias_param, (N,))
            H.check(H.lib().fcmf_colsum(H.ptr(dy), H.ptr(dbbuf), M, N, N, H.dt(dy), 1, H.stream()), "fcmf_colsum")
        else:
            db = colsum(dy, M, N, N)
    return dx, dw, db


class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b), act in {none, tanh}.  nn.Linear (+ BertPooler's tanh, mm_modeling.py:425-431)"""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        x2 = _rows(x)
        w = as_compute(weight, x2.dtype)
        epi = H.EPI_TANH if act == "tanh" else H.EPI_NONE
        y = _linear_fwd(x2, w, None if bias is None else bias.detach(), epi, master=weight)
        ctx.save_for_backward(x2, weight, y if act == "tanh" else None)
        ctx.act = act
        ctx.has_bias = bias is not None
        ctx.bias_param = bias
        ctx.xshape = x.shape
        return y.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, weight, y = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1]).contiguous()
        if ctx.act == "tanh":
            d = torch.empty_like(dy2)
            H.check(H.lib().fcmf_act_bwd(H.ptr(dy2), H.ptr(y), H.ptr(d), dy2.numel(), 0, H.dt(dy2), H.stream()), "act_bwd")
            dy2 = d
        w = as_compute(weight, x2.dtype)
        dx, dw, db = _linear_bwd(x2, w, dy2, ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                 ctx.has_bias and ctx.needs_input_grad[2], master=weight, bias_param=ctx.bias_param)
        return (None if dx is None else dx.view(ctx.xshape)), dw, db, None


def linear(x, weight, bias=None, act=None):
    return LinearFn.apply(x, weight, bias, act)


class SeqFanFn(torch.autograd.Function):
    """The three consumers of the text encoder's output in FCMFEncoder (fcmf_pretraining.py:97-124 after pruning): the key and
    the value projection of the text+ROI layer (two nn.Linear on the SAME [G, S, H] tensor) and the [CLS] row.  As separate
    autograd nodes their three gradient contributions to the sequence are materialised and added by the engine: a zero-filled
    [G, S, H] tensor for the row (select_backward) and two full-size adds -- 0.5 GB of traffic per step for nothing.  Here:
    dx = dy1 W1, then dx = dx + dy2 W2 in the second GEMM's epilogue (FCMF_EPI_ADD), then the [CLS] gradient added to row 0."""

    @staticmethod
    def forward(ctx, seq, w1, b1, w2, b2):
        G, S, Hd = seq.shape
        x2 = _rows(seq)
        c1, c2 = as_compute(w1, x2.dtype), as_compute(w2, x2.dtype)
        y1 = _linear_fwd(x2, c1, None if b1 is None else b1.detach(), master=w1)
        y2 = _linear_fwd(x2, c2, None if b2 is None else b2.detach(), master=w2)
        cls = seq[:, 0].contiguous()
        ctx.save_for_backward(x2, w1, w2)
        ctx.biases = (b1, b2)
        ctx.shape = (G, S, Hd)
        return y1.view(G, S, -1), y2.view(G, S, -1), cls

    @staticmethod
    def backward(ctx, dy1, dy2, dcls):
        x2, w1, w2 = ctx.saved_tensors
        b1, b2 = ctx.biases
        G, S, Hd = ctx.shape
        need_dx = ctx.needs_input_grad[0]
        dx = dw1 = db1 = dw2 = db2 = None
        if dy1 is not None:
            d1 = dy1.reshape(G * S, -1).contiguous()
            dx, dw1, db1 = _linear_bwd(x2, as_compute(w1, x2.dtype), d1, need_dx, ctx.needs_input_grad[1],
                                       b1 is not None and ctx.needs_input_grad[2], master=w1, bias_param=b1)
        if dy2 is not None:
            d2 = dy2.reshape(G * S, -1).contiguous()
            dx, dw2, db2 = _linear_bwd(x2, as_compute(w2, x2.dtype), d2, need_dx, ctx.needs_input_grad[3],
                                       b2 is not None and ctx.needs_input_grad[4], dx_epi=H.EPI_NONE if dx is None else H.EPI_ADD,
                                       dx_aux=dx, master=w2, bias_param=b2)
        if need_dx:
            if dx is None:
                dx = torch.zeros((G * S, Hd), dtype=x2.dtype, device=x2.device)
            dx = dx.view(G, S, Hd)
            if dcls is not None:
                dx[:, 0] += dcls
        return dx, dw1, db1, dw2, db2


class MultiLinearFn(torch.autograd.Function):
    """several nn.Linear on the SAME input (q / k / v of the box attention on the ROI features, roi_modeling.py:170-173; key / value
    of the cross attention on the patch features) as one autograd node: the input gradient is accumulated by the dX GEMMs' add
    epilogue instead of n - 1 full-size adds by the engine.  Arguments: x, w1, b1, w2, b2, ...; returns one output per pair."""

    @staticmethod
    def forward(ctx, x, *wb):
        x2 = _rows(x)
        ws, bs = wb[0::2], wb[1::2]
        ys = tuple(_linear_fwd(x2, as_compute(w, x2.dtype), None if b is None else b.detach(), master=w) for w, b in zip(ws, bs))
        ctx.save_for_backward(x2, *ws)
        ctx.biases = bs
        ctx.xshape = x.shape
        return tuple(y.view(*x.shape[:-1], w.shape[0]) for y, w in zip(ys, ws))

    @staticmethod
    def backward(ctx, *dys):
        x2, *ws = ctx.saved_tensors
        need_dx = ctx.needs_input_grad[0]
        dx = None
        grads = []
        for i, (dy, w, b) in enumerate(zip(dys, ws, ctx.biases)):
            if dy is None:
                grads += [None, None]
                continue
            d = dy.reshape(-1, dy.shape[-1]).contiguous()
            dx, dw, db = _linear_bwd(x2, as_compute(w, x2.dtype), d, need_dx, ctx.needs_input_grad[1 + 2 * i],
                                     b is not None and ctx.needs_input_grad[2 + 2 * i],
                                     dx_epi=H.EPI_NONE if dx is None else H.EPI_ADD, dx_aux=dx, master=w, bias_param=b)
            grads += [dw, db]
        return (None if dx is None else dx.view(ctx.xshape), *grads)


def linear_multi(x, *wb):
    """(x W1^T + b1, x W2^T + b2, ...) with ONE gradient tensor for x (see MultiLinearFn)"""
    return MultiLinearFn.apply(x, *wb)


def seq_fan(seq, w1, b1, w2, b2):
    """-> (seq W1^T + b1, seq W2^T + b2, seq[:, 0]) with ONE gradient tensor for `seq` (see SeqFanFn)"""
    return SeqFanFn.apply(seq, w1, b1, w2, b2)


class HeadLinearFn(torch.autograd.Function):
    """y[..., h*d + j] = sum_e x[..., e] * w[h, e, j]: the per-head projections of the IAOG decoder `Attention`
    (w_kx / w_qx [n_head, E, d], mm_modeling.py:57-58,79-92) as ONE GEMM against the [n_head*d, E] re-layout of the
    parameter instead of the reference's B-fold `repeat` + bmm.  The re-layouts (and their bf16 casts) are cached
    on the PARAMETER (`shadows.derived`), never on a temporary."""

    @staticmethod
    def _layouts(w, dtype):
        nh, E, d = w.shape

        def nk(src):      # [n_head*d, E]: nn.Linear layout, natural head order
            return cast(src.permute(0, 2, 1).reshape(nh * d, E), dtype)

        def kn(src):      # [E, n_head*d]: its transpose, the K-contiguous operand of dx = dy W
            return cast(src.permute(1, 0, 2).reshape(E, nh * d), dtype)
        return shadows.derived(w, ("head_nk", dtype), nk), (lambda: shadows.derived(w, ("head_kn", dtype), kn))

    @staticmethod
    def forward(ctx, x, w):
        x2 = _rows(x)
        nh, E, d = w.shape
        wl = shadows.head_nk([w]) if x2.dtype == torch.bfloat16 else HeadLinearFn._layouts(w, x2.dtype)[0]
        y = _linear_fwd(x2, wl, None)
        ctx.save_for_backward(x2, w)
        ctx.xshape = x.shape
        return y.view(*x.shape[:-1], nh * d)

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        nh, E, d = w.shape
        M, N = x2.shape[0], nh * d
        dy2 = dy.reshape(-1, N).contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, E), dtype=dy2.dtype, device=dy2.device)
            if dy2.dtype == torch.bfloat16:
                gemm(dy2, shadows.head_nk([w]), dx, M, E, N, N, E, E, 0, 1)      # NN: dx = dy W with W in the forward's [N, E] layout
            else:
                _, kn = HeadLinearFn._layouts(w, x2.dtype)
                gemm(dy2, kn(), dx, M, E, N, N, N, E, 0, 0)                  # NT: both operands K-contiguous
            dx = dx.view(ctx.xshape)
        if ctx.needs_input_grad[1]:
            direct = head_weight_grad(x2, dy2, [w])                          # straight into the parameter's [n_head, E, d] arena slice
            if direct is not None:
                dw = direct[0]
            else:
                dwl = torch.empty((N, E), dtype=torch.float32, device=dy2.device)    # (fresh buffer: written, not accumulated into)
                gemm(dy2, x2, dwl, N, E, M, N, _ld(x2), E, 1, 1)                 # [n_head*d, E] = dy^T x
                dw = dwl.view(nh, d, E).permute(0, 2, 1)                         # the parameter's [n_head, E, d] layout
        return dx, dw


def head_linear(x, w):
    return HeadLinearFn.apply(x, w)


class VocabLinearFn(torch.autograd.Function):
    """logits = x W^T + b for a weight whose row count is not a multiple of 8 (IAOG: the tied 64001 x 768
    vocabulary matrix, fcmf_pretraining.py:159-166), bf16 mode: every GEMM runs on the MFMA kernels over the
    row-padded weight copy and column-padded logits / logit gradients (the padding columns are exact zeros);
    without it all three GEMMs of the projection fall to the any-stride f32-MFMA kernel."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x2 = _rows(x)
        V, K = weight.shape
        wp = shadows.padded(weight)                       # [Vp, K] bf16, rows >= V zero
        Vp = wp.shape[0]
        M = x2.shape[0]
        bp = None
        if bias is not None:
            bp = torch.zeros(Vp, dtype=torch.float32, device=x2.device)
            bp[:V] = bias.detach()
        y = torch.empty((M, Vp), dtype=x2.dtype, device=x2.device)
        gemm(x2, wp, y, M, Vp, K, _ld(x2), K, Vp, 0, 0, bias=bp)
        ctx.save_for_backward(x2, weight)
        ctx.has_bias = bias is not None
        ctx.xshape = x.shape
        return y[:, :V].reshape(*x.shape[:-1], V)          # one compaction copy: consumers see the reference shape

    @staticmethod
    def backward(ctx, dy):
        x2, weight = ctx.saved_tensors
        V, K = weight.shape
        wp = shadows.padded(weight)
        Vp = wp.shape[0]
        M = x2.shape[0]
        dyp = torch.zeros((M, Vp), dtype=x2.dtype, device=x2.device)
        dyp[:, :V] = dy.reshape(M, V)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = gemm_dx_long_k(dyp, wp, M, K, Vp, K).view(ctx.xshape)
        if ctx.needs_input_grad[1]:
            dwp = torch.zeros((Vp, K), dtype=torch.float32, device=x2.device)
            gemm(dyp, x2, dwp, Vp, K, M, Vp, _ld(x2), K, 1, 1, acc=True)
            dw = dwp[:V]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = colsum(dyp, M, Vp, Vp)[:V]
        return dx, dw, db


def vocab_linear(x, weight, bias=None):
    """nn.Linear onto a vocabulary: the padded MFMA path for ragged vocabularies in bf16 mode, else `linear`"""
    if compute_dtype() == torch.bfloat16 and x.dtype == torch.bfloat16 and weight.shape[0] % VOCAB_PAD != 0:
        return VocabLinearFn.apply(x, weight, bias)
    return LinearFn.apply(x, weight, bias, None)


class VocabCrossEntropyFn(torch.autograd.Function):
    """loss = CrossEntropy(x W^T + b, labels; ignore_index) in ONE autograd node: the IAOG head
    (mm_modeling.py:662 logits -> run_pretraining_fcmf.py:322-324 loss).  The [rows, V] logits live only inside the
    node, in a column-padded buffer the MFMA kernels can write (64001 -> 64032 columns); the loss kernel reads it with
    its row stride, the logit gradient overwrites it IN PLACE and feeds the dX / dW GEMMs directly -- no compaction
    copy to the reference's [B, Ld, V] shape, no padded re-copy and no zero-filled gradient buffer in the backward."""

    @staticmethod
    def forward(ctx, x, weight, bias, labels, ignore_index):
        x2 = _rows(x)
        V, K = weight.shape
        M = x2.shape[0]
        if x2.dtype == torch.bfloat16 and V % VOCAB_PAD != 0:
            w = shadows.padded(weight)                     # [Vp, K] bf16, rows >= V zero
        else:
            w = as_compute(weight, x2.dtype)
        Vp = w.shape[0]
        bp = None
        if bias is not None:
            bp = bias.detach()
            if Vp != V:
                bp = torch.zeros(Vp, dtype=torch.float32, device=x2.device)
                bp[:V] = bias.detach()
        logits = torch.empty((M, Vp), dtype=x2.dtype, device=x2.device)
        gemm(x2, w, logits, M, Vp, K, _ld(x2), K, Vp, 0, 0, bias=bp)
        lb = labels.reshape(-1).contiguous()
        rows = torch.empty(M, dtype=torch.float32, device=x2.device)
        nvalid = torch.zeros(1, dtype=torch.float32, device=x2.device)
        H.check(H.lib().fcmf_xent_fwd(H.ptr(logits), Vp, H.ptr(lb), H.ptr(rows), H.ptr(nvalid), M, V, ignore_index,
                                      H.dt(logits), H.stream()), "fcmf_xent_fwd")
        ctx.save_for_backward(x2, weight, logits, lb, nvalid)
        ctx.cfg = (ignore_index, bias is not None, x.shape)
        return rows.sum() / nvalid[0]

    @staticmethod
    def backward(ctx, g):
        x2, weight, logits, lb, nvalid = ctx.saved_tensors
        ignore_index, has_bias, xshape = ctx.cfg
        V, K = weight.shape
        M, Vp = logits.shape
        w = shadows.padded(weight) if (x2.dtype == torch.bfloat16 and V % VOCAB_PAD != 0) else as_compute(weight, x2.dtype)
        scale = (g.float() / nvalid[0]).reshape(1).contiguous()
        d = logits                                          # in place: the node owns the buffer
        H.check(H.lib().fcmf_xent_bwd(H.ptr(logits), Vp, H.ptr(lb), H.ptr(d), Vp, H.ptr(scale), 1.0, M, V, ignore_index,
                                      H.dt(logits), H.stream()), "fcmf_xent_bwd")
        if Vp != V:
            d[:, V:].zero_()                                # the padding logits (= bias 0) must not leak into dX / dW
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = gemm_dx_long_k(d, w, M, K, Vp, K).view(xshape)
        if ctx.needs_input_grad[1]:
            # the tied vocabulary matrix: its arena slice has zeroed slack rows up to Vp (dp.GradArena pad_rows), so the padded product
            # accumulates straight into it -- no 196 MB zero fill, and the embedding lookup's gradient (the other producer of the tied
            # matrix) adds to the same memory in place instead of through a 196 MB autograd add
            got = _grad_arena.take_rows(weight, Vp) if (_grad_arena is not None and weight.dtype == torch.float32 and weight.is_contiguous()) else None
            if got is not None:
                dwp, dw = got
            else:
                dwp = torch.zeros((Vp, K), dtype=torch.float32, device=x2.device)
                dw = dwp[:V]
            gemm(d, x2, dwp, Vp, K, M, Vp, _ld(x2), K, 1, 1, acc=True)
        if has_bias and ctx.needs_input_grad[2]:
            db = colsum(d, M, Vp, Vp)[:V]
        return dx, dw, db, None, None


def vocab_cross_entropy(x, weight, bias, labels, ignore_index=-100):
    """mean CE of the vocabulary projection of x over the non-ignored positions (torch CrossEntropyLoss semantics)"""
    return VocabCrossEntropyFn.apply(x, weight, bias, labels, int(ignore_index))


class FFNFn(torch.autograd.Function):
    """y = gelu_erf(x W1^T + b1) W2^T + b2   (BertIntermediate + BertOutput.dense,
    mm_modeling.py:305-314,320; decoder PositionWiseFFN :558-565).  GELU is the epilogue of the
    first GEMM; in the backward gelu' is the epilogue of the dA GEMM."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        x2 = _rows(x)
        c1, c2 = as_compute(w1, x2.dtype), as_compute(w2, x2.dtype)
        M = x2.shape[0]
        u = torch.empty((M, w1.shape[0]), dtype=x2.dtype, device=x2.device)
        a = _linear_fwd(x2, c1, b1.detach(), H.EPI_GELU, aux=u, master=w1)
        y = _linear_fwd(a, c2, b2.detach(), master=w2)
        ctx.save_for_backward(x2, w1, w2, u, a)
        ctx.xshape = x.shape
        return y.view(*x.shape[:-1], w2.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w1, w2, u, a = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1]).contiguous()
        c1, c2 = as_compute(w1, x2.dtype), as_compute(w2, x2.dtype)
        du, dw2, db2 = _linear_bwd(a, c2, dy2, True, True, True, dx_epi=H.EPI_DGELU, dx_aux=u, master=w2)
        dx, dw1, db1 = _linear_bwd(x2, c1, du, ctx.needs_input_grad[0], True, True, master=w1)
        return (None if dx is None else dx.view(ctx.xshape)), dw1, db1, dw2, db2


def ffn(x, w1, b1, w2, b2):
    return FFNFn.apply(x, w1, b1, w2, b2)


# --------------------------------------------------------------------------------------
# residual + dropout + LayerNorm
# --------------------------------------------------------------------------------------
def ln_workspace(rows, Hd, device):
    """scratch for the per-workgroup partial column sums of fcmf_add_ln_bwd (stream-ordered reuse is safe:
    every call fully rewrites the part it reads)"""
    n = H.lib().fcmf_add_ln_bwd_workspace(rows, Hd)
    return torch.empty(n, dtype=torch.float32, device=device)


class AddLNFn(torch.autograd.Function):
    """LN(dropout(x) + res): BertSelfOutput / BertOutput / AddNorm (mm_modeling.py:276-280,
    324-328, 570-573) with FCMFLayerNorm (:167-171) or nn.LayerNorm (HF, eps 1e-5)."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, eps, p, seed):
        x2 = _rows(x).contiguous()
        rows, Hd = x2.shape
        r2 = None if res is None else _rows(res)
        y = torch.empty_like(x2)
        z = torch.empty_like(x2)
        mean = torch.empty(rows, dtype=torch.float32, device=x2.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x2.device)
        H.require_cuda(x2)
        H.check(H.lib().fcmf_add_ln_fwd(H.ptr(x2), H.ptr(r2), 0 if r2 is None else _ld(r2), H.ptr(gamma), H.ptr(beta),
                                        H.ptr(y), H.ptr(z), H.ptr(mean), H.ptr(rstd), rows, Hd, eps, p, seed,
                                        H.dt(x2), H.stream()), "fcmf_add_ln_fwd")
        ctx.save_for_backward(z, gamma, mean, rstd, beta)
        ctx.p, ctx.seed, ctx.xshape = p, seed, x.shape
        ctx.res_shape = None if res is None else res.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        z, gamma, mean, rstd, beta = ctx.saved_tensors
        rows, Hd = z.shape
        dy2 = dy.reshape(rows, Hd).contiguous()
        dz = torch.empty_like(z)
        dx = torch.empty_like(z) if ctx.p > 0 else None
        dgbuf, dg = alloc_grad_ex(gamma, (Hd,))       # the arena slices (already zero) where an arena is active: no fill launches
        dbbuf, db = alloc_grad_ex(beta, (Hd,))          # (dg / db = None: a later use of shared parameters, accumulated in place)
        H.check(H.lib().fcmf_add_ln_bwd(H.ptr(dy2), H.ptr(z), H.ptr(gamma), H.ptr(mean), H.ptr(rstd), H.ptr(dz),
                                        H.ptr(dx), H.ptr(dgbuf), H.ptr(dbbuf), 0, H.ptr(ln_workspace(rows, Hd, z.device)), rows, Hd, ctx.p, ctx.seed, H.dt(z),
                                        H.stream()), "fcmf_add_ln_bwd")
        dxo = (dx if dx is not None else dz).view(ctx.xshape)
        dres = None if ctx.res_shape is None else dz.view(ctx.res_shape)
        return dxo, dres, dg, db, None, None, None


def add_layer_norm(x, res, gamma, beta, eps, p=0.0, training=False):
    p = float(p) if training else 0.0
    return AddLNFn.apply(x, res, gamma, beta, float(eps), p, next_seed() if p > 0 else 0)


class DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        x2 = x.contiguous()
        y = torch.empty_like(x2)
        H.require_cuda(x2)
        H.check(H.lib().fcmf_dropout(H.ptr(x2), H.ptr(y), x2.numel(), p, seed, H.dt(x2), H.stream()), "fcmf_dropout")
        ctx.p, ctx.seed = p, seed
        return y

    @staticmethod
    def backward(ctx, dy):
        d = dy.contiguous()
        o = torch.empty_like(d)
        H.check(H.lib().fcmf_dropout(H.ptr(d), H.ptr(o), d.numel(), ctx.p, ctx.seed, H.dt(d), H.stream()), "fcmf_dropout")
        return o, None, None


def dropout(x, p, training):
    if not training or p <= 0:
        return x
    return DropoutFn.apply(x, float(p), next_seed())


# --------------------------------------------------------------------------------------
# RoBERTa embeddings
# --------------------------------------------------------------------------------------
def position_ids(input_ids, pad_id):
    """HF create_position_ids_from_input_ids"""
    ids = input_ids.contiguous()
    H.require_cuda(ids)
    pos = torch.empty_like(ids)
    H.check(H.lib().fcmf_position_ids(H.ptr(ids), H.ptr(pos), ids.shape[0], ids.shape[1], pad_id, H.stream()),
            "fcmf_position_ids")
    return pos


class EmbedLNFn(torch.autograd.Function):
    """RobertaEmbeddings: LN(word[ids] + type[tt] + pos[pos]) then dropout"""

    @staticmethod
    def forward(ctx, ids, pos, tt, word, ptab, ttab, gamma, beta, eps, p, seed, pad_id, out_dtype):
        ids, pos = ids.contiguous(), pos.contiguous()
        tt = None if tt is None else tt.contiguous()
        ntok, Hd = ids.numel(), word.shape[1]
        y = torch.empty((ntok, Hd), dtype=out_dtype, device=word.device)
        z = torch.empty_like(y)
        mean = torch.empty(ntok, dtype=torch.float32, device=word.device)
        rstd = torch.empty(ntok, dtype=torch.float32, device=word.device)
        H.require_cuda(ids, word)
        H.check(H.lib().fcmf_embed_ln_fwd(H.ptr(ids), H.ptr(pos), H.ptr(tt), H.ptr(word), H.ptr(ptab), H.ptr(ttab),
                                          H.ptr(gamma), H.ptr(beta), H.ptr(y), H.ptr(z), H.ptr(mean), H.ptr(rstd),
                                          ntok, Hd, eps, p, seed, H.dt(y), H.stream()), "fcmf_embed_ln_fwd")
        ctx.save_for_backward(ids, pos, tt, z, gamma, mean, rstd)
        ctx.p, ctx.seed, ctx.pad_id = p, seed, pad_id
        ctx.word = word if (word.dtype == torch.float32 and word.is_contiguous()) else None   # (the Parameter: arena lookup)
        ok = lambda t: t if (t is not None and t.dtype == torch.float32 and t.is_contiguous()) else None
        ctx.tabs = (ok(ptab), ok(ttab), ok(gamma), ok(beta))                                     # (likewise: their arena slices)
        ctx.shapes = (word.shape, ptab.shape, ttab.shape)
        return y.view(*ids.shape, Hd)

    @staticmethod
    def backward(ctx, dy):
        ids, pos, tt, z, gamma, mean, rstd = ctx.saved_tensors
        ntok, Hd = z.shape
        d = dy.reshape(ntok, Hd).contiguous()
        L = H.lib()
        if ctx.p > 0:
            o = torch.empty_like(d)
            H.check(L.fcmf_dropout(H.ptr(d), H.ptr(o), d.numel(), ctx.p, ctx.seed, H.dt(d), H.stream()), "fcmf_dropout")
            d = o
        dz = torch.empty_like(z)
        ptab, ttab, gpar, bpar = ctx.tabs
        zeros = lambda shape: torch.zeros(shape, dtype=torch.float32, device=z.device)
        dg = alloc_grad(gpar, (Hd,)) if gpar is not None else zeros(Hd)        # (arena slices: already zero, no fill launches)
        db = alloc_grad(bpar, (Hd,)) if bpar is not None else zeros(Hd)
        H.check(L.fcmf_add_ln_bwd(H.ptr(d), H.ptr(z), H.ptr(gamma), H.ptr(mean), H.ptr(rstd), H.ptr(dz), 0, H.ptr(dg),
                                  H.ptr(db), 0, H.ptr(ln_workspace(ntok, Hd, z.device)), ntok, Hd, 0.0, 0, H.dt(z), H.stream()), "fcmf_add_ln_bwd")
        ws, ps, ts = ctx.shapes
        if ctx.word is not None:
            dwordbuf, dword = alloc_grad_ex(ctx.word, ws)      # (dword = None: the tied matrix's slice was claimed by the vocabulary projection)
        else:
            dwordbuf = dword = torch.zeros(ws, dtype=torch.float32, device=z.device)
        dpos = alloc_grad(ptab, ps) if ptab is not None else zeros(ps)
        dtt = alloc_grad(ttab, ts) if ttab is not None else zeros(ts)
        two_d = ids.dim() == 2 and pos.is_contiguous()
        fused = False
        if two_d and (tt is None or tt.is_contiguous()):
            # [sequences, S] layout: position rows are shared by the offsets of all sequences; position and type gradients in ONE pass
            rc = L.fcmf_embed_pos_type_bwd(H.ptr(dz), H.ptr(pos), H.ptr(tt), H.ptr(dpos), H.ptr(dtt), ids.shape[0], ids.shape[1], Hd,
                                           ctx.pad_id, H.dt(dz), H.stream())
            fused = rc == 0
            if rc not in (0, H.ERR_UNSUPPORTED):
                H.check(rc, "fcmf_embed_pos_type_bwd")
        H.check(L.fcmf_embed_bwd(H.ptr(dz), H.ptr(ids), H.ptr(pos), H.ptr(tt), H.ptr(dwordbuf), None if two_d else H.ptr(dpos),
                                 None if fused else H.ptr(dtt), ntok, Hd, ctx.pad_id, H.dt(dz), H.stream()), "fcmf_embed_bwd")
        if two_d and not fused:
            H.check(L.fcmf_embed_pos_bwd(H.ptr(dz), H.ptr(pos), H.ptr(dpos), ids.shape[0], ids.shape[1], Hd, ctx.pad_id,
                                         H.dt(dz), H.stream()), "fcmf_embed_pos_bwd")
        return None, None, None, dword, dpos, dtt, dg, db, None, None, None, None, None


def embed_layer_norm(ids, pos, tt, word, ptab, ttab, gamma, beta, eps, p, training, pad_id, out_dtype):
    p = float(p) if training else 0.0
    return EmbedLNFn.apply(ids, pos, tt, word, ptab, ttab, gamma, beta, float(eps), p, next_seed() if p > 0 else 0,
                           int(pad_id), out_dtype)


# --------------------------------------------------------------------------------------
# attention
# --------------------------------------------------------------------------------------
def _desc(q, k1, v1, k2, v2, mask, bias, heads, group_div, scale, p, seed, causal, head_quirk):
    G, R, HD = q.shape
    d = HD // heads
    T1 = 0 if k1 is None else k1.shape[1]
    T2 = 0 if k2 is None else k2.shape[2]
    a = H.AttnDesc()
    a.dtype, a.G, a.heads, a.d, a.R, a.T1, a.T2, a.group_div = H.dt(q), G, heads, d, R, T1, T2, group_div
    a.q_sg, a.q_sr = q.stride(0), q.stride(1)
    if k1 is not None:
        a.k1_sg, a.k1_st = k1.stride(0), k1.stride(1)
    if k2 is not None:
        a.k2_sg, a.k2_sr, a.k2_st = k2.stride(0), k2.stride(1), k2.stride(2)
    a.o_sg, a.o_sr = R * HD, HD
    a.q, a.k1, a.v1, a.k2, a.v2 = H.ptr(q), H.ptr(k1), H.ptr(v1), H.ptr(k2), H.ptr(v2)
    a.mask, a.bias = H.ptr(mask), H.ptr(bias)
    a.scale, a.dropout_p, a.seed, a.causal, a.head_quirk = scale, p, seed, int(causal), int(head_quirk)
    return a


def _chk_same_layout(a, b):
    if a is not None and (a.stride() != b.stride() or a.shape != b.shape):
        raise H.HipLibraryError("attention: K and V of a segment must share shape and strides")


USE_MFMA_ATTENTION = True   # tests flip this to compare the MFMA kernel with the VALU kernel


class AttentionFn(torch.autograd.Function):
    """Two-segment multi-head attention (see fcmf_attn_desc in include/fcmf_hip.h).
      q  [G,R,heads*d]; k1/v1 [G,T1,heads*d] shared by the R rows of a group;
      k2/v2 [G/group_div,R,T2,heads*d] private per row; mask [G,T1+T2] additive float32;
      bias [G/group_div,heads,R,T1+T2] additive float32.  Returns [G,R,heads*d]."""

    @staticmethod
    def forward(ctx, q, k1, v1, k2, v2, mask, bias, heads, group_div, scale, p, seed, causal):
        H.require_cuda(q)
        # (row stride 0 = one query row expanded over R rows -- the [CLS] query against every image's keys: read in place)
        q = q if q.stride(2) == 1 and (q.stride(1) >= q.shape[2] or q.stride(1) == 0) else q.contiguous()
        k1 = None if k1 is None else (k1 if k1.stride(2) == 1 else k1.contiguous())
        v1 = None if v1 is None else (v1 if v1.stride() == k1.stride() else v1.contiguous())
        if k1 is not None and v1.stride() != k1.stride():
            k1 = k1.contiguous()
        k2 = None if k2 is None else (k2 if k2.stride(3) == 1 else k2.contiguous())
        v2 = None if v2 is None else (v2 if v2.stride() == k2.stride() else v2.contiguous())
        if k2 is not None and v2.stride() != k2.stride():
            k2 = k2.contiguous()
        mask = None if mask is None else mask.contiguous().float()
        bias = None if bias is None else bias.contiguous()
        G, R, HD = q.shape
        out = torch.empty((G, R, HD), dtype=q.dtype, device=q.device)
        lse = torch.empty((G, heads, R), dtype=torch.float32, device=q.device)
        # text-encoder shape (bf16, head dim 64, <=256 queries / keys, plain mask): MFMA kernel
        mfma = (q.dtype == torch.bfloat16 and HD // heads == 64 and k2 is None and bias is None and not causal
                and k1 is not None and k1.shape[1] <= 256 and R <= 256 and q.is_contiguous() and k1.is_contiguous()
                and v1.is_contiguous() and USE_MFMA_ATTENTION)
        if mfma:
            H.check(H.lib().fcmf_attn_mfma_fwd(H.ptr(q), H.ptr(k1), H.ptr(v1), H.ptr(mask), H.ptr(out), H.ptr(lse), G, heads,
                                               R, k1.shape[1], HD, HD, HD, scale, p, seed, H.stream()), "fcmf_attn_mfma_fwd")
        else:
            a = _desc(q, k1, v1, k2, v2, mask, bias, heads, group_div, scale, p, seed, causal, 0)
            H.check(H.lib().fcmf_attn_small_fwd(a, H.ptr(out), H.ptr(lse), H.stream()), "fcmf_attn_small_fwd")
        ctx.save_for_backward(q, k1, v1, k2, v2, mask, bias, out, lse)
        ctx.cfg = (heads, group_div, scale, p, seed, causal)
        ctx.mfma = mfma
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k1, v1, k2, v2, mask, bias, out, lse = ctx.saved_tensors
        heads, group_div, scale, p, seed, causal = ctx.cfg
        G, R, HD = q.shape
        dout = dout.contiguous()
        if ctx.mfma:
            dq, dk, dv = torch.empty_like(q), torch.empty_like(k1), torch.empty_like(v1)
            H.check(H.lib().fcmf_attn_mfma_bwd(H.ptr(q), H.ptr(k1), H.ptr(v1), H.ptr(mask), H.ptr(out), H.ptr(dout),
                                               H.ptr(lse), H.ptr(dq), H.ptr(dk), H.ptr(dv), G, heads, R, k1.shape[1], HD, HD,
                                               HD, scale, p, seed, None, H.stream()), "fcmf_attn_mfma_bwd")
            return dq, dk, dv, None, None, None, None, None, None, None, None, None, None
        T1 = 0 if k1 is None else k1.shape[1]
        nch = max(1, (T1 + 127) // 128)
        dq = torch.empty((nch, G, R, HD), dtype=q.dtype, device=q.device)   # one partial per 128-key chunk
        dk1 = dv1 = dk2 = dv2 = dbias = None
        if k1 is not None:
            dk1 = torch.empty((G, T1, HD), dtype=k1.dtype, device=q.device)
            dv1 = torch.empty((G, T1, HD), dtype=k1.dtype, device=q.device)
        T2 = 0 if k2 is None else k2.shape[2]
        # private keys shared by `group_div` groups: the library sums their gradients over the group (no per-group rows)
        grouped = k2 is not None and group_div <= 8 and G % group_div == 0
        if k2 is not None:
            G2 = G // group_div if grouped else G
            dk2 = torch.empty((G2, R, T2, HD), dtype=q.dtype, device=q.device)
            dv2 = torch.empty((G2, R, T2, HD), dtype=q.dtype, device=q.device)
        if bias is not None and ctx.needs_input_grad[6]:
            T = (0 if k1 is None else k1.shape[1]) + T2
            dbias = torch.empty((G, heads, R, T), dtype=torch.float32, device=q.device)
        a = _desc(q, k1, v1, k2, v2, mask, bias, heads, group_div, scale, p, seed, causal, 0)
        if grouped:
            scratch = torch.empty(2 * G * heads * R * T2, dtype=torch.float32, device=q.device)
            H.check(H.lib().fcmf_attn_small_bwd_grouped(a, H.ptr(out), H.ptr(dout), H.ptr(lse), H.ptr(dq), H.ptr(dk1), H.ptr(dv1),
                                                        H.ptr(dk2), H.ptr(dv2), H.ptr(dbias), H.ptr(scratch), scratch.numel() * 4,
                                                        H.stream()), "fcmf_attn_small_bwd_grouped")
        else:
            H.check(H.lib().fcmf_attn_small_bwd(a, H.ptr(out), H.ptr(dout), H.ptr(lse), H.ptr(dq), H.ptr(dk1), H.ptr(dv1),
                                                H.ptr(dk2), H.ptr(dv2), H.ptr(dbias), H.stream()), "fcmf_attn_small_bwd")
        dq = _sum_leading(dq)
        if group_div > 1:
            if dk2 is not None and not grouped:
                dk2, dv2 = _sum_groups(dk2, group_div), _sum_groups(dv2, group_div)
            if dbias is not None:
                dbias = _sum_groups(dbias, group_div)
        return dq, dk1, dv1, dk2, dv2, None, dbias, None, None, None, None, None, None


def _sum_leading(x):
    """[n, ...] -> [...] summing the leading axis (attention dq chunk partials)"""
    if x.shape[0] == 1:
        return x[0]
    inner = x[0].numel()
    out = torch.empty(x.shape[1:], dtype=x.dtype, device=x.device)
    H.check(H.lib().fcmf_sum_axis(H.ptr(x), H.ptr(out), 1, x.shape[0], inner, H.dt(x), H.stream()), "fcmf_sum_axis")
    return out


def _sum_groups(x, reps):
    """[G, ...] -> [G/reps, ...] summing consecutive groups"""
    G = x.shape[0]
    inner = x[0].numel()
    out = torch.empty((G // reps,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    H.check(H.lib().fcmf_sum_axis(H.ptr(x), H.ptr(out), G // reps, reps, inner, H.dt(x), H.stream()), "fcmf_sum_axis")
    return out


def attention(q, k1=None, v1=None, k2=None, v2=None, mask=None, bias=None, heads=12, group_div=1, scale=None,
              p=0.0, training=False, causal=False):
    d = q.shape[-1] // heads
    scale = 1.0 / math.sqrt(d) if scale is None else scale
    p = float(p) if training else 0.0
    return AttentionFn.apply(q, k1, v1, k2, v2, mask, bias, heads, group_div, float(scale), p,
                             next_seed() if p > 0 else 0, causal)


# --------------------------------------------------------------------------------------
# box geometry
# --------------------------------------------------------------------------------------
_dim_mat_cache = {}


def box_dim_mat(device):
    """1/1000^(k/8), k=0..7, rounded exactly as roi_modeling.py:123-125 does (float32)"""
    key = str(device)
    if key not in _dim_mat_cache:
        feat_range = torch.arange(64 / 8)
        dm = 1.0 / torch.pow(1000, feat_range / (64 / 8))
        _dim_mat_cache[key] = dm.float().to(device)
    return _dim_mat_cache[key]


class BoxBiasFn(torch.autograd.Function):
    """log(clamp(relu(WG(emb(boxes))), 1e-6)) -> [G,heads,N,N] float32 (roi_modeling.py:148-163,40)"""

    @staticmethod
    def forward(ctx, coords, wg_w, wg_b, fast_trig=False):
        c = coords.contiguous()
        H.require_cuda(c, wg_w)
        G, N, _ = c.shape
        heads = wg_w.shape[0]
        ww, wb = wg_w.detach().contiguous().float(), wg_b.detach().contiguous().float()
        bias = torch.empty((G, heads, N, N), dtype=torch.float32, device=c.device)
        ctx.cdt = H.dt(c) | (H.BOX_FAST_TRIG if fast_trig and c.dtype == torch.float32 else 0)
        H.check(H.lib().fcmf_box_bias_fwd(H.ptr(c), ctx.cdt, H.ptr(box_dim_mat(c.device)), H.ptr(ww), H.ptr(wb),
                                          H.ptr(bias), G, N, heads, H.stream()), "fcmf_box_bias_fwd")
        ctx.save_for_backward(c, ww, wb)
        return bias

    @staticmethod
    def backward(ctx, dbias):
        c, ww, wb = ctx.saved_tensors
        G, N, _ = c.shape
        heads = ww.shape[0]
        buf = torch.zeros(ww.numel() + wb.numel(), dtype=torch.float32, device=ww.device)       # (one fill for both)
        dw, db = buf[:ww.numel()].view_as(ww), buf[ww.numel():].view_as(wb)
        d = dbias.contiguous().float()
        H.check(H.lib().fcmf_box_bias_bwd(H.ptr(c), ctx.cdt, H.ptr(box_dim_mat(c.device)), H.ptr(ww), H.ptr(wb), H.ptr(d),
                                          H.ptr(dw), H.ptr(db), G, N, heads, H.stream()), "fcmf_box_bias_bwd")
        return None, dw, db, None


def box_bias(coords, wg_w, wg_b):
    # The reference does the geometry in the dtype of the coordinates (float64 from the data loader,
    # roi_modeling.py:79-138) and rounds to float32 (:150).  The parity (fp32) mode keeps that; the bf16 mode
    # runs the float32 instantiation of the same kernels (sincosf instead of f64 sincos: 5x faster, error ~1e-5
    # on a bias that is consumed in bf16).
    # ... and the hardware's sine / cosine (FCMF_BOX_FAST_TRIG: the 64 sincosf calls per box pair were two thirds of both kernels;
    # argument error <= 1e-4 on a bias whose bf16 rounding is 3e-2).
    bf16_mode = compute_dtype() == torch.bfloat16
    if bf16_mode and coords.dtype == torch.float64:
        coords = coords.float()
    return BoxBiasFn.apply(coords, wg_w, wg_b, bf16_mode)


def box_embedding(coords):
    c = coords.contiguous()
    H.require_cuda(c)
    G, N, _ = c.shape
    emb = torch.empty((G, N, N, 64), dtype=torch.float32, device=c.device)
    H.check(H.lib().fcmf_box_embedding(H.ptr(c), H.dt(c), H.ptr(box_dim_mat(c.device)), H.ptr(emb), G, N, H.stream()),
            "fcmf_box_embedding")
    return emb


# --------------------------------------------------------------------------------------
# cross entropy
# --------------------------------------------------------------------------------------
class XentFn(torch.autograd.Function):
    """mean cross entropy over the non-ignored rows (torch.nn.CrossEntropyLoss semantics) x `mult`: rows, their mean and the
    backward's scale in two launches, no torch arithmetic (fcmf_xent_fwd + fcmf_xent_mean)"""

    @staticmethod
    def forward(ctx, logits, labels, ignore_index, mult):
        lg = logits if logits.stride(-1) == 1 else logits.contiguous()
        lg = lg.reshape(-1, lg.shape[-1]) if lg.dim() != 2 else lg
        lb = labels.reshape(-1).contiguous()
        H.require_cuda(lg, lb)
        n, C = lg.shape
        rows = torch.empty(n, dtype=torch.float32, device=lg.device)
        out2 = torch.empty(2, dtype=torch.float32, device=lg.device)       # [loss, mult / nvalid]
        L = H.lib()
        H.check(L.fcmf_xent_fwd(H.ptr(lg), lg.stride(0), H.ptr(lb), H.ptr(rows), None, n, C, ignore_index, H.dt(lg), H.stream()),
                "fcmf_xent_fwd")
        H.check(L.fcmf_xent_mean(H.ptr(rows), H.ptr(lb), n, ignore_index, float(mult), H.ptr(out2), H.stream()), "fcmf_xent_mean")
        ctx.save_for_backward(lg, lb, out2)
        ctx.ignore_index = ignore_index
        ctx.lshape = logits.shape
        return out2[0]

    @staticmethod
    def backward(ctx, g):
        lg, lb, out2 = ctx.saved_tensors
        n, C = lg.shape
        d = torch.empty((n, C), dtype=lg.dtype, device=lg.device)
        scale = (g.float() * out2[1]).reshape(1)
        H.check(H.lib().fcmf_xent_bwd(H.ptr(lg), lg.stride(0), H.ptr(lb), H.ptr(d), C, H.ptr(scale), 1.0, n, C,
                                      ctx.ignore_index, H.dt(lg), H.stream()), "fcmf_xent_bwd")
        return d.view(ctx.lshape), None, None, None


def additive_mask(m, value):
    """(m - 1) * (-value) as float32 for a 0 / 1 int64 mask [G, L] (row stride free): fcmf_additive_mask"""
    G, Lc = m.shape
    out = torch.empty((G, Lc), dtype=torch.float32, device=m.device)
    H.check(H.lib().fcmf_additive_mask(H.ptr(m), m.stride(0), H.ptr(out), G, Lc, float(value), H.stream()), "fcmf_additive_mask")
    return out


def cross_entropy(logits, labels, ignore_index=-100, mult=1.0):
    return XentFn.apply(logits, labels, int(ignore_index), float(mult))

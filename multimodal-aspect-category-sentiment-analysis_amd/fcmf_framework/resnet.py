"""ResNet-152 trunk on the MI355X kernels: the network the reference's myResNetImg / myResNetRoI drive
(fcmf_framework/resnet_utils.py:13-30,39-56), i.e. torchvision's `resnet152` (third party, not installed here).

Same attribute tree and state-dict keys as torchvision's ResNet (conv1, bn1, relu, maxpool, layer1..4 of
Bottleneck{conv1,bn1,conv2,bn2,conv3,bn3,downsample.{0,1}}, avgpool, fc), so a torchvision checkpoint loads with
`load_state_dict`, and the sub-modules can be called one by one on NCHW-shaped tensors exactly as the reference's
wrappers do.  Underneath:
  * activations live in NHWC memory (an NCHW-shaped tensor in torch's channels_last format), so every convolution is
    ONE GEMM of the step's MFMA kernels against the weight re-laid as [Cout, kh*kw*Cin]: 1x1/stride-1 convolutions
    read the activation matrix in place, the others read a patch matrix (`fcmf_conv_im2col`);
  * BatchNorm is GROUPED: the reference calls the trunk once per image index and once per (image, ROI), B crops
    each, in train() mode (run_multimodal_fcmf.py:431,449-457), so batch statistics belong to each B-crop call.
    `trunk(x, groups=G)` runs all G calls as one batch with per-group statistics and G running-statistics updates in
    call order -- bit-for-bit the reference's semantics without its Python loop of num_imgs * num_rois launches;
  * with if_fine_tune=False (the default of both drivers) the reference detaches the features
    (resnet_utils.py:26-28) and the trunk runs forward only; with if_fine_tune=True (--fine_tune_cnn) `TrunkFn` records
    what the backward needs (inputs, raw convolution outputs, normalised outputs, batch statistics) and walks the
    network in reverse by hand: BatchNorm(+ReLU) backward kernels, dX = dY W / dW = dY^T A on the GEMM kernels (the
    patch matrix is rebuilt, not stored), col2im, max-pool and average-pool backward.
There is no torch (MIOpen) fallback: CPU tensors raise HipLibraryError.
"""
import torch
import torch.nn as nn

from . import _hip as H
from . import ops

MAX_CROPS_PER_PASS = 256      # keeps every GEMM operand below the kernels' 2^31-byte offset range


def _nhwc(x):
    """NCHW-shaped tensor -> contiguous [N, H, W, C] view in the compute dtype (copies only if x is not already
    channels_last / of another dtype)"""
    H.require_cuda(x)
    dt = ops.compute_dtype()
    v = x.permute(0, 2, 3, 1)
    if not v.is_contiguous() or v.dtype != dt:
        if v.dtype not in (torch.float32, torch.bfloat16):
            v = v.float()
        v = v.contiguous()
        v = ops.cast(v, dt)
    return v


def _nchw(v):
    """[N, H, W, C] -> NCHW-shaped view (channels_last memory)"""
    return v.permute(0, 3, 1, 2)


def _weight_matrix(conv, dtype):
    """[Cout, Kpad] compute-dtype matrix of a Conv2d weight [Cout, Cin, kh, kw], k = (r, s, c), zero tail"""
    w = conv.weight
    Cout, Cin, kh, kw = w.shape
    K = kh * kw * Cin
    Kpad = (K + 31) // 32 * 32

    def build(src):
        m = src.permute(0, 2, 3, 1).reshape(Cout, K)
        if Kpad != K:
            m = torch.cat((m, m.new_zeros(Cout, Kpad - K)), 1)
        return ops.cast(m.contiguous(), dtype)
    return ops.shadows.derived(w, ("conv_rsc", dtype), build), Kpad


IMPLICIT_CONV = True      # tests flip this to compare the implicit-GEMM convolutions with the patch-matrix path
_pad_cache = {}


def _implicit_ok(C, dt):
    return IMPLICIT_CONV and dt == torch.bfloat16 and C >= 64 and (C & (C - 1)) == 0


def padded_activation(N, Hh, Ww, C, dtype, device, pad=1):
    """NHWC buffer [N, H + 2 pad, W + 2 pad, C] whose border is zero: the input of an implicit-GEMM 3x3 convolution.  One buffer
    per shape, zeroed ONCE: every use rewrites the whole interior (fcmf_bn_apply_pad) and reads it on the same stream before
    the next block of that shape writes it again."""
    key = (N, Hh, Ww, C, dtype, str(device), pad)
    buf = _pad_cache.get(key)
    if buf is None:
        buf = _pad_cache[key] = torch.zeros((N, Hh + 2 * pad, Ww + 2 * pad, C), dtype=dtype, device=device)
    return buf


FUSED_BN_APPLY = True     # tests flip this to compare fcmf_bn_finalize_apply with the separate finalize / apply kernels
FUSED_BN_STATS = True     # tests flip this to compare the GEMM-epilogue block statistics with the separate statistics pass


def _block_stats(rows, Cout, K, dtype, device):
    """(buffer, rows per block) for the (sum, sum of squares) per block of output rows and channel that a colstats GEMM emits, or
    None where no kernel emits them (the caller then runs the statistics pass)"""
    if not FUSED_BN_STATS or dtype != torch.bfloat16:
        return None
    br = H.lib().fcmf_gemm_colstats_block_rows(H.gemm_ctx(), rows, Cout, K)
    if br <= 0:
        return None
    return torch.empty(((rows + br - 1) // br, Cout, 2), dtype=torch.float32, device=device), br


def conv2d_implicit(xp, conv, N, Ho, Wo, stats=False):
    """xp [N, Hp, Wp, C] (zero border included where the convolution pads) -> [N, Ho, Wo, Cout] through fcmf_conv_gemm: no patch
    matrix (the 3x3 / strided convolutions of the trunk spent 9 of 46.7 ms building them, 120 MB per crop).
    stats: -> (y, block statistics for batchnorm_nhwc_ or None)"""
    kh, kw = conv.kernel_size
    Cout, C = conv.out_channels, xp.shape[3]
    wm, Kpad = _weight_matrix(conv, xp.dtype)
    assert Kpad == kh * kw * C
    y = torch.empty((N * Ho * Wo, Cout), dtype=xp.dtype, device=xp.device)
    blocks = _block_stats(N * Ho * Wo, Cout, Kpad, xp.dtype, xp.device) if stats else None
    flops = 2.0 * N * Ho * Wo * Cout * Kpad
    if blocks is not None:
        with ops.trace_launch(flops):
            rc = H.lib().fcmf_conv_gemm_colstats(H.gemm_ctx(), H.ptr(xp), H.ptr(wm), H.ptr(y), H.ptr(blocks[0]), N, xp.shape[1], xp.shape[2],
                                                 C, Ho, Wo, kh, kw, conv.stride[0], Cout, H.stream())
        if rc == H.ERR_UNSUPPORTED:
            blocks = None
        else:
            H.check(rc, "fcmf_conv_gemm_colstats")
    if blocks is None:
        with ops.trace_launch(flops):
            H.check(H.lib().fcmf_conv_gemm(H.gemm_ctx(), H.ptr(xp), H.ptr(wm), H.ptr(y), N, xp.shape[1], xp.shape[2], C, Ho, Wo, kh, kw,
                                           conv.stride[0], Cout, H.stream()), "fcmf_conv_gemm")
    y = y.view(N, Ho, Wo, Cout)
    return (y, blocks) if stats else y


def _stem_runs_ok(conv, x, dt):
    return (IMPLICIT_CONV and dt == torch.bfloat16 and conv.in_channels == 3 and conv.kernel_size[0] == conv.kernel_size[1] <= 8
            and conv.stride[0] == 2 and x.dtype in (torch.float32, torch.float64, torch.bfloat16))


def conv2d_stem(v, conv, stats=False):
    """the stem (conv1: k x k, k <= 8, stride 2, 3 input channels) without a patch matrix: v = the crops as a strided NHWC view
    [N, H, W, 3] of any float dtype.  fcmf_pack_rgb0 writes them as bf16 RGB0 pixels into a zero-bordered buffer, and
    fcmf_conv_gemm_runs contracts, per kernel row, one 32-element run (8 pixels x 4) of it against w [Cout, kh, 8, 4] (zeros for
    the 8th pixel and the 4th channel).  im2col + GEMM read / wrote 4.1 GB per 448 crops here, this path 0.4 GB."""
    N, Hh, Ww, _ = v.shape
    k, pad, Cout = conv.kernel_size[0], conv.padding[0], conv.out_channels
    Ho, Wo = (Hh + 2 * pad - k) // 2 + 1, (Ww + 2 * pad - k) // 2 + 1
    # (one spare column on the right: the 8-pixel run of the last output pixel of a row may end past the padded row)
    Wp = max(Ww + 2 * pad, (Wo - 1) * 2 + 8)
    key = ("rgb0", N, Hh, Ww, pad, str(v.device))
    buf = _pad_cache.get(key)
    if buf is None:
        buf = _pad_cache[key] = torch.zeros((N, Hh + 2 * pad, Wp, 4), dtype=torch.bfloat16, device=v.device)
    sn, sh, sw, sc = v.stride()
    # (dst rows are Wp pixels wide: the pack kernel takes the padded width through W + 2 pad, so hand it the buffer's own geometry)
    H.check(H.lib().fcmf_pack_rgb0(H.ptr(v), H.dt(v), H.ptr(buf), N, Hh, Ww, sn, sh, sw, sc, pad, Wp, H.stream()), "fcmf_pack_rgb0")

    def build(src):
        w = torch.zeros((Cout, k, 8, 4), dtype=torch.float32, device=src.device)
        w[:, :, :k, :3] = src.detach().float().permute(0, 2, 3, 1)            # [Cout, 3, ky, kx] -> [Cout, ky, kx, c]
        return ops.cast(w.view(Cout, k * 32).contiguous(), torch.bfloat16)
    wm = ops.shadows.derived(conv.weight, ("stem_runs", torch.bfloat16), build)
    y = torch.empty((N * Ho * Wo, Cout), dtype=torch.bfloat16, device=v.device)
    blocks = _block_stats(N * Ho * Wo, Cout, k * 32, torch.bfloat16, v.device) if stats else None
    with ops.trace_launch(2.0 * N * Ho * Wo * Cout * k * 32):
        H.check(H.lib().fcmf_conv_gemm_runs(H.gemm_ctx(), H.ptr(buf), H.ptr(wm), H.ptr(y), H.ptr(blocks[0]) if blocks is not None else None, N, Hh + 2 * pad, Wp, 4, 32, Ho, Wo, k, 2,
                                            Cout, H.stream()), "fcmf_conv_gemm_runs")
    y = y.view(N, Ho, Wo, Cout)
    return (y, blocks) if stats else y


def conv2d_nhwc(x, conv, src_strides=None, stats=False):
    """x [N,H,W,C] (or any layout with `src_strides` = element strides of (n,h,w,c)) -> [N,Ho,Wo,Cout]
    stats: -> (y, block statistics of y for batchnorm_nhwc_, or None where the GEMM cannot emit them)"""
    if conv.bias is not None or conv.groups != 1 or conv.dilation != (1, 1):
        raise H.HipLibraryError("conv2d_nhwc: bias-free, ungrouped, undilated convolutions only (ResNet trunk)")
    dt = ops.compute_dtype()
    N, Hh, Ww, C = x.shape
    kh, kw = conv.kernel_size
    st, pad = conv.stride[0], conv.padding[0]
    Cout = conv.out_channels
    wm, Kpad = _weight_matrix(conv, dt)
    Ho, Wo = (Hh + 2 * pad - kh) // st + 1, (Ww + 2 * pad - kw) // st + 1
    rows = N * Ho * Wo
    if kh == 1 and kw == 1 and st == 1 and pad == 0 and src_strides is None and x.dtype == dt and x.is_contiguous():
        A = x.view(rows, C)
    elif (kh == 1 and kw == 1 and pad == 0 and src_strides is None and x.dtype == dt and x.is_contiguous() and _implicit_ok(C, dt)
          and not torch.is_grad_enabled()):
        return conv2d_implicit(x, conv, N, Ho, Wo, stats=stats)          # strided 1x1 shortcut: rows are gathered by the GEMM's DMA
    else:
        A = torch.empty((rows, Kpad), dtype=dt, device=x.device)
        sn, sh, sw, sc = src_strides if src_strides is not None else x.stride()
        H.check(H.lib().fcmf_conv_im2col(H.ptr(x), H.dt(x), H.ptr(A), H.dt(A), N, Hh, Ww, C, sn, sh, sw, sc, kh, kw, st,
                                         pad, Kpad, H.stream()), "fcmf_conv_im2col")
    y = torch.empty((rows, Cout), dtype=dt, device=x.device)
    blocks = _block_stats(rows, Cout, Kpad, dt, x.device) if stats else None
    if blocks is not None:
        with ops.trace_launch(2.0 * rows * Cout * Kpad):
            rc = H.lib().fcmf_gemm_colstats(H.gemm_ctx(), H.ptr(A), H.ptr(wm), H.ptr(y), H.ptr(blocks[0]), rows, Cout, Kpad, Kpad, Kpad, Cout,
                                            H.stream())
        if rc == H.ERR_UNSUPPORTED:
            blocks = None
        else:
            H.check(rc, "fcmf_gemm_colstats")
    if blocks is None:
        ops.gemm(A, wm, y, rows, Cout, Kpad, Kpad, Kpad, Cout, 0, 0)
    y = y.view(N, Ho, Wo, Cout)
    return (y, blocks) if stats else y


def batchnorm_nhwc_(y, bn, groups=1, res=None, relu=False, out=None, save=None, out_pad=0, blocks=None):
    """BatchNorm2d (+ residual, + ReLU) of y [N,H,W,C], in place (or into `out`); training mode: per-group batch
    statistics and `groups` running-statistics updates (module docstring).  save: dict that receives mean / rstd /
    groups / training for the backward.  blocks: (block statistics, rows per block) the producing convolution emitted
    (conv2d_nhwc(..., stats=True)): the statistics pass over y is skipped when every group is a whole number of blocks."""
    N, Hh, Ww, C = y.shape
    rows = N * Hh * Ww
    L, st = H.lib(), H.stream()
    dev = y.device
    training = bn.training or not bn.track_running_stats
    mean = rstd = sums = None
    if training:
        if N % groups != 0:
            raise H.HipLibraryError(f"grouped BatchNorm: {N} crops do not split into {groups} equal groups")
        rpg = rows // groups
        sums = torch.empty(L.fcmf_bn_stats_workspace(rpg, groups, C), dtype=torch.float64, device=dev)
        if blocks is not None and rpg % blocks[1] == 0:
            H.check(L.fcmf_bn_stats_blocks(H.ptr(blocks[0]), H.ptr(sums), rpg, groups, C, blocks[1], st), "fcmf_bn_stats_blocks")
        else:
            H.check(L.fcmf_bn_stats(H.ptr(y), H.ptr(sums), rpg, groups, C, H.dt(y), st), "fcmf_bn_stats")
        mom = 0.1 if bn.momentum is None else float(bn.momentum)
        if bn.track_running_stats:
            rm, rv = bn.running_mean, bn.running_var
        else:
            rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        if bn.track_running_stats and bn.num_batches_tracked is not None:
            bn._pending_batches = getattr(bn, "_pending_batches", 0) + groups      # flushed lazily (one add, not 155 per pass)
        shape = (groups, C)
    else:
        rpg, groups, mom = rows, 1, 0.0
        rm, rv = bn.running_mean, bn.running_var
        shape = (C,)
    if save is not None:
        mean, rstd = (torch.empty(shape, dtype=torch.float32, device=dev) for _ in range(2))
    z = y if out is None else out
    # scale / shift derivation + normalisation (+ residual, ReLU, zero-bordered output) in one launch where the shape allows
    rc = H.ERR_UNSUPPORTED
    if FUSED_BN_APPLY:
        rc = L.fcmf_bn_finalize_apply(H.ptr(y), H.ptr(res), H.ptr(z), H.ptr(sums), H.ptr(bn.weight), H.ptr(bn.bias), H.ptr(rm), H.ptr(rv),
                                      H.ptr(mean), H.ptr(rstd), C, groups, rpg, mom, float(bn.eps), int(relu), Hh, Ww, out_pad, H.dt(y), st)
    if rc == H.ERR_UNSUPPORTED:
        scale = torch.empty(shape, dtype=torch.float32, device=dev)
        shift = torch.empty(shape, dtype=torch.float32, device=dev)
        H.check(L.fcmf_bn_finalize(H.ptr(sums), H.ptr(bn.weight), H.ptr(bn.bias), H.ptr(rm), H.ptr(rv), H.ptr(scale), H.ptr(shift),
                                   H.ptr(mean), H.ptr(rstd), C, groups, rpg if training else 0, mom, float(bn.eps), st), "fcmf_bn_finalize")
        if out_pad > 0:           # `out` is a padded_activation buffer: the interior is written, the zero border stays
            H.check(L.fcmf_bn_apply_pad(H.ptr(y), H.ptr(res), H.ptr(z), H.ptr(scale), H.ptr(shift), rows, C, rpg, int(relu), Hh, Ww,
                                        out_pad, H.dt(y), st), "fcmf_bn_apply_pad")
        else:
            H.check(L.fcmf_bn_apply(H.ptr(y), H.ptr(res), H.ptr(z), H.ptr(scale), H.ptr(shift), rows, C, rpg, int(relu), H.dt(y),
                                    st), "fcmf_bn_apply")
    else:
        H.check(rc, "fcmf_bn_finalize_apply")
    if save is not None:
        save.update(mean=mean, rstd=rstd, groups=groups, training=training, rpg=rpg)
    return z


def flush_batch_counters(module):
    """apply the pending `num_batches_tracked` increments of every BatchNorm2d under `module`"""
    for m in module.modules():
        n = getattr(m, "_pending_batches", 0)
        if n and isinstance(m, BatchNorm2d):
            m.num_batches_tracked += n
            m._pending_batches = 0


def maxpool3x3s2_nhwc(x):
    N, Hh, Ww, C = x.shape
    Ho, Wo = (Hh - 1) // 2 + 1, (Ww - 1) // 2 + 1
    y = torch.empty((N, Ho, Wo, C), dtype=x.dtype, device=x.device)
    H.check(H.lib().fcmf_maxpool3x3s2(H.ptr(x), H.ptr(y), N, Hh, Ww, C, H.dt(x), H.stream()), "fcmf_maxpool3x3s2")
    return y


def adaptive_avgpool_nhwc(x, oh, ow, tokens=False):
    """float32 [N, C, oh, ow] (tokens=False: the reference's layout) or [N, oh*ow, C] (tokens=True)"""
    N, Hh, Ww, C = x.shape
    y = torch.empty((N, oh * ow, C) if tokens else (N, C, oh, ow), dtype=torch.float32, device=x.device)
    H.check(H.lib().fcmf_adaptive_avgpool(H.ptr(x), H.ptr(y), N, Hh, Ww, C, oh, ow, int(tokens), H.dt(x), H.stream()),
            "fcmf_adaptive_avgpool")
    return y


# ---------------------------------------------------------------------------------------
# backward building blocks (fine-tuning the CNN)
# ---------------------------------------------------------------------------------------
def _grad_of(grads, p):
    g = grads.get(p)
    if g is None:
        g = grads[p] = torch.zeros(p.shape, dtype=torch.float32, device=p.device)
    return g


def conv2d_bwd_nhwc(conv, x, dy, grads, need_dx=True, src_strides=None, add=None):
    """dy [N,Ho,Wo,Cout] -> dx [N,H,W,C] (+ `add`, the gradient arriving on a parallel branch); the weight gradient
    dW = dY^T A is accumulated into grads[conv.weight].  The patch matrix A is rebuilt from x, not stored."""
    dt = dy.dtype
    N, Hh, Ww, C = x.shape
    kh, kw = conv.kernel_size
    st, pad = conv.stride[0], conv.padding[0]
    Cout = conv.out_channels
    wm, Kpad = _weight_matrix(conv, dt)
    K = kh * kw * C
    Ho, Wo = dy.shape[1], dy.shape[2]
    rows = N * Ho * Wo
    dy2 = dy.reshape(rows, Cout)
    direct = kh == 1 and kw == 1 and st == 1 and pad == 0 and src_strides is None and x.dtype == dt and x.is_contiguous()
    L, stream = H.lib(), H.stream()
    if direct:
        A = x.view(rows, C)
    else:
        A = torch.empty((rows, Kpad), dtype=dt, device=dy.device)
        sn, sh, sw, sc = src_strides if src_strides is not None else x.stride()
        H.check(L.fcmf_conv_im2col(H.ptr(x), H.dt(x), H.ptr(A), H.dt(A), N, Hh, Ww, C, sn, sh, sw, sc, kh, kw, st, pad, Kpad,
                                   stream), "fcmf_conv_im2col")
    dwm = torch.zeros((Cout, Kpad), dtype=torch.float32, device=dy.device)
    ops.gemm(dy2, A, dwm, Cout, Kpad, rows, Cout, Kpad, Kpad, 1, 1, acc=True)
    _grad_of(grads, conv.weight).add_(dwm[:, :K].reshape(Cout, kh, kw, C).permute(0, 3, 1, 2))
    if not need_dx:
        return None
    if direct:
        dx = torch.empty((rows, C), dtype=dt, device=dy.device)
        if add is not None:
            ops.gemm(dy2, wm, dx, rows, Kpad, Cout, Cout, Kpad, Kpad, 0, 1, aux=add.reshape(rows, C), epi=H.EPI_ADD)
        else:
            ops.gemm(dy2, wm, dx, rows, Kpad, Cout, Cout, Kpad, Kpad, 0, 1)
        return dx.view(N, Hh, Ww, C)
    dA = torch.empty((rows, Kpad), dtype=dt, device=dy.device)
    ops.gemm(dy2, wm, dA, rows, Kpad, Cout, Cout, Kpad, Kpad, 0, 1)
    dx = torch.empty((N, Hh, Ww, C), dtype=dt, device=dy.device)
    H.check(L.fcmf_conv_col2im(H.ptr(dA), H.ptr(dx), N, Hh, Ww, C, kh, kw, st, pad, Kpad, H.dt(dx), stream), "fcmf_conv_col2im")
    if add is not None:
        raise H.HipLibraryError("conv2d_bwd_nhwc: `add` is only fused into 1x1 / stride-1 convolutions")
    return dx


def batchnorm_bwd_nhwc_(bn, rec, g, y, z, grads, want_gres=False):
    """g (gradient wrt the normalised, optionally ReLU'd output z) -> gradient wrt the convolution output y, IN PLACE in
    g; returns (dy, gres) with gres = the ReLU-masked g for the identity branch (a new tensor) when want_gres"""
    N, Hh, Ww, C = y.shape
    L = H.lib()
    groups, rpg = rec["groups"], rec["rpg"]
    ws = torch.empty(L.fcmf_bn_stats_workspace(rpg, groups, C), dtype=torch.float64, device=y.device)
    gres = torch.empty_like(g) if want_gres else None
    H.check(L.fcmf_bn_bwd(H.ptr(g), H.ptr(z), H.ptr(y), H.ptr(rec["mean"]), H.ptr(rec["rstd"]), H.ptr(bn.weight), H.ptr(ws),
                          H.ptr(g), H.ptr(gres), H.ptr(_grad_of(grads, bn.weight)), H.ptr(_grad_of(grads, bn.bias)), rpg, groups, C,
                          int(rec["training"]), H.dt(g), H.stream()), "fcmf_bn_bwd")
    return g, gres


class AvgPoolFn(torch.autograd.Function):
    """adaptive average pool of an NHWC activation with its backward kernel"""

    @staticmethod
    def forward(ctx, x, oh, ow, tokens):
        ctx.cfg = (x.shape, x.dtype, oh, ow, tokens)
        return adaptive_avgpool_nhwc(x, oh, ow, tokens)

    @staticmethod
    def backward(ctx, dy):
        (N, Hh, Ww, C), dt, oh, ow, tokens = ctx.cfg
        dx = torch.empty((N, Hh, Ww, C), dtype=dt, device=dy.device)
        d = dy.contiguous().float()
        H.check(H.lib().fcmf_adaptive_avgpool_bwd(H.ptr(d), H.ptr(dx), N, Hh, Ww, C, oh, ow, int(tokens), H.dt(dx), H.stream()),
                "fcmf_adaptive_avgpool_bwd")
        return dx, None, None, None


class TrunkFn(torch.autograd.Function):
    """conv1 -> bn1 -> relu -> maxpool -> layer1..4 of ONE pass (<= MAX_CROPS_PER_PASS crops) with a hand-written
    backward.  Inputs: the crops, the network, the group count, then every trunk parameter (so that autograd routes the
    gradients); output: the NHWC activation [N, h, w, 2048]."""

    @staticmethod
    def forward(ctx, x, net, groups, *params):
        tape = []
        y = net._trunk_pass(x, groups, tape)
        ctx.net, ctx.tape, ctx.nparams = net, tape, len(params)
        ctx.params = params
        return y

    @staticmethod
    def backward(ctx, g):
        grads = {}
        ctx.net._trunk_backward(ctx.tape, g.contiguous(), grads)
        ctx.tape = None
        return (None, None, None) + tuple(grads.get(p) for p in ctx.params)


# ---------------------------------------------------------------------------------------
# module tree (torchvision names); every forward takes / returns NCHW-shaped tensors
# ---------------------------------------------------------------------------------------
class Conv2d(nn.Conv2d):
    def forward(self, x):
        if x.dim() == 4 and not x.permute(0, 2, 3, 1).is_contiguous():
            # e.g. the float32 NCHW crops entering the stem: gathered straight from their layout
            xs = x if x.dtype in (torch.float32, torch.bfloat16) else x.float()
            H.require_cuda(xs)
            N, C, Hh, Ww = xs.shape
            v = xs.permute(0, 2, 3, 1)
            return _nchw(conv2d_nhwc(v, self, src_strides=v.stride()))
        return _nchw(conv2d_nhwc(_nhwc(x), self))


class BatchNorm2d(nn.BatchNorm2d):
    groups = 1     # number of reference calls packed into the batch (set by ResNet.trunk)

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self._pending_batches = 0
        self._register_state_dict_hook(BatchNorm2d._flush_hook)

    @staticmethod
    def _flush_hook(module, state_dict, prefix, local_metadata):
        n = module._pending_batches
        if n and module.num_batches_tracked is not None:
            module.num_batches_tracked += n
            module._pending_batches = 0
            state_dict[prefix + "num_batches_tracked"] = module.num_batches_tracked

    def forward(self, x):
        v = _nhwc(x)
        v = v.clone() if v.data_ptr() == x.data_ptr() else v        # module API: do not overwrite the caller's tensor
        out = _nchw(batchnorm_nhwc_(v, self, self.groups))
        flush_batch_counters(self)
        return out


class ReLU(nn.Module):
    def __init__(self, inplace=True):
        super().__init__()
        self.inplace = inplace

    def forward(self, x):
        v = _nhwc(x)
        N, Hh, Ww, C = v.shape
        y = v if (self.inplace and v.data_ptr() == x.data_ptr()) else torch.empty_like(v)
        one = torch.ones(C, dtype=torch.float32, device=v.device)
        H.check(H.lib().fcmf_bn_apply(H.ptr(v), 0, H.ptr(y), H.ptr(one), H.ptr(torch.zeros_like(one)), N * Hh * Ww, C,
                                      N * Hh * Ww, 1, H.dt(v), H.stream()), "fcmf_bn_apply")
        return _nchw(y)


class MaxPool2d(nn.Module):
    """nn.MaxPool2d(kernel_size=3, stride=2, padding=1) -- the only pooling the trunk uses"""
    kernel_size, stride, padding = 3, 2, 1

    def forward(self, x):
        return _nchw(maxpool3x3s2_nhwc(_nhwc(x)))


class AdaptiveAvgPool2d(nn.Module):
    def __init__(self, output_size=(1, 1)):
        super().__init__()
        self.output_size = (output_size, output_size) if isinstance(output_size, int) else tuple(output_size)

    def forward(self, x):
        return adaptive_avgpool_nhwc(_nhwc(x), *self.output_size)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)      # v1.5: stride on the 3x3
        self.bn2 = BatchNorm2d(planes)
        self.conv3 = Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = BatchNorm2d(planes * 4)
        self.relu = ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward_nhwc(self, x, groups):
        # (every convolution hands the statistics of its output to the BatchNorm that follows it: stats=True)
        y1, b1 = conv2d_nhwc(x, self.conv1, stats=True)
        N, Hh, Ww, C = y1.shape
        if _implicit_ok(C, y1.dtype) and not torch.is_grad_enabled():
            # bn1 + relu write straight into the zero-bordered input of the 3x3 convolution, which then runs as an implicit GEMM
            zp = padded_activation(N, Hh, Ww, C, y1.dtype, y1.device)
            batchnorm_nhwc_(y1, self.bn1, groups, relu=True, out=zp, out_pad=1, blocks=b1)
            st = self.conv2.stride[0]
            out, b2 = conv2d_implicit(zp, self.conv2, N, (Hh - 1) // st + 1, (Ww - 1) // st + 1, stats=True)
        else:
            out, b2 = conv2d_nhwc(batchnorm_nhwc_(y1, self.bn1, groups, relu=True, blocks=b1), self.conv2, stats=True)
        out = batchnorm_nhwc_(out, self.bn2, groups, relu=True, blocks=b2)
        out, b3 = conv2d_nhwc(out, self.conv3, stats=True)
        if self.downsample is not None:
            yd, bd = conv2d_nhwc(x, self.downsample[0], stats=True)
            x = batchnorm_nhwc_(yd, self.downsample[1], groups, blocks=bd)
        return batchnorm_nhwc_(out, self.bn3, groups, res=x, relu=True, blocks=b3)       # bn3 -> += identity -> relu, one pass

    def forward_rec(self, x, groups, tape):
        """forward_nhwc that keeps what the backward needs (raw convolution outputs are NOT overwritten)"""
        r = {"x": x, "blk": self}
        for i, (conv, bn, src) in enumerate(((self.conv1, self.bn1, "x"), (self.conv2, self.bn2, "z1"), (self.conv3, self.bn3, "z2")), 1):
            y, blk = conv2d_nhwc(r[src], conv, stats=True)
            r[f"y{i}"], r[f"s{i}"] = y, {}
            if i < 3:
                r[f"z{i}"] = batchnorm_nhwc_(y, bn, groups, relu=True, out=torch.empty_like(y), save=r[f"s{i}"], blocks=blk)
        idn = x
        if self.downsample is not None:
            (r["yd"], bd), r["sd"] = conv2d_nhwc(x, self.downsample[0], stats=True), {}
            idn = batchnorm_nhwc_(r["yd"], self.downsample[1], groups, out=torch.empty_like(r["yd"]), save=r["sd"], blocks=bd)
        r["z3"] = batchnorm_nhwc_(r["y3"], self.bn3, groups, res=idn, relu=True, out=torch.empty_like(r["y3"]), save=r["s3"], blocks=blk)
        tape.append(r)
        return r["z3"]

    def backward_rec(self, r, g, grads):
        """g: gradient wrt the block output -> gradient wrt the block input (parameter gradients into `grads`)"""
        dy3, gres = batchnorm_bwd_nhwc_(self.bn3, r["s3"], g, r["y3"], r["z3"], grads, want_gres=True)
        dz2 = conv2d_bwd_nhwc(self.conv3, r["z2"], dy3, grads)
        dy2, _ = batchnorm_bwd_nhwc_(self.bn2, r["s2"], dz2, r["y2"], r["z2"], grads)
        dz1 = conv2d_bwd_nhwc(self.conv2, r["z1"], dy2, grads)
        dy1, _ = batchnorm_bwd_nhwc_(self.bn1, r["s1"], dz1, r["y1"], r["z1"], grads)
        if self.downsample is not None:
            dyd, _ = batchnorm_bwd_nhwc_(self.downsample[1], r["sd"], gres, r["yd"], None, grads)
            conv_d = self.downsample[0]
            if conv_d.stride[0] == 1:        # layer1.0: both branches are 1x1 / stride 1 -> the second GEMM adds the first
                dxd = conv2d_bwd_nhwc(conv_d, r["x"], dyd, grads)
                return conv2d_bwd_nhwc(self.conv1, r["x"], dy1, grads, add=dxd)
            gres = conv2d_bwd_nhwc(conv_d, r["x"], dyd, grads)      # strided: col2im output, added by conv1's GEMM below
        return conv2d_bwd_nhwc(self.conv1, r["x"], dy1, grads, add=gres)

    def forward(self, x):
        return _nchw(self.forward_nhwc(_nhwc(x), self.bn1.groups))


class ResNet(nn.Module):
    def __init__(self, layers=(3, 8, 36, 3), num_classes=1000, base=64):
        super().__init__()
        self.inplanes = base
        self.conv1 = Conv2d(3, base, 7, stride=2, padding=3, bias=False)
        self.bn1 = BatchNorm2d(base)
        self.relu = ReLU(inplace=True)
        self.maxpool = MaxPool2d()
        self.layer1 = self._make_layer(base, layers[0], 1)
        self.layer2 = self._make_layer(base * 2, layers[1], 2)
        self.layer3 = self._make_layer(base * 4, layers[2], 2)
        self.layer4 = self._make_layer(base * 8, layers[3], 2)
        self.avgpool = AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(base * 8 * 4, num_classes)
        for m in self.modules():                       # torchvision's initialisation
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _make_layer(self, planes, blocks, stride):
        down = None
        if stride != 1 or self.inplanes != planes * 4:
            down = nn.Sequential(Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False), BatchNorm2d(planes * 4))
        mods = [Bottleneck(self.inplanes, planes, stride, down)]
        self.inplanes = planes * 4
        mods += [Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*mods)

    def set_groups(self, groups):
        for m in self.modules():
            if isinstance(m, BatchNorm2d):
                m.groups = groups

    def trunk_params(self):
        return [p for n, p in self.named_parameters() if not n.startswith("fc.")]

    def trunk_nhwc(self, x, groups=1, fine_tune=False):
        """x [N,3,H,W] (any float dtype / layout), N = groups * B crops packed group-major -> NHWC [N,h,w,C].
        fine_tune=False: forward only, detached (the reference's `Variable(x.data)`, resnet_utils.py:26-28);
        fine_tune=True under grad mode: differentiable wrt the trunk parameters (TrunkFn)."""
        H.require_cuda(x)
        grad = fine_tune and torch.is_grad_enabled() and any(p.requires_grad for p in self.trunk_params())
        if x.shape[0] % groups != 0:
            raise H.HipLibraryError(f"{x.shape[0]} crops do not split into {groups} equal groups")
        per = x.shape[0] // groups
        if per > MAX_CROPS_PER_PASS:
            raise H.HipLibraryError(f"more than {MAX_CROPS_PER_PASS} crops per BatchNorm group")
        gpp = max(1, MAX_CROPS_PER_PASS // per)           # whole groups per pass
        outs = []
        for g0 in range(0, groups, gpp):
            g1 = min(groups, g0 + gpp)
            xs = x[g0 * per:g1 * per]
            if grad:
                outs.append(TrunkFn.apply(xs, self, g1 - g0, *self.trunk_params()))
            else:
                with torch.no_grad():
                    outs.append(self._trunk_pass(xs, g1 - g0))
        return outs[0] if len(outs) == 1 else torch.cat(outs, 0)

    def _trunk_pass(self, x, groups, tape=None):
        """tape is None: activations are normalised in place, nothing is kept; else: recording forward for TrunkFn"""
        xs = x if x.dtype in (torch.float32, torch.bfloat16) else x.float()
        v = xs.permute(0, 2, 3, 1)                                       # strided NHWC view of the NCHW crops
        if tape is None and _stem_runs_ok(self.conv1, v, ops.compute_dtype()):
            y, blk = conv2d_stem(v, self.conv1, stats=True)
        else:
            y, blk = conv2d_nhwc(v, self.conv1, src_strides=v.stride(), stats=True)
        if tape is None:
            y = batchnorm_nhwc_(y, self.bn1, groups, relu=True, blocks=blk)
            y = maxpool3x3s2_nhwc(y)
            for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
                for blk in layer:
                    y = blk.forward_nhwc(y, groups)
            return y
        stem = {"v": v, "y": y, "s": {}}
        stem["z"] = batchnorm_nhwc_(y, self.bn1, groups, relu=True, out=torch.empty_like(y), save=stem["s"], blocks=blk)
        tape.append(stem)
        y = maxpool3x3s2_nhwc(stem["z"])
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                y = blk.forward_rec(y, groups, tape)
        return y

    def _trunk_backward(self, tape, g, grads):
        for r in reversed(tape[1:]):
            g = r["blk"].backward_rec(r, g, grads)
        stem = tape[0]
        z = stem["z"]
        N, Hh, Ww, C = z.shape
        dz = torch.empty_like(z)
        H.check(H.lib().fcmf_maxpool3x3s2_bwd(H.ptr(z), H.ptr(g), H.ptr(dz), N, Hh, Ww, C, H.dt(z), H.stream()), "fcmf_maxpool3x3s2_bwd")
        dy, _ = batchnorm_bwd_nhwc_(self.bn1, stem["s"], dz, stem["y"], z, grads)
        conv2d_bwd_nhwc(self.conv1, stem["v"], dy, grads, need_dx=False, src_strides=stem["v"].stride())

    def forward(self, x):
        """torchvision's ResNet.forward (classification head); not on the FCMF path, kept for API parity"""
        f = adaptive_avgpool_nhwc(self.trunk_nhwc(x), 1, 1).flatten(1)
        return ops.linear(ops.cast(f, ops.compute_dtype()), self.fc.weight, self.fc.bias)


def resnet152(weights=None, **kw):
    """torchvision.models.resnet152 stand-in (run_multimodal_fcmf.py:224-225).  `weights` must be None or a
    state dict: pretrained ImageNet weights need a download, which this environment cannot do -- load a local
    torchvision checkpoint with `load_state_dict` instead."""
    m = ResNet((3, 8, 36, 3), **kw)
    if weights is not None:
        if not isinstance(weights, dict):
            raise H.HipLibraryError("resnet152(weights=...): pass a state dict loaded from a local torchvision checkpoint")
        m.load_state_dict(weights)
    return m


def from_module(module):
    """our ResNet holding the parameters and BatchNorm buffers of any module with torchvision's ResNet state-dict
    keys (e.g. a torchvision resnet152 instance handed to myResNetImg by the reference's driver)"""
    if isinstance(module, ResNet):
        return module
    sd = module.state_dict()
    layers = []
    for li in range(1, 5):
        layers.append(len({k.split(".")[1] for k in sd if k.startswith(f"layer{li}.")}))
    base = sd["conv1.weight"].shape[0]
    m = ResNet(tuple(layers), num_classes=sd["fc.weight"].shape[0] if "fc.weight" in sd else 1000, base=base)
    m.load_state_dict(sd, strict="fc.weight" in sd)
    dev = sd["conv1.weight"].device
    m.train(module.training)
    return m.to(dev)

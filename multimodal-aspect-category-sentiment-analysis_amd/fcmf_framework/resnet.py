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
  * forward only: with if_fine_tune=False (the default of both drivers) the reference detaches the features
    (resnet_utils.py:26-28); --fine_tune_cnn needs convolution / BatchNorm backward kernels that do not exist yet,
    and asking for it raises instead of silently running something else.
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


def conv2d_nhwc(x, conv, src_strides=None):
    """x [N,H,W,C] (or any layout with `src_strides` = element strides of (n,h,w,c)) -> [N,Ho,Wo,Cout]"""
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
    else:
        A = torch.empty((rows, Kpad), dtype=dt, device=x.device)
        sn, sh, sw, sc = src_strides if src_strides is not None else x.stride()
        H.check(H.lib().fcmf_conv_im2col(H.ptr(x), H.dt(x), H.ptr(A), H.dt(A), N, Hh, Ww, C, sn, sh, sw, sc, kh, kw, st,
                                         pad, Kpad, H.stream()), "fcmf_conv_im2col")
    y = torch.empty((rows, Cout), dtype=dt, device=x.device)
    ops.gemm(A, wm, y, rows, Cout, Kpad, Kpad, Kpad, Cout, 0, 0)
    return y.view(N, Ho, Wo, Cout)


def batchnorm_nhwc_(y, bn, groups=1, res=None, relu=False):
    """in-place BatchNorm2d (+ residual, + ReLU) of y [N,H,W,C]; training mode: per-group batch statistics and
    `groups` running-statistics updates (module docstring)"""
    N, Hh, Ww, C = y.shape
    rows = N * Hh * Ww
    L, st = H.lib(), H.stream()
    dev = y.device
    training = bn.training or not bn.track_running_stats
    if training:
        if N % groups != 0:
            raise H.HipLibraryError(f"grouped BatchNorm: {N} crops do not split into {groups} equal groups")
        rpg = rows // groups
        sums = torch.empty((groups, C, 2), dtype=torch.float64, device=dev)
        scale = torch.empty((groups, C), dtype=torch.float32, device=dev)
        shift = torch.empty((groups, C), dtype=torch.float32, device=dev)
        H.check(L.fcmf_bn_stats(H.ptr(y), H.ptr(sums), rpg, groups, C, H.dt(y), st), "fcmf_bn_stats")
        mom = 0.1 if bn.momentum is None else float(bn.momentum)
        if bn.track_running_stats:
            rm, rv = bn.running_mean, bn.running_var
        else:
            rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        H.check(L.fcmf_bn_finalize(H.ptr(sums), H.ptr(bn.weight), H.ptr(bn.bias), H.ptr(rm), H.ptr(rv), H.ptr(scale),
                                   H.ptr(shift), C, groups, rpg, mom, float(bn.eps), st), "fcmf_bn_finalize")
        if bn.track_running_stats and bn.num_batches_tracked is not None:
            bn.num_batches_tracked += groups
    else:
        rpg = rows
        scale = torch.empty(C, dtype=torch.float32, device=dev)
        shift = torch.empty(C, dtype=torch.float32, device=dev)
        H.check(L.fcmf_bn_finalize(0, H.ptr(bn.weight), H.ptr(bn.bias), H.ptr(bn.running_mean), H.ptr(bn.running_var),
                                   H.ptr(scale), H.ptr(shift), C, 1, 0, 0.0, float(bn.eps), st), "fcmf_bn_finalize")
    H.check(L.fcmf_bn_apply(H.ptr(y), H.ptr(res), H.ptr(y), H.ptr(scale), H.ptr(shift), rows, C, rpg, int(relu), H.dt(y),
                            st), "fcmf_bn_apply")
    return y


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

    def forward(self, x):
        v = _nhwc(x)
        v = v.clone() if v.data_ptr() == x.data_ptr() else v        # module API: do not overwrite the caller's tensor
        return _nchw(batchnorm_nhwc_(v, self, self.groups))


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
        out = batchnorm_nhwc_(conv2d_nhwc(x, self.conv1), self.bn1, groups, relu=True)
        out = batchnorm_nhwc_(conv2d_nhwc(out, self.conv2), self.bn2, groups, relu=True)
        out = conv2d_nhwc(out, self.conv3)
        if self.downsample is not None:
            x = batchnorm_nhwc_(conv2d_nhwc(x, self.downsample[0]), self.downsample[1], groups)
        return batchnorm_nhwc_(out, self.bn3, groups, res=x, relu=True)       # bn3 -> += identity -> relu, one pass

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

    @torch.no_grad()
    def trunk_nhwc(self, x, groups=1):
        """x [N,3,H,W] (any float dtype / layout), N = groups * B crops packed group-major -> NHWC [N,h,w,C]"""
        H.require_cuda(x)
        if x.shape[0] % groups != 0:
            raise H.HipLibraryError(f"{x.shape[0]} crops do not split into {groups} equal groups")
        per = x.shape[0] // groups
        if per > MAX_CROPS_PER_PASS:
            raise H.HipLibraryError(f"more than {MAX_CROPS_PER_PASS} crops per BatchNorm group")
        gpp = max(1, MAX_CROPS_PER_PASS // per)           # whole groups per pass
        outs = []
        for g0 in range(0, groups, gpp):
            g1 = min(groups, g0 + gpp)
            outs.append(self._trunk_pass(x[g0 * per:g1 * per], g1 - g0))
        return outs[0] if len(outs) == 1 else torch.cat(outs, 0)

    def _trunk_pass(self, x, groups):
        xs = x if x.dtype in (torch.float32, torch.bfloat16) else x.float()
        v = xs.permute(0, 2, 3, 1)                                       # strided NHWC view of the NCHW crops
        y = conv2d_nhwc(v, self.conv1, src_strides=v.stride())
        y = batchnorm_nhwc_(y, self.bn1, groups, relu=True)
        y = maxpool3x3s2_nhwc(y)
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                y = blk.forward_nhwc(y, groups)
        return y

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

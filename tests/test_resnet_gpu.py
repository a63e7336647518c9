"""ResNet-152 trunk (reference resnet_utils.py:13-56 + torchvision resnet152) on the HIP kernels, through the C ABI.

Oracle = oracle/resnet_oracle.py, a torch-CPU restatement on F.conv2d / F.batch_norm / F.max_pool2d.  PARITY UNPINNED
vs torchvision (not installed, no reference fixture exists): what is checked is the build against that restatement.
Tolerances: fp32 mode 1e-3 relative to the output scale (north-star bound), bf16 mode 5e-2.
"""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import max_err
from oracle import resnet_oracle as RO
import synthetic_data as synth

pytestmark = pytest.mark.gpu


def _set(dtype):
    from fcmf_framework import ops
    ops.set_compute_dtype(dtype)
    ops.shadows.clear()


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def _build(layers, dev, seed=0):
    from fcmf_framework.resnet import ResNet
    shapes = synth.resnet_param_shapes(layers)
    P = synth.synth_resnet_params(shapes, seed)
    m = ResNet(layers)
    missing, unexpected = m.load_state_dict(P, strict=False)
    assert not unexpected and all(k.startswith("fc.") for k in missing), (missing, unexpected)
    return m.to(dev), P


@pytest.mark.parametrize("cin,cout,k,stride,pad,hw", [(3, 64, 7, 2, 3, 37), (64, 64, 3, 1, 1, 14), (128, 128, 3, 2, 1, 15),
                                                      (64, 256, 1, 1, 0, 9), (256, 512, 1, 2, 0, 14)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_matches_torch(dev, cin, cout, k, stride, pad, hw, dtype):
    from fcmf_framework import resnet as R
    _set(dtype)
    try:
        conv = R.Conv2d(cin, cout, k, stride=stride, padding=pad, bias=False)
        conv.weight.data = _rand(conv.weight.shape, 1, (2.0 / (cin * k * k)) ** 0.5)
        x = _rand((3, cin, hw, hw + 2), 2)
        ref = F.conv2d(x, conv.weight.data, stride=stride, padding=pad)
        y = conv.to(dev)(x.to(dev))                        # NCHW in, NCHW-shaped out (module API)
        assert y.shape == ref.shape
        tol = 1e-4 if dtype == torch.float32 else 3e-2
        assert max_err(y, ref) < tol * ref.abs().max().item()
        # channels_last input takes the in-place (no gather) path for 1x1 / stride 1
        y2 = conv(x.to(dev).contiguous(memory_format=torch.channels_last))
        assert max_err(y2, ref) < tol * ref.abs().max().item()
    finally:
        _set(torch.float32)


@pytest.mark.parametrize("C,groups", [(64, 1), (128, 3), (256, 2), (1024, 2)])
def test_grouped_batchnorm_train_and_eval(dev, C, groups):
    """training: per-group batch statistics + `groups` sequential running-statistics updates == `groups` separate
    nn.BatchNorm2d calls; eval: the running statistics"""
    from fcmf_framework import resnet as R
    _set(torch.float32)
    B, hw = 2, 5
    x = _rand((groups * B, C, hw, hw), 3) * 1.7 + 0.3
    bn = R.BatchNorm2d(C)
    bn.weight.data, bn.bias.data = 1 + 0.1 * _rand((C,), 4), 0.1 * _rand((C,), 5)
    bn.running_mean.data, bn.running_var.data = 0.1 * _rand((C,), 6), 0.5 + torch.rand(C, generator=torch.Generator().manual_seed(7))
    ref = copy.deepcopy(bn)
    ref.__class__ = torch.nn.BatchNorm2d
    bn = bn.to(dev)
    bn.groups = groups
    outs = [torch.nn.BatchNorm2d.forward(ref, xg) for xg in x.chunk(groups, 0)]
    y = bn(x.to(dev))
    assert max_err(y, torch.cat(outs, 0)) < 2e-5
    assert max_err(bn.running_mean, ref.running_mean) < 1e-6
    assert max_err(bn.running_var, ref.running_var) < 1e-5
    assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked) == groups
    bn.eval(); ref.eval()
    assert max_err(bn(x.to(dev)), torch.nn.BatchNorm2d.forward(ref, x)) < 2e-5


def test_pools_match_torch(dev):
    from fcmf_framework import resnet as R
    _set(torch.float32)
    x = _rand((2, 64, 13, 16), 8)
    assert max_err(R.MaxPool2d()(x.to(dev)), F.max_pool2d(x, 3, 2, 1)) == 0.0
    for size in ((7, 7), (1, 1), (3, 5)):
        assert max_err(R.AdaptiveAvgPool2d(size)(x.to(dev)), F.adaptive_avg_pool2d(x, size)) < 1e-6
    tok = R.adaptive_avgpool_nhwc(R._nhwc(x.to(dev)), 7, 7, tokens=True)
    assert max_err(tok, F.adaptive_avg_pool2d(x, (7, 7)).view(2, 64, 49).permute(0, 2, 1)) < 1e-6


@pytest.mark.parametrize("training", [False, True])
def test_small_trunk_matches_oracle_fp32(dev, training):
    """a (1,2,2,1)-block trunk of the same Bottleneck type on 64x64 crops, 3 call groups of 2 crops:
    myResNetImg / myResNetRoI outputs and (training) the BatchNorm buffers against the oracle"""
    from fcmf_framework.resnet_utils import myResNetImg, myResNetRoI
    _set(torch.float32)
    layers, groups, B = synth.RESNET_TINY_LAYERS, 3, 2
    m, P = _build(layers, dev)
    x = synth.synth_crops(groups * B, 64, seed=1)
    Po = {k: v.clone() for k, v in P.items()}
    o_img = RO.my_resnet_img(Po, x, layers, 2, training=training, groups=groups)
    Po2 = {k: v.clone() for k, v in P.items()}
    o_roi = RO.my_resnet_roi(Po2, x, layers, training=training, groups=groups)
    img, roi = myResNetImg(m, False, dev), myResNetRoI(copy.deepcopy(m), False, dev)
    img.train(training); roi.train(training)
    y_img = img.forward_groups(x.to(dev), groups, att_size=2)
    y_roi = roi.forward_groups(x.to(dev), groups)
    assert y_img.shape == o_img.shape and y_roi.shape == o_roi.shape
    assert not y_img.requires_grad
    assert max_err(y_img, o_img) < 1e-3 * o_img.abs().max().item()
    assert max_err(y_roi, o_roi) < 1e-3 * o_roi.abs().max().item()
    sd = img.resnet.state_dict()
    for k in ("bn1", "layer2.1.bn2", "layer4.0.downsample.1", "layer3.0.bn3"):
        tol = 1e-4 if training else 0.0
        assert max_err(sd[k + ".running_mean"], Po[k + ".running_mean"]) <= tol * (1 + Po[k + ".running_mean"].abs().max().item())
        assert max_err(sd[k + ".running_var"], Po[k + ".running_var"]) <= tol * (1 + Po[k + ".running_var"].abs().max().item())
        assert int(sd[k + ".num_batches_tracked"]) == int(Po[k + ".num_batches_tracked"])
    if training:
        # one forward() call per group (the reference's loop) == the batched pass
        m2, _ = _build(layers, dev)
        img2 = myResNetImg(m2, False, dev).train()
        ys = torch.cat([img2(xg.to(dev), att_size=2) for xg in x.chunk(groups, 0)], 0)
        assert max_err(ys, y_img) < 1e-5 * o_img.abs().max().item()


def test_resnet152_full_depth_eval_and_bf16(dev):
    """the real [3, 8, 36, 3] trunk on 224x224 crops: fp32 within 1e-3 of the oracle, bf16 (MFMA path) within 5e-2;
    output layouts of the reference ([B,2048,7,7] -> view(-1,2048,49).permute(0,2,1)) and of the batched driver"""
    from fcmf_framework.resnet import resnet152
    from fcmf_framework.resnet_utils import myResNetImg
    layers = synth.RESNET152_LAYERS
    P = synth.synth_resnet_params(synth.resnet_param_shapes(layers))
    x = synth.synth_crops(2, 224, seed=3)
    ref = RO.my_resnet_img({k: v.clone() for k, v in P.items()}, x, layers, 7)
    m = resnet152()
    m.load_state_dict(P, strict=False)
    img = myResNetImg(m.to(dev), False, dev).eval()
    scale = ref.abs().max().item()
    try:
        _set(torch.float32)
        y = img(x.to(dev))
        assert y.shape == (2, 2048, 7, 7)
        assert max_err(y, ref) < 1e-3 * scale, max_err(y, ref) / scale
        tok = img.forward_groups(x.to(dev), 1, 7, tokens=True)
        assert torch.equal(tok, y.view(-1, 2048, 49).permute(0, 2, 1))
        _set(torch.bfloat16)
        yb = img(x.to(dev))
        rel = (yb.cpu() - ref).norm().item() / ref.norm().item()
        assert rel < 5e-2, rel
    finally:
        _set(torch.float32)


def test_extract_features_matches_reference_loop_order(dev):
    """run_multimodal_fcmf.py:449-460: per-image and per-(image, ROI) calls in train() mode == two batched passes"""
    from fcmf_framework.resnet_utils import extract_features, myResNetImg, myResNetRoI
    _set(torch.float32)
    layers, B, NI, NR = synth.RESNET_TINY_LAYERS, 2, 2, 3
    m, P = _build(layers, dev)
    t_img = synth.synth_crops(B * NI, 224, seed=5).view(B, NI, 3, 224, 224)[..., :64, :64].contiguous()
    roi_img = synth.synth_crops(B * NI * NR, 64, seed=6).view(B, NI, NR, 3, 64, 64)
    Pi, Pr = {k: v.clone() for k, v in P.items()}, {k: v.clone() for k, v in P.items()}
    # the reference's loops, on the oracle (one call per group, BatchNorm buffers carried from call to call)
    vis = torch.stack([RO.my_resnet_img(Pi, t_img[:, i], layers, 7, training=True).view(-1, 2048, 49).permute(0, 2, 1)
                       for i in range(NI)], 1)
    roi = torch.stack([torch.stack([RO.my_resnet_roi(Pr, roi_img[:, i, r], layers, training=True) for r in range(NR)], 1)
                       for i in range(NI)], 1)
    ri, rr = myResNetImg(m, False, dev).train(), myResNetRoI(copy.deepcopy(m), False, dev).train()
    v, r = extract_features(ri, rr, t_img.to(dev), roi_img.to(dev).double())      # ROI crops arrive as float64
    assert v.shape == (B, NI, 49, 2048) and r.shape == (B, NI, NR, 2048)
    assert max_err(v, vis) < 1e-3 * vis.abs().max().item()
    assert max_err(r, roi) < 1e-3 * roi.abs().max().item()
    assert max_err(rr.resnet.bn1.running_var, Pr["bn1.running_var"]) < 1e-4


@pytest.mark.parametrize("training", [True, False])
def test_fine_tune_cnn_gradients_match_oracle_autograd(dev, training):
    """if_fine_tune=True (--fine_tune_cnn): the hand-written trunk backward (BatchNorm+ReLU backward with per-call-group
    statistics, dX / dW GEMMs, col2im, max-pool and average-pool backward, the bottleneck's two branches) against torch
    autograd through the CPU oracle -- gradient of every convolution weight and BatchNorm gamma / beta"""
    from fcmf_framework.resnet_utils import myResNetImg, myResNetRoI
    _set(torch.float32)
    layers, groups, B = synth.RESNET_TINY_LAYERS, 2, 2
    m, P = _build(layers, dev)
    x = synth.synth_crops(groups * B, 64, seed=11)
    w_img, w_roi = _rand((groups * B, 2048, 2, 2), 21), _rand((groups * B, 2048), 22)
    Po = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in P.items()}
    lo = (RO.my_resnet_img(Po, x, layers, 2, training=training, groups=groups) * w_img).sum()
    Po2 = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in P.items()}
    lo2 = (RO.my_resnet_roi(Po2, x, layers, training=training, groups=groups) * w_roi).sum()
    lo.backward(); lo2.backward()
    import copy
    img, roi = myResNetImg(m, True, dev), myResNetRoI(copy.deepcopy(m), True, dev)
    img.train(training); roi.train(training)
    y = img.forward_groups(x.to(dev), groups, att_size=2)
    assert y.requires_grad
    (y * w_img.to(dev)).sum().backward()
    (roi.forward_groups(x.to(dev), groups) * w_roi.to(dev)).sum().backward()
    for net, ref in ((img.resnet, Po), (roi.resnet, Po2)):
        worst = ("", 0.0)
        for n, p in net.named_parameters():
            if n.startswith("fc."):
                assert p.grad is None
                continue
            r = ref[n].grad
            e = (p.grad.cpu() - r).abs().max().item() / (r.abs().max().item() + 1e-12)
            if e > worst[1]:
                worst = (n, e)
        assert worst[1] < 2e-3, worst
    # the default (if_fine_tune=False) stays detached even under grad mode
    assert not myResNetImg(m, False, dev)(x.to(dev)).requires_grad


def test_fine_tune_cnn_bf16_runs_and_is_close(dev):
    from fcmf_framework.resnet_utils import myResNetRoI
    layers = synth.RESNET_TINY_LAYERS
    x = synth.synth_crops(8, 128, seed=12).to(dev)      # 8 crops x 4x4 positions behind layer4: sane batch statistics
    g = {}
    for dtype in (torch.float32, torch.bfloat16):
        _set(dtype)
        m, _ = _build(layers, dev)
        roi = myResNetRoI(m, True, dev).train()
        roi.forward_groups(x, 1).square().sum().backward()
        g[dtype] = torch.cat([p.grad.flatten() for n, p in m.named_parameters() if p.grad is not None])
    _set(torch.float32)
    a, b = g[torch.bfloat16], g[torch.float32]
    assert torch.isfinite(a).all()
    cos = (a @ b / (a.norm() * b.norm())).item()
    # gradients through 22 train-mode BatchNorm backward passes with bf16 activations (each subtracts two batch means from
    # bf16-rounded gradients): the direction is preserved (measured cos 0.963), the norm to 4 digits
    assert cos > 0.93 and 0.9 < (a.norm() / b.norm()).item() < 1.1, (cos, (a.norm() / b.norm()).item())


def test_cpu_tensors_are_loud(dev):
    from fcmf_framework._hip import HipLibraryError
    from fcmf_framework.resnet_utils import myResNetImg
    m, _ = _build(synth.RESNET_TINY_LAYERS, dev)
    with pytest.raises(HipLibraryError):
        myResNetImg(m, False, dev)(synth.synth_crops(1, 64))          # CPU tensor: no fallback


@pytest.mark.parametrize("C,Cout,k,stride,N,hw", [(64, 64, 3, 1, 5, 14), (128, 128, 3, 2, 3, 28), (256, 256, 3, 1, 70, 14), (512, 512, 3, 2, 9, 14),
                                                  (256, 512, 1, 2, 4, 56), (1024, 2048, 1, 2, 3, 14), (64, 64, 3, 1, 2, 56),
                                                  (64, 64, 3, 1, 3, 56), (128, 128, 3, 2, 12, 56)])
def test_implicit_gemm_convolution_matches_patch_matrix_and_torch(dev, C, Cout, k, stride, N, hw):
    """fcmf_conv_gemm (no patch matrix: the GEMM's LDS-DMA walks the taps of every receptive field in the zero-bordered NHWC
    activation) against (1) the explicit fcmf_conv_im2col + fcmf_gemm path on the same bf16 data -- same products, f32
    accumulation in a different k order at most -- and (2) F.conv2d in float32; the 128x128 kernel (Cout < 256, few rows), the
    narrow-output layouts of the persistent kernel (Cout <= 128 with >= 8192 output pixels: the last two cases) and the
    persistent 256 / 192-row kernels (Cout >= 256, M x N large), stride 1 and 2, ragged row counts, 3x3 (padded input
    written by fcmf_bn_apply_pad) and the strided 1x1 shortcut (unpadded input)."""
    from fcmf_framework import _hip as H, ops, resnet as R
    _set(torch.bfloat16)
    try:
        pad = 1 if k == 3 else 0
        conv = R.Conv2d(C, Cout, k, stride=stride, padding=pad, bias=False).to(dev)
        conv.weight.data = _rand(conv.weight.shape, 1, (2.0 / (C * k * k)) ** 0.5).to(dev)
        x = _rand((N, hw, hw + 2, C), 2).to(dev).bfloat16()                      # NHWC
        Hh, Ww = hw, hw + 2
        Ho, Wo = (Hh + 2 * pad - k) // stride + 1, (Ww + 2 * pad - k) // stride + 1
        with torch.no_grad():
            R.IMPLICIT_CONV = False
            y_exp = R.conv2d_nhwc(x, conv)
            R.IMPLICIT_CONV = True
            if k == 3:
                # the producer's path: an identity BatchNorm-apply writes x into the interior of the zero-bordered buffer
                xp = R.padded_activation(N, Hh, Ww, C, x.dtype, dev)
                one, zero = torch.ones(C, device=dev), torch.zeros(C, device=dev)
                H.check(H.lib().fcmf_bn_apply_pad(H.ptr(x), None, H.ptr(xp), H.ptr(one), H.ptr(zero), N * Hh * Ww, C, N * Hh * Ww, 0, Hh, Ww, 1,
                                                  H.dt(x), H.stream()), "fcmf_bn_apply_pad")
                assert torch.equal(xp[:, 1:-1, 1:-1], x) and not xp[:, 0].any() and not xp[:, :, 0].any() and not xp[:, -1].any() and not xp[:, :, -1].any()
                y_imp = R.conv2d_implicit(xp, conv, N, Ho, Wo)
            else:
                y_imp = R.conv2d_nhwc(x, conv)                                   # takes the implicit path by itself
        assert y_imp.shape == y_exp.shape == (N, Ho, Wo, Cout)
        scale = y_exp.float().abs().max().item()
        assert max_err(y_imp, y_exp) <= 2 ** -7 * scale                           # one bf16 ulp of the largest value
        ref = F.conv2d(x.float().permute(0, 3, 1, 2).cpu(), conv.weight.data.float().cpu(), stride=stride, padding=pad).permute(0, 2, 3, 1)
        assert max_err(y_imp, ref) < 3e-2 * ref.abs().max().item()
    finally:
        R.IMPLICIT_CONV = True
        _set(torch.float32)


@pytest.mark.parametrize("C,Cout,k,stride,N,hw,groups", [(64, 256, 1, 1, 8, 16, 2), (256, 256, 3, 1, 6, 16, 3), (512, 1024, 1, 2, 8, 16, 4),
                                                         (128, 512, 1, 1, 3, 20, 1), (256, 1024, 1, 1, 7, 14, 7),
                                                         (64, 64, 3, 1, 3, 56, 3), (256, 128, 1, 1, 12, 28, 3), (64, 256, 1, 1, 3, 56, 1)])
def test_convolution_emits_the_batchnorm_statistics_of_its_output(dev, C, Cout, k, stride, N, hw, groups):
    """fcmf_gemm_colstats / fcmf_conv_gemm_colstats + fcmf_bn_stats_blocks against the convolution followed by the separate
    statistics pass (fcmf_bn_stats): the output tensor is bit-identical; the block statistics are the exact f32 sums of the
    STORED bf16 values over fcmf_gemm_colstats_block_rows rows (128; 256 for the narrow layouts: the last cases) (checked against a float64 sum of the output); the BatchNorm that consumes them gives the
    same result as the two-pass path to bf16 rounding, running statistics included.  Cases: 1x1 plain GEMM, 3x3 implicit, strided
    1x1 implicit, a row count that is no multiple of 128 (ragged last block; the group then falls back to the pass), 7 groups."""
    from fcmf_framework import _hip as H, ops, resnet as R
    _set(torch.bfloat16)
    try:
        pad = 1 if k == 3 else 0
        conv = R.Conv2d(C, Cout, k, stride=stride, padding=pad, bias=False).to(dev)
        conv.weight.data = _rand(conv.weight.shape, 1, (2.0 / (C * k * k)) ** 0.5).to(dev)
        x = (_rand((N, hw, hw, C), 2) + 0.3).to(dev).bfloat16()
        Ho = (hw + 2 * pad - k) // stride + 1
        rows = N * Ho * Ho

        def run(fused):
            R.FUSED_BN_STATS = fused
            bn = R.BatchNorm2d(Cout).to(dev)
            bn.weight.data = _rand((Cout,), 3).to(dev) * 0.1 + 1.0
            bn.bias.data = _rand((Cout,), 4).to(dev) * 0.1
            bn.train()
            with torch.no_grad():
                if k == 3:
                    xp = R.padded_activation(N, hw, hw, C, x.dtype, dev)
                    one, zero = torch.ones(C, device=dev), torch.zeros(C, device=dev)
                    H.check(H.lib().fcmf_bn_apply_pad(H.ptr(x), None, H.ptr(xp), H.ptr(one), H.ptr(zero), N * hw * hw, C, N * hw * hw, 0, hw, hw, 1,
                                                      H.dt(x), H.stream()), "fcmf_bn_apply_pad")
                    y, blocks = R.conv2d_implicit(xp, conv, N, Ho, Ho, stats=True)
                else:
                    y, blocks = R.conv2d_nhwc(x, conv, stats=True)
                kern = H.last_gemm_kernel()
                z = R.batchnorm_nhwc_(y, bn, groups, relu=True, out=torch.empty_like(y), blocks=blocks)
            return y, blocks, z, bn.running_mean.clone(), bn.running_var.clone(), kern

        y0, b0, z0, rm0, rv0, _ = run(False)
        y1, b1, z1, rm1, rv1, kern = run(True)
        assert b0 is None and b1 is not None and "tile256" in kern, kern
        b1, br = b1                                              # (block statistics, rows per block: 128; narrow layouts: 256)
        assert br == (128 if Cout >= 256 else 256)
        assert torch.equal(y0, y1)
        yf = y1.view(rows, Cout).double().cpu()
        nb = (rows + br - 1) // br
        assert b1.shape == (nb, Cout, 2)
        padded = torch.zeros(nb * br, Cout, dtype=torch.float64)
        padded[:rows] = yf
        want_s, want_q = padded.view(nb, br, Cout).sum(1), (padded ** 2).view(nb, br, Cout).sum(1)
        assert max_err(b1[..., 0].cpu().double(), want_s) < 1e-5 * want_s.abs().max().item()
        assert max_err(b1[..., 1].cpu().double(), want_q) < 1e-5 * want_q.abs().max().item()
        assert max_err(z1, z0) <= 2 ** -7 * z0.float().abs().max().item()
        assert max_err(rm1, rm0) < 1e-5 and max_err(rv1, rv0) < 1e-5
    finally:
        R.FUSED_BN_STATS = True
        _set(torch.float32)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,groups,N,hw,pad,with_res,training", [(64, 7, 14, 9, 1, False, True), (256, 2, 4, 7, 0, True, True), (2048, 3, 3, 5, 0, True, True),
                                                                  (512, 1, 5, 6, 1, False, False), (4, 2, 4, 5, 0, False, True)])
def test_batchnorm_finalize_and_apply_in_one_launch(dev, dtype, C, groups, N, hw, pad, with_res, training):
    """fcmf_bn_finalize_apply against fcmf_bn_finalize + fcmf_bn_apply(_pad): the same arithmetic per element, so the normalised
    tensor and mean / rstd are bit-identical, the running statistics equal to one ulp; zero-bordered output, residual + ReLU, eval mode, 7 groups,
    and shapes the fused kernel refuses (bf16 with C = 4: less than one 16-byte vector; float32 with C = 2048: 512 lanes per row) that take the two-kernel path
    by themselves."""
    from fcmf_framework import resnet as R
    _set(dtype)
    try:
        x = (_rand((N, hw, hw, C), 1) * 1.3 + 0.2).to(dev).to(dtype)
        res = (_rand((N, hw, hw, C), 2)).to(dev).to(dtype) if with_res else None
        outs = []
        for fused in (False, True):
            R.FUSED_BN_APPLY = fused
            bn = R.BatchNorm2d(C).to(dev)
            bn.weight.data = _rand((C,), 3).to(dev) * 0.1 + 1.0
            bn.bias.data = _rand((C,), 4).to(dev) * 0.1
            bn.running_mean.data = _rand((C,), 5).to(dev) * 0.1
            bn.running_var.data = torch.rand(C, generator=torch.Generator().manual_seed(6)).to(dev) + 0.5
            bn.train(training)
            save = {}
            with torch.no_grad():
                out = R.padded_activation(N, hw, hw, C, dtype, dev).clone() if pad else torch.empty_like(x)
                z = R.batchnorm_nhwc_(x.clone(), bn, groups if training else 1, res=res, relu=True, out=out, save=save, out_pad=pad)
            outs.append((z, save["mean"], save["rstd"], bn.running_mean.clone(), bn.running_var.clone()))
        for a, b in zip(outs[0][:3], outs[1][:3]):
            assert torch.equal(a, b)
        for a, b in zip(outs[0][3:], outs[1][3:]):       # (running statistics: the two kernels contract a*b + c*d differently -- an ulp per group)
            assert max_err(a, b) <= 1e-6 * max(1.0, b.abs().max().item())
        if pad:
            z = outs[1][0]
            assert not z[:, 0].any() and not z[:, -1].any() and not z[:, :, 0].any() and not z[:, :, -1].any()
    finally:
        R.FUSED_BN_APPLY = True
        _set(torch.float32)


@pytest.mark.parametrize("N,hw,src", [(3, 64, torch.float32), (2, 37, torch.float64), (5, 224, torch.float32), (4, 30, torch.bfloat16)])
def test_stem_convolution_without_patch_matrix(dev, N, hw, src):
    """conv1 (7x7, stride 2, pad 3, RGB) through fcmf_pack_rgb0 + fcmf_conv_gemm_runs (one 8-pixel x RGB0 run per kernel row, zero
    weights for the 8th pixel and the padding channel) against the patch-matrix path on the same bf16-rounded crops and against
    F.conv2d in float32; NCHW crops read through their strides, odd sizes (the last run of a row needs the spare column),
    float64 crops (the reference's dataset dtype), border of the packed buffer still zero afterwards."""
    from fcmf_framework import ops, resnet as R
    _set(torch.bfloat16)
    try:
        conv = R.Conv2d(3, 64, 7, stride=2, padding=3, bias=False).to(dev)
        conv.weight.data = _rand(conv.weight.shape, 1, (2.0 / 147) ** 0.5).to(dev)
        x = _rand((N, 3, hw, hw + 2), 2).to(src).to(dev)                          # NCHW, as the dataset hands the crops over
        v = x.permute(0, 2, 3, 1)
        with torch.no_grad():
            assert R._stem_runs_ok(conv, v, ops.compute_dtype())
            y_runs, blk = R.conv2d_stem(v, conv, stats=True)
            assert blk is None                                  # (64 output channels: the statistics pass stays)
            ve = v if src != torch.float64 else x.float().permute(0, 2, 3, 1)   # (the patch-matrix kernel takes float32 / bf16 crops)
            y_exp = R.conv2d_nhwc(ve, conv, src_strides=ve.stride())
        assert y_runs.shape == y_exp.shape
        scale = y_exp.float().abs().max().item()
        assert max_err(y_runs, y_exp) <= 2 ** -7 * scale
        ref = F.conv2d(x.float().cpu().bfloat16().float(), conv.weight.data.float().cpu(), stride=2, padding=3).permute(0, 2, 3, 1)
        assert max_err(y_runs, ref) < 2e-2 * ref.abs().max().item()
        buf = R._pad_cache[("rgb0", N, hw, hw + 2, 3, str(v.device))]
        assert not buf[:, :3].any() and not buf[:, -3:].any() and not buf[:, :, :3].any() and not buf[:, :, hw + 2 + 3:].any() and not buf[..., 3].any()
    finally:
        _set(torch.float32)

"""CPU restatement (TEST INFRASTRUCTURE ONLY -- the product never imports this) of the ResNet-152 trunk that
the reference's myResNetImg / myResNetRoI drive (/root/reference/fcmf_framework/resnet_utils.py:13-30,39-56), i.e.
torchvision's `resnet152` (third party, `requirements.txt:35` torchvision 0.16.2 in a comment; NOT installed here and
its weights need a download).

PARITY UNPINNED vs torchvision: there is no reference test, fixture or importable torchvision to pin this file
against.  It restates the published ResNet v1.5 bottleneck architecture ([3, 8, 36, 3] blocks, expansion 4, stride on
the 3x3 convolution, 7x7/2 stem + 3x3/2 max-pool, BatchNorm eps 1e-5 momentum 0.1) on torch's own F.conv2d /
F.batch_norm / F.max_pool2d, which ARE the operators torchvision's module calls.

Call grouping: the reference calls the trunk once per image index with B crops (run_multimodal_fcmf.py:449-452) and
once per (image, ROI) with B crops (:454-457), in train() mode (:431), so BatchNorm batch statistics are those of
each B-crop call, and the running statistics are updated once per call, in call order.  `groups` = the number of
such calls packed into the leading axis (group-major), which the batched product path reproduces exactly.
"""
import torch
import torch.nn.functional as F


def _bn(P, prefix, x, training, groups, momentum=0.1, eps=1e-5):
    w, b = P[prefix + ".weight"], P[prefix + ".bias"]
    rm, rv = P[prefix + ".running_mean"], P[prefix + ".running_var"]
    if not training:
        return F.batch_norm(x, rm, rv, w, b, False, momentum, eps)
    outs = []
    for xg in x.chunk(groups, 0):                      # one reference call per group, in order
        outs.append(F.batch_norm(xg, rm, rv, w, b, True, momentum, eps))    # updates rm / rv in place
        if (prefix + ".num_batches_tracked") in P:
            P[prefix + ".num_batches_tracked"] += 1
    return torch.cat(outs, 0)


def _bottleneck(P, p, x, stride, training, groups):
    out = F.relu(_bn(P, p + ".bn1", F.conv2d(x, P[p + ".conv1.weight"]), training, groups))
    out = F.relu(_bn(P, p + ".bn2", F.conv2d(out, P[p + ".conv2.weight"], stride=stride, padding=1), training, groups))
    out = _bn(P, p + ".bn3", F.conv2d(out, P[p + ".conv3.weight"]), training, groups)
    if (p + ".downsample.0.weight") in P:
        x = _bn(P, p + ".downsample.1", F.conv2d(x, P[p + ".downsample.0.weight"], stride=stride), training, groups)
    return F.relu(out + x)


def resnet_trunk(P, x, layers, training=False, groups=1):
    """conv1 -> bn1 -> relu -> maxpool -> layer1..4 (resnet_utils.py:14-22).  P holds parameters and BN buffers
    (running statistics are updated IN PLACE in training mode, as nn.BatchNorm2d does)."""
    x = F.conv2d(x, P["conv1.weight"], stride=2, padding=3)
    x = F.relu(_bn(P, "bn1", x, training, groups))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for li, nblk in enumerate(layers):
        for b in range(nblk):
            x = _bottleneck(P, f"layer{li + 1}.{b}", x, 2 if (b == 0 and li > 0) else 1, training, groups)
    return x


def my_resnet_img(P, x, layers, att_size=7, training=False, groups=1):
    """myResNetImg.forward (resnet_utils.py:13-30): [N, 2048, att, att]"""
    return F.adaptive_avg_pool2d(resnet_trunk(P, x, layers, training, groups), [att_size, att_size])


def my_resnet_roi(P, x, layers, training=False, groups=1):
    """myResNetRoI.forward (resnet_utils.py:39-56): [N, 2048]"""
    return resnet_trunk(P, x, layers, training, groups).mean(3).mean(2)

"""myResNetImg / myResNetRoI (reference fcmf_framework/resnet_utils.py:6-56) on the MI355X ResNet trunk.

Same constructor `(resnet, if_fine_tune, device)` and forward signatures as the reference.  `resnet` may be
`fcmf_framework.resnet.resnet152()` or any module with torchvision's ResNet state-dict keys (the reference passes
torchvision's resnet152, run_multimodal_fcmf.py:224-227): its parameters and BatchNorm buffers are adopted by the
HIP trunk (`resnet.from_module`), so `.train()/.eval()`, `.state_dict()` (the `resimg_model` / `resroi_model`
checkpoints, :558-563) and the forward all act on the same weights.

One `forward(x)` call = one reference call = one BatchNorm group.  `forward_groups(x, groups)` is the MI355X-first
form: the reference's Python loop of num_imgs (+ num_imgs * num_rois) B-crop calls (:449-457) as ONE batched pass
with per-call BatchNorm statistics (see resnet.py).
"""
import torch
import torch.nn as nn

from . import resnet as R


class myResNetImg(nn.Module):
    def __init__(self, resnet, if_fine_tune, device):
        super().__init__()
        self.resnet = R.from_module(resnet)
        self.if_fine_tune = if_fine_tune
        self.device = device

    def _pool(self, feat, oh, ow, tokens=False):
        if feat.requires_grad:                         # if_fine_tune=True under grad mode: differentiable pooling
            return R.AvgPoolFn.apply(feat, oh, ow, tokens)
        return R.adaptive_avgpool_nhwc(feat, oh, ow, tokens=tokens)

    def forward_groups(self, x, groups=1, att_size=7, tokens=False):
        """x [groups*B, 3, H, W] packed group-major -> [groups*B, 2048, att, att] (tokens: [groups*B, att*att, 2048]);
        detached unless if_fine_tune (resnet_utils.py:26-28)"""
        return self._pool(self.resnet.trunk_nhwc(x, groups, fine_tune=bool(self.if_fine_tune)), att_size, att_size, tokens)

    def forward(self, x, att_size=7):
        return self.forward_groups(x, 1, att_size)       # detached by construction (resnet_utils.py:26-28)


class myResNetRoI(myResNetImg):
    def forward_groups(self, x, groups=1):
        """-> [groups*B, 2048]: x.mean(3).mean(2) of the trunk output (resnet_utils.py:48)"""
        return self._pool(self.resnet.trunk_nhwc(x, groups, fine_tune=bool(self.if_fine_tune)), 1, 1).flatten(1)

    def forward(self, x):
        return self.forward_groups(x, 1)


def extract_features(resnet_img, resnet_roi, t_img_features, roi_img_features):
    """The feature-extraction block of the step (run_multimodal_fcmf.py:449-460) as two batched trunk passes.
      t_img_features   [B, NI, 3, H, W]       -> vis_embeds [B, NI, 49, 2048]
      roi_img_features [B, NI, NR, 3, H, W]   -> roi_embeds [B, NI, NR, 2048]
    Reference call order = BatchNorm group order: image calls by img_idx; ROI calls by (img_idx, r)."""
    B, NI = t_img_features.shape[:2]
    NR = roi_img_features.shape[2]
    xi = t_img_features.transpose(0, 1).reshape(NI * B, *t_img_features.shape[2:])           # group-major
    vis = resnet_img.forward_groups(xi, NI, 7, tokens=True).view(NI, B, 49, -1).transpose(0, 1)
    xr = roi_img_features.permute(1, 2, 0, 3, 4, 5).reshape(NI * NR * B, *roi_img_features.shape[3:])
    roi = resnet_roi.forward_groups(xr, NI * NR).view(NI, NR, B, -1).permute(2, 0, 1, 3)
    return vis.contiguous(), roi.contiguous()

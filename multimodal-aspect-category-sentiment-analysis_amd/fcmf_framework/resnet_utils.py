"""ResNet trunk wrappers (reference fcmf_framework/resnet_utils.py:6-56).

BASELINE.json's configs use PRECOMPUTED ResNet-152 features, and torchvision's resnet152 (third
party, absent offline) is what the reference wraps, so the ResNet-152 trunk itself is the "next"
row of SURVEY.md section 8(f) and is not part of this round's hot path.  These wrappers keep the
reference's interface: they drive whatever `resnet` module they are given.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class myResNetImg(nn.Module):
    def __init__(self, resnet, if_fine_tune, device):
        super().__init__()
        self.resnet = resnet
        self.if_fine_tune = if_fine_tune
        self.device = device

    def _trunk(self, x):
        r = self.resnet
        x = r.maxpool(r.relu(r.bn1(r.conv1(x))))
        return r.layer4(r.layer3(r.layer2(r.layer1(x))))

    def forward(self, x, att_size=7):
        att = F.adaptive_avg_pool2d(self._trunk(x), [att_size, att_size])
        return att if self.if_fine_tune else att.detach()


class myResNetRoI(myResNetImg):
    def forward(self, x):
        fc = self._trunk(x).mean(3).mean(2)
        return fc if self.if_fine_tune else fc.detach()

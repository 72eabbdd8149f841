"""Step-invariant conditioning encoders (run once per item, SURVEY.md section 0 finding 2).

The reference re-evaluates both inside every denoising step (models/egohmr/egohmr.py:183,:214);
neither depends on x_t or t, so the build evaluates them once per sampled batch.  They are plain
library convolutions / GEMMs (MIOpen / rocBLAS through PyTorch-ROCm): the per-step hot ops are the
hand-written HIP kernels, these are not.  Sub-module and parameter names follow the reference so
its checkpoints load (``backbone.*`` models/resnet.py:97-136, ``scene_enc.*`` models/respointnet.py:13-27).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class _Bottleneck(nn.Module):
    def __init__(self, cin, width, stride, project):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, width * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(width * 4)
        self.downsample = None
        if project:
            self.downsample = nn.Sequential(nn.Conv2d(cin, width * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(width * 4))

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return F.relu(y + (x if self.downsample is None else self.downsample(x)))


class ResNet50Features(nn.Module):
    """models/resnet.py:139-150: ResNet-50 trunk, global average pool -> [B,2048]."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        cin = 64
        for i, (width, n, stride) in enumerate([(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)], 1):
            blocks = []
            for b in range(n):
                blocks.append(_Bottleneck(cin, width, stride if b == 0 else 1, project=(b == 0)))
                cin = width * 4
            setattr(self, f"layer{i}", nn.Sequential(*blocks))

    def forward(self, x):
        x = F.max_pool2d(F.relu(self.bn1(self.conv1(x))), 3, stride=2, padding=1)
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return x.mean(dim=(2, 3))


class _ResBlockFC(nn.Module):
    def __init__(self, cin, cout, hidden):
        super().__init__()
        self.fc_0 = nn.Linear(cin, hidden)
        self.fc_1 = nn.Linear(hidden, cout)
        self.shortcut = nn.Linear(cin, cout, bias=False)

    def forward(self, x):
        return self.shortcut(x) + self.fc_1(F.relu(self.fc_0(F.relu(x))))


class ResnetPointnet(nn.Module):
    """models/respointnet.py:33-59: per-point MLP with three global max-pool-concat stages -> [B,out_dim]."""

    def __init__(self, out_dim=512, hidden_dim=256):
        super().__init__()
        self.fc_pos_0 = nn.Linear(3, 2 * hidden_dim)
        for b in range(4):
            setattr(self, f"block_{b}", _ResBlockFC(2 * hidden_dim, hidden_dim, hidden_dim))
        self.fc_c = nn.Linear(hidden_dim, out_dim)

    def forward(self, p):
        net = self.block_0(self.fc_pos_0(p))
        for blk in (self.block_1, self.block_2, self.block_3):
            pooled = net.max(dim=1, keepdim=True)[0].expand_as(net)
            net = blk(torch.cat([net, pooled], dim=2))
        return self.fc_c(F.relu(net.max(dim=1)[0]))

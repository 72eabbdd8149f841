"""Eager PyTorch (MIOpen / hipBLASLt) forward of the ResNet-50 trunk for the yardstick tools ONLY: the package's modules have no eager route
(ResNet50Features.forward is the HIP path).  Works on the package's parameter containers."""
import torch.nn.functional as F


def _bn(x, bn):
    return F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)


def _conv(x, c):
    return F.conv2d(x, c.weight, None, c.stride, c.padding)


def resnet50_eager(net, x):
    x = F.max_pool2d(F.relu(_bn(_conv(x, net.conv1), net.bn1)), 3, stride=2, padding=1)
    for layer in (net.layer1, net.layer2, net.layer3, net.layer4):
        for b in layer:
            y = F.relu(_bn(_conv(x, b.conv1), b.bn1))
            y = F.relu(_bn(_conv(y, b.conv2), b.bn2))
            y = _bn(_conv(y, b.conv3), b.bn3)
            s = x if b.downsample is None else _bn(_conv(x, b.downsample[0]), b.downsample[1])
            x = F.relu(y + s)
    return x.mean(dim=(2, 3))

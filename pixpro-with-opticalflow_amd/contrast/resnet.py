"""ResNet encoders for the PixPro step, run by PyTorch-ROCm (MIOpen convolutions).

Out of the HIP hot path (BASELINE.json assigns the encoder forward/backward to PyTorch-ROCm); this
file only has to provide the constructor contract `arch(head_type='early_return')` and the state-dict
names (`conv1, bn1, layer1..4.<i>.conv1/bn1/..., downsample.0/1`) that the reference's checkpoints and
`transfer/detection/convert_pretrain_to_d2.py` rely on (reference: contrast/resnet.py).  Variants are
table-driven; `out_channels` exposes the c5 width so heads need no hard-coded 2048 (defect D3).
"""
import math

import torch.nn as nn

from .sync_bn import bn_act, bn_relu_pool, fork, residual_link

ARCHS = {
    # name: (block, depths, extra kwargs)
    "resnet18": ("basic", (2, 2, 2, 2), {}),
    "resnet34": ("basic", (3, 4, 6, 3), {}),
    "resnet50": ("bottleneck", (3, 4, 6, 3), {}),
    "resnet101": ("bottleneck", (3, 4, 23, 3), {}),
    "resnet152": ("bottleneck", (3, 8, 36, 3), {}),
    "resnet18_d": ("basic", (2, 2, 2, 2), dict(deep_stem=True, avg_down=True)),
    "resnet34_d": ("basic", (3, 4, 6, 3), dict(deep_stem=True, avg_down=True)),
    "resnet50_d": ("bottleneck", (3, 4, 6, 3), dict(deep_stem=True, avg_down=True)),
    "resnet101_d": ("bottleneck", (3, 4, 23, 3), dict(deep_stem=True, avg_down=True)),
    "resnet152_d": ("bottleneck", (3, 8, 36, 3), dict(deep_stem=True, avg_down=True)),
    "resnet50_16s": ("bottleneck", (3, 4, 6, 3), dict(layer4_dilation=2)),
    "resnet50_w2x": ("bottleneck", (3, 4, 6, 3), dict(width=2)),
    "resnext101_32x8d": ("bottleneck", (3, 4, 23, 3), dict(groups=32, width_per_group=8)),
    "resnext152_32x8d": ("bottleneck", (3, 8, 36, 3), dict(groups=32, width_per_group=8)),
}

__all__ = ["ResNet"] + list(ARCHS)


def _conv(cin, cout, k, stride=1, dilation=1, groups=1):
    pad = dilation * (k // 2)
    return nn.Conv2d(cin, cout, k, stride=stride, padding=pad, dilation=dilation, groups=groups, bias=False)


class _Residual(nn.Module):
    """Residual unit; `kind` selects the two-conv (expansion 1) or three-conv (expansion 4) body."""

    def __init__(self, kind, cin, planes, stride, downsample, groups, base_width, dilation):
        super().__init__()
        self.kind = kind
        if kind == "basic":
            if groups != 1 or base_width != 64:
                raise ValueError("basic residual units support groups=1, base_width=64 only")
            self.conv1, self.bn1 = _conv(cin, planes, 3, stride), nn.BatchNorm2d(planes)
            self.conv2, self.bn2 = _conv(planes, planes, 3), nn.BatchNorm2d(planes)
        else:
            mid = int(planes * (base_width / 64.)) * groups
            self.conv1, self.bn1 = _conv(cin, mid, 1), nn.BatchNorm2d(mid)
            self.conv2, self.bn2 = _conv(mid, mid, 3, stride, dilation, groups), nn.BatchNorm2d(mid)
            self.conv3, self.bn3 = _conv(mid, planes * 4, 1), nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        # bn_act(bn, t, relu, residual) = relu?(bn(t) (+ residual)): one fused launch pair with FusedSyncBatchNorm
        # link: with the fused modules the shortcut's gradient is never written as a tensor - the last BatchNorm hands (dy,
        # ReLU bits) to `fork` (identity shortcut) or to the shortcut's own BatchNorm (sync_bn.ResidualLink)
        link = residual_link(x, self.bn2 if self.kind == "basic" else self.bn3)
        if self.downsample is None:
            x, identity = fork(x, link) if link is not None else (x, x)
        elif link is not None and getattr(self.downsample[-1], "fused", False):
            identity = x
            for m in self.downsample[:-1]:
                identity = m(identity)
            identity = bn_act(self.downsample[-1], identity, link_in=link)
        else:
            identity = self.downsample(x)
        out = bn_act(self.bn1, self.conv1(x), relu=True)
        if self.kind == "basic":
            return bn_act(self.bn2, self.conv2(out), relu=True, residual=identity, link_out=link)
        out = bn_act(self.bn2, self.conv2(out), relu=True)
        return bn_act(self.bn3, self.conv3(out), relu=True, residual=identity, link_out=link)


class ResNet(nn.Module):
    def __init__(self, kind, depths, in_channel=3, width=1, groups=1, width_per_group=64, low_dim=128,
                 avg_down=False, deep_stem=False, head_type="mlp_head", layer4_dilation=1, **_unused):
        super().__init__()
        self.kind = kind
        self.expansion = 1 if kind == "basic" else 4
        base = int(64 * width)
        self._cin = base
        self._cfg = (groups, width_per_group, avg_down)
        if deep_stem:
            self.conv1 = nn.Sequential(
                nn.Sequential(_conv(in_channel, 32, 3, 2), nn.BatchNorm2d(32), nn.ReLU()),
                nn.Sequential(_conv(32, 32, 3), nn.BatchNorm2d(32), nn.ReLU()),
                _conv(32, 64, 3))
        else:
            self.conv1 = nn.Conv2d(in_channel, base, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(base)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._stage(base, depths[0], 1, 1)
        self.layer2 = self._stage(base * 2, depths[1], 2, 1)
        self.layer3 = self._stage(base * 4, depths[2], 2, 1)
        if layer4_dilation not in (1, 2):
            raise NotImplementedError
        self.layer4 = self._stage(base * 8, depths[3], 2 if layer4_dilation == 1 else 1, layer4_dilation)
        self.out_channels = base * 8 * self.expansion
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.head_type = head_type
        mid = self.out_channels
        if head_type == "mlp_head":
            self.fc1, self.relu2, self.fc2 = nn.Linear(mid, mid), nn.ReLU(inplace=True), nn.Linear(mid, low_dim)
        elif head_type == "reduce":
            self.fc = nn.Linear(mid, low_dim)
        elif head_type not in ("pass", "early_return", "multi_layer"):
            raise NotImplementedError(head_type)
        self._init_weights()

    def _stage(self, planes, count, stride, dilation):
        groups, base_width, avg_down = self._cfg
        cout = planes * self.expansion
        down = None
        if stride != 1 or self._cin != cout:
            if avg_down:
                down = nn.Sequential(nn.AvgPool2d(stride, stride), _conv(self._cin, cout, 1), nn.BatchNorm2d(cout))
            else:
                down = nn.Sequential(_conv(self._cin, cout, 1, stride), nn.BatchNorm2d(cout))
        units = [_Residual(self.kind, self._cin, planes, stride, down, groups, base_width, dilation)]
        self._cin = cout
        units += [_Residual(self.kind, cout, planes, 1, None, groups, base_width, dilation) for _ in range(count - 1)]
        return nn.Sequential(*units)

    def _init_weights(self):
        # He-normal convs, unit BN, zero gamma on the last BN of every residual unit
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                fan = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                nn.init.normal_(m.weight, 0.0, math.sqrt(2.0 / fan))
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
        for m in self.modules():
            if isinstance(m, _Residual):
                nn.init.zeros_((m.bn2 if m.kind == "basic" else m.bn3).weight)

    def forward(self, x):
        x = bn_relu_pool(self.bn1, self.conv1(x), self.maxpool)      # one fused pass with FusedSyncBatchNorm
        c2 = self.layer1(x)
        c3 = self.layer2(c2)
        c4 = self.layer3(c3)
        c5 = self.layer4(c4)
        if self.head_type == "multi_layer":
            return c2, c3, c4, c5
        if self.head_type in ("early_return", "pass"):
            return c5
        feat = self.avgpool(c5).flatten(1)
        if self.head_type == "mlp_head":
            return self.fc2(self.relu2(self.fc1(feat)))
        return self.fc(feat)


def _make(name):
    kind, depths, extra = ARCHS[name]

    def ctor(**kwargs):
        return ResNet(kind, depths, **{**extra, **kwargs})

    ctor.__name__ = name
    return ctor


for _name in ARCHS:
    globals()[_name] = _make(_name)

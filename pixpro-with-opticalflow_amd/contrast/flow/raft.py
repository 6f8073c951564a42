"""RAFT / RAFT-small inference for the online flow mode of `util.apply_optical_flow` (reference: contrast/flow/raft.py,
extractor.py, update.py).  Parameter names follow the reference (`fnet, cnet, update_block.encoder/gru/flow_head/mask`,
`layer{1,2,3}.{0,1}.conv*/norm*`), so the public RAFT checkpoints (`raft-small.pth`, `raft-things.pth`, stored from a
DataParallel wrapper, i.e. with a `module.` prefix) load unchanged.  Convolutions run in MIOpen; the correlation lookup of
every refinement iteration is one HIP launch (flow/corr.py).  Inference only (`test_mode=True`), fp32.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .corr import CorrBlock
from ..util import upflow8


def coords_grid(batch, ht, wd, device=None):
    """[B,2,h,w] with channel 0 = x, channel 1 = y (utils.py:81-84)."""
    ys, xs = torch.meshgrid(torch.arange(ht, device=device), torch.arange(wd, device=device), indexing="ij")
    return torch.stack([xs, ys], dim=0).float()[None].repeat(batch, 1, 1, 1)


def _norm(kind, planes, groups=None):
    if kind == "group":
        return nn.GroupNorm(num_groups=groups, num_channels=planes)
    if kind == "batch":
        return nn.BatchNorm2d(planes)
    if kind == "instance":
        return nn.InstanceNorm2d(planes)
    if kind == "none":
        return nn.Sequential()
    raise ValueError(kind)


class _Unit(nn.Module):
    """Residual unit of the encoders: two 3x3 convs (`bottleneck=False`, extractor.py:6-57) or 1x1-3x3-1x1 with a
    quarter-width body (`bottleneck=True`, extractor.py:60-118).  The shortcut gets a strided 1x1 conv + norm when the unit
    down-samples."""

    def __init__(self, cin, planes, norm_fn, stride, bottleneck):
        super().__init__()
        g = planes // 8
        if bottleneck:
            mid = planes // 4
            self.conv1 = nn.Conv2d(cin, mid, 1)
            self.conv2 = nn.Conv2d(mid, mid, 3, padding=1, stride=stride)
            self.conv3 = nn.Conv2d(mid, planes, 1)
            widths = [mid, mid, planes]
        else:
            self.conv1 = nn.Conv2d(cin, planes, 3, padding=1, stride=stride)
            self.conv2 = nn.Conv2d(planes, planes, 3, padding=1)
            widths = [planes, planes]
        self.relu = nn.ReLU(inplace=True)
        for i, c in enumerate(widths, 1):
            setattr(self, f"norm{i}", _norm(norm_fn, c, g))
        self.n_body = len(widths)
        self.downsample = None
        if stride != 1:
            last = _norm(norm_fn, planes, g)
            setattr(self, f"norm{len(widths) + 1}", last)
            self.downsample = nn.Sequential(nn.Conv2d(cin, planes, 1, stride=stride), last)

    def forward(self, x):
        y = x
        for i in range(1, self.n_body + 1):
            y = self.relu(getattr(self, f"norm{i}")(getattr(self, f"conv{i}")(y)))
        if self.downsample is not None:
            x = self.downsample(x)
        return self.relu(x + y)


class _Encoder(nn.Module):
    """7x7/2 stem + three stages of two units (strides 1, 2, 2 -> 1/8 resolution) + 1x1 output conv
    (BasicEncoder extractor.py:121-192 with widths 64/96/128, SmallEncoder :195-271 with 32/64/96 bottleneck units)."""

    def __init__(self, widths, bottleneck, output_dim, norm_fn, dropout=0.0):
        super().__init__()
        self.norm_fn = norm_fn
        self.norm1 = _norm(norm_fn, widths[0], 8)
        self.conv1 = nn.Conv2d(3, widths[0], 7, stride=2, padding=3)
        self.relu1 = nn.ReLU(inplace=True)
        cin = widths[0]
        for i, (wd, stride) in enumerate(zip(widths, (1, 2, 2)), 1):
            setattr(self, f"layer{i}", nn.Sequential(_Unit(cin, wd, norm_fn, stride, bottleneck),
                                                     _Unit(wd, wd, norm_fn, 1, bottleneck)))
            cin = wd
        self.conv2 = nn.Conv2d(cin, output_dim, 1)
        self.dropout = nn.Dropout2d(p=dropout) if dropout > 0 else None
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, (nn.BatchNorm2d, nn.InstanceNorm2d, nn.GroupNorm)) and m.weight is not None:
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        pair = isinstance(x, (tuple, list))
        if pair:
            n = x[0].shape[0]
            x = torch.cat(x, dim=0)
        x = self.relu1(self.norm1(self.conv1(x)))
        x = self.conv2(self.layer3(self.layer2(self.layer1(x))))
        if self.training and self.dropout is not None:
            x = self.dropout(x)
        return torch.split(x, [n, n], dim=0) if pair else x


def BasicEncoder(output_dim=128, norm_fn="batch", dropout=0.0):
    return _Encoder((64, 96, 128), False, output_dim, norm_fn, dropout)


def SmallEncoder(output_dim=128, norm_fn="batch", dropout=0.0):
    return _Encoder((32, 64, 96), True, output_dim, norm_fn, dropout)


class FlowHead(nn.Module):
    def __init__(self, input_dim=128, hidden_dim=256):
        super().__init__()
        self.conv1 = nn.Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.conv2 = nn.Conv2d(hidden_dim, 2, 3, padding=1)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        return self.conv2(self.relu(self.conv1(x)))


class _GRU(nn.Module):
    """Convolutional GRU; `passes` lists (suffix, kernel, padding): one 3x3 pass (ConvGRU, update.py:17-32) or a 1x5
    pass followed by a 5x1 pass (SepConvGRU, :35-73)."""

    def __init__(self, hidden_dim, input_dim, passes):
        super().__init__()
        self.suffixes = [s for s, _, _ in passes]
        for s, k, p in passes:
            for gate in "zrq":
                setattr(self, f"conv{gate}{s}", nn.Conv2d(hidden_dim + input_dim, hidden_dim, k, padding=p))

    def forward(self, h, x):
        for s in self.suffixes:
            hx = torch.cat([h, x], dim=1)
            z = torch.sigmoid(getattr(self, f"convz{s}")(hx))
            r = torch.sigmoid(getattr(self, f"convr{s}")(hx))
            q = torch.tanh(getattr(self, f"convq{s}")(torch.cat([r * h, x], dim=1)))
            h = (1 - z) * h + z * q
        return h


def ConvGRU(hidden_dim=128, input_dim=192 + 128):
    return _GRU(hidden_dim, input_dim, [("", 3, 1)])


def SepConvGRU(hidden_dim=128, input_dim=192 + 128):
    return _GRU(hidden_dim, input_dim, [("1", (1, 5), (0, 2)), ("2", (5, 1), (2, 0))])


class _MotionEncoder(nn.Module):
    """Correlation + flow features (SmallMotionEncoder update.py:76-90, BasicMotionEncoder :93-112)."""

    def __init__(self, cor_planes, small):
        super().__init__()
        if small:
            self.convc1 = nn.Conv2d(cor_planes, 96, 1)
            self.convf1, self.convf2 = nn.Conv2d(2, 64, 7, padding=3), nn.Conv2d(64, 32, 3, padding=1)
            self.conv = nn.Conv2d(128, 80, 3, padding=1)
        else:
            self.convc1, self.convc2 = nn.Conv2d(cor_planes, 256, 1), nn.Conv2d(256, 192, 3, padding=1)
            self.convf1, self.convf2 = nn.Conv2d(2, 128, 7, padding=3), nn.Conv2d(128, 64, 3, padding=1)
            self.conv = nn.Conv2d(64 + 192, 128 - 2, 3, padding=1)
        self.small = small

    def forward(self, flow, corr):
        cor = F.relu(self.convc1(corr))
        if not self.small:
            cor = F.relu(self.convc2(cor))
        flo = F.relu(self.convf2(F.relu(self.convf1(flow))))
        out = F.relu(self.conv(torch.cat([cor, flo], dim=1)))
        return torch.cat([out, flow], dim=1)


class _UpdateBlock(nn.Module):
    """SmallUpdateBlock (update.py:115-128: no up-sampling mask) / BasicUpdateBlock (:131-152)."""

    def __init__(self, args, hidden_dim, small):
        super().__init__()
        self.args = args
        cor_planes = args.corr_levels * (2 * args.corr_radius + 1) ** 2
        self.encoder = _MotionEncoder(cor_planes, small)
        if small:
            self.gru = ConvGRU(hidden_dim=hidden_dim, input_dim=82 + 64)
            self.flow_head = FlowHead(hidden_dim, hidden_dim=128)
            self.mask = None
        else:
            self.gru = SepConvGRU(hidden_dim=hidden_dim, input_dim=128 + hidden_dim)
            self.flow_head = FlowHead(hidden_dim, hidden_dim=256)
            self.mask = nn.Sequential(nn.Conv2d(128, 256, 3, padding=1), nn.ReLU(inplace=True), nn.Conv2d(256, 64 * 9, 1))

    def forward(self, net, inp, corr, flow):
        inp = torch.cat([inp, self.encoder(flow, corr)], dim=1)
        net = self.gru(net, inp)
        delta = self.flow_head(net)
        mask = None if self.mask is None else .25 * self.mask(net)      # scaled to balance gradients (:150)
        return net, mask, delta


def SmallUpdateBlock(args, hidden_dim=96):
    return _UpdateBlock(args, hidden_dim, True)


def BasicUpdateBlock(args, hidden_dim=128):
    return _UpdateBlock(args, hidden_dim, False)


class RAFT(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        for name, default in (("small", False), ("dropout", 0), ("alternate_corr", False), ("mixed_precision", False)):
            if not hasattr(args, name):          # the reference tests `name in args`; any attribute container works here
                setattr(args, name, default)
        if args.alternate_corr:
            raise NotImplementedError("alternate_corr: the volume is kept resident on MI355X (flow/corr.py)")
        if args.mixed_precision:
            raise NotImplementedError("RAFT runs in fp32 here (the reference sets mixed_precision=False, option.py:111)")
        args.corr_levels = 4
        if args.small:
            self.hidden_dim, self.context_dim, args.corr_radius = 96, 64, 3
            self.fnet = SmallEncoder(output_dim=128, norm_fn="instance", dropout=args.dropout)
            self.cnet = SmallEncoder(output_dim=self.hidden_dim + self.context_dim, norm_fn="none", dropout=args.dropout)
            self.update_block = SmallUpdateBlock(args, hidden_dim=self.hidden_dim)
        else:
            self.hidden_dim, self.context_dim, args.corr_radius = 128, 128, 4
            self.fnet = BasicEncoder(output_dim=256, norm_fn="instance", dropout=args.dropout)
            self.cnet = BasicEncoder(output_dim=self.hidden_dim + self.context_dim, norm_fn="batch", dropout=args.dropout)
            self.update_block = BasicUpdateBlock(args, hidden_dim=self.hidden_dim)

    def freeze_bn(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()

    def initialize_flow(self, img):
        N, _, H, W = img.shape
        return coords_grid(N, H // 8, W // 8, img.device), coords_grid(N, H // 8, W // 8, img.device)

    def upsample_flow(self, flow, mask):
        """Convex combination of the 3x3 neighbourhood, [h,w] -> [8h,8w] (raft.py:83-95)."""
        N, _, H, W = flow.shape
        mask = torch.softmax(mask.view(N, 1, 9, 8, 8, H, W), dim=2)
        up = F.unfold(8 * flow, [3, 3], padding=1).view(N, 2, 9, 1, 1, H, W)
        up = torch.sum(mask * up, dim=2).permute(0, 1, 4, 2, 5, 3)
        return up.reshape(N, 2, 8 * H, 8 * W)

    def forward(self, image1, image2, iters=12, flow_init=None, upsample=True, test_mode=False):
        image1 = (2 * (image1 / 255.0) - 1.0).contiguous()
        image2 = (2 * (image2 / 255.0) - 1.0).contiguous()
        fmap1, fmap2 = self.fnet([image1, image2])
        corr_fn = CorrBlock(fmap1.float(), fmap2.float(), radius=self.args.corr_radius)
        net, inp = torch.split(self.cnet(image1), [self.hidden_dim, self.context_dim], dim=1)
        net, inp = torch.tanh(net), torch.relu(inp)
        coords0, coords1 = self.initialize_flow(image1)
        if flow_init is not None:
            coords1 = coords1 + flow_init
        predictions = []
        flow_up = None
        for itr in range(iters):
            coords1 = coords1.detach()
            corr = corr_fn(coords1)
            net, up_mask, delta = self.update_block(net, inp, corr, coords1 - coords0)
            coords1 = coords1 + delta
            if not test_mode or itr == iters - 1:     # only the last prediction is used in test mode
                flow_up = upflow8(coords1 - coords0) if up_mask is None else self.upsample_flow(coords1 - coords0, up_mask)
                predictions.append(flow_up)
        if test_mode:
            return coords1 - coords0, flow_up
        return predictions

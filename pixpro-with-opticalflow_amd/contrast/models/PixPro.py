"""`contrast.models.PixPro` surface on top of the HIP hot path.

Mirrors the module surface of the reference's contrast/models/PixPro.py (class `PixPro`, module
functions `regression_loss`, `add_optical_flow`, `Proj_Head`, `Pred_Head`, `MLP2d`, `conv1x1`;
sub-module names `encoder, projector, encoder_k, projector_k, value_transform`) so checkpoints and
`main_pretrain.py` keep working, but the per-pixel math runs in libpixpro_hip.so:

  reference PixPro.forward (PixPro.py:368-439)        here
  -------------------------------------------------   ------------------------------------------
  featprop x2 + F.normalize x2 (:376-385)           }
  F.normalize of the two key projections (:403,407) }  ONE fused launch, both loss directions,
  regression_loss x2 incl. add_optical_flow (:429)  }  forward + backward (ops.ppm_loss)
  _momentum_update_key_encoder: 165 x 3 kernels       one multi-tensor launch (ops_multi.ema_update)

Fixes of reference defects that block the BASELINE configs: D3 (`Proj_Head`
in_dim follows the encoder's c5 width), D8 (no throw-away encoder).  The debug drawing mode (tuple
coords) is out of scope and raises.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.distributed import get_world_size, is_initialized

from .. import ops, wgrad_stream
from ..multi_tensor import ema_update
from ..sync_bn import FusedSyncBatchNorm, bn_act, view_groups
from .base import BaseModel


class Identity(nn.Module):
    def forward(self, input):
        return input


def conv1x1(in_planes, out_planes):
    return nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=1, padding=0, bias=True)


class MLP2d(nn.Module):
    """1x1-conv MLP: in -> inner (BN, ReLU) -> out.  Parameter names follow PixPro.py:25-43."""

    def __init__(self, in_dim, inner_dim=4096, out_dim=256):
        super().__init__()
        self.linear1 = conv1x1(in_dim, inner_dim)
        self.bn1 = nn.BatchNorm2d(inner_dim)
        self.relu1 = nn.ReLU(inplace=True)
        self.linear2 = conv1x1(inner_dim, out_dim)

    def forward(self, x):
        return self.linear2(bn_act(self.bn1, self.linear1(x), relu=True))


def Proj_Head(in_dim=2048, inner_dim=4096, out_dim=256):
    return MLP2d(in_dim, inner_dim, out_dim)


def Pred_Head(in_dim=256, inner_dim=4096, out_dim=256):
    return MLP2d(in_dim, inner_dim, out_dim)


def _unpack(coord_q, coord_k):
    """Argument polymorphism of regression_loss (PixPro.py:99-124): Tensor[N,10] | [Tensor, flow] with
    flow = Tensor[N,2,Hf,Wf] | [flow, size, mask]; mask = None | Tensor | [mask, cycle]."""
    if isinstance(coord_q, tuple):
        raise NotImplementedError("debug drawing mode (tuple coords) is not part of the HIP hot path")
    flow = mask = size = None
    if isinstance(coord_q, list):
        coord_q, flow = coord_q
        coord_k = coord_k[0]
        if isinstance(flow, list):
            flow, size, mask = flow
            if isinstance(mask, list):
                mask = mask[0]
        if hasattr(flow, "lo_own"):        # util.LazyFlow carries its own mask definition
            mask = None
    return coord_q, coord_k, flow, size, mask


def add_optical_flow(flow, x_grid, y_grid, size, mask=None, verbose=False):
    """Drop-in for PixPro.py:46-89 (one launch instead of ~17)."""
    if isinstance(size, torch.Tensor):
        size = (int(size[0].item()), int(size[1].item()))
    return ops.add_optical_flow_op(flow, x_grid, y_grid, size, mask)


def regression_loss(q, k, coord_q, coord_k, pos_ratio=0.5):
    """Drop-in for PixPro.py:92-247: returns (-2*mean_n loss_n, [pos_num[N], pos_mean[N]])."""
    if isinstance(k, tuple):
        raise NotImplementedError("debug drawing mode is not part of the HIP hot path")
    coord_q, coord_k, flow, size, mask = _unpack(coord_q, coord_k)
    loss, pos_num, pos_mean = ops.regression_loss_op(q, k, coord_q, coord_k, pos_ratio, flow, mask, size)
    return loss, [pos_num, pos_mean]


class PixPro(BaseModel):
    def __init__(self, base_encoder, args):
        super().__init__(base_encoder, args)
        self.pixpro_p = args.pixpro_p
        self.pixpro_momentum = args.pixpro_momentum
        self.pixpro_pos_ratio = args.pixpro_pos_ratio
        self.pixpro_clamp_value = args.pixpro_clamp_value
        self.pixpro_transform_layer = args.pixpro_transform_layer
        self.pixpro_ins_loss_weight = args.pixpro_ins_loss_weight
        self.output_root = getattr(args, "output_dir", None)
        self.overlap_momentum = bool(getattr(args, "overlap_momentum", False))
        # both views of an encoder in one pass, BatchNorm statistics still per view (sync_bn.view_groups): the per-layer
        # SyncBN collectives of the two views travel together, 162 instead of 324 per step
        self.view_batch = bool(getattr(args, "bn_view_batch", False))
        if self.view_batch and not getattr(args, "fused_bn", False):
            raise ValueError("--bn-view-batch needs --fused-bn: stock BatchNorm would pool the statistics of the two views")
        self._side_stream = None

        self.encoder = base_encoder(head_type="early_return")
        c5 = getattr(self.encoder, "out_channels", 2048)
        self.projector = Proj_Head(in_dim=c5)
        self.encoder_k = base_encoder(head_type="early_return")
        self.projector_k = Proj_Head(in_dim=c5)
        self._tie(self.encoder, self.encoder_k)
        self._tie(self.projector, self.projector_k)

        sync_bn = getattr(args, "sync_bn", True)
        # PixPro.py:289-292 (in-place on the children).  --fused-bn: the HIP SyncBN with ReLU / residual add folded in
        self._convert_bn = (FusedSyncBatchNorm.convert if getattr(args, "fused_bn", False)
                            else nn.SyncBatchNorm.convert_sync_batchnorm)
        # With several ranks the SyncBN collectives of a process group are queued on that group's ONE internal stream in host
        # order: the key branch on its second HIP stream (overlap_momentum) would wait head-of-line behind the online
        # branch's collectives and the other way round.  --key-bn-group gives the key encoder's BatchNorm layers a process
        # group of their own (same ranks, a second RCCL communicator with its own stream): the two branches then exchange
        # statistics independently.  Same numbers - only which communicator carries them changes.
        self.key_bn_group = None
        if sync_bn and getattr(args, "key_bn_group", False) and torch.distributed.is_available() \
                and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            self.key_bn_group = torch.distributed.new_group()
        if sync_bn:
            for m in (self.encoder, self.projector):
                self._convert_bn(m)
            for m in (self.encoder_k, self.projector_k):
                self._convert_bn(m, self.key_bn_group)

        world = get_world_size() if is_initialized() else 1
        # weight gradients of the online encoder's convolutions on a second stream (contrast/wgrad_stream.py): one rank only -
        # DistributedDataParallel copies a gradient into its bucket the moment autograd delivers it
        self.wgrad_stream = bool(getattr(args, "wgrad_stream", False)) and world == 1
        if self.wgrad_stream:
            wgrad_stream.convert(self.encoder)
            wgrad_stream.convert(self.projector)
        self._steps_per_epoch = args.num_instances * 1. / world / args.batch_size
        self.K = int(self._steps_per_epoch * args.epochs)           # PixPro.py:294
        self.k = self.steps_before_epoch(args.start_epoch)          # PixPro.py:295

        if self.pixpro_transform_layer == 0:
            self.value_transform = Identity()
        elif self.pixpro_transform_layer == 1:
            self.value_transform = conv1x1(in_planes=256, out_planes=256)
        elif self.pixpro_transform_layer == 2:
            self.value_transform = MLP2d(in_dim=256, inner_dim=256, out_dim=256)
        else:
            raise NotImplementedError

        if self.pixpro_ins_loss_weight > 0.:
            self.projector_instance = Proj_Head(in_dim=c5)
            self.projector_instance_k = Proj_Head(in_dim=c5)
            self.predictor = Pred_Head()
            self._tie(self.projector_instance, self.projector_instance_k)
            if sync_bn:
                for m in (self.projector_instance, self.projector_instance_k, self.predictor):
                    self._convert_bn(m)
            self.avgpool = nn.AdaptiveAvgPool2d(1)   # reference hard-codes AvgPool2d(7) (defect D4)

    @staticmethod
    def _tie(online, key):
        for pq, pk in zip(online.parameters(), key.parameters()):
            pk.data.copy_(pq.data)
            pk.requires_grad = False

    def _ema_pairs(self):
        pairs = [(self.encoder, self.encoder_k), (self.projector, self.projector_k)]
        if self.pixpro_ins_loss_weight > 0.:
            pairs.append((self.projector_instance, self.projector_instance_k))
        for a, b in pairs:
            yield from zip(a.parameters(), b.parameters())

    def steps_before_epoch(self, epoch):
        """EMA step counter at the start of `epoch` (PixPro.py:295); also what a resume restores."""
        return int(self._steps_per_epoch * (epoch - 1))

    def momentum_coefficient(self):
        """1 - (1-m)(cos(pi k/K)+1)/2, PixPro.py:326.  k/K is held at 1 once the schedule has run out (a loader with more
        steps per epoch than num_instances / (world * batch) would otherwise wrap the cosine back towards m)."""
        frac = min(self.k / self.K, 1.) if self.K > 0 else 1.
        return 1. - (1. - self.pixpro_momentum) * (math.cos(math.pi * frac) + 1) / 2.

    @torch.no_grad()
    def _momentum_update_key_encoder(self):
        m = self.momentum_coefficient()
        self.k = self.k + 1
        pq, pk = zip(*self._ema_pairs())
        ema_update(pq, pk, m)

    def featprop(self, feat):
        """Public PPM (PixPro.py:339-363): returns the un-normalised propagated features."""
        return ops.featprop(feat, self.value_transform(feat), self.pixpro_p, self.pixpro_clamp_value)

    def regression_loss(self, x, y):
        return -2. * torch.einsum('nc, nc->n', [x, y]).mean()

    def _two_views(self, encoder, projector, im_1, im_2):
        """(feat_1, proj_1, feat_2, proj_2) of one encoder + projector: two passes as in the reference (PixPro.py:376-384),
        or one pass over both views with per-view BatchNorm statistics."""
        if self.view_batch:
            with view_groups(2):
                feat = encoder(torch.cat([im_1, im_2]))
                proj = projector(feat)
            (feat_1, feat_2), (proj_1, proj_2) = feat.chunk(2), proj.chunk(2)
            return feat_1, proj_1, feat_2, proj_2
        feat_1 = encoder(im_1)
        proj_1 = projector(feat_1)
        feat_2 = encoder(im_2)
        proj_2 = projector(feat_2)
        return feat_1, proj_1, feat_2, proj_2

    def _momentum_passes(self, im_1, im_2, is_update_momentum):
        with torch.no_grad():
            if is_update_momentum:
                self._momentum_update_key_encoder()
            if self.view_batch:
                feat_1_ng, key_1, feat_2_ng, key_2 = self._two_views(self.encoder_k, self.projector_k, im_1, im_2)
                return key_1, key_2, feat_1_ng, feat_2_ng
            feat_1_ng = self.encoder_k(im_1)
            key_1 = self.projector_k(feat_1_ng)          # normalised inside the kernel (:403)
            feat_2_ng = self.encoder_k(im_2)
            key_2 = self.projector_k(feat_2_ng)
        return key_1, key_2, feat_1_ng, feat_2_ng

    def forward(self, im_1, im_2, coord1, coord2, is_update_momentum=True):
        overlap = self.overlap_momentum and im_1.is_cuda
        if overlap:
            # the key branch (EMA update + two no-grad passes) depends on nothing the online branch produces: it runs on a
            # second stream, filling the CUs the small late-stage kernels leave idle.  Measured with one rank only.  With
            # several ranks every SyncBN collective of the process is queued on ProcessGroupNCCL's one internal stream in
            # host order, so the online branch's first collective waits for all of the key branch's: no overlap is to be
            # expected there and bench.py / main_pretrain leave the flag off for world > 1 unless it is forced.
            main = torch.cuda.current_stream()
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(device=im_1.device)
            side = self._side_stream
            side.wait_stream(main)
            with torch.cuda.stream(side):
                key_1, key_2, feat_1_ng, feat_2_ng = self._momentum_passes(im_1, im_2, is_update_momentum)
        feat_1, proj_1, feat_2, proj_2 = self._two_views(self.encoder, self.projector, im_1, im_2)
        val_1 = self.value_transform(proj_1)
        val_2 = self.value_transform(proj_2)
        if overlap:
            main.wait_stream(side)
            for t in (key_1, key_2, feat_1_ng, feat_2_ng):
                t.record_stream(main)
        else:
            key_1, key_2, feat_1_ng, feat_2_ng = self._momentum_passes(im_1, im_2, is_update_momentum)

        c1, c2, flow_fwd, size, mask_fwd = _unpack(coord1, coord2)
        _, _, flow_bwd, _, mask_bwd = _unpack(coord2, coord1)
        d1 = ops.Direction(proj_1, val_1, key_2, c1, c2, flow_fwd, mask_fwd, size)
        d2 = ops.Direction(proj_2, val_2, key_1, c2, c1, flow_bwd, mask_bwd, size)
        loss, pos_num, pos_mean, _ = ops.ppm_loss([d1, d2], self.pixpro_p, self.pixpro_clamp_value,
                                                  self.pixpro_pos_ratio, normalize_k=True)
        pos_num_list = [[pos_num[0], pos_mean[0]], [pos_num[1], pos_mean[1]]]

        if self.pixpro_ins_loss_weight > 0.:
            def pooled(t):
                return F.normalize(self.avgpool(t).flatten(1), dim=1)
            pred_i1 = pooled(self.predictor(self.projector_instance(feat_1)))
            pred_i2 = pooled(self.predictor(self.projector_instance(feat_2)))
            with torch.no_grad():
                key_i1 = pooled(self.projector_instance_k(feat_1_ng))
                key_i2 = pooled(self.projector_instance_k(feat_2_ng))
            loss_instance = self.regression_loss(pred_i1, key_i2) + self.regression_loss(pred_i2, key_i1)
            loss = loss + self.pixpro_ins_loss_weight * loss_instance

        return loss, pos_num_list

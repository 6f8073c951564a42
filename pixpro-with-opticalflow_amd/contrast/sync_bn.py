"""(Sync)BatchNorm on the HIP library with the ReLU / residual add of the ResNet blocks folded in.

The reference turns every BatchNorm of the encoders and projectors into `torch.nn.SyncBatchNorm`
(contrast/models/PixPro.py:289-292).  With more than one rank that module runs ATen's native kernels instead of MIOpen
(torch/nn/modules/_functions.py:SyncBatchNorm), which is ~3x slower on MI355X (profiles/r01_syncbn_microbench.txt), and
every BN is followed by a separate ReLU (and, at the end of a residual unit, a separate add).  `FusedSyncBatchNorm`
keeps torch's split - local statistics | all_gather | normalise ; local sums | all_reduce | input gradient - so the
collectives and their payloads ([2C+1] and [2C] floats per layer) are exactly torch's, but the element-wise passes are
the streaming kernels of csrc/sync_bn.hip:

    y = relu?( batch_norm(x) (+ residual) )                    3 (4) tensor passes instead of 5 (8)
    backward: dz = dy * (y > 0)?; dx, dweight, dbias, dresidual = dz

Three things keep traffic out of the backward pass (round 4): behind a residual add the ReLU decisions are
kept as a bit record instead of being read back from the saved output (`pixpro_bn_apply_bits`); the shortcut gradient of a
residual unit is handed to its consumer as (dy, bits) over a `ResidualLink` instead of being written and added by
autograd (`fork`, `link_in` / `link_out`); and the stem's `maxpool(relu(bn1(x)))` is one pass in each direction
(`bn_relu_pool`).  All three are bit-compatible with the plain route and have an environment switch for A/B runs
(PIXPRO_BN_RELU_BITS, PIXPRO_BN_LAZY_SHORTCUT, PIXPRO_BN_FUSED_POOL = 0).

Parameter / buffer names are those of `nn.BatchNorm2d` (`weight, bias, running_mean, running_var, num_batches_tracked`),
so checkpoints load either way.  Activations must be channels_last (the layout the encoders run in); CPU tensors raise.
"""
import ctypes
import os

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from . import _native as nv


def _rows_view(t, name):
    """[N,C,H,W] channels_last-dense (or any layout when H = W = 1) -> (tensor, rows, C)."""
    if t.dim() == 2:
        t = t if t.is_contiguous() else t.contiguous()
        return t, t.shape[0], t.shape[1]
    if t.dim() != 4:
        raise RuntimeError(f"{name}: expected [N,C,H,W] or [N,C], got {tuple(t.shape)}")
    if not t.is_cuda or t.dtype != torch.float32:
        raise RuntimeError(f"{name}: FusedSyncBatchNorm runs on fp32 GPU tensors only (got {t.dtype} on {t.device}); "
                           "there is no CPU fallback")
    if not t.is_contiguous(memory_format=torch.channels_last):
        t = t.contiguous(memory_format=torch.channels_last)
    return t, t.shape[0] * t.shape[2] * t.shape[3], t.shape[1]


def _world(group):
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    return dist.get_world_size(group)


_VIEW_GROUPS = 1
# PIXPRO_BN_RELU_BITS=0: keep reading the saved output for the ReLU decisions behind a residual add (A/B switch, tests)
_RELU_BITS = os.environ.get("PIXPRO_BN_RELU_BITS", "1") != "0"


class view_groups:
    """`with view_groups(2): encoder(torch.cat([im_1, im_2]))` - the batch holds G views of N samples each (view g =
    samples [g*N, (g+1)*N)) and every FusedSyncBatchNorm inside normalises each view with ITS OWN statistics, exactly as
    G separate forward passes would (same kernels on the same rows, running statistics updated view after view), but
    the G per-view records of a layer cross the process group in ONE collective per direction: all_gather of
    [G, 2C+1] floats forward, all_reduce of [G, 2C] floats backward.  PixPro runs two views through each encoder
    (contrast/models/PixPro.py:376-407), so this halves the 324 SyncBN collectives of a step without
    changing a statistic."""

    def __init__(self, groups):
        self.groups = int(groups)

    def __enter__(self):
        global _VIEW_GROUPS
        self.prev, _VIEW_GROUPS = _VIEW_GROUPS, self.groups

    def __exit__(self, *exc):
        global _VIEW_GROUPS
        _VIEW_GROUPS = self.prev


class ResidualLink:
    """Carries the shortcut gradient of ONE residual unit from the unit's last BatchNorm to whatever produced the shortcut
    tensor, WITHOUT materialising it.  That gradient is `dy` masked by the unit's ReLU decisions; with the bit record
    (pixpro_bn_apply_bits) the last BatchNorm's backward hands over (dy, bits) instead of writing `dy * mask` for autograd
    to add: the consumer - `fork` at the unit's input (identity shortcut) or the shortcut's own BatchNorm (downsample.1) -
    reads dy and the bits in the pass it makes anyway.  One sweep over the unit's widest tensor less; same numbers bit for
    bit (a masked gradient added, or read, is the same arithmetic).  The consumer sets `ready` in its forward (it runs
    before the last BatchNorm's forward), so a producer never leaves a gradient with a consumer that is not there."""
    __slots__ = ("ready", "_lazy")

    def __init__(self):
        self.ready = False
        self._lazy = None

    def push(self, dy, bits):
        self._lazy = (dy, bits)

    def pop(self):
        lazy, self._lazy = self._lazy, None
        return lazy


# PIXPRO_BN_LAZY_SHORTCUT=0: the last BatchNorm of a unit writes the shortcut gradient as a tensor (A/B switch, tests)
_LAZY_SHORTCUT = os.environ.get("PIXPRO_BN_LAZY_SHORTCUT", "1") != "0"


def residual_link(x, last_bn):
    """A link for the unit whose input is x and whose last BatchNorm is last_bn, or None when it cannot be used."""
    if _LAZY_SHORTCUT and _RELU_BITS and getattr(last_bn, "fused", False) and last_bn.training \
            and torch.is_grad_enabled() and x.requires_grad and x.is_cuda:
        return ResidualLink()
    return None


class _Fork(torch.autograd.Function):
    """x -> (x, x): the unit's input goes to its first convolution and to the shortcut.  Backward adds the two gradients
    as autograd would - except that the shortcut's may arrive over the link as (dy, bits)."""

    @staticmethod
    def forward(ctx, x, link):
        ctx.link = link
        ctx.set_materialize_grads(False)
        link.ready = True
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, ga, gb):
        lazy = ctx.link.pop()
        base = ga if gb is None else (gb if ga is None else ga + gb)
        if lazy is None:
            return base, None
        dy, bits = lazy
        G = bits.shape[0]
        C = dy.shape[1]
        rows = dy.numel() // C // G
        if base is not None and not base.is_contiguous(memory_format=torch.channels_last):
            base = base.contiguous(memory_format=torch.channels_last)
        out = torch.empty_like(dy)
        for g in range(G):
            nv.call("pixpro_bn_masked_add", nv.ptr(_part(base, g, G)), nv.ptr(_part(dy, g, G)), nv.ptr(bits[g]),
                    nv.ptr(_part(out, g, G)), rows, C, nv.stream())
        return out, None


def fork(x, link):
    return _Fork.apply(x, link)


def _part(t, g, G):
    """View g of G along the batch dimension (a contiguous block of rows in channels_last / row-major storage)."""
    if t is None or G == 1:
        return t
    n = t.shape[0] // G
    return t.narrow(0, g * n, n)


def _statistics(x, rows, C, G, group):
    """Local statistics of the G views of x (pixpro_bn_stats) and their exchange: returns (small, gathered, world) - `small`
    is the one allocation behind the per-layer records (its tail holds mean / invstd / scale_shift), `gathered[g]` the
    [world, 2C+1] records of view g as the apply kernels read them."""
    dev = x.device
    f32 = dict(dtype=torch.float32, device=dev)
    wsb = nv.load().pixpro_bn_workspace_bytes(rows, C)
    if wsb == 0:
        raise RuntimeError(f"FusedSyncBatchNorm: unsupported shape rows={rows}, C={C} (C must be a multiple of 4)")
    ws = nv.zeroed_workspace(wsb, dev)            # tickets + partial sums: one buffer per stream, reused by every layer
    # one allocation for the four small per-layer records (16-B aligned pieces: C % 4 == 0)
    small = torch.empty(G * (2 * C + 4) + G * 4 * C, **f32)
    packed = small[:G * (2 * C + 1)].view(G, 2 * C + 1)
    for g in range(G):
        nv.call("pixpro_bn_stats", nv.ptr(_part(x, g, G)), rows, C, nv.ptr(ws), wsb, nv.ptr(packed[g]), nv.stream())
    world = _world(group)
    if world > 1:
        gathered = torch.empty(world, G, 2 * C + 1, **f32)
        if dist.get_backend(group) == "gloo":      # rehearsal / tests: no all_gather_into_tensor on gloo
            parts = list(gathered.unbind(0))
            dist.all_gather(parts, packed, group=group)
            gathered = torch.stack(parts)
        else:
            dist.all_gather_into_tensor(gathered, packed, group=group)
        # the kernels read one view's records as [world, 2C+1]
        gathered = gathered.transpose(0, 1).contiguous() if G > 1 else gathered.reshape(1, world, 2 * C + 1)
    else:
        gathered = packed.reshape(G, 1, 2 * C + 1)
    return small, gathered, world


class _FusedBN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, running_mean, running_var, momentum, eps, relu, group, G, grad_mode=True,
                link_out=None, link_in=None):
        x, rows, C = _rows_view(x, "input")
        if not x.is_cuda or x.dtype != torch.float32:
            raise RuntimeError("FusedSyncBatchNorm runs on fp32 GPU tensors only; there is no CPU fallback")
        if residual is not None:
            residual, r_rows, r_C = _rows_view(residual, "residual")
            if (r_rows, r_C) != (rows, C):
                raise RuntimeError(f"residual shape {tuple(residual.shape)} does not match input {tuple(x.shape)}")
        if x.shape[0] % G:
            raise RuntimeError(f"view_groups({G}): batch {x.shape[0]} is not a multiple of the number of views")
        rows //= G
        dev = x.device
        small, gathered, world = _statistics(x, rows, C, G, group)
        y = torch.empty_like(x)
        rest = small[G * (2 * C + 4):]
        mean, invstd, scale_shift = rest[:G * C].view(G, C), rest[G * C:2 * G * C].view(G, C), rest[2 * G * C:].view(G, 2 * C)
        # The ReLU decision is recomputed from x in backward unless a residual entered it.  Then it is kept as a bit record
        # (1/32 of the tensor) where the shape has one, so that backward does not read the output y twice for it; needed
        # only when something upstream wants a gradient.
        bits, bits_bytes = None, 0
        if relu and residual is not None and _RELU_BITS and grad_mode and any(ctx.needs_input_grad[:4]):
            bits_bytes = nv.load().pixpro_bn_relu_bits_bytes(rows, C)
            if bits_bytes:
                bits = torch.empty(G, bits_bytes, dtype=torch.uint8, device=dev)
        for g in range(G):         # view after view: the running statistics see the same sequence of updates as G passes
            if bits is not None:
                nv.call("pixpro_bn_apply_bits", nv.ptr(gathered[g]), world, nv.ptr(_part(x, g, G)),
                        nv.ptr(_part(residual, g, G)), nv.ptr(weight), nv.ptr(bias), nv.ptr(running_mean),
                        nv.ptr(running_var), float(momentum), float(eps), rows, C, nv.ptr(_part(y, g, G)), nv.ptr(mean[g]),
                        nv.ptr(invstd[g]), nv.ptr(scale_shift[g]), nv.ptr(bits[g]), nv.stream())
                continue
            nv.call("pixpro_bn_apply", nv.ptr(gathered[g]), world, nv.ptr(_part(x, g, G)), nv.ptr(_part(residual, g, G)),
                    nv.ptr(weight), nv.ptr(bias), nv.ptr(running_mean), nv.ptr(running_var), float(momentum), float(eps),
                    int(relu), rows, C, nv.ptr(_part(y, g, G)), nv.ptr(mean[g]), nv.ptr(invstd[g]), nv.ptr(scale_shift[g]),
                    nv.stream())
        ctx.save_for_backward(x, y if (relu and residual is not None and bits is None) else None, weight, mean, invstd,
                              scale_shift if (relu and residual is None) else None, gathered, bits)
        ctx.meta = (rows, C, bool(relu), residual is not None, group, world, G)
        # the shortcut gradient over a ResidualLink: as its producer (this is the unit's last BatchNorm, the bits exist, the
        # consumer has announced itself) and / or as its consumer (this is the BatchNorm on the shortcut: no ReLU, no residual)
        ctx.lazy_out = link_out if (link_out is not None and link_out.ready and bits is not None
                                    and ctx.needs_input_grad[3]) else None
        ctx.lazy_in = None
        if link_in is not None and not relu and residual is None and dev.type == "cuda" and x.dim() == 4 \
                and nv.load().pixpro_bn_relu_bits_bytes(rows, C):
            ctx.lazy_in = link_in
            link_in.ready = True
            ctx.set_materialize_grads(False)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, weight, mean, invstd, scale_shift, gathered, bits = ctx.saved_tensors
        rows, C, relu, has_res, group, world, G = ctx.meta
        if ctx.lazy_in is not None:
            lazy = ctx.lazy_in.pop()
            if lazy is not None:
                # the unit's last BatchNorm left (its dy, its ReLU bits): this layer's dy is their product, never written
                if dy is not None:
                    raise RuntimeError("FusedSyncBatchNorm: a shortcut gradient arrived both as a tensor and over its link")
                dy, bits = lazy
                relu = True          # (the _bits entry points mask dy by the record; this layer itself has no ReLU)
            elif dy is None:
                return (None,) * 14
        if dy.dim() == 4 and not dy.is_contiguous(memory_format=torch.channels_last) or dy.dim() == 2 and not dy.is_contiguous():
            dy = dy.contiguous(memory_format=torch.channels_last) if dy.dim() == 4 else dy.contiguous()
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        wsb = nv.load().pixpro_bn_workspace_bytes(rows, C)
        ws = nv.zeroed_workspace(wsb, dev)
        need_w = weight is not None and ctx.needs_input_grad[1]
        need_b = ctx.needs_input_grad[2]
        small = torch.empty(G * 4 * C, **f32)         # sums [G,2C] | grad_weight [G,C] | grad_bias [G,C]
        sums = small[:2 * G * C].view(G, 2 * C)
        gw = small[2 * G * C:3 * G * C].view(G, C) if need_w else None
        gb = small[3 * G * C:].view(G, C) if need_b else None
        for g in range(G):
            if bits is not None:
                nv.call("pixpro_bn_backward_reduce_bits", nv.ptr(_part(dy, g, G)), nv.ptr(_part(x, g, G)), nv.ptr(bits[g]),
                        nv.ptr(mean[g]), nv.ptr(invstd[g]), rows, C, nv.ptr(ws), wsb, nv.ptr(sums[g]),
                        nv.ptr(gw[g] if need_w else None), nv.ptr(gb[g] if need_b else None), nv.stream())
                continue
            nv.call("pixpro_bn_backward_reduce", nv.ptr(_part(dy, g, G)), nv.ptr(_part(x, g, G)), nv.ptr(_part(y, g, G)),
                    nv.ptr(scale_shift[g] if scale_shift is not None else None), nv.ptr(mean[g]), nv.ptr(invstd[g]),
                    int(relu), rows, C, nv.ptr(ws), wsb, nv.ptr(sums[g]), nv.ptr(gw[g] if need_w else None),
                    nv.ptr(gb[g] if need_b else None), nv.stream())
        if world > 1:
            dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)     # ONE collective for all views of the layer
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dres = torch.empty_like(x) if (has_res and ctx.needs_input_grad[3] and ctx.lazy_out is None) else None
        if ctx.lazy_out is not None:
            ctx.lazy_out.push(dy, bits)       # the consumer reads dy and the bits itself; nothing is written here
        if dx is None and dres is not None:
            dx = torch.empty_like(x)
        if dx is not None:
            for g in range(G):
                # the total row count is the sum of what the ranks exchanged in forward (element 2C of each packed record)
                counts = ctypes.c_void_p(gathered[g].data_ptr() + 2 * C * 4)
                if bits is not None:
                    nv.call("pixpro_bn_backward_apply_bits", nv.ptr(_part(dy, g, G)), nv.ptr(_part(x, g, G)), nv.ptr(bits[g]),
                            nv.ptr(mean[g]), nv.ptr(invstd[g]), nv.ptr(weight), nv.ptr(sums[g]), counts, world, 2 * C + 1,
                            rows, C, nv.ptr(_part(dx, g, G)), nv.ptr(_part(dres, g, G)), nv.stream())
                    continue
                nv.call("pixpro_bn_backward_apply", nv.ptr(_part(dy, g, G)), nv.ptr(_part(x, g, G)), nv.ptr(_part(y, g, G)),
                        nv.ptr(scale_shift[g] if scale_shift is not None else None), nv.ptr(mean[g]), nv.ptr(invstd[g]),
                        nv.ptr(weight), nv.ptr(sums[g]), counts, world, 2 * C + 1, int(relu), rows, C,
                        nv.ptr(_part(dx, g, G)), nv.ptr(_part(dres, g, G)), nv.stream())
        if G > 1:                # the views' contributions in pass order, as autograd would accumulate them
            gw = gw.sum(0) if need_w else None
            gb = gb.sum(0) if need_b else None
        else:
            gw = gw[0] if need_w else None
            gb = gb[0] if need_b else None
        return dx, gw, gb, dres, None, None, None, None, None, None, None, None, None, None


# PIXPRO_BN_FUSED_POOL=0: the stem as BatchNorm + ReLU, then ATen's max pooling (A/B switch, tests)
_FUSED_POOL = os.environ.get("PIXPRO_BN_FUSED_POOL", "1") != "0"


class _FusedBNPool(torch.autograd.Function):
    """maxpool3x3/2/1(relu(batch_norm(x))) of the ResNet stem without the full-resolution tensor in between
    (csrc/sync_bn.hip: k_apply_pool, k_pool_bwd_reduce, k_pool_bwd_apply).  Statistics, collectives and the per-view
    treatment are those of _FusedBN."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, group, G):
        x, rows, C = _rows_view(x, "input")
        N, _, H, W = x.shape
        if N % G:
            raise RuntimeError(f"view_groups({G}): batch {N} is not a multiple of the number of views")
        Nv, rows = N // G, rows // G
        small, gathered, world = _statistics(x, rows, C, G, group)
        PH, PW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        pooled = torch.empty((N, C, PH, PW), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        rest = small[G * (2 * C + 4):]
        mean, invstd, scale_shift = rest[:G * C].view(G, C), rest[G * C:2 * G * C].view(G, C), rest[2 * G * C:].view(G, 2 * C)
        for g in range(G):
            nv.call("pixpro_bn_apply_pool", nv.ptr(gathered[g]), world, nv.ptr(_part(x, g, G)), nv.ptr(weight), nv.ptr(bias),
                    nv.ptr(running_mean), nv.ptr(running_var), float(momentum), float(eps), Nv, H, W, C,
                    nv.ptr(_part(pooled, g, G)), nv.ptr(mean[g]), nv.ptr(invstd[g]), nv.ptr(scale_shift[g]), nv.stream())
        ctx.save_for_backward(x, weight, mean, invstd, scale_shift, gathered)
        ctx.meta = (Nv, H, W, C, rows, group, world, G)
        return pooled

    @staticmethod
    def backward(ctx, dp):
        x, weight, mean, invstd, scale_shift, gathered = ctx.saved_tensors
        Nv, H, W, C, rows, group, world, G = ctx.meta
        if not dp.is_contiguous(memory_format=torch.channels_last):
            dp = dp.contiguous(memory_format=torch.channels_last)
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        wsb = nv.load().pixpro_bn_workspace_bytes(rows, C)
        ws = nv.zeroed_workspace(wsb, dev)
        need_w = weight is not None and ctx.needs_input_grad[1]
        need_b = ctx.needs_input_grad[2]
        small = torch.empty(G * 4 * C, **f32)
        sums = small[:2 * G * C].view(G, 2 * C)
        gw = small[2 * G * C:3 * G * C].view(G, C) if need_w else None
        gb = small[3 * G * C:].view(G, C) if need_b else None
        for g in range(G):
            nv.call("pixpro_bn_pool_backward_reduce", nv.ptr(_part(dp, g, G)), nv.ptr(_part(x, g, G)), nv.ptr(scale_shift[g]),
                    nv.ptr(mean[g]), nv.ptr(invstd[g]), Nv, H, W, C, nv.ptr(ws), wsb, nv.ptr(sums[g]),
                    nv.ptr(gw[g] if need_w else None), nv.ptr(gb[g] if need_b else None), nv.stream())
        if world > 1:
            dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            for g in range(G):
                counts = ctypes.c_void_p(gathered[g].data_ptr() + 2 * C * 4)
                nv.call("pixpro_bn_pool_backward_apply", nv.ptr(_part(dp, g, G)), nv.ptr(_part(x, g, G)), nv.ptr(scale_shift[g]),
                        nv.ptr(mean[g]), nv.ptr(invstd[g]), nv.ptr(weight), nv.ptr(sums[g]), counts, world, 2 * C + 1,
                        Nv, H, W, C, nv.ptr(_part(dx, g, G)), nv.stream())
        if G > 1:
            gw = gw.sum(0) if need_w else None
            gb = gb.sum(0) if need_b else None
        else:
            gw = gw[0] if need_w else None
            gb = gb[0] if need_b else None
        return dx, gw, gb, None, None, None, None, None, None


def _flush_counter_hook(module, prefix, keep_vars):
    module._flush_counter()


def flush_counters(module):
    """Apply every pending `num_batches_tracked` increment below `module`.  Call it before anything that reads buffers
    WITHOUT going through attribute access or state_dict() - `module.buffers()` / `named_buffers()` / `_buffers` walks,
    DistributedDataParallel(broadcast_buffers=True), buffer-copying EMA code: those see the tensor as it is."""
    for m in module.modules():
        if isinstance(m, FusedSyncBatchNorm):
            m._flush_counter()
    return module


class FusedSyncBatchNorm(nn.modules.batchnorm._BatchNorm):
    """Drop-in for nn.BatchNorm2d / nn.SyncBatchNorm with optional fused tail: `bn(x, residual=None, relu=False)`.

    Limitation: `num_batches_tracked` is incremented lazily (on attribute access, state_dict(), load); a reader that walks
    `buffers()` / `named_buffers()` directly sees a stale counter unless it calls `sync_bn.flush_counters(model)` first
    (running_mean / running_var are always current)."""

    fused = True

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True, process_group=None):
        super().__init__(num_features, eps, momentum, affine, track_running_stats)
        self.process_group = process_group
        if momentum is None:
            raise NotImplementedError("cumulative moving average (momentum=None) is not provided")
        # `num_batches_tracked` is bookkeeping here (the momentum is fixed): a device-side `add_` per forward call is one
        # tiny launch per layer - 216 per ResNet-50 step, 1.6 ms.  The increments are collected on the host and applied
        # whenever the buffer is looked at (attribute access, state_dict): same values, no launches in the step.
        self._nbt_pending = 0
        self.register_state_dict_pre_hook(_flush_counter_hook)     # a plain function: the module stays picklable

    def _flush_counter(self):
        nbt = self._buffers.get("num_batches_tracked")
        if self._nbt_pending and nbt is not None:
            nbt.add_(self._nbt_pending)
        self._nbt_pending = 0

    def __getattr__(self, name):
        if name == "num_batches_tracked" and self.__dict__.get("_nbt_pending"):
            self._flush_counter()
        return super().__getattr__(name)

    def _load_from_state_dict(self, *args, **kwargs):
        self._nbt_pending = 0                   # a loaded counter replaces whatever was pending
        return super()._load_from_state_dict(*args, **kwargs)

    def _check_input_dim(self, input):
        if input.dim() not in (2, 4):
            raise ValueError(f"expected 2D or 4D input (got {input.dim()}D input)")

    def forward(self, input, residual=None, relu=False, link_out=None, link_in=None):
        self._check_input_dim(input)
        if self.training or not self.track_running_stats:
            if self.training and self.track_running_stats and self._buffers.get("num_batches_tracked") is not None:
                self._nbt_pending += _VIEW_GROUPS
            rm = self.running_mean if (self.training and self.track_running_stats) else None
            rv = self.running_var if (self.training and self.track_running_stats) else None
            # (grad mode is always off inside Function.forward: it is read here)
            return _FusedBN.apply(input, self.weight, self.bias, residual, rm, rv, self.momentum, self.eps, relu,
                                  self.process_group, _VIEW_GROUPS, torch.is_grad_enabled(), link_out, link_in)
        out = F.batch_norm(input, self.running_mean, self.running_var, self.weight, self.bias, False, 0.0, self.eps)
        if residual is not None:
            out = out + residual
        return F.relu(out) if relu else out

    def forward_pool(self, input):
        """maxpool3x3/2/1(relu(bn(input))) in one pass (training mode; see bn_relu_pool)."""
        self._check_input_dim(input)
        if self._buffers.get("num_batches_tracked") is not None:
            self._nbt_pending += _VIEW_GROUPS
        return _FusedBNPool.apply(input, self.weight, self.bias, self.running_mean, self.running_var, self.momentum, self.eps,
                                  self.process_group, _VIEW_GROUPS)

    @classmethod
    def convert(cls, module, process_group=None):
        """Replace every BatchNorm / SyncBatchNorm below `module` (in place on the children, like
        nn.SyncBatchNorm.convert_sync_batchnorm); parameters and buffers are shared, not copied."""
        out = module
        if isinstance(module, nn.modules.batchnorm._BatchNorm) and not isinstance(module, cls):
            out = cls(module.num_features, module.eps, module.momentum, module.affine, module.track_running_stats,
                      process_group)
            if module.affine:
                out.weight, out.bias = module.weight, module.bias
            out.running_mean, out.running_var = module.running_mean, module.running_var
            out.num_batches_tracked = module.num_batches_tracked
            out.training = module.training
        for name, child in module.named_children():
            out.add_module(name, cls.convert(child, process_group))
        return out


def bn_relu_pool(bn, x, pool):
    """`pool(relu(bn(x)))` of the ResNet stem: one fused pass when `bn` is a FusedSyncBatchNorm in training mode and `pool` is
    the 3x3 / stride 2 / padding 1 max pooling (the kernels' only shape); the separate ops otherwise."""
    if (_FUSED_POOL and getattr(bn, "fused", False) and bn.training and bn.track_running_stats and x.is_cuda and x.dim() == 4
            and isinstance(pool, nn.MaxPool2d) and _pair(pool.kernel_size) == (3, 3) and _pair(pool.stride) == (2, 2)
            and _pair(pool.padding) == (1, 1) and _pair(pool.dilation) == (1, 1) and not pool.ceil_mode
            and not pool.return_indices
            and nv.load().pixpro_bn_pool_supported(x.shape[0] // _VIEW_GROUPS, x.shape[2], x.shape[3], x.shape[1])):
        return bn.forward_pool(x)
    return pool(bn_act(bn, x, relu=True))


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def bn_act(bn, x, relu=False, residual=None, link_out=None, link_in=None):
    """`relu?(bn(x) (+ residual))` through the fused module when `bn` is one, through the stock ops otherwise.
    link_out / link_in: the unit's ResidualLink on its last BatchNorm / on the BatchNorm of its shortcut."""
    if getattr(bn, "fused", False):
        return bn(x, residual, relu, link_out, link_in)
    out = bn(x)
    if residual is not None:
        out = out + residual
    return F.relu(out, inplace=True) if relu else out

"""Weight gradients of the encoder convolutions on a second HIP stream.

In the backward pass of the pre-training step (reference: main_pretrain.py:259-262, one `loss.backward()`) every
convolution produces two things: the input gradient, which the next layer's backward needs at once, and the weight
gradient, which nothing needs before `optimizer.step()`.  PyTorch computes both on the one stream of the backward pass,
so the HBM-bound BatchNorm backward kernels of the next layer wait for a matrix-core-bound weight-gradient convolution
they do not depend on.  `Conv2dSplit` launches the weight-gradient convolution on a side stream and returns at once:
on MI355X the two kinds of kernels then share the chip (the streaming kernels fill what the implicit-GEMM work-groups
leave idle).  `join()` - called by the training step before the optimizer reads the gradients - makes the main stream
wait for the side stream.

Conditions (checked, otherwise the stock single-stream path runs):
  * one use of a weight per backward pass: autograd sums the gradients of a weight that was used twice with an `add` on
    the MAIN stream, which would read a tensor the side stream is still writing.  PixPro runs two views through each
    encoder (contrast/models/PixPro.py:376-384): use it with `--bn-view-batch` (both views in one pass).  A second use
    inside one pass is detected and falls back to waiting for the side stream before the gradient is returned.
  * one rank: DistributedDataParallel's reducer copies a gradient into its bucket the moment autograd delivers it.

Numerics: the same MIOpen kernels on the same operands; only the stream changes.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

_side = {}            # device index -> side stream
_dirty = set()        # device indices with weight-gradient work the main stream has not waited for yet
_enabled = True


def set_enabled(flag):
    global _enabled
    _enabled = bool(flag)


def _side_stream(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    s = _side.get(idx)
    if s is None:
        s = _side[idx] = torch.cuda.Stream(device=device)
    return idx, s


def join():
    """Main stream waits for every weight gradient still in flight on a side stream (call before optimizer.step())."""
    for idx in list(_dirty):
        torch.cuda.current_stream(idx).wait_stream(_side[idx])
    _dirty.clear()


class _ConvSplitBackward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, stride, padding, dilation, groups):
        ctx.save_for_backward(x, weight)
        ctx.cfg = (stride, padding, dilation, groups)
        return F.conv2d(x, weight, None, stride, padding, dilation, groups)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        stride, padding, dilation, groups = ctx.cfg
        dx = dw = None

        def conv_bwd(mask):
            return torch.ops.aten.convolution_backward(dy, x, weight, None, list(stride), list(padding), list(dilation),
                                                       False, [0, 0], groups, mask)
        if ctx.needs_input_grad[1]:
            main = torch.cuda.current_stream(dy.device)
            idx, side = _side_stream(dy.device)
            side.wait_stream(main)                     # dy (and everything before it) is ready
            with torch.cuda.stream(side):
                dw = conv_bwd([False, True, False])[1]
            # the allocator must not hand these blocks to later main-stream work while the side stream still reads them
            dy.record_stream(side)
            x.record_stream(side)
            dw.record_stream(main)
            _dirty.add(idx)
            # a weight used twice in one pass: autograd is about to add the two gradients on the main stream
            # (or to accumulate into a gradient that is already there)
            pending = getattr(weight, "_wgrad_pending", 0)
            if pending or weight.grad is not None:
                main.wait_stream(side)
            weight._wgrad_pending = pending + 1
        if ctx.needs_input_grad[0]:
            dx = conv_bwd([True, False, False])[0]
        return dx, dw, None, None, None, None


class Conv2dSplit(nn.Conv2d):
    """nn.Conv2d (same parameters, same state-dict names) whose weight gradient is computed on the side stream."""

    def forward(self, x):
        if (_enabled and self.bias is None and x.is_cuda and torch.is_grad_enabled() and self.weight.requires_grad
                and self.padding_mode == "zeros" and not isinstance(self.padding, str)):
            self.weight._wgrad_pending = 0             # a new forward pass: nothing in flight for this weight
            return _ConvSplitBackward.apply(x, self.weight, self.stride, self.padding, self.dilation, self.groups)
        return super().forward(x)


def convert(module):
    """Turn every bias-free nn.Conv2d below `module` into a Conv2dSplit, in place (parameters untouched)."""
    for m in module.modules():
        if type(m) is nn.Conv2d and m.bias is None:
            m.__class__ = Conv2dSplit
    return module

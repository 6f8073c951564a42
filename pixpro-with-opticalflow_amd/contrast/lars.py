"""LARS-wrapped SGD for the pre-training step, without per-parameter host syncs.

Reference semantics (contrast/lars.py:109-151 + torch.optim.SGD): per tensor, `g += wd*p`; for tensors in
the decayed (>= 2-D) group `g *= trust*|p|/(|g|+eps)` when both norms are non-zero; then SGD with
momentum (no dampening, no nesterov).  The reference does this with two norm kernels and one
GPU->CPU sync per tensor (56 per step for ResNet-50 PixPro); here the whole step is two launches
over a device pointer table (multi_tensor.LarsSgdStep).  The class keeps the `torch.optim.Optimizer`
surface that main_pretrain.py and the LR schedulers use (param_groups, state_dict, zero_grad, step).
"""
import torch

from .multi_tensor import LarsSgdStep

__all__ = ['LARS', 'add_weight_decay']


def add_weight_decay(model, weight_decay=1e-5, skip_list=()):
    """Split trainable params into the no-decay / LARS-exempt group (1-D tensors: biases, norm gains)
    and the decayed, LARS-adapted group (contrast/lars.py:7-31)."""
    plain, decayed = [], []
    for name, p in model.named_parameters():
        if p.requires_grad:
            (plain if p.ndim == 1 or name in skip_list else decayed).append(p)
    return [{'params': plain, 'weight_decay': 0, 'ignore': True},
            {'params': decayed, 'weight_decay': weight_decay, 'ignore': False}]


class LARS(torch.optim.Optimizer):
    """`LARS(torch.optim.SGD(groups, lr=..., momentum=...))` as in the reference, or
    `LARS(groups, lr=..., momentum=...)` directly."""

    def __init__(self, optimizer_or_params, lr=None, momentum=0.9, eps=1e-8, trust_coef=0.001):
        if eps < 0.0 or trust_coef < 0.0:
            raise ValueError("eps and trust_coef must be non-negative")
        if isinstance(optimizer_or_params, torch.optim.Optimizer):
            inner = optimizer_or_params
            groups = [{k: v for k, v in g.items()} for g in inner.param_groups]
            defaults = dict(inner.defaults)
        else:
            groups = list(optimizer_or_params)
            defaults = dict(lr=lr, momentum=momentum, weight_decay=0)
        for g in groups:
            for k in ("dampening", "nesterov"):
                if g.get(k):
                    raise NotImplementedError("LARS here implements plain momentum SGD only")
        defaults.setdefault("weight_decay", 0)
        super().__init__(groups, defaults)
        self.eps, self.trust_coef = eps, trust_coef
        self._fused = None

    def _build(self):
        params, wds, adapts = [], [], []
        lr0, mom0 = self.param_groups[0]['lr'], self.param_groups[0].get('momentum', 0.0)
        for g in self.param_groups:
            if g.get('momentum', 0.0) != mom0:
                raise NotImplementedError("all groups must share one momentum")
            ignore = g.get('ignore', None)
            for p in g['params']:
                params.append(p)
                wds.append(float(g['weight_decay']))
                adapts.append(ignore is not None and not ignore)
        self._fused = LarsSgdStep(params, wds, adapts, mom0, self.trust_coef, self.eps)
        for p, b in zip(params, self._fused.bufs):
            self.state[p]['momentum_buffer'] = b
        del lr0

    def zero_grad(self, set_to_none=True):
        # None instead of zeros (torch's own default since 2.0; the reference's torch 1.8 zeroes - same gradients either
        # way): 161 fill launches and 161 accumulate-into-zeros adds per ResNet-50 step disappear, autograd hands its
        # gradient tensors over as they are.  They may sit at new addresses: the fused step's pointer table follows
        # (multi_tensor._Table, an asynchronous 5-KB upload when a pointer moved).
        super().zero_grad(set_to_none=set_to_none)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        if self._fused is None:
            self._build()
        lrs = {g['lr'] for g in self.param_groups}
        if len(lrs) != 1:
            raise NotImplementedError("all groups must share one learning rate")
        self._fused.step(lrs.pop())
        return loss

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        saved = {p: self.state[p].get('momentum_buffer') for g in self.param_groups for p in g['params']}
        self._build()                      # fresh fused buffers (registered as the state's momentum_buffer)
        have = 0
        for p, b in zip(self._fused.params, self._fused.bufs):
            if saved.get(p) is not None:
                b.copy_(saved[p])
                have += 1
        if have:
            self._fused.first = False

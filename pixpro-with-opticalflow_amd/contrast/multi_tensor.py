"""Multi-tensor HIP updates (one launch over a device-side pointer table).

`ema_update`     momentum update of the key encoder           <- PixPro.py:321-337
`LarsSgdStep`    LARS scaling + SGD-momentum step, sync-free   <- contrast/lars.py:109-151
"""
import torch

from . import _native as nv

MAX_TENSORS = 2048     # per C call (include/pixpro_hip.h)


class _Table:
    """Device arrays of pointers / sizes for a list of tensors; rebuilt only when a pointer moves.  A rebuild is an
    asynchronous copy from pinned memory on the current stream (gradients that autograd allocates anew every step may move:
    the step must not wait for the device because of it); the table in use by an earlier launch is a different tensor."""

    def __init__(self):
        self.key = None
        self.dev = None

    def get(self, groups, device):
        key = tuple(t.data_ptr() for g in groups for t in g)
        if key != self.key:
            n = len(groups[0])
            host = torch.tensor([[t.data_ptr() for t in g] for g in groups] + [[t.numel() for t in groups[0]]],
                                dtype=torch.int64)
            if torch.device(device).type == "cuda":
                host = host.pin_memory()
            self.dev = host.to(device, non_blocking=True)
            self.key = key
            self.n = n
        return self.dev


_ema_table = _Table()


def _dense(t):
    return t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))


def _check(ts, what):
    for t in ts:
        if not (t.is_cuda and t.dtype == torch.float32 and _dense(t)):
            raise RuntimeError(f"{what}: needs dense fp32 CUDA tensors (no CPU fallback)")


def _same_layout(groups, what):
    """element-wise kernels walk the storage linearly: every group must share the first one's strides"""
    for g in groups[1:]:
        for a, b in zip(groups[0], g):
            if a.stride() != b.stride() or a.shape != b.shape:
                raise RuntimeError(f"{what}: tensors of one slot differ in shape/strides {a.stride()} vs {b.stride()}")


@torch.no_grad()
def ema_update(params_q, params_k, m):
    """param_k <- param_k * m + param_q * (1 - m) for every pair, one launch."""
    pq = [p.data for p in params_q]
    pk = [p.data for p in params_k]
    _check(pq + pk, "ema_update")
    _same_layout([pq, pk], "ema_update")
    tab = _ema_table.get([pq, pk], pq[0].device)
    for a in range(0, len(pq), MAX_TENSORS):        # the kernel's chunk table holds MAX_TENSORS entries per call
        b = min(a + MAX_TENSORS, len(pq))
        nv.call("pixpro_ema_update", nv.ptr(tab[0][a:b]), nv.ptr(tab[1][a:b]), nv.ptr(tab[2][a:b]), b - a, float(m),
                nv.stream())


class LarsSgdStep:
    """Fused step for a fixed list of parameters.  wd/adapt per tensor as contrast/lars.py groups them
    (1-D params: no decay, no adaptation)."""

    def __init__(self, params, weight_decays, adapts, momentum, trust_coef=0.001, eps=1e-8):
        self.params = list(params)
        self.momentum, self.trust, self.eps = momentum, trust_coef, eps
        dev = self.params[0].device
        _check([p.data for p in self.params], "LarsSgdStep")
        self.bufs = [torch.zeros_like(p.data) for p in self.params]
        self.wd = torch.tensor(weight_decays, dtype=torch.float32, device=dev)
        self.adapt = torch.tensor([int(a) for a in adapts], dtype=torch.int32, device=dev)
        self.norms2 = torch.zeros(2 * len(self.params), dtype=torch.float32, device=dev)
        self.table = _Table()
        self.first = True

    @torch.no_grad()
    def step(self, lr):
        grads = []
        for p in self.params:
            if p.grad is None:
                raise RuntimeError("LarsSgdStep: every parameter needs a gradient")
            grads.append(p.grad if p.grad.stride() == p.stride() else p.grad.contiguous(
                memory_format=torch.channels_last if p.dim() == 4 and not p.is_contiguous() else torch.contiguous_format))
        tab = self.table.get([[p.data for p in self.params], grads, self.bufs], self.params[0].device)
        for a in range(0, len(self.params), MAX_TENSORS):
            b = min(a + MAX_TENSORS, len(self.params))
            nv.call("pixpro_lars_sgd_step", nv.ptr(tab[0][a:b]), nv.ptr(tab[1][a:b]), nv.ptr(tab[2][a:b]),
                    nv.ptr(tab[3][a:b]), nv.ptr(self.wd[a:b]), nv.ptr(self.adapt[a:b]), b - a, float(lr),
                    float(self.momentum), float(self.trust), float(self.eps), int(self.first),
                    nv.ptr(self.norms2[2 * a:2 * b]), nv.stream())
        self.first = False

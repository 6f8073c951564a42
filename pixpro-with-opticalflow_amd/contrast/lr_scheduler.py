"""Per-iteration LR schedule: linear warm-up then cosine / multi-step (reference:
contrast/lr_scheduler.py).  Written as one closed-form LambdaLR so it can be evaluated at any step."""
import math

from torch.optim.lr_scheduler import LambdaLR


def lr_factor(step, n_iter_per_epoch, args):
    """Multiplier on the base LR after `step` scheduler steps."""
    warm = args.warmup_epoch * n_iter_per_epoch
    if warm > 0 and step <= warm:
        mult = args.warmup_multiplier
        return ((mult - 1.) * step / warm + 1.) / mult
    t = step - warm
    if "cosine" in args.lr_scheduler:
        t_max = (args.epochs - args.warmup_epoch) * n_iter_per_epoch
        return None, t, t_max  # resolved in get_scheduler (needs base lr for eta_min)
    if "step" in args.lr_scheduler:
        k = sum(1 for m in args.lr_decay_epochs if t >= (m - args.warmup_epoch) * n_iter_per_epoch)
        return args.lr_decay_rate ** k
    raise NotImplementedError(f"scheduler {args.lr_scheduler} not supported")


def get_scheduler(optimizer, n_iter_per_epoch, args):
    eta_min = 0.000001
    base_lrs = [g['lr'] for g in optimizer.param_groups]

    def make(base):
        def fn(step):
            f = lr_factor(step, n_iter_per_epoch, args)
            if isinstance(f, tuple):
                _, t, t_max = f
                lr = eta_min + (base - eta_min) * (1 + math.cos(math.pi * t / t_max)) / 2
                return lr / base
            return f
        return fn

    return LambdaLR(optimizer, [make(b) for b in base_lrs])

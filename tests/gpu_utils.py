import torch


def to_dev(x, dev):
    if isinstance(x, torch.Tensor):
        return x.to(dev)
    if isinstance(x, (list, tuple)):
        return type(x)(to_dev(t, dev) for t in x)
    return x


def close(a, b, rel=1e-4, abs_=1e-9):
    """max-abs error relative to the reference's max-abs (the SURVEY.md §8c gradient bar)."""
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = (a - b).abs().max().item()
    bound = rel * b.abs().max().item() + abs_
    assert err <= bound, f"max err {err:.3e} > {bound:.3e}"
    return err

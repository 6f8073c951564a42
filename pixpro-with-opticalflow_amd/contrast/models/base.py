import torch.nn as nn


class BaseModel(nn.Module):
    """Constructor contract `(base_encoder, args)` of the reference (contrast/models/base.py).
    The reference builds a throw-away encoder here that PixPro overwrites (defect D8); sub-classes
    build their own encoders instead."""

    def __init__(self, base_encoder, args):
        super().__init__()
        self.feature_dim = getattr(args, "feature_dim", 256)
        self.head_type = getattr(args, "head_type", "early_return")

    def forward(self, x1, x2):
        raise NotImplementedError

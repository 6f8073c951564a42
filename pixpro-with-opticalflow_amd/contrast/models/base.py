"""Constructor contract shared by the pre-training models: ``Model(base_encoder, args)`` where
``base_encoder`` is a ResNet factory and ``args`` the parsed options (reference: contrast/models/base.py).
The reference's base class also instantiates an encoder that PixPro immediately replaces (it only consumes
RNG and memory); here sub-classes create the encoders they need."""
import torch.nn as nn


class BaseModel(nn.Module):
    def __init__(self, base_encoder, args):
        super().__init__()
        self.feature_dim = getattr(args, "feature_dim", 256)
        self.head_type = getattr(args, "head_type", "early_return")

    def forward(self, *views):
        raise NotImplementedError("sub-classes define the pre-training forward pass")

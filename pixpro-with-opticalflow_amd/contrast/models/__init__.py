"""Model registry looked up by the entry point as ``models.__dict__[args.model]``; only the PixPro
pre-training model is part of this build (the reference registers nothing else either)."""
from . import PixPro as _pixpro_module

PixPro = _pixpro_module.PixPro
__all__ = ("PixPro",)

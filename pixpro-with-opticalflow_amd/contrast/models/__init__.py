from .PixPro import PixPro

__all__ = ['PixPro']

"""ctypes binding of libpixpro_hip.so (include/pixpro_hip.h).

There is deliberately no fallback: if the shared library is missing, or a tensor is not a
contiguous fp32 CUDA tensor, the call raises.  PyTorch is used here only as the owner of device
memory and of the current HIP stream.
"""
import ctypes
import os

import torch

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_PKG_DIR, "lib", "libpixpro_hip.so")

_vp = ctypes.c_void_p
_i32 = ctypes.c_int32
_i64 = ctypes.c_int64
_f32 = ctypes.c_float
_f64 = ctypes.c_double
_sz = ctypes.c_size_t


class Problem(ctypes.Structure):
    _fields_ = [("N", _i32), ("C", _i32), ("H", _i32), ("W", _i32), ("Hf", _i32), ("Wf", _i32),
                ("Ho", _i32), ("Wo", _i32), ("p", _f32), ("clamp", _f32), ("pos_ratio", _f32),
                ("normalize_k", _i32), ("ndir", _i32), ("grad_scale", _f32), ("layout", _i32),
                ("lo_h", _i32), ("lo_w", _i32), ("lazy_mask", _i32), ("alpha1", _f32), ("lo_frames", _i32),
                ("alpha2", _f64)]


class Dir(ctypes.Structure):
    _fields_ = [(n, _vp) for n in ("x", "v", "k", "coord_q", "coord_k", "flow", "mask", "size_dev", "lo_own", "lo_opp",
                                    "pred", "dx", "dv")]


_PP = ctypes.POINTER(Problem)
_DP = ctypes.POINTER(Dir)

# name -> (restype, argtypes): every symbol include/pixpro_hip.h declares
SIGNATURES = {
    "pixpro_version": (ctypes.c_int, []),
    "pixpro_last_error": (ctypes.c_char_p, []),
    "pixpro_ppm_loss_workspace_bytes": (_sz, [_PP]),
    "pixpro_ppm_loss_layout_supported": (ctypes.c_int, [_PP]),
    "pixpro_ppm_loss_generic_workspace_bytes": (_sz, [_PP]),
    "pixpro_ppm_loss": (ctypes.c_int, [_PP, _DP, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pixpro_ppm_loss_generic": (ctypes.c_int, [_PP, _DP, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pixpro_scale_by_device_scalar": (ctypes.c_int, [_vp, _sz, _vp, _vp]),
    "pixpro_featprop_workspace_bytes": (_sz, [_PP]),
    "pixpro_featprop_fwd": (ctypes.c_int, [_PP, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pixpro_featprop_bwd": (ctypes.c_int, [_PP, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pixpro_regression_loss_workspace_bytes": (_sz, [_PP]),
    "pixpro_regression_loss": (ctypes.c_int, [_PP, _DP, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pixpro_add_optical_flow": (ctypes.c_int, [_PP, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pixpro_upflow8": (ctypes.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    "pixpro_concat_flow": (ctypes.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "pixpro_fbc_mask": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _f64, _i32, _vp]),
    "pixpro_flow_prepare": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32,
                                           _f32, _f64, _vp]),
    "pixpro_flow_prepare_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32, _i32]),
    "pixpro_flow_prepare_ws": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32,
                                              _f32, _f64, _vp, _sz, _vp]),
    "pixpro_mask_ratio": (ctypes.c_int, [_vp, _vp, _i32, _i64, _vp]),
    "pixpro_ema_update": (ctypes.c_int, [_vp, _vp, _vp, _i32, _f64, _vp]),
    "pixpro_lars_sgd_step": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _f32, _f32, _f32, _f32, _i32,
                                            _vp, _vp]),
    "pixpro_corr_lookup": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "pixpro_bn_workspace_bytes": (_sz, [_i64, _i32]),
    "pixpro_bn_stats": (ctypes.c_int, [_vp, _i64, _i32, _vp, _sz, _vp, _vp]),
    "pixpro_bn_apply": (ctypes.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _i32, _i64, _i32,
                                       _vp, _vp, _vp, _vp, _vp]),
    "pixpro_bn_backward_reduce": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _vp, _sz, _vp, _vp,
                                                 _vp, _vp]),
    "pixpro_bn_backward_apply": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i64, _i32,
                                                _vp, _vp, _vp]),
    "pixpro_bn_relu_bits_bytes": (_sz, [_i64, _i32]),
    "pixpro_bn_apply_bits": (ctypes.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _i64, _i32,
                                            _vp, _vp, _vp, _vp, _vp, _vp]),
    "pixpro_bn_backward_reduce_bits": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _sz, _vp, _vp, _vp, _vp]),
    "pixpro_bn_backward_apply_bits": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i64, _i32,
                                                     _vp, _vp, _vp]),
    "pixpro_bn_masked_add": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "pixpro_bn_pool_supported": (ctypes.c_int, [_i32, _i32, _i32, _i32]),
    "pixpro_bn_apply_pool": (ctypes.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _i32, _i32, _i32, _i32,
                                            _vp, _vp, _vp, _vp, _vp]),
    "pixpro_bn_pool_backward_reduce": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _sz, _vp, _vp, _vp,
                                                      _vp]),
    "pixpro_bn_pool_backward_apply": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _i32, _i32,
                                                     _vp, _vp]),
}

_lib = None


def load():
    """Load the shared library (no GPU needed for loading) and attach the signatures."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C "
                "pixpro-with-opticalflow_amd/csrc`).  There is no CPU fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {lib.pixpro_last_error().decode()}")


def stream():
    return _vp(torch.cuda.current_stream().cuda_stream)


def is_nhwc(t):
    """4-D tensor stored channels_last (and not also NCHW-contiguous, as 1x1 grids are)."""
    return t.dim() == 4 and not t.is_contiguous() and t.is_contiguous(memory_format=torch.channels_last)


def dev_f32(t, name, shape=None, nhwc=False):
    """Validate a device fp32 operand (the reference's implicit torch errors become explicit).  nhwc=True keeps /
    makes the tensor channels_last-dense instead of NCHW-contiguous."""
    if not isinstance(t, torch.Tensor):
        raise RuntimeError(f"{name}: expected a tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: the PixPro hot path runs on the GPU only (got a {t.device} tensor); "
                           "there is no CPU fallback")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: expected float32, got {t.dtype}")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise RuntimeError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    if nhwc:
        return t if is_nhwc(t) else t.contiguous(memory_format=torch.channels_last)
    return t if t.is_contiguous() else t.contiguous()


def dev_mask(t, name, shape):
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a CUDA tensor")
    if t.dtype not in (torch.bool, torch.uint8):
        raise RuntimeError(f"{name}: expected bool/uint8, got {t.dtype}")
    if tuple(t.shape) != tuple(shape):
        raise RuntimeError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t if t.is_contiguous() else t.contiguous()


def ptr(t):
    return _vp(t.data_ptr()) if t is not None else None


def workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


_zeroed = {}


def zeroed_workspace(nbytes, device):
    """A persistent workspace per (device, current stream) that was zero when it was created: for entry points whose kernels
    need zeroed tickets on entry and leave them zeroed (pixpro_bn_stats, pixpro_bn_backward_reduce).  Calls on one stream
    are ordered, so they share the buffer; another stream gets its own.  Grows by reallocation (the caching allocator
    keeps the old block alive until the stream has passed the kernels that use it)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)
    buf = _zeroed.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.zeros(max(int(nbytes), 4096), dtype=torch.uint8, device=device)
        _zeroed[key] = buf
    return buf

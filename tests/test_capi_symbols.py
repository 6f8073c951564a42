"""The C-ABI library loads without a GPU and exports every symbol include/pixpro_hip.h declares."""
import ctypes
import os
import re

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pixpro_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pixpro_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    from contrast import _native
    lib = _native.load()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
    assert sorted(_native.SIGNATURES) == names       # the ctypes table binds exactly the header
    assert lib.pixpro_version() >= 100


def test_struct_layout_matches_header():
    from contrast import _native
    assert ctypes.sizeof(_native.Problem) == 88          # 20 x 4 B, one double
    assert ctypes.sizeof(_native.Dir) == 13 * ctypes.sizeof(ctypes.c_void_p)


def test_argument_errors_without_gpu():
    """Validation paths return error codes (no launch, no GPU needed)."""
    from contrast import _native
    lib = _native.load()
    pb = _native.Problem()
    assert lib.pixpro_ppm_loss(ctypes.byref(pb), None, None, None, None, None, None, 0, None) == 1
    assert b"null" in lib.pixpro_last_error()
    assert lib.pixpro_upflow8(None, None, 1, 2, 2, None) == 1
    assert lib.pixpro_concat_flow(ctypes.c_void_p(16), ctypes.c_void_p(16), 0, 1, 4, 4, 0, None) == 1
    pb.N, pb.C, pb.H, pb.W, pb.ndir = 4, 256, 7, 7, 2
    assert lib.pixpro_ppm_loss_workspace_bytes(ctypes.byref(pb)) == 0           # MFMA kernel, no workspace
    pb.H = pb.W = 14
    need = lib.pixpro_ppm_loss_workspace_bytes(ctypes.byref(pb))
    assert need == lib.pixpro_ppm_loss_generic_workspace_bytes(ctypes.byref(pb)) > 6 * 8 * 256 * 196 * 4


def test_missing_library_fails_loudly(monkeypatch):
    from contrast import _native
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", "/nonexistent/libpixpro_hip.so")
    try:
        _native.load()
    except RuntimeError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("loading a missing library must raise")

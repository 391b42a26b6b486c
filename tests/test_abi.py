"""CPU-side checks of the C-ABI boundary: the library builds for gfx950, loads, and exports
every symbol include/artdeco_hip.h declares (no compute calls -- there is no GPU here)."""
import ctypes
import os
import subprocess

from artdeco_amd import _lib, build


def test_header_parses_and_lists_symbols():
    protos = _lib.parse_header()
    assert "adk_abi_version" in protos and protos["adk_abi_version"] == []
    for must in ("adk_fused_ssim_fwd", "adk_fused_ssim_bwd", "adk_adam_update", "adk_adam_update_basic"):
        assert must in protos, must
    # every pointer/scalar type used in the header is one the binding understands
    for name, args in protos.items():
        for t, _ in args:
            assert t == "ptr" or t in _lib._CTYPE, (name, t)


def test_library_exports_every_declared_symbol(lib):
    protos = _lib.parse_header()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(raw, name), f"{name} declared in artdeco_hip.h but not exported"
    assert lib.adk_abi_version() == _lib._header_abi_version()


def test_library_is_gfx950_code_object():
    path = build.lib_path()
    assert os.path.exists(path)
    out = subprocess.run(["strings", "-n", "6", path], capture_output=True, text=True).stdout
    assert "gfx950" in out


def test_no_cpu_fallback_in_product_path():
    """The product package must not import the oracle (parity claims depend on it)."""
    pkg = os.path.dirname(os.path.abspath(_lib.__file__))
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, os.path.join(root, f)


def test_cpu_tensor_is_rejected_loudly():
    import pytest
    import torch
    from fused_ssim import fused_ssim
    with pytest.raises(_lib.AdkError):
        fused_ssim(torch.rand(1, 3, 16, 16), torch.rand(1, 3, 16, 16))

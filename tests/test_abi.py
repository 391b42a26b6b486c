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


def test_argument_validation_returns_error_codes_without_touching_the_gpu(lib):
    """Every entry point rejects bad arguments (negative sizes, NULL where data is required, unsupported shapes)
    with a negative adk error code BEFORE any HIP call, so these calls are safe on a machine without a GPU."""
    EINVAL, EUNSUPPORTED = -1, -3
    N = None  # NULL
    assert lib.adk_scatter_argmax(-1, N, 2, N, 4, 0, N, N, N) == EINVAL
    assert lib.adk_scatter_argmax(4, N, 2, N, 4, 0, N, N, N) == EINVAL            # out / arg missing
    assert lib.adk_gauss_newton(7, 2, 1, 8, N, N, N, N, N, N, N, N, N, 0, 0, 0, 0.0, 1.0, 1.0, 0.0, 0.0, 10, 1e-8, 1,
                                N, N, N, N, 0, N) == EINVAL                        # unknown factor kind
    assert lib.adk_gauss_newton(1, 1, 0, 8, N, N, N, N, N, N, N, N, N, 0, 0, 0, 0.0, 1.0, 1.0, 0.0, 0.0, 10, 1e-8, 1,
                                N, N, N, N, 0, N) == 0                             # nothing to optimise: one pose, fixed
    assert lib.adk_gn_workspace_bytes(-1, 0, 0) == EINVAL and lib.adk_gn_workspace_bytes(16, 54, 196608) > 0
    assert lib.adk_photometric_workspace_bytes(-1, 4) == EINVAL and lib.adk_photometric_workspace_bytes(1920, 1080) > 0
    assert lib.adk_photometric_fwd(-1, 4, N, N, N, N, N, N, N, 0, N, N, N, N, 0, N) == EINVAL
    assert lib.adk_pose6d_fwd(N, N, N, N) == EINVAL and lib.adk_pose6d_bwd(N, N, N, N, N) == EINVAL
    assert lib.adk_visibility_masks(-1, N, N, 0, N, N, N) == EINVAL
    assert lib.adk_bin_count_isects(-1, N, N, N) == EINVAL
    assert lib.adk_lod_params_bwd_workspace_bytes(-1) == EINVAL
    assert lib.adk_lod_params_fwd(8, N, N, N, N, N, N, N, N, 8, 8, 32, N, N, N, N, N, N, N, N, N, N) == EUNSUPPORTED  # only 16+16 features
    assert lib.adk_project_bwd_adam(8, N, N, N, N, N, 16, 3, N, N, 64, 64, 0.01, 0.01, 1e10, 0, N, N, N, N, N, N, N, N,
                                    N, N, N, N, N, N, 0.5, 0.99, 1e-15, N) == EINVAL
    assert lib.adk_rope_2d(N, N, 1, 1, 4, 0, 0, 2, 6, 100.0, 1.0, N) == EINVAL     # D must be a multiple of 4

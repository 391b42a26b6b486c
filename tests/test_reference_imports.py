"""ARTDECO's own modules import against the drop-ins (CPU, only where /root/reference is mounted): the reference's
`from diff_gaussian_rasterization import adamUpdate, adamUpdateBasic` (Reconstruct/scene/optimizers.py:14) and
`import mast3r_slam_backends` (VSLAM/utils_matching.py:3) resolve to this package, and the names the reference
calls exist with the arity it calls them with."""
import inspect
import os
import sys

import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted")


@pytest.fixture()
def ref_path():
    import artdeco_amd
    artdeco_amd.install_dropins()
    sys.path.insert(0, REF)
    yield
    sys.path.remove(REF)


def test_reference_optimizers_import_and_bind_the_dropin(ref_path):
    import diff_gaussian_rasterization as dgr
    mod = __import__("Reconstruct.scene.optimizers", fromlist=["SparseGaussianAdam"])
    assert mod.adamUpdate is dgr.adamUpdate and mod.adamUpdateBasic is dgr.adamUpdateBasic
    assert dgr.__file__.startswith(os.path.dirname(os.path.abspath(__import__("artdeco_amd").__file__)))
    # call-site arity: optimizers.py:116-128 passes 11 positional arguments, :48-57 passes 8
    assert len(inspect.signature(dgr.adamUpdate).parameters) == 11
    assert len(inspect.signature(dgr.adamUpdateBasic).parameters) == 8
    assert hasattr(mod, "SparseGaussianAdam") and hasattr(mod, "BaseAdam")


def test_reference_matching_imports_the_backend_dropin(ref_path):
    import mast3r_slam_backends as msb
    mod = __import__("VSLAM.utils_matching", fromlist=["match_iterative_proj"])
    assert mod.mast3r_slam_backends is msb
    for name, n_args in (("iter_proj", 6), ("refine_matches", 5), ("gauss_newton_rays", 14), ("gauss_newton_calib", 19),
                         ("gauss_newton_points", 13)):
        assert len(inspect.signature(getattr(msb, name)).parameters) == n_args, name  # gn.cpp:3-114


def test_install_tracker_redirects_the_frontends_import(ref_path):
    """VSLAM/Frontend.py:9 does `from VSLAM.CameraTracker import CameraTracker`; after install_tracker() that name is the
    drop-in class, with the constructor arguments Frontend.py:34-37 passes."""
    import artdeco_amd.tracker as T
    saved = sys.modules.get("VSLAM.CameraTracker")
    try:
        T.install_tracker()
        from VSLAM.CameraTracker import CameraTracker
        assert CameraTracker is T.CameraTracker
        params = list(inspect.signature(CameraTracker.__init__).parameters)
        assert params[1:11] == ["args", "config", "min_displacement", "thres_keyframe", "model", "frames", "H_slam", "W_slam", "K_slam", "device"]
    finally:
        if saved is None:
            sys.modules.pop("VSLAM.CameraTracker", None)
        else:
            sys.modules["VSLAM.CameraTracker"] = saved

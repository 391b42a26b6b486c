"""artdeco_amd.small_inverse: torch.linalg.inv / torch.inverse / Tensor.inverse of 4x4 fp32 CUDA matrices as one launch of adk_inv4x4
(run_system.py:221-223 inverts three per mapper keyframe on a SLAM keyframe, h3dgsv3.py:1000 one per add_keyframe).

The checker is the inverse itself in float64 (numpy.linalg.inv, LAPACK dgesv): an fp32 inverse computed by ANY backward-stable method
differs from it by about cond(A) * 2^-24 relative to |A^-1|, so the bar is  |X - inv64(A)| <= 8 * cond(A) * eps32 * max|inv64(A)|
per matrix -- and, measured beside it, torch's own fp32 LU on the same device must not be closer by more than that bound either.
Everything that is not a plain CUDA fp32 [..., 4, 4] must still reach torch's own functions -- and (round 6) so must every CALLER that is not
one of ARTDECO's own two pose-inverting files; an argument inside autograd takes the same launch with the analytic backward; a singular matrix
raises at the next host wait (`check()`), late but not never."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

EPS32 = float(np.finfo(np.float32).eps)


def _rigid(K, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    q = torch.nn.functional.normalize(torch.randn(K, 4, generator=g), dim=-1)
    w, x, y, z = q.unbind(-1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                     2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                     2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1).reshape(K, 3, 3)
    Rt = torch.eye(4).repeat(K, 1, 1)
    Rt[:, :3, :3] = R * scale
    Rt[:, :3, 3] = 5 * torch.randn(K, 3, generator=g)
    return Rt


def _bound(A64):
    inv = np.linalg.inv(A64)
    cond = np.linalg.cond(A64)
    return inv, 8.0 * cond[..., None, None] * EPS32 * np.abs(inv).max(axis=(-1, -2), keepdims=True)


@pytest.fixture
def patched():
    from artdeco_amd import small_inverse
    was, any_was = small_inverse.installed(), small_inverse._ANY_CALLER
    small_inverse.install(force=True, any_caller=True)      # these tests call from THIS file: lift the caller restriction for them
    yield small_inverse
    small_inverse.install(any_caller=any_was)
    if not was:
        small_inverse.uninstall()


def test_everything_but_cuda_fp32_4x4_reaches_torch(patched):
    """CPU tensors, other sizes / dtypes, `out=`, autograd: torch's own functions, bit for bit (no GPU needed to see that)."""
    g = torch.Generator().manual_seed(0)
    A = torch.randn(5, 4, 4, generator=g) + 4 * torch.eye(4)
    before = dict(patched.STATS)
    want = patched._ORIG["linalg.inv"](A)
    assert torch.equal(torch.linalg.inv(A), want) and torch.equal(torch.inverse(A), want) and torch.equal(A.inverse(), want)
    out = torch.empty_like(A)
    torch.linalg.inv(A, out=out)
    assert torch.equal(out, want)
    B = torch.randn(3, 3, generator=g).double() + 3 * torch.eye(3).double()
    assert torch.equal(torch.linalg.inv(B), patched._ORIG["linalg.inv"](B))
    L = (torch.randn(4, 4, generator=g) + 4 * torch.eye(4)).requires_grad_(True)
    torch.linalg.inv(L).sum().backward()
    assert L.grad is not None and torch.isfinite(L.grad).all()
    with pytest.raises(torch.linalg.LinAlgError):
        torch.linalg.inv(torch.zeros(4, 4))                 # a CPU singular matrix still raises: torch's function
    assert patched.STATS["fast"] == before["fast"] and patched.STATS["torch"] > before["torch"]


def test_only_artdecos_own_files_take_the_fast_path():
    """The wrapped entry points look at the CALLING frame's file: run_system.py / h3dgsv3.py (and what allow_caller registered), nobody else."""
    from artdeco_amd import small_inverse as si
    src = "def call(si):\n    def wrapper():\n        return si._caller_ok()\n    return wrapper()\n"
    any_was = si._ANY_CALLER
    si._ANY_CALLER = False
    try:
        verdicts = {}
        for fn in ("/somewhere/ARTDECO/run_system.py", "/x/Reconstruct/scene/scene_models/h3dgsv3.py", "/site-packages/pypose/lietensor/operation.py",
                   "/x/webviewer/scene_models.py", __file__):
            ns = {}
            exec(compile(src, fn, "exec"), ns)
            verdicts[os.path.basename(fn)] = ns["call"](si)
        assert verdicts == {"run_system.py": True, "h3dgsv3.py": True, "operation.py": False, "scene_models.py": False,
                            os.path.basename(__file__): False}
        si._ANY_CALLER = True
        ns = {}
        exec(compile(src, "/anything.py", "exec"), ns)
        assert ns["call"](si) is True
    finally:
        si._ANY_CALLER = any_was


def test_install_is_idempotent_and_reversible():
    from artdeco_amd import small_inverse
    was = small_inverse.installed()
    if was:
        small_inverse.uninstall()
    o = (torch.linalg.inv, torch.inverse, torch.Tensor.inverse)
    try:
        assert small_inverse.install(force=True) and small_inverse.install(force=True)
        assert torch.linalg.inv is small_inverse._linalg_inv and torch.Tensor.inverse is small_inverse._tensor_inverse
        small_inverse.uninstall()
        assert (torch.linalg.inv, torch.inverse, torch.Tensor.inverse) == o
        os.environ["ARTDECO_AMD_FAST_INV4"] = "0"
        assert small_inverse.install() is False and torch.linalg.inv is o[0]
    finally:
        os.environ.pop("ARTDECO_AMD_FAST_INV4", None)
        small_inverse.uninstall()
        if was:
            small_inverse.install(force=True)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["rigid", "scaled_rigid", "general", "needs_pivoting"])
def test_inverse_against_float64(patched, kind):
    dev = torch.device("cuda:0")
    K = 257
    g = torch.Generator().manual_seed(7)
    if kind == "rigid":
        A = _rigid(K, 3)
    elif kind == "scaled_rigid":
        A = _rigid(K, 4, scale=1.0003)                     # what arrives from the SLAM graph is not exactly orthonormal
    elif kind == "general":
        A = torch.randn(K, 4, 4, generator=g)
    else:
        A = torch.randn(K, 4, 4, generator=g)
        A[:, 0, 0] = 0.0                                   # the first pivot is zero without a row exchange
        A[::2, 1, 1] = 1e-6
    inv64, bound = _bound(A.double().numpy())
    before = patched.STATS["fast"]
    got = torch.linalg.inv(A.to(dev))
    assert patched.STATS["fast"] == before + 1
    assert got.shape == A.shape and got.is_contiguous() and got.dtype == torch.float32
    err = np.abs(got.cpu().double().numpy() - inv64)
    assert (err <= bound).all(), float((err / bound).max())
    # torch's own LU on the same device is held to the same bar (so the two agree to twice it)
    ref = patched._ORIG["linalg.inv"](A.to(dev))
    assert (np.abs(ref.cpu().double().numpy() - inv64) <= bound).all()
    # the three entry points, a single matrix, and a transposed view (run_system.py:220-221 inverts get_Rt().transpose(0, 1))
    assert torch.equal(torch.inverse(A.to(dev)), got) and torch.equal(A.to(dev).inverse(), got)
    assert torch.equal(torch.linalg.inv(A[5].to(dev)), got[5])
    At = A.to(dev).transpose(1, 2)
    assert not At.is_contiguous()
    assert torch.equal(At.inverse(), torch.linalg.inv(At.contiguous()))
    assert torch.equal(A[5].to(dev).transpose(0, 1).inverse(), torch.linalg.inv(At[5].contiguous()))
    B = A.to(dev).reshape(K, 1, 4, 4).expand(K, 2, 4, 4)   # batch dimensions that do not collapse into a view
    assert torch.equal(torch.linalg.inv(B)[:, 1], got)


@pytest.mark.gpu
def test_singular_matrices_give_nan_not_an_exception_and_no_synchronisation(patched):
    dev = torch.device("cuda:0")
    A = _rigid(6, 1).to(dev)
    A[2] = 0.0
    A[4, :, 1] = 0.0                                         # a zero column: the second pivot is exactly zero
    info = torch.full((6,), -1, dtype=torch.int32, device=dev)
    got = patched.inv4x4(A, info)
    assert info.tolist() == [0, 0, 1, 0, 2, 0]
    assert torch.isnan(got[2]).all() and torch.isnan(got[4]).all() and torch.isfinite(got[[0, 1, 3, 5]]).all()
    torch.cuda.synchronize()
    assert patched.singular_seen() >= 2
    with pytest.raises(torch.linalg.LinAlgError, match="singular 4x4"):    # ... and the host hears of it at its next wait
        patched.check()
    patched.check()                                                        # reported once
    assert torch.isnan(torch.linalg.inv(A)[2]).all()                       # the wrapped entry point: NaN at the call, where torch raises
    torch.cuda.synchronize()
    with pytest.raises(torch.linalg.LinAlgError):
        patched.check()
    with pytest.raises(torch.linalg.LinAlgError):
        patched._ORIG["linalg.inv"](A)


@pytest.mark.gpu
def test_autograd_takes_the_same_launch_and_other_dtypes_stay_with_torch(patched):
    """run_system.py:222 inverts `frame.get_Rt()` with grad enabled: one launch + the analytic backward -A^-T G A^-T, equal to torch's."""
    dev = torch.device("cuda:0")
    A = _rigid(3, 2).to(dev).requires_grad_(True)
    w = torch.randn(3, 4, 4, generator=torch.Generator().manual_seed(5)).to(dev)
    before = dict(patched.STATS)
    (torch.linalg.inv(A) * w).sum().backward()
    assert patched.STATS["fast_grad"] == before["fast_grad"] + 1 and patched.STATS["torch"] == before["torch"]
    B = A.detach().clone().requires_grad_(True)
    (patched._ORIG["linalg.inv"](B) * w).sum().backward()
    assert torch.allclose(A.grad, B.grad, rtol=1e-4, atol=1e-5 * float(B.grad.abs().max()))
    with torch.no_grad():
        torch.linalg.inv(A)                                                  # outside autograd the same tensor takes the plain launch
    assert patched.STATS["fast"] == before["fast"] + 1
    assert torch.linalg.inv(A.detach().double()).dtype == torch.float64 and patched.STATS["fast"] == before["fast"] + 1


@pytest.mark.gpu
def test_a_foreign_caller_keeps_torchs_inverse_on_the_device():
    from artdeco_amd import small_inverse as si
    dev = torch.device("cuda:0")
    was, any_was = si.installed(), si._ANY_CALLER
    si.install(force=True, any_caller=False)
    try:
        before = dict(si.STATS)
        A = _rigid(2, 8).to(dev)
        torch.linalg.inv(A); A.inverse(); torch.inverse(A)                   # this file is not run_system.py
        assert si.STATS["foreign_caller"] == before["foreign_caller"] + 3 and si.STATS["fast"] == before["fast"]
        with pytest.raises(torch.linalg.LinAlgError):
            torch.linalg.inv(torch.zeros(4, 4, device=dev))                  # ... error behaviour included
    finally:
        si.install(any_caller=any_was)
        if not was:
            si.uninstall()


@pytest.mark.gpu
def test_the_slam_keyframe_loop_as_written_gives_the_same_map_with_and_without_the_wrapped_inverse():
    """run_system.py:194-227 (harness/stream.slam_pose_update) with torch's inverse and with the wrapped one: same keyframe poses, same
    camera centres, same moved Gaussians to the inverse's rounding; and the wrapped loop contains no torch LU."""
    import artdeco_amd
    artdeco_amd.install_dropins()
    from artdeco_amd import fused, small_inverse
    from harness import mapper, stream
    dev = torch.device("cuda:0")
    was = small_inverse.installed()
    out = []
    try:
        for fast in (False, True):
            os.environ["ARTDECO_AMD_FAST_INV4"] = "1" if fast else "0"
            small_inverse.uninstall()
            scene = mapper.build_synthetic_mapper(3_000, 96, 64, dev, seed=2, n_keyframes=0, targets="random")
            fused.patch_scene_model(scene)
            assert small_inverse.installed() == fast
            frames = stream.synthetic_frames(scene, 5, seed=2, slam_hw=(48, 64))
            for i, fr in enumerate(frames):
                scene.add_keyframe(stream.make_keyframe(scene, fr, i))
            xyz0 = scene.xyz.detach().clone()
            n_torch = small_inverse.STATS["torch"]
            stream.slam_pose_update(scene, delta=1e-2, seed=9)
            if fast:
                assert small_inverse.STATS["torch"] == n_torch          # 15 inversions, none of them torch's
            out.append((torch.stack([kf.get_Rt().detach() for kf in scene.keyframes]), scene.xyz.detach().clone(),
                        scene.rotation.detach().clone(), scene.cam_centres.clone(), xyz0))
    finally:
        os.environ.pop("ARTDECO_AMD_FAST_INV4", None)
        small_inverse.uninstall()
        if was:
            small_inverse.install(force=True)
    (Rt_a, xyz_a, rot_a, cc_a, xyz0), (Rt_b, xyz_b, rot_b, cc_b, _) = out
    assert float((xyz_a - xyz0).abs().max()) > 1e-3
    assert torch.equal(Rt_a, Rt_b) and torch.allclose(cc_a, cc_b, atol=1e-5)
    assert torch.allclose(xyz_a, xyz_b, atol=1e-5) and torch.allclose(rot_a, rot_b, atol=1e-5)

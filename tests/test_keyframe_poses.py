"""artdeco_amd.keyframe_poses.update_keyframe_poses against the per-keyframe loop of run_system.py:194-227 (restated in
harness/stream.slam_pose_update with the reference Keyframe's own get_Rt / set_Rt): same old / new camera-to-world matrices,
same camera centres, same parameters written.  CPU tensors: the helper is plain torch."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class _KF:
    """The pose fields of scene/keyframe.py's Keyframe and its two pose methods (:150-159)."""

    def __init__(self, Rt):
        self.rW2C = torch.nn.Parameter(Rt[:3, :2].clone().contiguous())
        self.tW2C = torch.nn.Parameter(Rt[:3, 3].clone().contiguous())
        self.approx_centre = -Rt[:3, :3].T @ Rt[:3, 3]

    def get_Rt(self):
        from artdeco_amd.keyframe_poses import six_d_to_matrix
        Rt = torch.eye(4)
        Rt[:3, :3] = six_d_to_matrix(self.rW2C)
        Rt[:3, 3] = self.tW2C
        return Rt

    def set_Rt(self, Rt):
        self.rW2C.data.copy_(Rt[:3, :2])
        self.tW2C.data.copy_(Rt[:3, 3])
        self.approx_centre = -Rt[:3, :3].T @ Rt[:3, 3]


def _poses(K, seed):
    g = torch.Generator().manual_seed(seed)
    q = torch.nn.functional.normalize(torch.randn(K, 4, generator=g), dim=-1)
    w, x, y, z = q.unbind(-1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                     2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                     2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1).reshape(K, 3, 3)
    Rt = torch.eye(4).repeat(K, 1, 1)
    Rt[:, :3, :3] = R
    Rt[:, :3, 3] = 3 * torch.randn(K, 3, generator=g)
    return Rt


@pytest.mark.parametrize("K", [1, 7, 64])
def test_batched_pose_reread_equals_the_per_keyframe_loop(K):
    from artdeco_amd.keyframe_poses import update_keyframe_poses
    old, new = _poses(K, 1), _poses(K, 2)
    new[:, :3, :3] = new[:, :3, :3] * 1.0003        # what arrives from the SLAM graph is not exactly orthonormal
    loop_kfs, bat_kfs = [_KF(m) for m in old], [_KF(m) for m in old]
    old_c2ws, new_c2ws, centres = torch.zeros(K, 4, 4), torch.zeros(K, 4, 4), torch.zeros(K, 3)
    with torch.no_grad():
        for k, kf in enumerate(loop_kfs):           # run_system.py:198-221, one keyframe at a time
            old_Rt = kf.get_Rt()
            kf.set_Rt(new[k])
            view_matrix = kf.get_Rt().transpose(0, 1)
            centres[k] = view_matrix.inverse()[3, :3]
            old_c2ws[k] = torch.linalg.inv(old_Rt)
            new_c2ws[k] = torch.linalg.inv(new[k])
    o, n, c = update_keyframe_poses(bat_kfs, new.clone())
    assert torch.allclose(o, old_c2ws, atol=2e-6, rtol=1e-5) and torch.allclose(n, new_c2ws, atol=2e-6, rtol=1e-5)
    assert torch.allclose(c, centres, atol=5e-6, rtol=1e-5)
    for a, b in zip(loop_kfs, bat_kfs):
        assert torch.equal(a.rW2C.data, b.rW2C.data) and torch.equal(a.tW2C.data, b.tW2C.data)
        assert torch.allclose(a.approx_centre, b.approx_centre, atol=2e-6, rtol=1e-5)


def test_batched_pose_reread_with_no_keyframes():
    from artdeco_amd.keyframe_poses import update_keyframe_poses
    o, n, c = update_keyframe_poses([], torch.zeros(0, 4, 4))
    assert o.shape == (0, 4, 4) and n.shape == (0, 4, 4) and c.shape == (0, 3)


@pytest.mark.gpu
def test_batched_pose_reread_on_the_device_feeds_rigid_transform_gs():
    """The helper on device tensors, through the harness' frame loop: a stream whose SLAM keyframes use the batched re-read ends with
    the same keyframe poses and (to the inverse's rounding) the same moved Gaussians as one that runs run_system.py's loop."""
    import artdeco_amd
    artdeco_amd.install_dropins()
    from artdeco_amd import fused
    from harness import mapper, stream
    dev = torch.device("cuda:0")
    out = []
    for batched in (False, True):
        scene = mapper.build_synthetic_mapper(3_000, 96, 64, dev, seed=2, n_keyframes=0, targets="random")
        fused.patch_scene_model(scene)
        frames = stream.synthetic_frames(scene, 4, seed=2, slam_hw=(48, 64))
        clock = stream.StageClock(False)
        for i, fr in enumerate(frames):            # keyframes only, no optimisation: the two runs see identical maps
            scene.add_keyframe(stream.make_keyframe(scene, fr, i))
        xyz0 = scene.xyz.detach().clone()
        (stream.slam_pose_update_batched if batched else stream.slam_pose_update)(scene, delta=1e-2, seed=9)
        out.append((torch.stack([kf.get_Rt().detach() for kf in scene.keyframes]), scene.xyz.detach().clone(), scene.rotation.detach().clone(),
                    scene.cam_centres.clone(), xyz0))
    (Rt_a, xyz_a, rot_a, cc_a, xyz0), (Rt_b, xyz_b, rot_b, cc_b, _) = out
    assert float((xyz_a - xyz0).abs().max()) > 1e-3          # the update moved the map
    assert torch.allclose(Rt_a, Rt_b, atol=1e-6) and torch.allclose(cc_a, cc_b, atol=1e-5)
    assert torch.allclose(xyz_a, xyz_b, atol=1e-5) and torch.allclose(rot_a, rot_b, atol=1e-5)

"""Generate tests/golden/densify_*.npz by running the REFERENCE's own `SceneModel.add_new_gaussians`
(/root/reference/Reconstruct/scene/scene_models/h3dgsv3.py:766-940) -- the real class, imported from the reference tree, with a real
`Keyframe` (scene/keyframe.py) -- on CPU.  The natives it reaches (gsplat rasterization through render_from_id, scatter_max
inside update_voxel, adamUpdate*) are bound to the CPU oracles; `device="cuda"` literals mean the CPU here.  Recorded per LoD level:
the uniform draw (`torch.rand_like`), the two `get_lapla_norm` maps, the full-resolution render the penalty was computed from; and
the tensors handed to `optimizer.add_and_prune` (every new Gaussian's attributes + the prune mask).  Build container only.

    ARTDECO_AMD_AUTOFUSE=0 python tests/golden/make_golden_densify.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from harness import mapper, ref_env  # noqa: E402


def ref_args(**over):
    a = dict(num_prev_keyframes_check=5, sh_degree=3, lambda_dssim=0.2, init_proba_scaler=2.0, max_active_keyframes=200,
             use_last_frame_proba=0.0, scaling_reg_factor=0.0, rad_decay=float(np.sqrt(5.0)), position_lr_init=5e-5,
             position_lr_decay=1 - 2e-5, feature_lr=5e-3, scaling_lr=0.01, rotation_lr=2e-3, opacity_lr=0.1, feat_lr=4e-3,
             local_feat_dim=16, global_feat_dim=16, mlp_cov_lr_init=4e-3, mlp_cov_lr_decay=1 - 2e-5, voxel_size=0.1,
             visible_threshold=0.0, low_pass_filter_eps=0.01, gs_add_ratio=1.0, pyr_levels=2, depth_loss_weight_init=1e-2,
             depth_loss_weight_decay=0.9, lr_poses=1e-4, lr_exposure=1e-3)
    a.update(over)
    return types.SimpleNamespace(**a)


def cases():
    """name -> dict(seed, W, H, Ws, Hs, N, texture, neg_conf_frac, near_frac, voxel_size)."""
    return {
        "densify_mixed": dict(seed=0, W=96, H=72, Ws=56, Hs=40, N=2500, texture=0.25, neg_conf=0.08, near=0.0, n_prev=1),
        "densify_ragged": dict(seed=1, W=103, H=67, Ws=64, Hs=48, N=1800, texture=0.4, neg_conf=0.0, near=0.05, n_prev=2),
        "densify_cold": dict(seed=2, W=64, H=48, Ws=64, Hs=48, N=0, texture=0.3, neg_conf=0.02, near=0.0, n_prev=0),
    }


def make_inputs(c):
    """Seeded inputs of one case (shared with tests/test_densify.py): the map's Gaussians, the new keyframe's image, point map,
    confidence and pose, earlier keyframe poses."""
    g = torch.Generator().manual_seed(c["seed"])
    W, H, Ws, Hs, N = c["W"], c["H"], c["Ws"], c["Hs"], c["N"]
    fx = 0.8 * W
    cloud = mapper.synthetic_cloud(max(N, 1), W, H, seed=c["seed"], sigma_px=1.5)
    cloud = {k: (v[:N] if torch.is_tensor(v) else v) for k, v in cloud.items()}
    # a smooth image with textured patches: Laplacian responses spread over (0, 1) after the x2 scaler
    base = torch.rand(3, H // 8 + 2, W // 8 + 2, generator=g)
    smooth = torch.nn.functional.interpolate(base[None], (H, W), mode="bicubic", align_corners=True)[0]
    tex = torch.rand(3, H, W, generator=g) - 0.5
    gate = torch.nn.functional.interpolate(torch.rand(1, 1, H // 12 + 2, W // 12 + 2, generator=g), (H, W), mode="bilinear", align_corners=True)[0]
    image = (smooth + c["texture"] * tex * (gate > 0.45)).clamp(0, 1).contiguous()
    depth = 2.0 + 3.0 * torch.nn.functional.interpolate(torch.rand(1, 1, Hs // 6 + 2, Ws // 6 + 2, generator=g), (Hs, Ws), mode="bilinear",
                                                          align_corners=True)[0, 0]
    if c["near"] > 0:   # a fraction of near-zero depths: the 2 % quantile then falls below 1e-2 and the interpolated branch matters
        depth = torch.where(torch.rand(Hs, Ws, generator=g) < c["near"], 1e-3 * torch.rand(Hs, Ws, generator=g), depth)
    ys, xs = torch.meshgrid(torch.arange(Hs, dtype=torch.float32), torch.arange(Ws, dtype=torch.float32), indexing="ij")
    fs = fx * Ws / W
    point_map = torch.stack([(xs - (Ws - 1) / 2) / fs * depth, (ys - (Hs - 1) / 2) / fs * depth, depth], -1).contiguous()
    conf = 0.3 + 0.6 * torch.rand(Hs, Ws, generator=g)
    if c["neg_conf"] > 0:
        conf = torch.where(torch.rand(Hs, Ws, generator=g) < c["neg_conf"], -torch.ones(Hs, Ws), conf)
    ang = 0.03 * torch.randn(3, generator=g)
    Rx = torch.tensor([[1, 0, 0], [0, torch.cos(ang[0]), -torch.sin(ang[0])], [0, torch.sin(ang[0]), torch.cos(ang[0])]])
    Ry = torch.tensor([[torch.cos(ang[1]), 0, torch.sin(ang[1])], [0, 1, 0], [-torch.sin(ang[1]), 0, torch.cos(ang[1])]])
    Rt = torch.eye(4)
    Rt[:3, :3] = Rx @ Ry
    Rt[:3, 3] = 0.05 * torch.randn(3, generator=g)
    prev = []
    for _ in range(c["n_prev"]):
        P = torch.eye(4)
        P[:3, 3] = 0.1 * torch.randn(3, generator=g)
        prev.append(P)
    feats = dict(local=0.3 * torch.randn(N, 16, generator=g), glob=0.3 * torch.randn(max(N // 8, 1), 16, generator=g),
                 cls=torch.randint(0, max(N // 8, 1), (N, 1), generator=g), d_max=3.0 + 4.0 * torch.rand(N, 1, generator=g))
    return dict(W=W, H=H, fx=fx, cloud=cloud, image=image, point_map=point_map, conf=conf.contiguous(), Rt=Rt, prev=prev, feats=feats)


def populate(scene, inp, seed):
    """The same N Gaussians, features and mlp weights into a scene model (mirror or real) through ITS add_and_prune."""
    c, fe = inp["cloud"], inp["feats"]
    N = c["means"].shape[0]
    op = c["opacities"].clamp(1e-4, 1 - 1e-4)
    ext = {"id": torch.zeros(N, 1, dtype=torch.long), "cls_id": fe["cls"].clone(), "d_max": fe["d_max"].clone(), "xyz": c["means"].clone(),
           "f_dc": c["sh"][:, :1, :].clone(), "f_rest": c["sh"][:, 1:, :].clone(), "opacity": torch.log(op / (1 - op)).reshape(N, 1),
           "scaling": torch.log(2.0 * c["scales"]), "rotation": c["quats"].clone(), "local_feat": fe["local"].clone(),
           "global_feat": fe["glob"].clone() if N else fe["glob"][:0].clone()}
    scene.optimizer.add_and_prune(ext, torch.ones(scene.xyz.shape[0], dtype=torch.bool))
    torch.manual_seed(seed + 77)
    with torch.no_grad():
        for p in scene.mlp_cov.parameters():
            p.copy_(0.3 * torch.randn(p.shape))
        scene.mlp_cov[2].bias.copy_(torch.tensor([0.0, 0.0, 0.0, 1.0, 1.0, 1.0, 1.0]))


def empty_mirror(inp):
    sc = mapper.MapperScene(inp["W"], inp["H"], inp["fx"], "cpu")
    z = torch.zeros
    sc.set_gaussians(z(0, 3), z(0, 4), z(0, 3), z(0), z(0, 16, 3), n_voxels=1)
    sc.gaussian_params["global_feat"]["val"] = z(0, 16).requires_grad_(True)
    sc.optimizer.params["global_feat"]["exp_avg"] = z(0, 16)
    sc.optimizer.params["global_feat"]["exp_avg_sq"] = z(0, 16)
    return sc


def bind_cpu_natives(*modules):
    ras, ssim, au, aub, smax = ref_env.cpu_natives()
    fake_gsplat = types.SimpleNamespace(rendering=types.SimpleNamespace(rasterization=ras))
    for m in modules:
        for name, val in (("gsplat", fake_gsplat), ("fused_ssim", ssim), ("scatter_max", smax), ("adamUpdate", au), ("adamUpdateBasic", aub)):
            if hasattr(m, name):
                setattr(m, name, val)


def run_reference(inp, seed, record):
    """The real class: returns (scene, captured add_and_prune arguments of the FIRST call = the densification itself)."""
    mod = ref_env.import_scene_module()
    opt_mod, kf_mod = sys.modules["Reconstruct.scene.optimizers"], sys.modules["Reconstruct.scene.keyframe"]
    bind_cpu_natives(mod, opt_mod, mapper)
    mod.torch = ref_env.TorchNoCuda()
    W, H, fx = inp["W"], inp["H"], inp["fx"]
    K = torch.tensor([[fx, 0, (W - 1) / 2], [0, fx, (H - 1) / 2], [0, 0, 1.0]])
    args = ref_args()
    real = mod.SceneModel(W, H, K, args, device="cpu")
    populate(real, inp, seed)
    f = torch.tensor([fx])
    for i, P in enumerate(inp["prev"]):
        real.add_keyframe(kf_mod.Keyframe(inp["image"].clone(), f"prev{i}", False, P, i, i, 0, 0, False, f, args, point_map=inp["point_map"].clone(),
                                          point_conf=inp["conf"].clone(), device_mapper="cpu"))
    kf = kf_mod.Keyframe(inp["image"].clone(), "new", False, inp["Rt"], len(inp["prev"]), len(inp["prev"]), 0, 0, False, f, args,
                         point_map=inp["point_map"].clone(), point_conf=inp["conf"].clone(), device_mapper="cpu")
    real.add_keyframe(kf)
    captured = {}
    orig_aap = real.optimizer.add_and_prune

    def spy(ext, mask):
        if "ext" not in captured:
            captured["ext"] = {k: v.detach().clone() for k, v in ext.items()}
            captured["mask"] = mask.clone()
        return orig_aap(ext, mask)
    real.optimizer.add_and_prune = spy
    orig_lap, orig_rfi, orig_rand = mod.get_lapla_norm, real.render_from_id, torch.rand_like
    mod.get_lapla_norm = lambda img, k, device="cpu": record("lap", orig_lap(img, k, device="cpu"))
    real.render_from_id = lambda *a, **k: record("render", orig_rfi(*a, **k))
    torch.rand_like = lambda t, **k: record("rand", orig_rand(t, **k))
    try:
        torch.manual_seed(seed + 1000)
        real.add_new_gaussians()
    finally:
        mod.get_lapla_norm, torch.rand_like = orig_lap, orig_rand
    return real, captured


def main():
    os.environ["ARTDECO_AMD_AUTOFUSE"] = "0"   # the reference's OWN methods are what is recorded (set here, not at import: the tests import this module)
    # torch's CPU bilinear resize rounds differently in the vectorised body and the scalar tail of a thread's range, so its last bit
    # depends on the thread count: the goldens are generated with the thread count the tests run with (tests/conftest.py)
    torch.set_num_threads(int(os.environ.get("ADK_TEST_THREADS", "1")))
    for name, c in cases().items():
        inp = make_inputs(c)
        rec = {"lap": [], "render": [], "rand": []}

        def record(kind, val):
            rec[kind].append((val["render"] if kind == "render" else val).detach().clone())
            return val
        real, cap = run_reference(inp, c["seed"], record)
        out = {}
        n_lap = 2 if c["N"] > 0 else 1
        for li, lod in enumerate((1, 2, 4, 8)):
            out[f"init_proba_{lod}"] = rec["lap"][n_lap * li].numpy()
            if c["N"] > 0:
                out[f"penalty_{lod}"] = rec["lap"][n_lap * li + 1].numpy()
                out[f"render_{lod}"] = rec["render"][li].numpy()
            out[f"rand_{lod}"] = rec["rand"][li].numpy()
        for k, v in cap["ext"].items():
            out["ext_" + k] = v.numpy()
        out["valid_gs_mask"] = cap["mask"].numpy()
        out["final_n"] = np.int64(real.xyz.shape[0])
        out["final_cls_id"] = real.cls_id.numpy()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "new Gaussians:", cap["ext"]["xyz"].shape[0], "pruned:", int((~cap["mask"]).sum()), "final:", int(real.xyz.shape[0]),
              "global_feat rows added:", cap["ext"]["global_feat"].shape[0])


if __name__ == "__main__":
    main()

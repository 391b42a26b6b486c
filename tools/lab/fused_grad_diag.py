#!/usr/bin/env python
"""DEV TOOL (round 4): where does the fused-vs-unfused gradient difference of tests/test_fused_glue.py sit for a given scene seed?
    python tools/lab/fused_grad_diag.py 6"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import artdeco_amd

artdeco_amd.install_dropins()
import test_fused_glue as T
from artdeco_amd import fused

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda:0")
a, b = T._scene(dev, N=8000, seed=seed), T._scene(dev, N=8000, seed=seed)
assert fused.patch_scene_model(b)
keys = ("xyz", "scaling", "rotation", "opacity", "local_feat", "global_feat")
for i in range(3):
    T._sync_state(a, b)
    grads = {}
    for name, sc in (("a", a), ("b", b)):
        orig = sc.optimizer.step

        def spy(*args, _o=orig, _sc=sc, _n=name, **kw):
            grads[_n] = {k: _sc.gaussian_params[k]["val"].grad.clone() for k in keys}
            return _o(*args, **kw)
        sc.optimizer.step = spy
        torch.manual_seed(i)
        loss = float(sc.optimization_step(i % 2, is_important=(i != 1)))
        sc.optimizer.step = orig
        grads[name]["loss"] = loss
    print(f"step {i}: loss a {grads['a']['loss']:.8f} b {grads['b']['loss']:.8f}")
    inv_a, inv_b = a.keyframes[i % 2].latest_invdepth, b.keyframes[i % 2].latest_invdepth
    da, db = torch.nan_to_num(1.0 / inv_a, posinf=0.0), torch.nan_to_num(1.0 / inv_b, posinf=0.0)
    dd = (da - db).abs()
    print(f"   depth diff max {float(dd.max()):.3e}  pixels > 1e-5: {int((dd > 1e-5).sum())}  > 1e-6: {int((dd > 1e-6).sum())}  > 1e-7: {int((dd > 1e-7).sum())}")
    for k in keys:
        x, y = grads["a"][k].double(), grads["b"][k].double()
        d = (x - y).reshape(x.shape[0], -1)
        rel = float(d.norm() / (x.norm() + 1e-30))
        rows = d.norm(dim=1)
        top = torch.topk(rows, min(5, rows.numel()))
        share = float((top.values ** 2).sum() / (rows ** 2).sum().clamp_min(1e-300))
        xr = x.reshape(x.shape[0], -1).norm(dim=1)
        print(f"   {k:12s} rel_l2 {rel:.3e}  top-5 rows {top.indices.tolist()} carry {share:.3f} of the squared error; their |grad| {xr[top.indices].tolist()} vs median {float(xr.median()):.3e} max {float(xr.max()):.3e}")
    if i == 0:
        with torch.no_grad():
            j = int(torch.topk((grads["a"]["xyz"] - grads["b"]["xyz"]).norm(dim=1), 1).indices[0])
            print("   worst row", j, "xyz", a.xyz[j].tolist(), "scaling", a.gaussian_params["scaling"]["val"][j].tolist(), "opacity", a.gaussian_params["opacity"]["val"][j].tolist())
            print("   grad a", grads["a"]["xyz"][j].tolist(), "grad b", grads["b"]["xyz"][j].tolist())

# ---- which path is right?  the same step on the CPU-oracle path (harness/mapper.py with natives = oracle/), from the same state
if "--oracle" in sys.argv:
    from harness import psnr_proxy as PP
    cmap = PP.cpu_mapper()
    a, b = T._scene(dev, N=8000, seed=seed), T._scene(dev, N=8000, seed=seed)
    assert fused.patch_scene_model(b)
    c = cmap.build_synthetic_mapper(8000, 160, 112, "cpu", seed=seed, n_keyframes=2)
    T._sync_state(a, b)
    with torch.no_grad():   # c <- a (CPU copy of every parameter / moment / keyframe state / mlp)
        for k, pd in a.optimizer.params.items():
            qd = c.optimizer.params[k]
            for name in ("val", "exp_avg", "exp_avg_sq"):
                if name in pd and torch.is_tensor(pd[name]):
                    qd[name].copy_(pd[name].cpu())
            if torch.is_tensor(pd.get("lr")):
                qd["lr"].copy_(pd["lr"].cpu())
        for pa, pc in zip(a.mlp_cov.parameters(), c.mlp_cov.parameters()):
            pc.copy_(pa.cpu())
        for ka, kc in zip(a.keyframes, c.keyframes):
            for name in ("rW2C", "tW2C", "exposure"):
                getattr(kc, name).copy_(getattr(ka, name).cpu())
            kc.image_pyr = [t.cpu() for t in ka.image_pyr]
            kc.idepth_pyr = [t.cpu() for t in ka.idepth_pyr]
            kc.depth_loss_weight = ka.depth_loss_weight
    grads = {}
    for name, sc in (("a", a), ("b", b), ("c", c)):
        orig = sc.optimizer.step

        def spy(*args, _o=orig, _sc=sc, _n=name, **kw):
            grads[_n] = {k: _sc.gaussian_params[k]["val"].grad.detach().cpu().clone() for k in keys}
            return _o(*args, **kw)
        sc.optimizer.step = spy
        torch.manual_seed(0)
        # the step draws its background with torch.rand(3, device=...): make it the same on every side
        real_rand = torch.rand
        torch.rand = lambda *size, **kw: (torch.tensor([0.3, 0.6, 0.1], device=kw.get("device", "cpu")) if size == (3,) else real_rand(*size, **kw))
        try:
            loss = float(sc.optimization_step(0, is_important=True))
        finally:
            torch.rand = real_rand
        print(f"[oracle check] path {name}: loss {loss:.8f}")
    for k in keys:
        ga, gb, gc = (grads[n][k].double() for n in "abc")
        ra = float((ga - gc).norm() / gc.norm().clamp_min(1e-300)); rb = float((gb - gc).norm() / gc.norm().clamp_min(1e-300)); rab = float((ga - gb).norm() / ga.norm().clamp_min(1e-300))
        print(f"[oracle check] {k:12s} |unfused - oracle| {ra:.3e}   |fused - oracle| {rb:.3e}   |unfused - fused| {rab:.3e}")
    j = int(torch.topk((grads["a"]["xyz"] - grads["b"]["xyz"]).norm(dim=1), 1).indices[0])
    print("[oracle check] worst row", j, "unfused", grads["a"]["xyz"][j].tolist(), "fused", grads["b"]["xyz"][j].tolist(), "oracle", grads["c"]["xyz"][j].tolist())
    with torch.no_grad():
        xyz = a.xyz[j].cpu(); dm = float(a.gaussian_params["d_max"]["val"][j])
        print("[oracle check] worst row d_max", dm, "|xyz|", float(xyz.norm()), "|xyz| / d_max", float(xyz.norm()) / dm)

#!/usr/bin/env python
"""DEV TOOL: how large is the fused-vs-unfused gradient difference of tests/test_fused_glue.py from run to run?  Prints the largest
rel_l2 per key over the 3 steps, for the fused-vs-unfused pair AND for unfused-vs-unfused (two scenes from the same state: pure
run-to-run noise of the float atomics)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_fused_glue as T
from artdeco_amd import fused
dev = torch.device("cuda:0")
a, b, c = T._scene(dev, N=8000, seed=3), T._scene(dev, N=8000, seed=3), T._scene(dev, N=8000, seed=3)
assert fused.patch_scene_model(b)
keys = ("xyz", "scaling", "rotation", "opacity", "local_feat", "global_feat")
worst = {}
for i in range(3):
    T._sync_state(a, b); T._sync_state(a, c)
    grads = {}
    for name, sc in (("a", a), ("b", b), ("c", c)):
        orig = sc.optimizer.step
        def spy(*args, _o=orig, _sc=sc, _n=name, **kw):
            grads[_n] = {k: _sc.gaussian_params[k]["val"].grad.clone() for k in keys}
            return _o(*args, **kw)
        sc.optimizer.step = spy
        torch.manual_seed(i)
        sc.optimization_step(i % 2, is_important=(i != 1))
        sc.optimizer.step = orig
    for k in keys:
        ga, gb, gc = grads["a"][k].double(), grads["b"][k].double(), grads["c"][k].double()
        worst[k] = max(worst.get(k, (0, 0)), (float((ga - gb).norm() / ga.norm()), float((ga - gc).norm() / ga.norm())))
print(" ".join(f"{k}: fused {v[0]:.1e} unfused-repeat {v[1]:.1e}" for k, v in worst.items()))

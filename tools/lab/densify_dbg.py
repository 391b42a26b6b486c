import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import artdeco_amd; artdeco_amd.install_dropins()
import test_densify as T
from artdeco_amd import fused
dev = torch.device("cuda:0")
name = "densify_mixed"
sc, inp, c = T._gpu_mirror(name, dev)
fused.patch_scene_model(sc)
g = T.golden(name)
draws = [torch.from_numpy(g[f"rand_{lod}"]).to(dev) for lod in T.LODS]
torch.rand_like = lambda t, **k: draws.pop(0)
renders = [torch.from_numpy(g[f"render_{lod}"]).to(dev) for lod in T.LODS]
sc.render_from_id = lambda *a, **k: {"render": renders.pop(0)}
cap = {}
orig = sc.optimizer.add_and_prune
def spy(ext, mask):
    if not cap: cap.update(ext={k: v.detach().clone() for k, v in ext.items()})
    return orig(ext, mask)
sc.optimizer.add_and_prune = spy
sc.add_new_gaussians()
mine = cap["ext"]["opacity"].cpu().numpy()[:, 0]; ref = g["ext_opacity"][:, 0]
outs = T._oracle_levels(c, inp, g)
oconf = np.concatenate([o["conf"] for o in outs]); oop = np.concatenate([o["opacity"][:, 0] for o in outs])
d = np.abs(mine - ref); idx = np.argsort(-d)[:6]
sig = lambda x: 1 / (1 + np.exp(-x.astype(np.float64)))
for i in idx:
    print(i, "mine", mine[i], "ref", ref[i], "oracle", oop[i], "conf(mine)", sig(mine[i]) / 0.2, "conf(ref)", sig(ref[i]) / 0.2, "conf(oracle)", oconf[i],
          "d_max mine/ref", cap["ext"]["d_max"].cpu().numpy()[i, 0], g["ext_d_max"][i, 0])

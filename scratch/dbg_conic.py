import sys, torch, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import artdeco_amd; artdeco_amd.install_dropins()
from oracle import gsplat_oracle as go
from test_raster import _run_hip, _scene
dev = torch.device('cuda:0')
sc = _scene(5000, 160, 112, 0)
_, _, ometa = go.rasterization(**sc, eps2d=0.01)
r, a, meta, _ = _run_hip(sc, dev)
vis = ometa["p32"]["valid"]
ch = meta["conics"][0].cpu()[vis]; co = ometa["p32"]["conics"][vis]
d = (ch - co)
print("neq per comp", (d != 0).sum(0), "of", len(co))
i = (d != 0).any(1).nonzero()[:5, 0]
for k in i:
    print(ch[k].numpy().view(np.uint32), co[k].numpy().view(np.uint32), ch[k].numpy(), )

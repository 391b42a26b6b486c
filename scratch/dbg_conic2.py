import sys, torch, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import artdeco_amd; artdeco_amd.install_dropins()
from oracle import gsplat_oracle as go
from test_raster import _run_hip, _scene
dev = torch.device('cuda:0')
sc = _scene(5000, 160, 112, 0)
pc = go.project(sc['means'], sc['quats'], sc['scales'], sc['opacities'], sc['viewmat'], sc['K'], 160, 112, 0.01)
g = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in sc.items()}
try:
    pg = go.project(g['means'], g['quats'], g['scales'], g['opacities'], g['viewmat'], g['K'], 160, 112, 0.01)
    print("oracle cpu vs oracle gpu conics neq:", (pg['conics'].cpu() != pc['conics']).sum().item())
except Exception as e:
    print("oracle on gpu failed", e)
r, a, meta, _ = _run_hip(sc, dev)
print("kernel vs oracle-gpu neq:", (meta['conics'][0] != pg['conics']).sum().item())
# bisect pieces on cpu vs gpu
q = sc['quats']; 
def rot(q):
    return go._quat_to_rotmat(q)
Rc = rot(q); Rg = rot(q.to(dev))
print("rotmat neq", sum((Rg[i][j].cpu() != Rc[i][j]).sum().item() for i in range(3) for j in range(3)))
x = q[:,1]; xg = x.to(dev)
n2 = ((q[:,1]*q[:,1] + q[:,2]*q[:,2]) + q[:,3]*q[:,3]) + q[:,0]*q[:,0]
n2g = ((xg*xg + q[:,2].to(dev)*q[:,2].to(dev)) + q[:,3].to(dev)*q[:,3].to(dev)) + q[:,0].to(dev)*q[:,0].to(dev)
print("n2 neq", (n2g.cpu()!=n2).sum().item(), "sqrt neq", (torch.sqrt(n2g).cpu()!=torch.sqrt(n2)).sum().item(), "1/sqrt neq", ((1.0/torch.sqrt(n2g)).cpu()!=(1.0/torch.sqrt(n2))).sum().item())
a = torch.rand(100000)*3+0.1; b = torch.rand(100000)+0.1
print("div neq", ((a.to(dev)/b.to(dev)).cpu() != a/b).sum().item(), "recip neq", ((1.0/b.to(dev)).cpu() != 1.0/b).sum().item(), "mul-add neq", ((a.to(dev)*b.to(dev)+a.to(dev)).cpu() != a*b+a).sum().item())

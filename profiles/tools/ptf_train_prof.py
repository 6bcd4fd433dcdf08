import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_ptf_hip import _scene
from freesplat_amd.ptf import PixelwiseTripletFusion
dev = torch.device("cuda:0")
V, h, w = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2, 384, 512)
E, Kn, depths, lat, dens, wts, coords = _scene(V, h, w, seed=5)
torch.manual_seed(1)
m = PixelwiseTripletFusion().to(dev)
d = lambda t: t.to(dev)
a = ([d(lat)], [d(coords)], d(dens), d(wts), d(depths), d(E)[None], d(Kn)[None], (h, w))
import time
ins = [t.detach().clone().requires_grad_(True) for t in (a[0][0], a[1][0], a[2], a[3], a[4])]
def step():
    out = m.fuse_gaussians([ins[0]], [ins[1]], ins[2], ins[3], ins[4], *a[5:])
    sum(o.sum() for o in out).backward()
    for t in ins: t.grad = None
    for q in m.gru.parameters(): q.grad = None
for _ in range(3): step()
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); print(f"fold {V} views {h}x{w}: ms/step", (time.perf_counter()-t)*100)

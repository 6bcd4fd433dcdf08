"""Loop for the rocprofv3 kernel trace of the adapter kernels (unprojection, Gaussian head) and the depth-regression tail at
the native 2-view size: 384x512 pixels per view, 290 044 fused Gaussians, 128 planes @ 192x256."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from freesplat_amd.gaussian_adapter import _Head, _Unproject
from freesplat_amd.depth_tail import depth_regression_tail
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
M = int(os.environ.get("ADAPTER_M", 290044))
raw = torch.randn(M, 34, generator=g).to(dev).requires_grad_(True)
dep = (1.0 + torch.rand(M, generator=g)).to(dev).requires_grad_(True)
E = (torch.eye(4).repeat(M, 1, 1) + 0.1 * torch.randn(M, 4, 4, generator=g)).to(dev).requires_grad_(True)
mult = torch.tensor([0.0123], device=dev)
mask = torch.tensor([1.0, .025, .025, .025, .00625, .00625, .00625, .00625, .00625], device=dev)
gs = [torch.randn(M, 3, 3, generator=g).to(dev), torch.randn(M, 3, 9, generator=g).to(dev), torch.randn(M, 3, generator=g).to(dev),
      torch.randn(M, 4, generator=g).to(dev)]
V, h, w = 2, 384, 512
depths = (1.0 + torch.rand(V, h * w, generator=g)).to(dev).requires_grad_(True)
Ev = torch.eye(4).repeat(V, 1, 1).to(dev)
k0 = torch.tensor([400.0, 400.0, 256.0, 192.0], device=dev)
gx = torch.randn(V, h * w, 3, generator=g).to(dev)
lg = (3.0 * torch.randn(V, 128, 192, 256, generator=g)).to(dev).requires_grad_(True)
cd = (torch.log(torch.tensor(0.5)) + torch.linspace(0, 1, 128) * torch.log(torch.tensor(30.0))).to(dev)

def step():
    o = _Head.apply(raw, dep, E, mult, mask, 0.5, 15.0)
    torch.autograd.backward(list(o), gs)
    x = _Unproject.apply(depths, Ev, k0, h, w)
    x.backward(gx)
    t = depth_regression_tail(lg, cd, True)
    torch.autograd.backward([t["depth_map"], t["depth_weights"], t["depth"]],
                            [torch.ones_like(t["depth_map"]), torch.ones_like(t["depth_weights"]), torch.ones_like(t["depth"])])
    for q in (raw, dep, E, depths, lg):
        q.grad = None

for _ in range(3):
    step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t0) * 100)

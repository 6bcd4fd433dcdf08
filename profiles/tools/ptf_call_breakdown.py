"""Host-side timeline of one inference fold call (2 views @ 384x512): where the 0.09 ms between the kernels' 0.19 ms and
the call's 0.28 ms goes.  Times perf_counter marks inside a copy of freesplat_amd.ptf._fuse_gaussians_fused's body."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_ptf_hip import _scene
from freesplat_amd import _lib, ptf as P

dev = torch.device("cuda:0")
V, h, w = 2, 384, 512
E, Kn, depths, lat, dens, wts, coords = _scene(V, h, w, seed=5)
m = P.PixelwiseTripletFusion().to(dev)
d = lambda t: t.to(dev)
a = ([d(lat)], [d(coords)], d(dens), d(wts), d(depths), d(E)[None], d(Kn)[None], (h, w))
marks = {}

def call():
    with torch.no_grad():
        return m.fuse_gaussians(*a)

for _ in range(5):
    call()
torch.cuda.synchronize()
# whole calls
t0 = time.perf_counter()
for _ in range(50):
    call()
torch.cuda.synchronize()
print("call: %.1f us" % ((time.perf_counter() - t0) / 50 * 1e6))

# instrumented: wrap the library call and the sync
L = _lib.lib()
orig_fold = L.fs_ptf_fold
stamps = []
class Wrap:
    def __call__(self, *args):
        stamps.append(("before_fold", time.perf_counter()))
        r = orig_fold(*args)
        stamps.append(("after_fold", time.perf_counter()))
        return r
P._lib.lib = lambda: type("LL", (), {"__getattr__": lambda s, k: Wrap() if k == "fs_ptf_fold" else getattr(L, k)})()
acc = {}
N = 50
for _ in range(N):
    stamps.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    call()
    t1 = time.perf_counter()
    s = dict(stamps)
    for k, v in (("pre_launch", s["before_fold"] - t0), ("fold_call", s["after_fold"] - s["before_fold"]),
                 ("wait_and_views", t1 - s["after_fold"]), ("total", t1 - t0)):
        acc[k] = acc.get(k, 0.0) + v
print({k: round(v / N * 1e6, 1) for k, v in acc.items()}, "us (the wrapper itself adds a few us)")

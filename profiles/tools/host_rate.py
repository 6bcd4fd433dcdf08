import sys, os, time, torch
sys.path.insert(0, os.getcwd())
from freesplat_amd import synthetic
from freesplat_amd.decoder import render_views, check_deferred
dev = torch.device("cuda:0")
H, W, N = synthetic.WORKLOADS["c3_968x1296_1M"]
scene = synthetic.make_scene(N)
cams = {k: v.to(dev) for k, v in synthetic.target_cameras(16).items()}
g = {k: scene[k].to(dev) for k in ("means", "covariances", "harmonics", "opacities")}
bg = torch.zeros(16, 3, device=dev)
def step():
    with torch.no_grad():
        return render_views(cams["extrinsics"], cams["intrinsics"], cams["near"], cams["far"], (H, W), bg,
                            g["means"], g["covariances"], g["harmonics"], g["opacities"], check="deferred")
for _ in range(5): step()
check_deferred(); torch.cuda.synchronize()
for n in (20, 100, 300):
    t0 = time.perf_counter()
    for _ in range(n): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    check_deferred()
    print(n, "steps: enqueue %.3f ms/step, total %.3f ms/step -> %.1f views/s" % ((t1-t0)/n*1e3, (t2-t0)/n*1e3, 16*n/(t2-t0)))

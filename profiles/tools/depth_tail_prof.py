"""Loop for the rocprofv3 kernel trace of the depth-regression tail (csrc/depth_tail.hip) alone:
python profiles/tools/depth_tail_prof.py <B> <h2> <w2> [steps]   -- logits [B,128,h2,w2], forward + backward of all three outputs.
Native 2-view size: 2 192 256; BASELINE config 3: 3 484 648 (481 MB of logits)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from freesplat_amd.depth_tail import depth_regression_tail
B, h2, w2 = (int(a) for a in sys.argv[1:4])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
lg = (3.0 * torch.randn(B, 128, h2, w2, generator=g)).to(dev).requires_grad_(True)
cd = (torch.log(torch.tensor(0.5)) + torch.linspace(0, 1, 128) * torch.log(torch.tensor(30.0))).to(dev)
with torch.no_grad():
    t = depth_regression_tail(lg, cd, True)
cot = [torch.ones_like(t["depth_map"]), torch.ones_like(t["depth_weights"]), torch.ones_like(t["depth"])]

def step():
    t = depth_regression_tail(lg, cd, True)
    torch.autograd.backward([t["depth_map"], t["depth_weights"], t["depth"]], cot)
    lg.grad = None

for _ in range(3):
    step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize(); print(f"depth tail B={B} {h2}x{w2}: ms/step {(time.perf_counter() - t0) * 1e3 / steps:.4f}")

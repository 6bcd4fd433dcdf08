import ctypes as C, json, os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests", "golden"))
import numpy as np, torch, inputs
from freesplat_amd import _lib
from freesplat_amd.cost_volume import AVGFeatureVolumeManager
V, K, h4, w4, D = 3, 2, 242, 324, 128
dev = torch.device("cuda:0")
L = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * (16384 * 6))()
torch.manual_seed(0)
m = AVGFeatureVolumeManager(h4, w4, num_depth_bins=D, mlp_channels=[202, 32, 32, 1], matching_dim_size=48).to(dev)
kw = {k: v.to(dev) for k, v in inputs.cv_inputs(V, K, h4, w4, 48, seed=1).items()}
kw = {k: (v.clone().requires_grad_(True) if k in ("cur_feats", "src_feats") else v) for k, v in kw.items()}
for i in range(3):
    if i == 2:
        torch.cuda.synchronize(); L.fs_debug_cvb_trace(buf, 1)
    o = m(**kw); o.backward(torch.ones_like(o))
torch.cuda.synchronize(); L.fs_debug_cvb_trace(buf, 0)
a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 6).astype(np.float64)
a = a[a[:, 4] > 0]; planes = a[:, 4].sum()
names = ["gather", "x_handover_layer1", "layer2_dz2_dh1", "dW2_dz1_dx_dW1"]
out = {"wavefronts": int(len(a)), "items_per_wavefront": float(a[:, 4].mean())}
out.update({nm: round(float(a[:, i].sum() / planes)) for i, nm in enumerate(names)})
out["total_per_item"] = round(float(a[:, 5].sum() / planes))
print(os.environ.get("FREESPLAT_LIB", ""), json.dumps(out))

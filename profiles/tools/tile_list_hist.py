import torch, numpy as np, sys
sys.path.insert(0,'.')
from freesplat_amd import synthetic
from freesplat_amd.decoder import render_views
dev=torch.device('cuda:0')
for name in ("c3_968x1296_1M","c2_640x480_300k"):
    H,W,N=synthetic.WORKLOADS[name]
    sc=synthetic.workload_scene(name); cams=synthetic.target_cameras(16)
    g={k:sc[k].to(dev).requires_grad_(True) for k in ("means","covariances","harmonics","opacities")}
    c={k:v.to(dev) for k,v in cams.items()}
    col,dep=render_views(c["extrinsics"],c["intrinsics"],c["near"],c["far"],(H,W),torch.zeros(16,3,device=dev),g["means"],g["covariances"],g["harmonics"],g["opacities"])
    st=col.grad_fn.states
    T=((W+15)//16)*((H+15)//16)
    for i in (0,7,15):
        off=st[i].binning[:(T+1)*4].view(torch.int32).cpu().numpy().astype(np.int64)
        cnt=np.diff(off)
        print(name,"view",i,"tiles",T,"mean",cnt.mean(),"p50",np.percentile(cnt,50),"p90",np.percentile(cnt,90),"p99",np.percentile(cnt,99),"max",cnt.max(),">1536",(cnt>1536).sum(),">1792",(cnt>1792).sum(),">2048",(cnt>2048).sum())

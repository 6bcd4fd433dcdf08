"""Worker of tests/test_multi_rank_one_gpu.py: launched by torch.distributed.run with 2 ranks that SHARE cuda:0
(gloo collectives; RCCL refuses two ranks on one device).  Exercises on real HIP kernels what the gloo CPU tests
exercise with stand-ins: the process-group path of DecoderSplattingCUDA (view-sharded rendering, image all-gather,
gradient sum) and sharded_cost_volume, each against the single-process result computed in the same process."""
import json
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE, os.path.join(HERE, "golden")]


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {}
    from freesplat_amd.decoder import DecoderSplattingCUDA, Gaussians
    from util_raster import small_scene
    H, W, v = 48, 64, 5                      # 5 views over 2 ranks: ragged shards (3 + 2)
    scene, cams = small_scene(N=2000, H=H, W=W, seed=17, n_views=v)
    cam = {k: t.to(dev)[None] for k, t in cams.items()}
    wgt = torch.randn(1, v, 3, H, W, generator=torch.Generator().manual_seed(5)).to(dev)
    wd = torch.randn(1, v, H, W, generator=torch.Generator().manual_seed(6)).to(dev)

    def run(group):
        leaves = {k: scene[k].to(dev)[None].clone().requires_grad_(True) for k in ("means", "covariances", "harmonics", "opacities")}
        dec = DecoderSplattingCUDA(None, None, background_color=(0.1, 0.2, 0.3), group=group).to(dev)
        out = dec(Gaussians(**leaves), cam["extrinsics"], cam["intrinsics"], cam["near"], cam["far"], (H, W), depth_mode="depth")
        ((out.color * wgt).sum() + (out.depth * wd).sum()).backward()
        return out, {k: t.grad for k, t in leaves.items()}

    ref, gref = run(None)
    out, g = run(True)
    res["decoder_color_equal"] = bool(torch.equal(out.color, ref.color))
    res["decoder_depth_equal"] = bool(torch.equal(out.depth, ref.depth))
    res["decoder_grad_err"] = {k: float((g[k] - gref[k]).abs().max() / (gref[k].abs().max() + 1e-20)) for k in g}

    import inputs
    from freesplat_amd.cost_volume import AVGFeatureVolumeManager, sharded_cost_volume
    V, h4, w4, D, C = 5, 24, 32, 16, 48
    E, Kn = inputs.cameras(V, h4, w4, baseline=1.0, seed=3)
    feats = torch.randn(V, C, h4, w4, generator=torch.Generator().manual_seed(2)).to(dev)
    torch.manual_seed(3)
    m = AVGFeatureVolumeManager(h4, w4, num_depth_bins=D, mlp_channels=[202, 32, 32, 1], matching_dim_size=C).to(dev)
    near, far = torch.full((1, V), 0.5, device=dev), torch.full((1, V), 15.0, device=dev)
    from freesplat_amd.encoder_glue import prepare_cost_volume_inputs
    f_ref = feats.clone().requires_grad_(True)
    vol = m(**prepare_cost_volume_inputs(E[None].to(dev), Kn[None].to(dev), f_ref, near, far, (4 * h4, 4 * w4), 3))
    wv = torch.randn(vol.shape, generator=torch.Generator().manual_seed(4)).to(dev)
    (vol * wv).sum().backward()
    from freesplat_amd.view_sharding import shard_range
    mine = shard_range(V, rank, world)
    f_loc = feats[mine.start: mine.stop].clone().requires_grad_(True)
    loc = sharded_cost_volume(m, f_loc, E[None].to(dev), Kn[None].to(dev), near, far, (4 * h4, 4 * w4), 3)
    res["cv_rows_equal"] = bool(torch.equal(loc, vol.detach()[mine.start: mine.stop]))
    (loc * wv[mine.start: mine.stop]).sum().backward()
    gr = f_ref.grad[mine.start: mine.stop]
    res["cv_feat_grad_err"] = float((f_loc.grad - gr).abs().max() / (gr.abs().max() + 1e-20))
    torch.cuda.synchronize()
    allres = [None] * world
    dist.all_gather_object(allres, res)
    if rank == 0:
        print("DIST_WORKER_RESULT " + json.dumps(allres), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

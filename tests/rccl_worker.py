"""Worker of tests/test_rccl_world1.py: ONE rank, launched by torch.distributed.run exactly as the driver launches
bench.py, with backend "nccl" (= RCCL on ROCm) bound to cuda:0.  Every collective entry point of
freesplat_amd.view_sharding and the process-group paths of the decoder / the cost volume run on DEVICE tensors
through RCCL (no host staging: `_stage` is the identity for this backend); with one rank each collective must
reproduce the unsharded result, so every check is an equality against the plain single-process computation."""
import json
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE, os.path.join(HERE, "golden")]


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert world == 1
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    res = {"backend": dist.get_backend()}
    from freesplat_amd import view_sharding as vs
    gen = torch.Generator().manual_seed(1)
    imgs = torch.rand(5, 3, 24, 32, generator=gen).to(dev)
    res["gather_views"] = bool(torch.equal(vs.gather_views(imgs, 5), imgs))
    ag = vs.AsyncViewGather(5, device=dev)
    outs = []
    for k in range(3):                 # overlapped with "rendering" (a few kernels on the current stream)
        frame = imgs * float(k + 1)
        ag.launch(frame)
        _ = (imgs @ imgs.transpose(-1, -2)).sum()
        got = ag.wait()
        outs.append(bool(torch.equal(got, frame)))
    res["async_view_gather"] = all(outs) and ag.wait() is None
    g = [torch.randn(1000, 3, generator=gen).to(dev), None, torch.randn(1000, 3, 9, generator=gen).to(dev),
         torch.randn(1000, generator=gen).to(dev), torch.randn(1000, 3, 3, generator=gen).to(dev)]
    ref = [None if t is None else t.clone() for t in g]
    sh = vs.reduce_scatter_gaussian_grads(g)
    res["reduce_scatter"] = all((a is None and b is None) or torch.equal(a, b) for a, b in zip(sh, ref))
    sh2 = vs.reduce_scatter_gaussian_grads(g)       # second call reuses the bucket
    res["reduce_scatter_bucket_reuse"] = all((a is None and b is None) or torch.equal(a, b) for a, b in zip(sh2, ref))
    vs.allreduce_gaussian_grads(g)
    res["all_reduce"] = all((a is None and b is None) or torch.equal(a, b) for a, b in zip(g, ref))
    res["grad_exchange"] = all(torch.equal(a, b) for a, b in zip(vs.GradExchange("reduce_scatter")([ref[0], ref[2]]), [ref[0], ref[2]]))
    # autograd-aware gathers
    x = torch.randn(4, 6, 5, generator=gen).to(dev).requires_grad_(True)
    w = torch.randn(4, 6, 5, generator=gen).to(dev)
    (vs.gather_views_autograd(x, 4) * w).sum().backward()
    res["gather_views_autograd_grad"] = bool(torch.equal(x.grad, w))
    x2 = torch.randn(4, 6, 5, generator=gen).to(dev).requires_grad_(True)
    (vs.gather_features_autograd(x2, 4) * w).sum().backward()       # backward = reduce-scatter
    res["gather_features_autograd_grad"] = bool(torch.equal(x2.grad, w))

    # the decoder's process-group path on the real kernels (forced onto the sharded branch with one rank)
    from freesplat_amd.decoder import DecoderSplattingCUDA, Gaussians
    from util_raster import small_scene
    H, W, v = 48, 64, 5
    scene, cams = small_scene(N=2000, H=H, W=W, seed=17, n_views=v)
    cam = {k: t.to(dev)[None] for k, t in cams.items()}
    wgt = torch.randn(1, v, 3, H, W, generator=torch.Generator().manual_seed(5)).to(dev)
    wd = torch.randn(1, v, H, W, generator=torch.Generator().manual_seed(6)).to(dev)

    def run(group, depth_mode="depth"):
        leaves = {k: scene[k].to(dev)[None].clone().requires_grad_(True) for k in ("means", "covariances", "harmonics", "opacities")}
        dec = DecoderSplattingCUDA(None, None, background_color=(0.1, 0.2, 0.3), group=group,
                                   single_rank_collectives=group is not None).to(dev)
        out = dec(Gaussians(**leaves), cam["extrinsics"], cam["intrinsics"], cam["near"], cam["far"], (H, W), depth_mode=depth_mode)
        loss = (out.color * wgt).sum() + ((out.depth * wd).sum() if out.depth is not None else 0.0)
        loss.backward()
        return out, {k: t.grad for k, t in leaves.items()}

    ref_o, gref = run(None)
    out, gg = run(True)
    res["decoder_color_equal"] = bool(torch.equal(out.color, ref_o.color))
    res["decoder_depth_equal"] = bool(torch.equal(out.depth, ref_o.depth))
    res["decoder_grad_err"] = {k: float((gg[k] - gref[k]).abs().max() / (gref[k].abs().max() + 1e-20)) for k in gg}
    out_nd, _ = run(True, depth_mode=None)          # colour-only gather
    res["decoder_color_only_equal"] = bool(out_nd.depth is None and torch.equal(out_nd.color, ref_o.color))
    # replica check: a rank-dependent scene must be refused (with one rank: exercise the collective itself)
    dec = DecoderSplattingCUDA(None, None, group=True, single_rank_collectives=True).to(dev)
    dec._check_replicas(None, dist, Gaussians(**{k: scene[k].to(dev)[None] for k in ("means", "covariances", "harmonics", "opacities")}),
                        cam["extrinsics"])
    res["replica_check_ran"] = True

    import inputs
    from freesplat_amd.cost_volume import AVGFeatureVolumeManager, sharded_cost_volume
    from freesplat_amd.encoder_glue import prepare_cost_volume_inputs
    V, h4, w4, D, Cc = 4, 24, 32, 16, 48
    E, Kn = inputs.cameras(V, h4, w4, baseline=1.0, seed=3)
    feats = torch.randn(V, Cc, h4, w4, generator=torch.Generator().manual_seed(2)).to(dev)
    torch.manual_seed(3)
    m = AVGFeatureVolumeManager(h4, w4, num_depth_bins=D, mlp_channels=[202, 32, 32, 1], matching_dim_size=Cc).to(dev)
    near, far = torch.full((1, V), 0.5, device=dev), torch.full((1, V), 15.0, device=dev)
    f_ref = feats.clone().requires_grad_(True)
    vol = m(**prepare_cost_volume_inputs(E[None].to(dev), Kn[None].to(dev), f_ref, near, far, (4 * h4, 4 * w4), 3))
    wv = torch.randn(vol.shape, generator=torch.Generator().manual_seed(4)).to(dev)
    (vol * wv).sum().backward()
    f_loc = feats.clone().requires_grad_(True)
    loc = sharded_cost_volume(m, f_loc, E[None].to(dev), Kn[None].to(dev), near, far, (4 * h4, 4 * w4), 3)
    res["cv_rows_equal"] = bool(torch.equal(loc, vol.detach()))
    (loc * wv).sum().backward()
    res["cv_feat_grad_err"] = float((f_loc.grad - f_ref.grad).abs().max() / (f_ref.grad.abs().max() + 1e-20))
    torch.cuda.synchronize()
    dist.barrier()
    print("RCCL_WORKER_RESULT " + json.dumps(res), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

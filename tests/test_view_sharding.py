"""World-size-2 gloo tests (CPU) of the N>1 path: view partition, ragged image all-gather, flat
gradient all-reduce -- the exact functions bench.py / the sharded decoder use under RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from freesplat_amd.view_sharding import (AsyncViewGather, GradExchange, allreduce_gaussian_grads, chunk_row_ranges,
                                         chunked_owned_rows, gather_views, gather_views_autograd,
                                         reduce_scatter_gaussian_grads, replicate_gaussians, shard_counts, shard_range)


def test_shard_range_partition():
    for n in (0, 1, 5, 8, 17):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                seen += list(shard_range(n, r, world))
            assert seen == list(range(n))
            c = shard_counts(n, world)
            assert sum(c) == n and max(c) - min(c) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_render(view_id, h=6, w=8):
    g = torch.Generator().manual_seed(1000 + view_id)
    return torch.rand(3, h, w, generator=g)


def _worker(rank, world, port, n_views, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = shard_range(n_views, rank, world)
        local = torch.stack([_fake_render(v) for v in mine]) if len(mine) else torch.zeros(0, 3, 6, 8)
        full = gather_views(local, n_views)
        expect = torch.stack([_fake_render(v) for v in range(n_views)])
        ok_gather = torch.equal(full, expect)
        ag = AsyncViewGather(n_views, device=torch.device("cpu"))
        ag.launch(local)
        ok_async = torch.equal(ag.wait(), expect) and ag.wait() is None
        # gradient all-reduce: rank r contributes (r+1) * pattern
        g1 = torch.arange(12.0).reshape(4, 3) * (rank + 1)
        g2 = torch.ones(4, 6) * (rank + 1)
        allreduce_gaussian_grads([g1, None, g2])
        tot = sum(r + 1 for r in range(world))
        ok_red = torch.equal(g1, torch.arange(12.0).reshape(4, 3) * tot) and torch.equal(g2, torch.ones(4, 6) * tot)
        q.put((rank, ok_gather, ok_async, ok_red))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_views", [4, 5])
def test_gloo_world2_gather_and_allreduce(n_views):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_views, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(all(r[1:]) for r in res), res


def _run(world, target, *args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res, key=lambda r: r[0])


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _rs_worker(rank, world, port, q, N):
    _init(rank, world, port)
    try:
        g = torch.Generator().manual_seed(7 + rank)
        grads = [torch.randn(N, 3, generator=g), torch.randn(N, 3, 3, generator=g), None,
                 torch.randn(N, 3, 9, generator=g), torch.randn(N, generator=g)]
        shards = reduce_scatter_gaussian_grads([None if t is None else t.clone() for t in grads])
        full = [None if t is None else t.clone() for t in grads]
        allreduce_gaussian_grads(full)
        mine = shard_range(N, rank, world)
        ok = shards[2] is None
        for sh, fu in zip(shards, full):
            if fu is not None:
                ok = ok and sh.shape == fu[mine.start: mine.stop].shape and torch.allclose(sh, fu[mine.start: mine.stop], atol=1e-6)
        ex = GradExchange("reduce_scatter")([None if t is None else t.clone() for t in grads])
        ok = ok and all((a is None and b is None) or torch.equal(a, b) for a, b in zip(ex, shards))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("N", [10, 7, 1])
def test_gloo_world2_reduce_scatter_equals_allreduce_shards(N):
    """The reduce-scatter gradient exchange: every rank receives the total for the Gaussian
    rows it owns; ragged row counts (N not divisible by the world size, N < world) included."""
    assert all(ok for _, ok in _run(2, _rs_worker, N))


def _toy_render(means, cov, sh, op, cams):
    """A differentiable stand-in for render_views on CPU: one [4, 2, 3] 'image' per camera, nonlinear in every
    Gaussian tensor (the HIP rasterizer cannot run here; what is under test is the sharding + autograd plumbing)."""
    feat = torch.cat([means, cov.reshape(-1, 9), sh.reshape(-1, 27), op[:, None]], dim=1)      # [N,40]
    w = torch.sin(cams[:, None, :] * torch.arange(1, 41, dtype=torch.float32)[None, :, None] * 0.1)   # [v,40,24]
    return torch.tanh(torch.einsum("nf,vfk->vk", feat, w)).reshape(-1, 4, 2, 3)


def _decoder_worker(rank, world, port, q, v):
    _init(rank, world, port)
    try:
        g = torch.Generator().manual_seed(3)          # identical Gaussians / cameras on every rank
        N = 13
        leaves = [torch.randn(N, 3, generator=g), torch.randn(N, 3, 3, generator=g), torch.randn(N, 3, 9, generator=g),
                  torch.rand(N, generator=g)]
        cams = torch.randn(v, 24, generator=g)
        wgt = torch.randn(v, 4, 2, 3, generator=g)
        ref_leaves = [t.clone().requires_grad_(True) for t in leaves]
        ref = _toy_render(*ref_leaves, cams)
        (ref * wgt).sum().backward()
        # sharded: this rank renders its block of the views, images gathered, gradients summed over ranks
        sh_leaves = [t.clone().requires_grad_(True) for t in leaves]
        mine = shard_range(v, rank, world)
        rep = replicate_gaussians(sh_leaves)
        local = _toy_render(*rep, cams[mine.start: mine.stop]) if len(mine) else torch.zeros(0, 4, 2, 3) + 0.0 * sum(t.sum() for t in rep)
        full = gather_views_autograd(local, v)
        ok = torch.allclose(full, ref.detach(), atol=1e-6)
        (full * wgt).sum().backward()                 # every rank computes the same loss on the gathered views
        for a, b in zip(sh_leaves, ref_leaves):
            ok = ok and torch.allclose(a.grad, b.grad, atol=1e-5)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("v", [4, 3, 1])
def test_gloo_world2_sharded_decoder_autograd(v):
    """The process-group path of DecoderSplattingCUDA (gather_views_autograd + replicate_gaussians) reproduces the
    single-process outputs and Gaussian gradients on every rank; ragged and under-subscribed view counts included."""
    assert all(ok for _, ok in _run(2, _decoder_worker, v))


def _cv_worker(rank, world, port, q, V, ncv):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    _init(rank, world, port)
    try:
        import inputs
        from oracle import cost_volume_oracle as cvo
        from freesplat_amd.cost_volume import sharded_cost_volume
        from freesplat_amd.encoder_glue import prepare_cost_volume_inputs
        h4, w4, D, C = 6, 8, 4, 48
        E, Kn = inputs.cameras(V, h4, w4, baseline=0.8, seed=5)
        feats = torch.randn(V, C, h4, w4, generator=torch.Generator().manual_seed(9))
        g = torch.Generator().manual_seed(1)
        mlp = cvo.mlp_from_state({"mlp__net__0__weight": torch.randn(32, 49, generator=g) * 0.2, "mlp__net__0__bias": torch.randn(32, generator=g) * 0.1,
                                  "mlp__net__2__weight": torch.randn(32, 32, generator=g) * 0.2, "mlp__net__2__bias": torch.randn(32, generator=g) * 0.1,
                                  "mlp__net__4__weight": torch.randn(1, 32, generator=g) * 0.2, "mlp__net__4__bias": torch.randn(1, generator=g) * 0.1})

        def rows(cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK, min_depth, max_depth):
            return cvo.cost_volume(cur_feats, src_feats, src_extrinsics, src_Ks, cur_invK, min_depth, max_depth, D, mlp)
        near, far = torch.full((1, V), 0.5), torch.full((1, V), 15.0)
        f_ref = feats.clone().requires_grad_(True)
        ref = rows(**prepare_cost_volume_inputs(E[None], Kn[None], f_ref, near, far, (4 * h4, 4 * w4), ncv))
        wgt = torch.randn(ref.shape, generator=torch.Generator().manual_seed(2))
        (ref * wgt).sum().backward()
        mine = shard_range(V, rank, world)
        f_loc = feats[mine.start: mine.stop].clone().requires_grad_(True)
        out = sharded_cost_volume(rows, f_loc, E[None], Kn[None], near, far, (4 * h4, 4 * w4), ncv)
        ok = out.shape[0] == len(mine) and torch.allclose(out, ref.detach()[mine.start: mine.stop], atol=1e-6)
        (out * wgt[mine.start: mine.stop]).sum().backward()          # each rank: the loss of ITS volumes
        ok = ok and torch.allclose(f_loc.grad, f_ref.grad[mine.start: mine.stop], atol=1e-5)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("V,ncv", [(4, 9), (5, 3)])
def test_gloo_world2_sharded_cost_volume(V, ncv):
    """Cost-volume view sharding: all-gather of the 1/4-resolution features, every rank sweeps
    its own current views; volumes equal the unsharded rows and the feature gradients (which cross ranks through the
    source views: reduce-scatter in backward) equal the unsharded ones.  Row compute = the reference-pinned oracle
    (the HIP kernel cannot run on CPU); V=5 with 3-nearest source selection covers ragged shards."""
    assert all(ok for _, ok in _run(2, _cv_worker, V, ncv))


def _world3_worker(rank, world, port, q, n_views, N):
    _init(rank, world, port)
    try:
        mine = shard_range(n_views, rank, world)
        local = torch.stack([_fake_render(v) for v in mine]) if len(mine) else torch.zeros(0, 3, 6, 8)
        expect = torch.stack([_fake_render(v) for v in range(n_views)])
        ok = torch.equal(gather_views(local, n_views), expect)          # ragged: 4 views -> [2, 1, 1]; trimmed by block copies
        g = torch.Generator().manual_seed(11 + rank)
        grads = [torch.randn(N, 3, generator=g), torch.randn(N, 3, 9, generator=g), torch.randn(N, generator=g)]
        for _ in range(2):                                              # second call reuses the bucket
            shards = reduce_scatter_gaussian_grads([t.clone() for t in grads])
            full = [t.clone() for t in grads]
            allreduce_gaussian_grads(full)
            rows = shard_range(N, rank, world)
            ok = ok and all(torch.allclose(sh, fu[rows.start: rows.stop], atol=1e-6) for sh, fu in zip(shards, full))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_views,N", [(4, 9), (7, 10), (2, 2)])
def test_gloo_world3_ragged_gather_and_bucket_packing(n_views, N):
    """Three ranks: the trimmed all-gather of ragged view shards (more than one short shard), the reduce-scatter bucket with
    evenly divisible rows (N = 9: one strided copy per tensor) and ragged ones (N = 10, N < world), and its reuse."""
    assert all(ok for _, ok in _run(3, _world3_worker, n_views, N))


def _toy_render_views(extrinsics, intrinsics, near, far, image_shape, bg, means, cov, sh, op, **_):
    """render_views' signature on CPU tensors (the HIP rasterizer cannot run here): [v,3,h,w] colour and [v,1,h,w] depth that
    depend nonlinearly on every Gaussian tensor and on the view's camera."""
    h, w = image_shape
    v = extrinsics.shape[0]
    feat = torch.cat([means, cov.reshape(-1, 9), sh.reshape(-1, 27), op[:, None]], dim=1)                 # [N,40]
    cam = extrinsics.reshape(v, 16)[:, :12]                                                                # [v,12]
    k = torch.arange(1, 4 * h * w + 1, dtype=torch.float32).reshape(1, 1, -1) * 0.01
    wgt = torch.sin(cam.sum(1)[:, None, None] + k * torch.arange(1, 41, dtype=torch.float32)[None, :, None])   # [v,40,4hw]
    img = torch.tanh(torch.einsum("nf,vfk->vk", feat, wgt) * 0.1).reshape(v, 4, h, w)
    return img[:, :3] + bg[:, :, None, None], img[:, 3:4]


def _world8_worker(rank, world, port, q):
    """BASELINE config 4's real split on 8 ranks: 10 context views (shards 2,2,1,1,1,1,1,1) through sharded_cost_volume, 4 target
    views (four EMPTY shards) through gather_views, reduce_scatter_gaussian_grads and DecoderSplattingCUDA(group=...)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    _init(rank, world, port)
    try:
        import inputs
        from oracle import cost_volume_oracle as cvo
        from freesplat_amd import decoder as D_
        from freesplat_amd.cost_volume import sharded_cost_volume
        from freesplat_amd.encoder_glue import prepare_cost_volume_inputs
        ok = shard_counts(10, 8) == [2, 2, 1, 1, 1, 1, 1, 1] and shard_counts(4, 8) == [1, 1, 1, 1, 0, 0, 0, 0]
        # ---- cost volume: 10 context views, the 9 pose-nearest as sources (K = 8) ----
        V, ncv, h4, w4, Dp, C = 10, 9, 6, 8, 4, 48
        E, Kn = inputs.cameras(V, h4, w4, baseline=1.2, seed=5)
        feats = torch.randn(V, C, h4, w4, generator=torch.Generator().manual_seed(9))
        g = torch.Generator().manual_seed(1)
        mlp = cvo.mlp_from_state({"mlp__net__0__weight": torch.randn(32, 49, generator=g) * 0.2, "mlp__net__0__bias": torch.randn(32, generator=g) * 0.1,
                                  "mlp__net__2__weight": torch.randn(32, 32, generator=g) * 0.2, "mlp__net__2__bias": torch.randn(32, generator=g) * 0.1,
                                  "mlp__net__4__weight": torch.randn(1, 32, generator=g) * 0.2, "mlp__net__4__bias": torch.randn(1, generator=g) * 0.1})

        def rows(cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK, min_depth, max_depth):
            return cvo.cost_volume(cur_feats, src_feats, src_extrinsics, src_Ks, cur_invK, min_depth, max_depth, Dp, mlp)
        near, far = torch.full((1, V), 0.5), torch.full((1, V), 15.0)
        f_ref = feats.clone().requires_grad_(True)
        ref = rows(**prepare_cost_volume_inputs(E[None], Kn[None], f_ref, near, far, (4 * h4, 4 * w4), ncv))
        wgt = torch.randn(ref.shape, generator=torch.Generator().manual_seed(2))
        (ref * wgt).sum().backward()
        mine = shard_range(V, rank, world)
        f_loc = feats[mine.start: mine.stop].clone().requires_grad_(True)
        out = sharded_cost_volume(rows, f_loc, E[None], Kn[None], near, far, (4 * h4, 4 * w4), ncv)
        ok = ok and out.shape[0] == len(mine) and torch.allclose(out, ref.detach()[mine.start: mine.stop], atol=1e-6)
        (out * wgt[mine.start: mine.stop]).sum().backward()
        ok = ok and torch.allclose(f_loc.grad, f_ref.grad[mine.start: mine.stop], atol=1e-5)
        # ---- 4 target views on 8 ranks: plain gather with four empty shards ----
        tv = 4
        tmine = shard_range(tv, rank, world)
        local = torch.stack([_fake_render(v) for v in tmine]) if len(tmine) else torch.zeros(0, 3, 6, 8)
        ok = ok and torch.equal(gather_views(local, tv), torch.stack([_fake_render(v) for v in range(tv)]))
        # ---- Gaussian-gradient reduce-scatter, rows not divisible by 8 ----
        N = 21
        gg = torch.Generator().manual_seed(7 + rank)
        grads = [torch.randn(N, 3, generator=gg), torch.randn(N, 3, 3, generator=gg), torch.randn(N, 3, 9, generator=gg), torch.randn(N, generator=gg)]
        shards = reduce_scatter_gaussian_grads([t.clone() for t in grads])
        full = [t.clone() for t in grads]
        allreduce_gaussian_grads(full)
        rws = shard_range(N, rank, world)
        ok = ok and all(torch.allclose(sh, fu[rws.start: rws.stop], atol=1e-5) for sh, fu in zip(shards, full))
        # ---- the decoder itself: DecoderSplattingCUDA(group=True)._forward_sharded with the CPU stand-in renderer ----
        D_.render_views = _toy_render_views
        gen = torch.Generator().manual_seed(3)          # the same scene on every rank (the decoder checks that)
        Ng = 13
        leaves = [torch.randn(Ng, 3, generator=gen), torch.randn(Ng, 3, 3, generator=gen), torch.randn(Ng, 3, 9, generator=gen),
                  torch.rand(Ng, generator=gen)]
        Et = torch.randn(1, tv, 4, 4, generator=gen)
        Kt = torch.rand(1, tv, 3, 3, generator=gen)
        nf = torch.full((1, tv), 0.5), torch.full((1, tv), 15.0)
        wimg = torch.randn(1, tv, 3, 5, 6, generator=gen)
        wdep = torch.randn(1, tv, 5, 6, generator=gen)
        bgc = (0.1, 0.2, 0.3)
        ref_l = [t.clone().requires_grad_(True) for t in leaves]
        c_ref, d_ref = _toy_render_views(Et[0], Kt[0], nf[0][0], nf[1][0], (5, 6), torch.tensor(bgc)[None].expand(tv, 3), *ref_l)
        ((c_ref[None] * wimg).sum() + (d_ref[None, :, 0] / 2 * wdep).sum()).backward()
        sh_l = [t.clone().requires_grad_(True) for t in leaves]
        dec = D_.DecoderSplattingCUDA(background_color=bgc, group=True)
        gs = D_.Gaussians(*(t[None] for t in sh_l))
        o = dec(gs, Et, Kt, nf[0], nf[1], (5, 6), depth_mode="depth")
        ok = ok and torch.allclose(o.color[0], c_ref.detach(), atol=1e-6) and torch.allclose(o.depth[0], d_ref.detach()[:, 0] / 2, atol=1e-6)
        ((o.color * wimg).sum() + (o.depth * wdep).sum()).backward()
        for a, b_ in zip(sh_l, ref_l):
            ok = ok and torch.allclose(a.grad, b_.grad, atol=1e-5)
        # a rank holding a DIFFERENT scene must be refused (Lightning DDP hands every rank its own batch)
        if world > 1:
            bad = [t.clone() for t in leaves]
            if rank == 5:
                bad[0] = bad[0] + 1.0
            dec2 = D_.DecoderSplattingCUDA(background_color=bgc, group=True)
            try:
                dec2(D_.Gaussians(*(t[None] for t in bad)), Et, Kt, nf[0], nf[1], (5, 6))
                ok = False
            except RuntimeError:
                pass
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_gloo_world8_config4_split():
    """the multi-GPU code paths at the world size they will first meet on hardware, with BASELINE config 4's
    real numbers -- 10 context views on 8 ranks (2,2,1,1,1,1,1,1), 4 target views on 8 ranks (four ranks render nothing) --
    through sharded_cost_volume (all-gather of features, reduce-scatter in backward), gather_views, the Gaussian-gradient
    reduce-scatter and DecoderSplattingCUDA(group=...) forward + backward (outputs and gradients equal the unsharded ones on
    every rank, empty shards keep the graph connected, a rank with a different scene is refused)."""
    assert all(ok for _, ok in _run(8, _world8_worker))


def test_chunk_row_ranges_partition():
    for n in (0, 1, 255, 256, 1000, 100_000):
        for c in (1, 3, 4, 8):
            ch = chunk_row_ranges(n, c)
            assert [r for c0, c1 in ch for r in range(c0, c1)] == list(range(n)) and len(ch) <= c
            assert all(c0 % 256 == 0 for c0, _ in ch)
            for world in (1, 2, 8):
                owned = [r for rk in range(world) for rg in chunked_owned_rows(n, rk, world, c) for r in rg]
                assert sorted(owned) == list(range(n))


def _chunked_worker(rank, world, port, q, N, chunks):
    _init(rank, world, port)
    try:
        g = torch.Generator().manual_seed(21 + rank)
        grads = [torch.randn(N, 3, generator=g), torch.randn(N, 3, 3, generator=g), torch.randn(N, 3, 9, generator=g), torch.randn(N, generator=g)]
        full = [t.clone() for t in grads]
        allreduce_gaussian_grads(full)
        ex = GradExchange("chunked", chunks=chunks)
        # what decoder._RenderViews.backward does through the hook: rows become final chunk by chunk
        ex.begin(N)
        for c0, c1 in ex.chunk_rows(N):
            ex.chunk_ready(c0, c1, grads)
        got = ex(grads)
        rows = [r for rg in chunked_owned_rows(N, rank, world, chunks) for r in rg]
        ok = all(torch.allclose(sh, fu[rows], atol=1e-5) and sh.shape[0] == len(rows) for sh, fu in zip(got, full))
        raised = False
        try:
            ex(grads)           # nothing ran through the hook since the last call
        except RuntimeError:
            raised = True
        q.put((rank, bool(ok and raised)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("N,chunks", [(1000, 4), (700, 3), (100, 4)])
def test_gloo_world2_chunked_exchange_equals_allreduce_rows(N, chunks):
    """GradExchange("chunked"): the reduce-scatter issued chunk by chunk of the rows leaves every rank with
    the all-reduced values of the rows chunked_owned_rows names -- together every row exactly once."""
    assert all(ok for _, ok in _run(2, _chunked_worker, N, chunks))

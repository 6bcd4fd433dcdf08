"""World-size-2 gloo tests (CPU) of the N>1 path: view partition, ragged image all-gather, flat
gradient all-reduce -- the exact functions bench.py / the sharded decoder use under RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from freesplat_amd.view_sharding import (AsyncViewGather, allreduce_gaussian_grads, gather_views,
                                         shard_counts, shard_range)


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

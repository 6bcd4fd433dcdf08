"""Multi-GPU sharding of the rasterizer: one process per GPU, target views of one
Gaussian set are split across ranks; the only exchanges are an all-gather of the rendered images
(forward) and an all-reduce of the per-Gaussian gradients (training).  `torch.distributed` backend
"nccl" is RCCL over xGMI on ROCm; the same code runs on "gloo" for the CPU tests.

The reference has no explicit distributed code (Lightning DDP over scenes only, src/main.py:98-103);
this is the view-sharded decoder the north_star asks for.

Exchanges:
  rasterizer forward   all-gather of the rendered colour+depth images       gather_views / AsyncViewGather / gather_views_autograd
  rasterizer backward  sum of the per-Gaussian gradients of the view shards
                         - reduce-scatter by Gaussian rows (each rank receives the total for the rows it owns; 1/G of
                           the bucket crosses each link, the direct pattern on the 7 point-to-point xGMI links)
                                                                            reduce_scatter_gaussian_grads
                         - all-reduce when every rank needs every row (replicated encoder)
                                                                            allreduce_gaussian_grads / replicate_gaussians
  cost volume          all-gather of the 48-channel 1/4-resolution feature maps, then rank r sweeps its contiguous
                       block of current views (shard_range)
                                                                            freesplat_amd.cost_volume.sharded_cost_volume

xGMI is point-to-point (7 links/GPU): one all-gather of a whole step's images per rank (tens of MB)
keeps every link busy with a few large messages rather than many per-view ones, and it runs on a
side stream so that the next step's rasterization overlaps it.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist
from torch import Tensor


def _is_gloo(group=None) -> bool:
    return dist.get_backend(group) == "gloo"


def _stage(t: Tensor, group=None) -> Tensor:
    """gloo (the CPU test backend; also what lets the N>1 control flow run with several ranks sharing ONE GPU, which
    RCCL refuses) moves host memory: device tensors are staged through the host for it.  RCCL takes them as they are."""
    return t.cpu() if (t.is_cuda and _is_gloo(group)) else t


def _reduce_scatter_sum(flat: Tensor, chunk: int, group=None) -> Tensor:
    """[world*chunk] -> this rank's [chunk] of the element-wise sum over ranks.  RCCL: one reduce_scatter.  gloo has
    no reduce-scatter: all-reduce and slice (same result; test backend only)."""
    if _is_gloo(group):
        buf = _stage(flat, group).clone()
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        r = dist.get_rank(group)
        return buf[r * chunk: (r + 1) * chunk].to(flat.device)
    mine = torch.empty(chunk, dtype=flat.dtype, device=flat.device)
    dist.reduce_scatter_tensor(mine, flat, op=dist.ReduceOp.SUM, group=group)
    return mine


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous block partition; the first n_items % world ranks get one extra item."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


def shard_counts(n_items: int, world: int) -> list[int]:
    return [len(shard_range(n_items, r, world)) for r in range(world)]


def gather_views(local: Tensor, n_total: int, group=None) -> Tensor:
    """All-gather per-view tensors: `local` is [n_local, ...] for this rank's shard_range of
    n_total views; returns [n_total, ...] in global view order on every rank.  Ragged shards are
    padded to the largest shard for the collective and trimmed afterwards."""
    world = dist.get_world_size(group)
    counts = shard_counts(n_total, world)
    mx = max(counts)
    if local.shape[0] != counts[dist.get_rank(group)]:
        raise ValueError(f"rank holds {local.shape[0]} views, expected {counts[dist.get_rank(group)]}")
    dev = local.device
    tail = tuple(local.shape[1:])
    if local.shape[0] < mx:   # ragged: this rank sends one padded block (the pad rows are trimmed below, never read)
        send = torch.empty((mx,) + tail, dtype=local.dtype, device=dev)
        send[: local.shape[0]].copy_(local)
        send[local.shape[0]:].zero_()
        local = send
    local = _stage(local.contiguous(), group)
    out = torch.empty((world * mx,) + tail, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local, group=group)
    out = out.to(dev)
    if all(c == mx for c in counts):
        return out
    # the first n_total % world ranks hold mx rows, the others mx - 1: two block copies
    big = n_total % world
    res = torch.empty((n_total,) + tail, dtype=out.dtype, device=dev)
    res[: big * mx].copy_(out[: big * mx])
    res[big * mx:].view((world - big, mx - 1) + tail).copy_(out[big * mx:].view((world - big, mx) + tail)[:, : mx - 1])
    return res


class AsyncViewGather:
    """Overlaps the all-gather of step k's images with the rasterization of step k+1: the collective
    is enqueued on a side stream behind an event recorded on the render stream."""

    def __init__(self, n_total: int, group=None, device: Optional[torch.device] = None):
        self.n_total = n_total
        self.group = group
        self.cuda = device is not None and device.type == "cuda"
        self.stream = torch.cuda.Stream(device=device) if self.cuda else None
        self.pending: Optional[Tensor] = None
        # diagnostics (bench.py --gpus N): with `timing` set, every launch / wait is bracketed by events -- (start, end) of
        # the collective on the side stream, (before, after) of the render stream's wait for it (= the EXPOSED part)
        self.timing = False
        self.gather_events: list = []
        self.wait_events: list = []

    def launch(self, local: Tensor) -> None:
        if self.cuda:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ready)
                local.record_stream(self.stream)
                if self.timing:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(self.stream)
                self.pending = gather_views(local, self.n_total, self.group)
                if self.timing:
                    e1.record(self.stream)
                    self.gather_events.append((e0, e1))
        else:
            self.pending = gather_views(local, self.n_total, self.group)

    def wait(self) -> Optional[Tensor]:
        out, self.pending = self.pending, None
        if out is not None and self.cuda:
            cur = torch.cuda.current_stream()
            if self.timing:
                w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                w0.record(cur)
            cur.wait_stream(self.stream)
            if self.timing:
                w1.record(cur)
                self.wait_events.append((w0, w1))
            out.record_stream(cur)   # allocated on the side stream, consumed on this one
        return out


def allreduce_gaussian_grads(grads: list[Optional[Tensor]], group=None) -> None:
    """Sum the view-sharded gradients of the shared Gaussian set across ranks in ONE flat bucket
    (N*(3+6+27+1) floats = 148 MB at 1 M Gaussians) instead of one collective per tensor."""
    ts = [g for g in grads if g is not None]
    if not ts:
        return
    flat = _stage(torch.cat([t.reshape(-1) for t in ts]), group)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat = flat.to(ts[0].device)
    off = 0
    for t in ts:
        n = t.numel()
        t.copy_(flat[off: off + n].view_as(t))
        off += n


_buckets: dict = {}


def _bucket(numel: int, dtype, device) -> Tensor:
    """The flat exchange bucket, allocated (zeroed) once per (device, stream, dtype) and grown on demand: a training step
    reuses it instead of allocating world*chunk floats every call.  Pad elements (ragged shards) may hold stale values of
    an earlier call: they are summed into positions no receiver reads.  Reuse is safe because the collective is ordered
    on the current stream before the next call's packing copies (exchanges issued from different streams get different
    buckets)."""
    key = (str(device), torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0, dtype)
    b = _buckets.get(key)
    if b is None or b.numel() < numel:
        b = _buckets[key] = torch.zeros(numel, dtype=dtype, device=device)
    return b[:numel]


def reduce_scatter_gaussian_grads(grads: list[Optional[Tensor]], group=None) -> list[Optional[Tensor]]:
    """Sum the view-sharded gradients of the shared Gaussian set across ranks and leave every rank with the total
    for the Gaussian rows it OWNS (shard_range over dim 0 of each tensor): one reduce-scatter of one flat bucket
    laid out rank-major ([rows of rank 0 of every tensor | rows of rank 1 ...], padded to equal chunks).
    Returns the row shards [rows_r, ...] in the order of `grads` (None stays None).  Concatenating the shards of
    all ranks reproduces allreduce_gaussian_grads.  The returned shards are views of a fresh [chunk] tensor.

    Packing: a tensor whose rows divide evenly over the ranks enters the bucket with ONE strided copy
    ([world, rows/world * k] -> column block of the [world, chunk] bucket); ragged tensors take one copy per rank."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    ts = [g for g in grads if g is not None]
    if not ts:
        return list(grads)
    rows = [[shard_range(t.shape[0], r, world) for t in ts] for r in range(world)]
    per = [[len(rr) * (t[0].numel() if t.shape[0] else 0) for rr, t in zip(rows[r], ts)] for r in range(world)]
    chunk = max(sum(p) for p in per)
    flat = _bucket(world * chunk, ts[0].dtype, ts[0].device)
    flat2 = flat.view(world, chunk)
    offs = [[sum(per[r][:j]) for j in range(len(ts))] for r in range(world)]
    for j, t in enumerate(ts):
        if t.shape[0] == 0:
            continue
        n0 = per[0][j]
        if t.shape[0] % world == 0 and all(offs[r][j] == offs[0][j] for r in range(world)):
            flat2[:, offs[0][j]: offs[0][j] + n0].copy_(t.reshape(world, n0))
        else:
            for r in range(world):
                rr, n = rows[r][j], per[r][j]
                if n:
                    flat2[r, offs[r][j]: offs[r][j] + n].copy_(t[rr.start: rr.stop].reshape(-1))
    mine = _reduce_scatter_sum(flat, chunk, group)
    out, off, it = [], 0, iter(zip(rows[rank], ts, per[rank]))
    for g in grads:
        if g is None:
            out.append(None)
            continue
        rr, t, n = next(it)
        out.append(mine[off: off + n].view((len(rr),) + tuple(t.shape[1:])))
        off += n
    return out


def chunk_row_ranges(n_rows: int, n_chunks: int) -> list[tuple[int, int]]:
    """Contiguous row chunks [c0, c1) of a chunked gradient exchange: boundaries at multiples of 256 rows (the per-Gaussian
    backward pass works in 256-row workgroups), the last chunk takes the remainder; empty chunks are dropped."""
    n_chunks = max(1, int(n_chunks))
    per = -(-n_rows // n_chunks)
    per = -(-per // 256) * 256
    return [(c0, min(n_rows, c0 + per)) for c0 in range(0, n_rows, per)] if n_rows > 0 else []


def chunked_owned_rows(n_rows: int, rank: int, world: int, n_chunks: int) -> list[range]:
    """The rows rank `rank` owns after a CHUNKED reduce-scatter: its shard_range of every chunk (interleaved over the set,
    not one contiguous block: a sharded optimizer must use this map with GradExchange("chunked"))."""
    out = []
    for c0, c1 in chunk_row_ranges(n_rows, n_chunks):
        r = shard_range(c1 - c0, rank, world)
        out.append(range(c0 + r.start, c0 + r.stop))
    return out


class GradExchange:
    """The gradient exchange of a view-sharded training step, by name: "all_reduce" sums in place (every rank ends
    with every row), "reduce_scatter" returns the row shards this rank owns, "chunked" is the reduce-scatter issued CHUNK BY CHUNK
    of the rows from inside the rasterizer's backward (install() hooks decoder._RenderViews.backward): the per-Gaussian pass
    over chunk c + 1 (fs_raster_backward_views_rows) runs while chunk c is summed over the ranks on a side stream.  Calling
    the exchange after backward() then only waits for the side stream and returns, per tensor, the concatenation of this
    rank's pieces (rows chunked_owned_rows(N, rank, world, chunks)).  Built for the first 8-GPU run to A/B against
    "reduce_scatter" (VERDICT r5 item 7); on one GPU it can only be checked for correctness."""

    def __init__(self, kind: str = "reduce_scatter", group=None, chunks: int = 4):
        if kind not in ("reduce_scatter", "all_reduce", "chunked"):
            raise ValueError(kind)
        self.kind, self.group, self.chunks = kind, group, int(chunks)
        self._pieces, self._stream, self._n = None, None, 0

    # ---- chunked: the hook interface of decoder._RenderViews.backward ----
    def install(self) -> "GradExchange":
        if self.kind == "chunked":
            from . import decoder
            decoder.GRAD_EXCHANGE_HOOK = self
        return self

    def uninstall(self) -> None:
        from . import decoder
        if decoder.GRAD_EXCHANGE_HOOK is self:
            decoder.GRAD_EXCHANGE_HOOK = None

    def begin(self, n_rows: int) -> None:
        self._pieces, self._n = [], n_rows

    def chunk_rows(self, n_rows: int) -> list[tuple[int, int]]:
        return chunk_row_ranges(n_rows, self.chunks)

    def chunk_ready(self, c0: int, c1: int, tensors: list[Tensor]) -> None:
        """Rows [c0, c1) of the gradient tensors are final on the current stream: sum them over the ranks on the side stream."""
        dev = tensors[0].device
        if dev.type != "cuda":
            self._pieces.append(reduce_scatter_gaussian_grads([t[c0:c1] for t in tensors], self.group))
            return
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=dev)
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self._stream):
            self._stream.wait_event(ev)
            for t in tensors:
                t.record_stream(self._stream)
            self._pieces.append(reduce_scatter_gaussian_grads([t[c0:c1] for t in tensors], self.group))

    def __call__(self, grads: list[Optional[Tensor]]):
        if self.kind == "all_reduce":
            allreduce_gaussian_grads(grads, self.group)
            return grads
        if self.kind == "reduce_scatter":
            return reduce_scatter_gaussian_grads(grads, self.group)
        if self._pieces is None:
            raise RuntimeError('GradExchange("chunked"): no backward ran through the hook since the last call (install() it, and '
                               "render through decoder.render_views)")
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)
        pieces, self._pieces = self._pieces, None
        if not pieces:
            return [None if g is None else g[:0] for g in grads]
        out = [torch.cat([pc[j] for pc in pieces]) for j in range(len(pieces[0]))]
        it = iter(out)
        return [None if g is None else next(it) for g in grads]


class _GatherViewsFn(torch.autograd.Function):
    """All-gather of per-view tensors that autograd can cross: every rank goes on to compute the SAME loss on the
    gathered views, so the gradient of the local shard is simply its slice of the incoming gradient (no collective
    in backward; the sum over ranks happens once, on the Gaussian gradients -- replicate_gaussians)."""

    @staticmethod
    def forward(ctx, local, n_total, group):
        ctx.group, ctx.n_total = group, n_total
        return gather_views(local, n_total, group)

    @staticmethod
    def backward(ctx, g):
        r = shard_range(ctx.n_total, dist.get_rank(ctx.group), dist.get_world_size(ctx.group))
        return g[r.start: r.stop].contiguous(), None, None


def gather_views_autograd(local: Tensor, n_total: int, group=None) -> Tensor:
    return _GatherViewsFn.apply(local, n_total, group)


class _ReplicateFn(torch.autograd.Function):
    """Identity on the (replicated) Gaussian tensors whose backward sums their gradients over the ranks in ONE flat
    bucket: with the views of a scene sharded over the ranks each rank holds the partial gradient of its own views;
    afterwards every rank holds the total, so a replicated encoder continues backward identically everywhere."""

    @staticmethod
    def forward(ctx, group, *tensors):
        ctx.group = group
        return tuple(t.view_as(t) for t in tensors)

    @staticmethod
    def backward(ctx, *grads):
        gs = [None if g is None else g.contiguous() for g in grads]
        allreduce_gaussian_grads(gs, ctx.group)
        return (None,) + tuple(gs)


def replicate_gaussians(tensors: list[Tensor], group=None) -> list[Tensor]:
    return list(_ReplicateFn.apply(group, *tensors))


class _GatherFeatsFn(torch.autograd.Function):
    """All-gather of per-view feature maps whose consumers differ per rank (cost-volume sharding: rank r uses the
    gathered maps as SOURCES of its own current views): the gradient of a local map is the SUM over the ranks of
    their gradients for it -- backward is a reduce-scatter."""

    @staticmethod
    def forward(ctx, local, n_total, group):
        ctx.group, ctx.n_total, ctx.tail = group, n_total, tuple(local.shape[1:])
        return gather_views(local, n_total, group)

    @staticmethod
    def backward(ctx, g):
        world, rank = dist.get_world_size(ctx.group), dist.get_rank(ctx.group)
        counts = shard_counts(ctx.n_total, world)
        mx = max(counts)
        per = 1
        for d in ctx.tail:
            per *= d
        g = g.contiguous()
        flat = _bucket(world * mx * per, g.dtype, g.device)
        if all(c == mx for c in counts):
            flat.copy_(g.reshape(-1))
        else:
            lo = 0
            for r, c in enumerate(counts):
                flat[r * mx * per: (r * mx + c) * per].copy_(g[lo: lo + c].reshape(-1))
                lo += c
        mine = _reduce_scatter_sum(flat, mx * per, ctx.group)
        return mine[: counts[rank] * per].view((counts[rank],) + ctx.tail), None, None


def gather_features_autograd(local: Tensor, n_total: int, group=None) -> Tensor:
    return _GatherFeatsFn.apply(local, n_total, group)

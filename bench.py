#!/usr/bin/env python
"""bench.py -- rendered views/sec of the MI355X rasterizer on BASELINE.json's metric config, plus (N=1) the
other rows of the hot path under the same clock.

  python bench.py [--gpus N --steps K --warmup W]            (N=1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W    (N>1, one rank per GPU)

A "step" renders `--views` target views (default 16) of the synthetic 1.0 M-Gaussian scene at
968x1296 on every rank (weak scaling, view-sharded: rank r renders its own block of the N*views
target cameras) and, for N>1, all-gathers the rendered colour images (`--gather-depth`: + depth; `--gather-dtype
fp32 | fp16 | uint8`) over RCCL on a side stream, overlapped with the next step.  Inputs are resident in HBM before the
timed region.  The timed region is EXACTLY `--steps` steps between two barriers (max over ranks); at the default sizes it
lasts ~80 ms, so it is repeated until `--min-time` (0.6 s) has been timed and the MEDIAN region is reported
(`timed_regions_ms` lists them all).  `--single-rank-collectives` (under `torch.distributed.run --nproc-per-node 1`) takes
the N>1 code path on RCCL with one rank.

Rank 0 prints ONE JSON line.  Top level = the headline metric (forward rendering, config 3's size) with
`roofline` (render kernel, HIP-event timed inside the timed region through the library's fs_profile_* hooks),
`cpu_baseline` (the CPU oracle, OpenMP, on whole views of the same workload) and `parity`.  At N=1 the same line
carries sub-objects measured in the same run, each with its own roofline / cpu_baseline / parity:
  "train"        fwd+bwd training step at config 3 (roofline: render_bwd kernel)
  "c2"           config 2 (640x480, 300 k Gaussians, forward)
  "cost_volume"  plane-sweep cost volume: native 96x128 K=1 and config-3 scale 242x324 K=2 (roofline: fp32 MFMA)
  "c3_fp16_sh"   the headline workload with the SH coefficients stored in fp16 (BASELINE config 5's storage option)
  "ptf"          Pixel-wise Triplet Fusion folds: 2, 10 and 30 views at 384x512, 3 views at 968x1296 (roofline: HBM)
  "encoder_tail" the depth-regression tail of the DepthDecoder and the latent -> Gaussian head at the native size (roofline: HBM)
(`--sections raster` restricts the run to the top-level metric.)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--views", type=int, default=16, help="target views per step per GPU (one decoder call)")
    ap.add_argument("--workload", default="c3_968x1296_1M",
                    help="c3_968x1296_1M (metric config) | c2_640x480_300k | c1_256x256_plumbing")
    ap.add_argument("--mode", default="fwd", choices=["fwd", "train"],
                    help="top-level measurement. fwd: forward rendering (the metric); train: fwd + bwd (+ grad exchange)")
    ap.add_argument("--sections", default="all",
                    help="N=1 only: comma list of extra sub-objects (train,c2,closeup,c5,cost_volume,ptf,encoder_tail,c3_step), 'all', or 'raster' for none")
    ap.add_argument("--grad-exchange", default="reduce_scatter", choices=["reduce_scatter", "all_reduce", "chunked"],
                    help="N>1 train mode: how the per-Gaussian gradients of the view shards are summed.  chunked = the reduce-scatter "
                         "issued chunk by chunk of the rows from inside the backward (--grad-chunks), overlapping the per-Gaussian pass")
    ap.add_argument("--grad-chunks", type=int, default=4, help="row chunks of --grad-exchange chunked")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the image all-gather")
    ap.add_argument("--single-rank-collectives", action="store_true",
                    help="N=1 under torch.distributed.run --nproc-per-node 1: initialise RCCL with one rank and take the "
                         "N>1 code path (image all-gather on the side stream / gradient exchange, barrier, max-over-ranks "
                         "timing) -- the RCCL branch on a one-GPU box")
    ap.add_argument("--gather-depth", action="store_true",
                    help="N>1: all-gather the depth maps too (the reference's decoder returns depth only when depth_mode is "
                         "set, decoder_splatting_cuda.py:64-70; colour alone is 15 MB per view, with depth 20 MB)")
    ap.add_argument("--min-time", type=float, default=0.6,
                    help="repeat the timed region of `--steps` steps until this many seconds have been timed; the median "
                         "region is reported (0: one region)")
    ap.add_argument("--gather-dtype", default="fp32", choices=["fp32", "fp16", "uint8"],
                    help="N>1: element type of the gathered images.  fp32 (default) = what the decoder returns; fp16 / uint8 "
                         "(clamped to [0,1], x255, rounded) halve / quarter the bytes every GPU receives over its 7 xGMI links "
                         "(241 MB per 16-view step and rank in fp32) for evaluation runs that end in 8-bit images anyway")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--no-graph", action="store_true", help="skip the hipGraph capture / replay measurement")
    return ap.parse_args()


class Ctx:
    """Process-wide state of one bench run (device, ranks)."""

    def __init__(self, args):
        self.args = args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            args.gpus = self.world     # (the launcher's environment wins; `--gpus N` alone re-launches itself, main())
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device (no CPU fallback in the product path)")
        # FS_DIST_BACKEND=gloo + FS_SHARE_GPU=1 exercise the N>1 control flow on a single-GPU box (tests only)
        backend = os.environ.get("FS_DIST_BACKEND", "nccl")
        if os.environ.get("FS_SHARE_GPU") == "1":
            local_rank = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
        self.dev = torch.device("cuda", local_rank)
        self.dist_on = self.world > 1 or args.single_rank_collectives
        if self.dist_on:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            if backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.dev)
            else:
                dist.init_process_group(backend, rank=self.rank, world_size=self.world)

    def barrier(self):
        torch.cuda.synchronize()
        if self.dist_on:
            dist.barrier()
        torch.cuda.synchronize()


def bench_raster(cx: Ctx, workload: str, mode: str, views: int, steps: int, warmup: int, cpu_baseline: bool,
                 sh_fp16: bool = False) -> dict | None:
    """One rasterizer measurement (forward, or forward+backward): returns the JSON object on rank 0.
    `sh_fp16` (forward only): the SH coefficients are STORED in fp16 (BASELINE config 5; FS_RASTER_SH_FP16: converted on
    load, fp32 arithmetic) -- algorithmic bytes N*94 + P*16, parity against the oracle fed the fp16-rounded coefficients."""
    args = cx.args
    world, rank, dev = cx.world, cx.rank, cx.dev
    from freesplat_amd import _lib, synthetic
    from freesplat_amd.decoder import check_deferred, render_views
    from freesplat_amd.view_sharding import AsyncViewGather, GradExchange, shard_range

    H, W, N = synthetic.WORKLOADS[workload]
    from freesplat_amd.rasterizer import _state as _rstate
    _rstate(dev).last_instances = 0      # (capacity history of a previous workload in this process)
    _rstate(dev).retry_cap = 0
    scene = synthetic.workload_scene(workload)
    n_total_views = views * world
    cams_all = synthetic.target_cameras(n_total_views)
    mine = shard_range(n_total_views, rank, world)
    sl = slice(mine.start, mine.stop)
    cams = {k: v[sl].to(dev) for k, v in cams_all.items()}
    g = {k: scene[k].to(dev) for k in ("means", "covariances", "harmonics", "opacities")}
    if sh_fp16:
        g["harmonics"] = g["harmonics"].half()
        scene = dict(scene, harmonics=scene["harmonics"].half().float())     # (what the oracle is fed)
    bg = torch.zeros(len(mine), 3, device=dev)
    train = mode == "train"
    if train:
        for t in g.values():
            t.requires_grad_(True)
        target = torch.rand(len(mine), 3, H, W, device=dev)
    gather = AsyncViewGather(n_total_views, device=dev) if (cx.dist_on and not args.no_gather and not train) else None
    exchange = GradExchange(args.grad_exchange, chunks=args.grad_chunks).install() if (cx.dist_on and train) else None
    diag_events = None      # (set for the few extra untimed steps that time the gradient exchange per rank)

    def step():
        nonlocal diag_events
        if train:
            for t in g.values():
                t.grad = None
            color, depth = render_views(cams["extrinsics"], cams["intrinsics"], cams["near"], cams["far"],
                                        (H, W), bg, g["means"], g["covariances"], g["harmonics"], g["opacities"])
            # (the fused library loss: `((color - target) ** 2).mean()` ran ~10 elementwise torch kernels over the step's images,
            #  22 % of the step's GPU time in profiles/r6_train_kernel_stats.csv -- glue, not rasterizer)
            loss = torch.nn.functional.mse_loss(color, target)
            loss.backward()
            if exchange is not None:
                if diag_events is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                exchange([t.grad for t in g.values()])
                if diag_events is not None:
                    e1.record()
                    diag_events.append((e0, e1))
            return color, depth
        with torch.no_grad():
            # capacity check deferred to the end of the timed region (check_deferred below): the GPU
            # queue stays full across steps; all launched work still completes inside the region
            color, depth = render_views(cams["extrinsics"], cams["intrinsics"], cams["near"], cams["far"],
                                        (H, W), bg, g["means"], g["covariances"], g["harmonics"], g["opacities"],
                                        check="deferred")
            if gather is not None:
                gather.wait()  # previous step's gather must be done before its buffers are dropped
                payload = torch.cat([color, depth], dim=1) if args.gather_depth else color
                if args.gather_dtype == "fp16":
                    payload = payload.half()
                elif args.gather_dtype == "uint8":
                    payload = (payload.clamp(0, 1) * 255.0 + 0.5).to(torch.uint8)
                gather.launch(payload)
        return color, depth

    capacity_retries = 0
    for attempt in range(2):
        for _ in range(max(warmup, 1) if attempt else warmup):
            color, depth = step()
        if gather is not None:
            gather.wait()
        try:
            check_deferred()
            break
        except _lib.FreeSplatHipError:
            # first contact with a workload denser than the default instance capacity: the capacity history is
            # updated by the check, warm up again with it (the timed region below still fails loudly on overflow)
            if attempt:
                raise
            capacity_retries += 1
    cx.barrier()
    profile = not args.no_profile
    dominant = "render_bwd" if train else "render"
    if profile:
        # inside the timed region only the dominant kernel (the roofline kernel) is bracketed by HIP events:
        # every timed launch puts two event records on the stream, and timing all five stages of every view
        # costs ~8 % of the throughput being measured
        _lib.profile_collect()
        _lib.profile_enable(True, stages=[dominant])
    # The timed region = EXACTLY `steps` steps between two barriers.  One region of the default run lasts ~0.1 s, and
    # box-to-box / run-to-run noise is of the size of the kernel deltas being measured: the region is repeated (every
    # repeat bracketed the same way) until >= ~0.6 s have been timed, and the MEDIAN region is reported (all of them
    # are listed in `timed_regions_ms`).
    regions = []
    while True:
        t0 = time.perf_counter()
        failure = None
        try:
            for _ in range(steps):
                color, depth = step()
            if gather is not None:
                gather.wait()
            check_deferred()  # raises if any view of the timed region overflowed its instance capacity
        except Exception as e:  # noqa: BLE001 -- re-raised below, on EVERY rank
            failure = e
        # a failure on one rank (capacity overflow, out of memory) must not leave the others in the barrier / all-reduce
        # below: every rank learns of it and all of them stop
        flag = torch.tensor([1.0 if failure is not None else 0.0], device=dev)
        if cx.dist_on:
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if flag.item() > 0:
            raise failure if failure is not None else RuntimeError("another rank failed inside the timed region")
        cx.barrier()
        regions.append(time.perf_counter() - t0)
        enough = torch.tensor([1.0 if (sum(regions) >= args.min_time or len(regions) >= 15) else 0.0], device=dev)
        if cx.dist_on:
            dist.all_reduce(enough, op=dist.ReduceOp.MIN)   # every rank runs the same number of regions
        if enough.item() > 0:
            break
    dt = sorted(regions)[len(regions) // 2]
    stages, breakdown = {}, {}
    if profile:
        _lib.profile_enable(False)
        stages = _lib.profile_collect()
        # per-stage breakdown from two extra, untimed steps with every stage bracketed
        from freesplat_amd import rasterizer as _R
        streams, _R.NUM_STREAMS = _R.NUM_STREAMS, 1      # one stream: launches back to back, durations not shared
        _lib.profile_enable(True)
        for _ in range(2):
            step()
        if gather is not None:
            gather.wait()
        check_deferred()
        torch.cuda.synchronize()
        _lib.profile_enable(False)
        _R.NUM_STREAMS = streams
        breakdown = _lib.profile_collect()
    # N > 1 diagnostics: a few extra untimed steps with the render, the all-gather and the render
    # stream's wait for the gather (the EXPOSED, non-overlapped part) event-timed separately on every rank
    multi = None
    if gather is not None:
        gather.timing = True
        spans = []
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            step()
            e1.record()
            spans.append((e0, e1))
        gather.wait()
        check_deferred()
        torch.cuda.synchronize()
        gather.timing = False
        mean = lambda evs: sum(a.elapsed_time(b) for a, b in evs) / max(len(evs), 1)
        mine_ms = [mean(spans), mean(gather.gather_events), mean(gather.wait_events)]
        allv = torch.tensor(mine_ms, device=dev, dtype=torch.float64)
        got = [torch.zeros_like(allv) for _ in range(world)]
        if cx.dist_on:
            dist.all_gather(got, allv)
        else:
            got = [allv]
        multi = {"per_rank": [{"step_ms": round(float(t[0]), 3), "gather_ms": round(float(t[1]), 3),
                               "exposed_gather_ms": round(float(t[2]), 3)} for t in got],
                 "what": "4 extra untimed steps, event-timed per rank: one step on the render stream (render of this rank's views "
                         "+ its wait for the PREVIOUS step's gather), the all-gather of one step's images on the side stream, "
                         "and that wait alone = the part of the gather the rendering did not hide",
                 "gather_bytes_per_rank_per_step": int(n_total_views * H * W * (4 if args.gather_depth else 3)
                                                       * {"fp32": 4, "fp16": 2, "uint8": 1}[args.gather_dtype])}
    if exchange is not None:
        # training mode: the gradient exchange runs on the render stream after the backward -- all of it is exposed
        diag_events, spans = [], []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            step()
            e1.record()
            spans.append((e0, e1))
        torch.cuda.synchronize()
        mean = lambda evs: sum(a.elapsed_time(b) for a, b in evs) / max(len(evs), 1)
        allv = torch.tensor([mean(spans), mean(diag_events)], device=dev, dtype=torch.float64)
        diag_events = None
        got = [torch.zeros_like(allv) for _ in range(world)]
        dist.all_gather(got, allv)
        multi = {"per_rank": [{"step_ms": round(float(t[0]), 3), "grad_exchange_ms": round(float(t[1]), 3)} for t in got],
                 "what": "3 extra untimed training steps, event-timed per rank: one whole step (forward + loss + backward + exchange) "
                         f"and the {args.grad_exchange} of the Gaussian gradients alone (on the render stream: none of it is hidden)",
                 "exchange_bytes_per_rank_per_step": int(N * 37 * 4)}
    graph_views_per_s = None
    if world == 1 and not cx.dist_on and not train and not args.no_graph:
        # the same step recorded once into a hipGraph (the C ABI never allocates or syncs: framing + 5 kernels per view
        # over two forked streams capture as they are) and replayed: what is left when the host-side launch train is
        # taken out of the loop
        from freesplat_amd import decoder as _D
        with torch.no_grad():
            cap = torch.cuda.Stream(device=dev)
            graph = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.stream(cap):
                with torch.cuda.graph(graph, stream=cap):
                    gc_color, gc_depth = step()
            _D._pending_checks.clear()
            graph.replay()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(steps):
                graph.replay()
            torch.cuda.synchronize()
            graph_views_per_s = n_total_views * steps / (time.perf_counter() - t1)
            graph_ok = bool(torch.equal(gc_color, color))
            del graph
    if cx.dist_on:
        # max over ranks, region by region; then the median region
        tmax = torch.tensor(regions, device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        regions = tmax.tolist()
        dt = sorted(regions)[len(regions) // 2]
    if rank != 0:
        return None

    from freesplat_amd.rasterizer import NUM_STREAMS as R_NUM_STREAMS, _state
    n_inst = _state(dev).last_instances
    n_views_done = n_total_views * steps      # per timed region
    # algorithmic bytes per rendered view: N*(12+24+4+12*d_sh) + P*(12+4)
    alg_fwd = N * (94 if sh_fp16 else 148) + H * W * 16
    alg_bwd = 2 * N * 148 + H * W * 20
    out = {
        "metric": f"rendered views/sec @ {H}x{W}, {N / 1e6:.1f}M Gaussians" + (" (fwd+bwd)" if train else ""),
        "value": n_views_done / dt, "unit": "views/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": 1e3 * dt / steps, "timed_regions_ms": [round(1e3 * r, 3) for r in regions],
        "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload, "mode": mode, "image_hw": [H, W], "gaussians": N,
                   "sh_degree": 2, "sh_storage": "fp16" if sh_fp16 else "fp32", "views_per_step_per_gpu": views,
                   "raster_streams": R_NUM_STREAMS,
                   "projection": "legacy per-view kernel (FREESPLAT_PREPROCESS=legacy)" if os.environ.get("FREESPLAT_PREPROCESS") == "legacy" else
                                 f"all views of a call in one launch, {os.environ.get('FREESPLAT_RASTER_BATCH', '16')} views per batch",
                   "blend_exp": "hardware v_exp_f32" if _R_FAST() else "contract polynomial",
                   "blend": ("training instantiation (tracks n_contrib, writes the sorted lists for the backward)" if train else
                             "inference instantiation (torch.no_grad: no n_contrib tracking, sorted lists stay in LDS; same image bits)"),
                   "instances_per_view": int(n_inst), "instances_per_gaussian": round(n_inst / max(N, 1), 2),
                   "parallelism": f"view-sharded x{world}" + (f" + {args.grad_exchange}(gaussian grads)" if exchange else
                                                              (" + all_gather(color,depth)" if args.gather_depth else " + all_gather(color)")
                                                              + ("" if args.gather_dtype == "fp32" else f"[{args.gather_dtype}]") if gather else "")},
    }
    # what the fixed per-tile key areas cost in memory at this workload's instance capacity (fs_raster_buffer_sizes), and how
    # often the warm-up had to be repeated with a capacity taken from the overflow counters
    from freesplat_amd import rasterizer as _Rz
    cap_now = _Rz.default_capacity(N, _state(dev), H, W)
    bsz = _Rz._buffer_sizes(N, H, W, cap_now)
    slots = int(_lib.lib().fs_raster_scratch_slots(len(mine), R_NUM_STREAMS if R_NUM_STREAMS > 1 else 0))
    out["raster_buffers"] = {"instance_capacity": int(cap_now), "geom_bytes_per_view": bsz[0], "binning_bytes_per_view": bsz[1],
                             "image_bytes_per_view": bsz[2], "scratch_bytes_per_stream": bsz[3],
                             "scratch_slots_per_call": slots, "scratch_bytes_per_call": slots * bsz[3],
                             "scratch_note": "one key area (scratch_bytes_per_stream: the field's name since round 3) per view in flight: "
                                             "the projection + binning of all views of a call is ONE launch (round 6), at most 16 slots",
                             "capacity_retries": capacity_retries,
                             "note": "capacity_retries = warm-up passes repeated after an instance-capacity overflow (first contact "
                                     "with a workload denser than 8 entries per Gaussian); the timed region never retries"}
    if multi is not None:
        out["multi_gpu"] = multi
    if graph_views_per_s is not None:
        out["hipgraph_replay"] = {"value": graph_views_per_s, "unit": "views/s", "same_image_as_eager": graph_ok,
                                  "what": "one step captured into a hipGraph, replayed `steps` times"}
    if stages:
        ms, cnt = stages.get(dominant, (0.0, 0))
        per = ms / max(cnt, 1) * 1e-3
        alg = alg_bwd if train else alg_fwd
        ach = alg / per / 1e9 if per > 0 else 0.0
        kname = "render_bwd_kernel" if train else "sort_blend_kernel"
        if train:
            traffic, traffic_src = (committed_traffic("fs::render_bwd_kernel<" + ("true" if _R_FAST() else "false"))
                                    if workload.startswith("c3") else (None, None))
        elif sh_fp16:
            traffic, traffic_src = None, None
        elif "closeup" in workload:
            traffic, traffic_src = traffic_lookup("raster_closeup", "fs::sort_blend_kernel")
        else:   # per launch of the fused sort + blend kernel (profiles/tools/fwd_traffic.py)
            traffic, traffic_src = traffic_lookup("raster_" + workload[:2], "fs::sort_blend_kernel")
        out["roofline"] = {"bound": "hbm", "kernel": kname, "achieved": ach, "peak": 8000.0,
                           "unit": "GB/s", "frac": ach / 8000.0, "traffic": traffic,
                           "traffic_source": traffic_src,
                           "algorithmic_bytes_per_launch": alg, "avg_launch_ms": per * 1e3,
                           "launches": cnt}
        out["kernel_ms_per_view"] = {k: v[0] / max(v[1], 1) for k, v in breakdown.items() if v[1]}
        out["kernel_ms_per_view_note"] = ("2 extra untimed single-stream steps with every stage event-timed (isolated "
                                          "durations); roofline.avg_launch_ms is from the timed region, where the "
                                          "launches of adjacent views overlap on config.raster_streams streams")
        iso = breakdown.get(dominant, (0.0, 0))
        out["roofline"]["isolated_launch_ms"] = iso[0] / max(iso[1], 1)
        # the three numbers one could mean by "fraction of the HBM roofline", named:
        #   frac_overlapped    = `frac`: algorithmic bytes / AVERAGE launch duration inside the timed region, where the launches
        #                        of adjacent views run concurrently on the raster streams (each is slowed by its neighbour)
        #   frac_isolated      = the same bytes / the duration of a launch running alone (the extra single-stream steps)
        #   pipeline_frac_wall = algorithmic bytes of all views of a timed region / its wall time: the whole pipeline
        #                        (projection + binning + scan + sort + blend, launch gaps included) by the driver's clock
        out["roofline"]["frac_overlapped"] = out["roofline"]["frac"]
        if iso[1] and iso[0] > 0:
            out["roofline"]["frac_isolated"] = alg / (iso[0] / iso[1] * 1e-3) / 8e12
        out["roofline"]["pipeline_frac_wall"] = alg * n_views_done / dt / 8e12
        if not train and not sh_fp16:
            out["roofline"]["pipeline_traffic_per_view"] = traffic_lookup("raster_closeup" if "closeup" in workload else "raster_" + workload[:2])[0]
        # VALU-issue roofline of the blend kernel: the HBM figure above cannot move for a kernel whose limit is
        # instruction issue.  Issue cycles per launch = hardware instruction-class counters x microbenchmarked cycles per class
        # (profiles/tools/valu_roofline.py -> the newest committed profiles/*_valu_roofline.json; PMC counters cannot be
        # collected inside this run) against 1024 SIMDs x 2.4 GHz x THIS run's isolated launch time.
        rv = committed_valu_roofline("fs::render_bwd_kernel<" if train else "fs::sort_blend_kernel<")
        if rv is not None and not sh_fp16 and workload.startswith("c3") and "closeup" not in workload and iso[1] and iso[0] > 0:
            t_iso = iso[0] / iso[1] * 1e-3
            peak_cyc = 1024 * 2.4e9 * t_iso
            out["roofline_valu"] = {"bound": "valu_issue", "kernel": rv["kernel"], "achieved": rv["issue_cycles"] / t_iso / 1e12,
                                    "peak": 1024 * 2.4e9 / 1e12, "unit": "T SIMD-cycles/s", "frac": rv["issue_cycles"] / peak_cyc,
                                    "issue_cycles_per_launch": rv["issue_cycles"], "valu_instructions_per_launch": rv["SQ_INSTS_VALU"],
                                    "mean_cycles_per_instruction": rv["mean_cycles_per_valu_instruction"],
                                    "hw_valu_busy_frac": rv.get("hw_valu_busy_frac"),
                                    "hw_dual_issue_frac": rv.get("hw_dual_issue_frac_of_valu_quads"),
                                    "isolated_launch_ms": t_iso * 1e3, "source": rv["source"],
                                    "note": "frac = microbenchmark-priced issue cycles / available SIMD cycles; hw_valu_busy_frac = "
                                            "SQ_ACTIVE_INST_VALU x 4 / (1024 x GRBM_GUI_ACTIVE per XCD): the wave-cycles spent IN "
                                            "VALU instructions, which at 6 wavefronts per SIMD includes their latency"}
        ksum = sum(out["kernel_ms_per_view"].values())
        if ksum > 0:   # the whole pipeline of one view against the same algorithmic bytes
            out["roofline"]["pipeline_frac_isolated"] = (alg_fwd + (alg_bwd if train else 0)) / (ksum * 1e-3) / 8e12
    if world == 1 and cpu_baseline:
        out.update(cpu_baseline_and_parity(scene, cams_all, H, W, workload, train=train, sh_fp16=sh_fp16))
    return out


def _R_FAST() -> bool:
    from freesplat_amd import rasterizer
    return bool(rasterizer.FAST_EXP)


def traffic_lookup(workload: str, kernel_prefix=None):
    sys.path.insert(0, os.path.join(ROOT, "profiles", "tools"))
    try:
        import fwd_traffic
        return fwd_traffic.lookup(workload, kernel_prefix)
    except Exception:
        return None, None


def committed_valu_roofline(kernel_prefix: str):
    """The kernel's entry of the newest committed profiles/*_valu_roofline.json (profiles/tools/valu_roofline.py), or None."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_valu_roofline.json")), reverse=True):
        try:
            ks = json.load(open(f))["kernels"]
        except Exception:
            continue
        # the inference instantiation for the forward metric (<false, false>), the training one otherwise
        for name in sorted(ks):
            if name.startswith(kernel_prefix) and (not kernel_prefix.startswith("fs::sort_blend") or name.endswith("<false, false>")):
                return dict(ks[name], kernel=name, source=os.path.relpath(f, ROOT))
    return None


def committed_traffic(kernel: str):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary under profiles/
    (FETCH_SIZE / WRITE_SIZE passes of this same command, profiles/run_rocprof.sh); PMC counters cannot be
    collected from inside the timed run.  (None, None) if no summary is present."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json")))
    if not files:
        return None, None
    for f in reversed(files):   # newest summary that has this kernel (forward and backward are separate files)
        try:
            ks = json.load(open(f))["kernels"]
            # `kernel` is a prefix up to the first template argument (e.g. "fs::render_kernel<false"): the forward
            # blend has a second one (contributor-count tracking) that differs between inference and training runs
            k = next((v for name, v in ks.items() if name == kernel or name.startswith(kernel + ",") or name.startswith(kernel + ">")), None)
        except Exception:
            continue
        if k:
            return float(k["hbm_bytes_per_launch"]), os.path.relpath(f, ROOT)
    return None, None


def cpu_baseline_and_parity(scene, cams_all, H, W, workload, train=False, sh_fp16=False):
    """Times the CPU oracle (kind "port": OpenMP restatement of the reference algorithm, all host
    cores) on a bounded sample -- whole views of the same workload (forward, or forward + backward in train mode)
    until >= 10 s of CPU work or 3 views -- and checks the GPU result of view 0 against it: image max-abs / PSNR /
    bit-exactness, in train mode every gradient, and (forward) the sensitivity of the image to the private exp()
    contract the oracle and the kernels share (oracle re-run with libm expf)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import raster_oracle as ro
    from util_raster import hip_forward, oracle_forward, view_inputs
    cores = os.cpu_count() or 1
    os.environ["OMP_NUM_THREADS"] = str(cores)
    rng = np.random.default_rng(0)
    g_color = rng.normal(size=(3, H, W)).astype(np.float32)
    vi = view_inputs(scene, cams_all, 0, H, W)
    oracle_forward(vi) if H * W <= 640 * 480 else None  # warm small workloads only
    n, t_tot, st0, ref0 = 0, 0.0, None, None
    while n < 3 and t_tot < 10.0:
        vi_n = view_inputs(scene, cams_all, n % cams_all["extrinsics"].shape[0], H, W)
        t = time.perf_counter()
        st = oracle_forward(vi_n)
        ref = ro.backward(st, g_color) if train else None
        t_tot += time.perf_counter() - t
        if n == 0:
            st0, ref0 = st, ref
        n += 1
    # parity of the kernels: the SAME (CPU-framed) inputs through the product rasterizer
    dev = torch.device("cuda", torch.cuda.current_device())
    vi_hip = dict(vi, shs=vi["shs"].half()) if sh_fp16 else vi      # (`scene` already holds the fp16-rounded values)
    (gc, _, _, _), leaves = hip_forward(vi_hip, dev, requires_grad=train)
    g = gc.detach().cpu().numpy()
    err = float(np.abs(g - st0["color"]).max())
    mse = float(((g.clip(0, 1) - st0["color"].clip(0, 1)) ** 2).mean())
    psnr = None if mse == 0 else float(-10 * np.log10(mse))
    from freesplat_amd import rasterizer as _R
    from freesplat_amd import rasterizer as _R
    dpx = np.abs(g - st0["color"]).max(axis=0)
    parity = {"mode": "FS_RASTER_FAST_EXP (hardware exp, guarded alpha threshold; opt-in)" if _R.FAST_EXP else "contract exp (default)",
              "max_abs_err_vs_oracle": err, "pixels_above_1e-4": int((dpx > 1e-4).sum()), "pixels": H * W,
              "psnr_db_vs_oracle": psnr if psnr is not None else "inf", "bit_exact": bool((g == st0["color"]).all())}
    if train:
        (gc * torch.from_numpy(g_color).to(dev)).sum().backward()
        rel = {}
        for k in ("means3D", "cov3D", "shs", "opacities"):
            a = leaves[k].grad.cpu().numpy().reshape(ref0[k].shape)
            rel[k] = float(np.abs(a - ref0[k]).max() / (np.abs(ref0[k]).max() + 1e-20))
        parity["grad_err_over_max_abs"] = rel
        parity["grad_tolerance"] = 2e-4
    else:
        try:
            ro.set_exp_mode(True)
            st_libm = oracle_forward(vi)
        finally:
            ro.set_exp_mode(False)
        d = np.abs(g - st_libm["color"]).max(axis=0)
        mse2 = float(((g.clip(0, 1) - st_libm["color"].clip(0, 1)) ** 2).mean())
        parity["exp_contract"] = {"what": "HIP image (" + ("guarded hardware exp" if _R.FAST_EXP else "contract exp")
                                          + ") vs the oracle with libm expf in the blend loop",
                                  "max_abs": float(d.max()), "pixels_above_1e-4": int((d > 1e-4).sum()), "pixels": H * W,
                                  "psnr_db": "inf" if mse2 == 0 else float(-10 * np.log10(mse2))}
    return {
        "cpu_baseline": {"value": n / t_tot, "unit": "views/s", "cores": cores, "kind": "port",
                         "sample": f"{n} {'forward+backward' if train else 'forward'} view(s) of {workload} through "
                                   "oracle/raster_oracle.c (OpenMP)"},
        "parity": parity,
    }


HEADLINE_MAX_BYTES = 4096      # the driver parses the LAST stdout line; round 4's single 28 KB line came back `parsed: null`


def _num(x, digits=5):
    """Floats to `digits` significant digits (the compact line carries figures, not 17-digit reprs)."""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    return float(f"{x:.{digits}g}")


def _pick(d, keys):
    return {k: _num(d[k]) for k in keys if isinstance(d, dict) and k in d}


def headline(full: dict) -> dict:
    """The compact driver-format object (<= HEADLINE_MAX_BYTES as JSON): every key of the bench contract, the headline's
    `roofline` / `cpu_baseline` / `parity`, and ONE small digest per section of the full line (which is printed before it and
    written to gpurun_out/bench_full.json).  Digest entries drop first if the object would not fit."""
    h = {k: _num(full[k]) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data") if k in full}
    h["config"] = _pick(full.get("config", {}), ("workload", "mode", "image_hw", "gaussians", "sh_degree", "sh_storage",
                                                  "views_per_step_per_gpu", "raster_streams", "instances_per_gaussian", "parallelism"))
    if "roofline" in full:
        h["roofline"] = _pick(full["roofline"], ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic",
                                                 "algorithmic_bytes_per_launch", "avg_launch_ms", "frac_isolated",
                                                 "pipeline_frac_wall", "pipeline_traffic_per_view"))
    if "roofline_valu" in full:
        h["roofline_valu"] = _pick(full["roofline_valu"], ("bound", "kernel", "achieved", "peak", "unit", "frac", "source"))
    if "cpu_baseline" in full:
        h["cpu_baseline"] = dict(full["cpu_baseline"], value=_num(full["cpu_baseline"]["value"]))
    if "parity" in full:
        h["parity"] = _pick(full["parity"], ("max_abs_err_vs_oracle", "pixels_above_1e-4", "pixels", "bit_exact", "psnr_db_vs_oracle"))
    if "multi_gpu" in full:
        pr = full["multi_gpu"].get("per_rank", [])
        h["multi_gpu"] = {k: _num(max(r[k] for r in pr), 4) for k in (pr[0] if pr else {})}      # max over ranks
    digest = {}

    def rast(name, o):
        if not isinstance(o, dict):
            return
        if "error" in o:
            digest[name] = {"error": str(o["error"])[:80]}
            return
        e = _pick(o, ("value", "unit"))
        if isinstance(o.get("roofline"), dict):
            e["frac"] = _num(o["roofline"].get("frac_isolated", o["roofline"].get("frac")), 3)
        if isinstance(o.get("roofline_valu"), dict):
            e["valu_frac"] = _num(o["roofline_valu"].get("frac"), 3)
        p = o.get("parity") or {}
        if "bit_exact" in p:
            e["bit_exact"] = p["bit_exact"]
        if "grad_err_over_max_abs" in p:
            e["grad_err"] = _num(max(p["grad_err_over_max_abs"].values()), 2)
        for k in ("instances_per_gaussian",):
            if k in o.get("config", {}):
                e[k] = o["config"][k]
        for k in ("scratch_bytes", "capacity_retries", "views_per_s_per_instance_vs_headline"):
            if k in o:
                e[k] = _num(o[k], 3)
        digest[name] = e

    for name in ("train", "c2", "c3_fp16_sh", "fast_exp", "c3_closeup"):
        if name in full:
            rast(name, full[name])
    if "hipgraph_replay" in full:
        digest["hipgraph_replay"] = _num(full["hipgraph_replay"]["value"], 4)
    for group in ("cost_volume", "ptf", "encoder_tail"):
        for name, o in (full.get(group) or {}).items():
            if not isinstance(o, dict):
                continue
            if "error" in o:
                digest[f"{group}.{name}"] = {"error": str(o["error"])[:80]}
                continue
            e = {"ms": _num(o.get("ms_per_call"), 4)}
            if isinstance(o.get("roofline"), dict):
                e["frac"] = _num(o["roofline"].get("frac"), 3)
                if isinstance(o["roofline"].get("l1_tap_roofline"), dict):
                    e["l1_frac"] = _num(o["roofline"]["l1_tap_roofline"]["frac"], 3)
            if isinstance(o.get("roofline_mfma"), dict):      # (the fold's GRU on the fp32 matrix pipe: the ruler that binds it)
                e["mfma_frac"] = _num(o["roofline_mfma"].get("frac"), 3)
            t = o.get("train_fwd_bwd") or {}
            tm = t.get("ms", t.get("hip_ms"))
            if tm is not None:
                e["train_ms"] = _num(tm, 4)
                if isinstance(t.get("roofline"), dict):
                    e["train_frac"] = _num(t["roofline"].get("frac"), 3)
            if "ms_fwd_bwd" in o:
                e["train_ms"] = _num(o["ms_fwd_bwd"], 4)
            p = o.get("parity") or {}
            for k in ("max_abs_err_vs_oracle", "max_rel_err_vs_oracle"):
                if k in p:
                    e["err"] = _num(p[k], 2)
            if "cells_above_1e-4" in p:      # (the one cost-volume cell in 10 M whose err is a border validity flip: say how many)
                e["cells_above_1e-4"] = p["cells_above_1e-4"]
            if "same_count_and_order" in p:
                e["order_ok"] = p["same_count_and_order"]
                e["views_cmp"] = p.get("views_compared")
            digest[f"{group}.{name}"] = e
    for name in ("c3_train_step_hotpath", "c4_eval_step_hotpath", "c5_eval_step_hotpath"):
        if isinstance(full.get(name), dict):
            o = full[name]
            digest[name] = ({"error": str(o["error"])[:80]} if "error" in o else
                            {k: v for k, v in _pick(o, ("ms_per_step", "library_kernel_ms", "glue_ms", "glue_frac_of_gpu_time",
                                                        "gaussians", "target_views")).items() if v is not None})
    h["sections"] = digest
    h["full_line"] = "previous stdout line; gpurun_out/bench_full.json"
    # never exceed the driver's window: drop digest entries (last first) until the line fits
    while len(json.dumps(h)) > HEADLINE_MAX_BYTES and h["sections"]:
        h["sections"].pop(next(reversed(h["sections"])))
        h["sections_truncated"] = True
    return h


def emit(full: dict):
    """Rank 0's output: the full sectioned object on one line FIRST (also written to gpurun_out/bench_full.json), then -- as the
    LAST line, the one the driver parses -- the compact headline object."""
    line = json.dumps(full)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_full.json"), "w") as f:
            f.write(line + "\n")
    except OSError:
        pass
    print(line, flush=True)
    h = json.dumps(headline(full))
    assert len(h) <= HEADLINE_MAX_BYTES, len(h)
    print(h, flush=True)


def respawn_under_launcher(args) -> int:
    """`python bench.py --gpus N` (N > 1) without a launcher's environment: start the N ranks ourselves --
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free port>
    bench.py <same arguments>` -- and return its exit status (the driver's command shape is `python3 bench.py --gpus N ...`;
    the reference's own launch is one process that spawns its ranks, src/main.py:96-110)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(respawn_under_launcher(args))
    cx = Ctx(args)
    cpu = not args.no_cpu_baseline
    out = bench_raster(cx, args.workload, args.mode, args.views, args.steps, args.warmup, cpu)
    sections = []
    if cx.world == 1 and args.sections != "raster":
        sections = ["train", "c2", "closeup", "c5", "cost_volume", "ptf", "encoder_tail", "c3_step"] if args.sections == "all" else args.sections.split(",")

    def section(fn):
        """A secondary measurement must never take the headline line down with it."""
        try:
            return fn()
        except Exception as e:  # noqa: BLE001 -- reported in the line
            import traceback
            return {"error": f"{type(e).__name__}: {e}", "traceback": traceback.format_exc(limit=6)}

    if sections and args.mode == "fwd" and not _R_FAST():
        # the opt-in hardware exp of the blend (FS_RASTER_FAST_EXP): same workload, throughput + what it costs in parity
        from freesplat_amd import rasterizer as _R
        _R.FAST_EXP = True
        try:
            fx = section(lambda: bench_raster(cx, args.workload, "fwd", args.views, args.steps, args.warmup, cpu))
        finally:
            _R.FAST_EXP = False
        out["fast_exp"] = {k: fx[k] for k in ("value", "unit", "ms_per_step", "roofline", "kernel_ms_per_view", "parity", "error",
                                              "traceback") if k in fx}
        out["fast_exp"]["what"] = ("rasterizer.FAST_EXP = True: hardware v_exp_f32 in the blend loops, alpha >= 1/255 decisions "
                                   "guarded (re-evaluated with the contract exp inside +-16 ulp of the threshold); opt-in "
                                   "because the image is within ~1e-6 of, not bit-identical to, the exact mode's")
    if "train" in sections and args.mode != "train":
        out["train"] = section(lambda: bench_raster(cx, args.workload, "train", min(args.views, 8), max(3, args.steps // 2), 2, cpu))
    if "c2" in sections and not args.workload.startswith("c2"):
        out["c2"] = section(lambda: bench_raster(cx, "c2_640x480_300k", "fwd", args.views, args.steps, args.warmup, cpu))
    if "closeup" in sections and args.mode == "fwd":
        # the unfriendly config-3 workload (synthetic.WORKLOADS): every splat 2.5x its size, every tile list beyond the 2 048
        # entries that sort in LDS -> the global-memory sort path of sort_blend_kernel
        cu = section(lambda: bench_raster(cx, "c3_closeup_968x1296_1M", "fwd", min(args.views, 4), max(3, args.steps // 4), 1, cpu))
        if "error" not in cu:
            per_inst = lambda o: o["value"] * o["config"]["instances_per_view"]
            cu["views_per_s_per_instance_vs_headline"] = per_inst(cu) / per_inst(out)
            cu["scratch_bytes"] = cu["raster_buffers"]["scratch_bytes_per_stream"]
            cu["capacity_retries"] = cu["raster_buffers"]["capacity_retries"]
        out["c3_closeup"] = cu
    if "c5" in sections and args.mode == "fwd":
        # BASELINE config 5's storage option on the headline workload: SH coefficients held in fp16
        out["c3_fp16_sh"] = section(lambda: bench_raster(cx, args.workload, "fwd", args.views, args.steps, args.warmup, cpu, sh_fp16=True))
    if "cost_volume" in sections or "ptf" in sections or "encoder_tail" in sections:
        import bench_encoder as be
        if "encoder_tail" in sections:
            # the rows SURVEY.md 8(f) marks "next": the DepthDecoder's regression tail and the latent -> Gaussian head
            out["encoder_tail"] = {
                "depth_tail": section(lambda: be.bench_depth_tail(cx.dev, args.steps, args.warmup)),
                "gaussian_head": section(lambda: be.bench_gaussian_head(cx.dev, args.steps, args.warmup, cpu=cpu)),
            }
        if "cost_volume" in sections:
            out["cost_volume"] = {
                "native_96x128_K1": section(lambda: be.bench_cost_volume(cx.dev, args.steps, args.warmup, cpu=cpu)),
                "c3scale_242x324_K2": section(lambda: be.bench_cost_volume(cx.dev, max(3, args.steps // 4), 1, V=3, K=2, h4=242,
                                                                           w4=324, cpu=cpu, cpu_views=1)),
                # config 4's shape: 10 context views, the 9 pose-nearest as sources (K = 8); CPU baseline / parity on ONE of
                # its current views (the whole volume also has a parity test: tests/test_configs_4_5.py)
                "fvt10_96x128_K8": section(lambda: be.bench_cost_volume(cx.dev, max(3, args.steps // 4), 1, V=10, K=8, cpu=cpu,
                                                                        cpu_views=1)),
            }
        if "ptf" in sections:
            out["ptf"] = {
                "fold_2_views": section(lambda: be.bench_ptf(cx.dev, args.steps, args.warmup, cpu=cpu)),
                "fold_10_views": section(lambda: be.bench_ptf(cx.dev, max(2, args.steps // 4), 1, V=10, cpu=cpu)),
                # BASELINE config 3's image: 3 views at 968x1296 = 3.76 M raw Gaussians
                "fold_3_views_968x1296": section(lambda: be.bench_ptf(cx.dev, max(2, args.steps // 4), 1, V=3, h=968, w=1296, cpu=cpu)),
                # BASELINE config 5's long sequence: 30 views at 384x512, CPU baseline / parity over ALL 30 views (~8 s of oracle)
                "fold_30_views": section(lambda: be.bench_ptf(cx.dev, 2, 1, V=30, cpu=cpu, train=False)),
            }
    if "c3_step" in sections:
        # BASELINE config 3 as written: ONE composed training step at 3 x 968x1296 (bench_c3_step.py)
        import bench_c3_step as bc
        out["c3_train_step_hotpath"] = section(lambda: bc.bench_c3_step(cx.dev, steps=5, warmup=3))
        # BASELINE config 4's work on ONE of its GPUs as one composed EVALUATION step: 10 context views at the native 384x512, each
        # matched against its 8 pose-nearest views (num_views = 9), the 10-view fold, 8 rendered target views; no autograd
        out["c4_eval_step_hotpath"] = section(lambda: bc.bench_c3_step(cx.dev, steps=5, warmup=2, H=384, W=512, V=10, n_targets=8,
                                                                       train=False, num_views=9, workload="c4_eval_step_hotpath"))
        # BASELINE config 5 ("Replica 10-views eval, 30-view long-sequence fusion, fp16 SH coeffs") on ONE of its GPUs as one composed
        # evaluation step: 30 context views at 384x512, each matched against its 8 pose-nearest views, the 30-view fold, 8 rendered
        # target views with the SH coefficients stored in fp16; no autograd
        out["c5_eval_step_hotpath"] = section(lambda: bc.bench_c3_step(cx.dev, steps=3, warmup=1, H=384, W=512, V=30, n_targets=8,
                                                                       train=False, num_views=9, workload="c5_eval_step_hotpath",
                                                                       sh_fp16=True))
    if cx.rank == 0:
        emit(out)
    if cx.dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

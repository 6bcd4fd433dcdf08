#!/usr/bin/env python
"""Secondary benchmark (not the headline metric): the encoder-side rows of SURVEY.md section 8 --
plane-sweep cost volume, one PTF fold and the depth-regression tail -- on the shipped native shapes, with the reference-pinned
CPU oracles timed beside them.  Prints one JSON line per workload.

  python bench_encoder.py [--steps 20 --warmup 3]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import torch


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / max(steps, 1)


_EVENT_OVERHEAD = {}


def _event_pair_overhead_ms(dev) -> float:
    """Mean elapsed time of an EMPTY event bracket on the current stream (record, record): what every timed launch span of the
    library's stage hooks contains besides the kernel.  Measured once per run (median of 200 brackets)."""
    key = str(dev)
    if key not in _EVENT_OVERHEAD:
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
        for a_, b_ in evs:
            a_.record(); b_.record()
        torch.cuda.synchronize()
        ts = sorted(a_.elapsed_time(b_) for a_, b_ in evs)
        _EVENT_OVERHEAD[key] = float(ts[len(ts) // 2])
    return _EVENT_OVERHEAD[key]


def _traffic(workload):
    """(bytes per call / fold, source) from the newest committed profiles/*_traffic.json (profiles/tools/fwd_traffic.py)."""
    sys.path.insert(0, os.path.join(ROOT, "profiles", "tools"))
    try:
        import fwd_traffic
        return fwd_traffic.lookup(workload)
    except Exception:
        return None, None


def _mfma_busy(kernel_prefix):
    """Matrix-pipe busy fraction of the sweep kernel from the newest committed PMC summary (SQ_VALU_MFMA_BUSY_CYCLES /
    (1024 SIMDs x kernel-trace duration x 2.4 GHz); profiles/run_rocprof_encoder.sh: native 96x128 K = 1 only)."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_cv_sq_counters.json")), reverse=True):
        try:
            d = json.load(open(f))
            for k, v in d.get("kernels", d).items():
                if isinstance(v, dict) and k.startswith(kernel_prefix) and "mfma_busy_frac_at_2.4GHz" in v:
                    return float(v["mfma_busy_frac_at_2.4GHz"]), os.path.relpath(f, ROOT)
        except Exception:
            continue
    return None, None


def bench_cost_volume(dev, steps, warmup, V=2, K=1, h4=96, w4=128, D=128, C=48, cpu=True, cpu_views=None):
    """`cpu_views`: how many of the V current views the CPU baseline sweeps (bounded sample; default all)."""
    import inputs
    from freesplat_amd import _lib
    from freesplat_amd.cost_volume import AVGFeatureVolumeManager
    from oracle import cost_volume_oracle as cvo
    torch.manual_seed(0)
    m = AVGFeatureVolumeManager(matching_height=h4, matching_width=w4, num_depth_bins=D,
                                mlp_channels=[202, 32, 32, 1], matching_dim_size=C)
    kw = inputs.cv_inputs(V, K, h4, w4, C, seed=1)
    sd = {k.replace(".", "__"): v for k, v in m.state_dict().items()}
    mg = AVGFeatureVolumeManager(matching_height=h4, matching_width=w4, num_depth_bins=D,
                                 mlp_channels=[202, 32, 32, 1], matching_dim_size=C)
    mg.load_state_dict(m.state_dict())
    mg = mg.to(dev)
    args = {k: v.to(dev) for k, v in kw.items()}
    with torch.no_grad():
        out = mg(**args).cpu()
        # the call rate is timed with the stage events OFF (two hipEventRecords per stage of every call are host work the
        # product does not do); a second loop of the same length, events on, gives the kernels' durations
        dt = timed(lambda: mg(**args), steps, warmup)
        _lib.profile_collect(); _lib.profile_enable(True)
        timed(lambda: mg(**args), steps, 0)
        _lib.profile_enable(False)
        ms, cnt = _lib.profile_collect()["cost_volume"]
    flops = V * h4 * w4 * D * (480 * K + 5248)
    # the same inference call on channels_last feature maps (K >= 2: read in place by fs_cost_volume_forward_layout, no re-layout pass)
    dt_cl = None
    if K >= 2:
        args_cl = dict(args, cur_feats=args["cur_feats"].contiguous(memory_format=torch.channels_last),
                       src_feats=args["src_feats"].permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3))
        with torch.no_grad():
            same = bool(torch.equal(mg(**args_cl).cpu(), out))
            dt_cl = timed(lambda: mg(**args_cl), steps, warmup)
    wl_name = {(2, 1, 96): "cv_native_K1", (3, 2, 242): "cv_c3scale_K2", (10, 8, 96): "cv_fvt10_K8"}.get((V, K, h4), "")
    kern = ms / max(cnt, 1) * 1e-3
    # training step of the volume: forward + backward w.r.t. both feature maps and the six MLP tensors
    ga = {k: (v.clone().requires_grad_(True) if k in ("cur_feats", "src_feats") else v) for k, v in args.items()}

    cot = torch.ones(V, D, h4, w4, device=dev)      # the volume's cotangent (in training it arrives from the depth network)
    leaves = [ga["cur_feats"], ga["src_feats"]] + list(mg.parameters())

    def train_step():
        # gradients start from None, as after optimizer.zero_grad(set_to_none=True): rounds 4 - 5 let every backward ACCUMULATE into
        # the previous step's .grad -- eight torch `add` kernels per step, two of them over the feature maps' 45 + 91 MB at
        # config-3 scale, plus a 120 MB ones_like fill: ~0.15 ms of harness inside `train_ms`
        for t in leaves:
            t.grad = None
        o = mg(**ga)
        o.backward(cot)
    dt_train = timed(train_step, max(2, steps // 4), 2)
    # the backward alone, event-timed through the library's stage hooks (every launch of a training step is in the
    # cost_volume stage: forward sweep + relayouts, then the backward's two passes + relayouts)
    n_tr = max(2, steps // 4)
    _lib.profile_collect(); _lib.profile_enable(True)
    timed(train_step, n_tr, 0)
    _lib.profile_enable(False)
    ms_tr, _ = _lib.profile_collect()["cost_volume"]
    bwd_ms = max(ms_tr / n_tr - kern * 1e3, 1e-6)
    ws_bwd = _lib.lib().fs_cost_volume_backward_workspace_bytes(V, K, C, h4, w4, D)
    extra = {}
    if cpu:
        nv = V if cpu_views is None else min(V, cpu_views)
        cores = min(64, os.cpu_count() or 1)     # (large fused tensor ops: more threads than this only add synchronisation)
        torch.set_num_threads(cores)
        t0 = time.perf_counter()
        ref = cvo.cost_volume(kw["cur_feats"][:nv], kw["src_feats"][:nv], kw["src_extrinsics"][:nv], kw["src_Ks"][:nv],
                              kw["cur_invK"][:nv], kw["min_depth"], kw["max_depth"], D, cvo.mlp_from_state(sd))
        t_cpu = time.perf_counter() - t0
        torch.set_num_threads(8)
        e = (out[:nv] - ref).abs()
        extra = {"cpu_baseline": {"value": nv / t_cpu, "unit": "views/s", "cores": cores, "kind": "port",
                                  "sample": f"{nv} of {V} current view(s) through oracle/cost_volume_oracle.py (torch CPU, "
                                            "vectorised over D; pinned by the reference's golden volumes)"},
                 "parity": {"max_abs_err_vs_oracle": float(e.max()), "median_abs_err": float(e.median()),
                            "cells_above_1e-4": int((e > 1e-4).sum()), "cells": e.numel(),
                            "note": "cells above 1e-4 are validity flips of bilinear taps on the image border "
                                    "(tests/test_cost_volume_hip.py bounds them)"}}
    return dict({"metric": f"cost-volume views/sec @ {h4}x{w4} match res, D={D}, K={K}", "value": V / dt, "unit": "views/s",
            "ms_per_call": dt * 1e3, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cv_native", "views": V, "sources": K, "channels": C},
            "train_fwd_bwd": {"ms": dt_train * 1e3, "backward_workspace_bytes": int(ws_bwd),
                              "what": "forward + backward (features and all six MLP tensors); the backward runs in two passes "
                                      "-- records of C + 2 floats per (view, plane, pixel), then a sweep over tiles of source "
                                      "texels that accumulates in LDS: no global float atomics on the source maps",
                              "roofline": {"bound": "mfma", "kernel": "cost-volume backward (relayouts + cost_volume_bwd_kernel + "
                                                                      "cv_src_grad_kernel)",
                                           "algorithmic_flops_per_launch": 2 * flops, "avg_launch_ms": bwd_ms,
                                           "achieved": 2 * flops / (bwd_ms * 1e-3) / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                                           "frac": 2 * flops / (bwd_ms * 1e-3) / 1e12 / 157.3,
                                           "flops_note": "priced at twice the forward's algorithmic flops; the first pass issues 137 "
                                                         "fp32 MFMAs per 32 points (17.5 kFLOP per point: forward recompute, three "
                                                         "transposed layers, two weight-gradient products)",
                                           "traffic": _traffic("cvt_" + wl_name[3:])[0] if wl_name else None,
                                           "traffic_source": _traffic("cvt_" + wl_name[3:])[1] if wl_name else None,
                                           "traffic_unit": "L2-miss bytes (FETCH_SIZE / WRITE_SIZE) of one whole training step, forward "
                                                           "included; the tile sweep re-reads the records once per source",
                                           "global_float_atomics_on_source_maps": 0}},
            "roofline": {"bound": "mfma", "kernel": "cost_volume (relayout + sweep)", "achieved": flops / kern / 1e12,
                         "peak": 157.3, "unit": "TFLOP/s", "frac": flops / kern / 1e12 / 157.3,
                         "algorithmic_flops_per_launch": flops, "avg_launch_ms": kern * 1e3, "launches": cnt,
                         "mfma_busy_counter": _mfma_busy("fs::cost_volume_proj_kernel")[0] if wl_name == "cv_native_K1" else None,
                         "mfma_busy_note": "matrix-pipe busy fraction from SQ_VALU_MFMA_BUSY_CYCLES (native K = 1 profile); `frac` "
                                           "prices the reference formulation's flops (41 MFMAs per cell), the K = 1 sweep issues 16",
                         "traffic": _traffic(wl_name)[0], "traffic_source": _traffic(wl_name)[1],
                         "channels_last": None if dt_cl is None else {
                             "ms_per_call": dt_cl * 1e3, "same_volume_bit_for_bit": same,
                             "traffic": _traffic(wl_name + "_cl")[0], "traffic_source": _traffic(wl_name + "_cl")[1],
                             "what": "the same call with cur_feats / src_feats given as channels_last tensors: the 16-pixel sweep reads "
                                     "them in place, the two NCHW -> pixel-major re-layout launches are gone"},
                         "traffic_unit": "HBM bytes per call (all current views)",
                         "algorithmic_bytes_per_call": 4 * V * ((1 + K) * C + D) * h4 * w4,
                         # the general (K >= 2) sweep gathers 4 bilinear taps x C channels per (pixel, plane, source) through the
                         # vector L1: its real ceiling is the L1's 64 B per clock per CU, not the matrix pipe
                         "l1_tap_roofline": None if K < 2 else {
                             "bound": "vector L1 bandwidth", "tap_bytes_per_launch": V * K * D * h4 * w4 * 4 * C * 4,
                             "achieved": V * K * D * h4 * w4 * 4 * C * 4 / kern / 1e12, "peak": 256 * 64 * 2.4e9 / 1e12,
                             "unit": "TB/s", "frac": V * K * D * h4 * w4 * 4 * C * 4 / kern / (256 * 64 * 2.4e9)}}}, **extra)


def _ptf_w2c(E):
    from freesplat_amd.ptf import world_to_camera
    return world_to_camera(E).view(-1, 4, 4).cpu()


def bench_ptf(dev, steps, warmup, V=2, h=384, w=512, cpu=True, cpu_steps=None, train=True):
    """`cpu_steps`: bound the CPU baseline to the fold of the first cpu_steps + 1 views (the later steps of a long fold
    are larger -- the state grows -- so scaling that time to all V - 1 steps UNDER-estimates the CPU time)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_ptf_hip import _scene
    from freesplat_amd import _lib
    from freesplat_amd.ptf import PixelwiseTripletFusion
    from oracle import ptf_oracle as po
    E, Kn, depths, lat, dens, wts, coords = _scene(V, h, w, seed=5)
    torch.manual_seed(1)
    m = PixelwiseTripletFusion()
    params = {k: v.detach().clone() for k, v in m.gru.state_dict().items()}
    m = m.to(dev)
    d = lambda t: t.to(dev)
    a = ([d(lat)], [d(coords)], d(dens), d(wts), d(depths), d(E)[None], d(Kn)[None], (h, w))
    with torch.no_grad():
        out = [x.cpu() for x in m.fuse_gaussians(*a)]
        dt = timed(lambda: m.fuse_gaussians(*a), steps, warmup + 1)    # (stage events off: see bench_cost_volume)
        _lib.profile_collect(); _lib.profile_enable(True)
        timed(lambda: m.fuse_gaussians(*a), steps, 0)
        _lib.profile_enable(False)
        ms, cnt = _lib.profile_collect()["ptf"]
    # training step of the fold (forward + backward, every differentiable input and the GRU parameters): the HIP path (_PtfFold)
    # (the differentiable inputs are made ONCE; a step = forward, loss, backward, then dropping the gradients)
    ins = [t.detach().clone().requires_grad_(True) for t in (a[0][0], a[1][0], a[2], a[3], a[4])]

    def train_step_sum_loss(fn):
        """Rounds 3 - 4's harness: a sum() loss over the four outputs, leaves accumulating .grad -- per step 4 reductions, 3 adds,
        4 copies of the expanded cotangents and 5 clones in AccumulateGrad (the op returns views of ONE zero-filled buffer, which a
        leaf cannot adopt): ~0.2 ms of harness at 2 views that no caller of the op pays (upstream autograd nodes take the views)."""
        out = fn([ins[0]], [ins[1]], ins[2], ins[3], ins[4], *a[5:])
        sum(o.sum() for o in out).backward()
        for t in ins:
            t.grad = None
        for q in m.gru.parameters():
            q.grad = None
    cot = None

    def train_step(fn):
        """forward + backward of the fold with FIXED contiguous cotangents on its four outputs (torch.autograd.grad w.r.t. the five
        differentiable inputs and the 12 GRU tensors): the op's own training cost."""
        nonlocal cot
        out = fn([ins[0]], [ins[1]], ins[2], ins[3], ins[4], *a[5:])
        if cot is None or cot[0].shape != out[0].shape:
            gg = torch.Generator(device=dev).manual_seed(3)
            cot = [torch.randn(o.shape, device=dev, generator=gg) for o in out]
        torch.autograd.grad(out, ins + list(m.gru.parameters()), cot, allow_unused=True)
    n_tr = max(2, steps // 4)
    dt_train = timed(lambda: train_step(m.fuse_gaussians), n_tr, 2) if train else float("nan")
    dt_train_sum = timed(lambda: train_step_sum_loss(m.fuse_gaussians), n_tr, 1) if train else float("nan")
    ms_tr_stage = float("nan")
    if train:
        # the library kernels of a training step, event-timed through the stage hooks (a second loop: events off above)
        _lib.profile_collect(); _lib.profile_enable(True)
        timed(lambda: train_step(m.fuse_gaussians), n_tr, 0)
        _lib.profile_enable(False)
        ms_tr_stage = _lib.profile_collect()["ptf"][0] / n_tr
    with torch.no_grad():
        m.fuse_gaussians(*a)            # (LAST_FOLD_COUNTS of the inference fold)
    from freesplat_amd import ptf as _ptf
    steps_counts = _ptf.LAST_FOLD_COUNTS.tolist()      # [V,4]: kept, fused, appended, state rows after the step
    # algorithmic bytes of the fold (SURVEY.md 8(d), 344-byte state record): per step read M*12 (xyz) + P*4 (depth),
    # gather kept + both sides of every fused pair + appended pixels, write the new state
    P, REC = h * w, 344
    alg, M = 0, P
    for i in range(1, V):
        k, f, n_app, m_out = steps_counts[i]
        alg += M * 12 + P * 4 + (k + 2 * f + n_app) * REC + (k + f + n_app) * REC
        M = m_out
    # every launch of the library's ptf stage (event-bracketed), per fold call.  The event pairs themselves cost time inside the
    # bracketed spans (a 30-view fold has 58 of them): the roofline's denominator is never more than the un-instrumented wall
    # time of the call (fold_30_views reported kernel_ms_per_fold > ms_per_call)
    kern_ms_events = ms / steps
    # (no clamp -- the roofline's denominator is what the events measured; a per-pair event overhead, measured on an
    #  empty bracket in this run, is taken off instead)
    kern_ms = max(kern_ms_events - _event_pair_overhead_ms(dev) * (cnt / max(steps, 1)), 1e-6)
    extra = {}
    if cpu:
        cores = min(16, os.cpu_count() or 1)     # (the fold is many small operations: on all 256 host threads their
        torch.set_num_threads(cores)             #  synchronisation made it ~50x slower -- round 3's figures)
        nv = V if cpu_steps is None else min(V, cpu_steps + 1)
        with torch.no_grad():
            t0 = time.perf_counter()
            # (both sides project with the same world-to-camera matrices -- the product's GPU inverse: see
            #  tests/test_configs_4_5.py::test_config4_fold_at_its_real_size)
            ref = po.fuse_gaussians(params, lat[:, :nv], coords[:, :nv], dens[:, :nv], wts[:, :nv], depths[:nv], E[None, :nv],
                                    Kn[None, :nv], (h, w), w2c_all=_ptf_w2c(E[:nv].to(dev)))
            t_cpu = time.perf_counter() - t0
            got = out if nv == V else [x.cpu() for x in m.fuse_gaussians([a[0][0][:, :nv]], [a[1][0][:, :nv]], a[2][:, :nv], a[3][:, :nv],
                                                                         a[4][:nv], a[5][:, :nv], a[6][:, :nv], (h, w))]
        torch.set_num_threads(8)
        err = max(float((x - y).abs().max()) for x, y in zip(got, ref))
        scale = (V - 1) / (nv - 1)
        extra = {"cpu_baseline": {"value": 1.0 / (t_cpu * scale), "unit": "folds/s", "cores": cores, "kind": "port",
                                  "sample": (f"1 fold of {V} views" if nv == V else
                                             f"the fold of the first {nv} of {V} views ({nv - 1} of {V - 1} steps, time scaled by "
                                             f"{scale:.1f}: an under-estimate, later steps fold into a larger state)")
                                            + " through oracle/ptf_oracle.py (numpy match + torch CPU GRU; pinned by the "
                                              "reference's golden folds)"},
                 "parity": {"max_abs_err_vs_oracle": err, "same_count_and_order": bool(got[0].shape == ref[0].shape),
                            "views_compared": nv}}
    M_in, M_out = V * h * w, out[0].shape[1]
    return dict({"metric": f"PTF folds/sec, {V} views @ {h}x{w}", "value": 1.0 / dt, "unit": "folds/s", "ms_per_call": dt * 1e3,
                 "dtype": "f32 / int64 indices", "data": "synthetic",
                 "config": {"workload": f"ptf_{V}_views_{h}x{w}", "views": V, "gaussians_in": M_in, "gaussians_out": M_out,
                            "fused_pairs_per_step": [c[1] for c in steps_counts[1:]]},
                 "train_fwd_bwd": {"hip_ms": dt_train * 1e3 if train else None,
                                   "hip_ms_sum_loss_harness": dt_train_sum * 1e3 if train else None,
                                   "harness": "hip_ms: torch.autograd.grad with fixed cotangents on the four outputs; "
                                              "hip_ms_sum_loss_harness: rounds 3 - 4's sum() loss on leaf inputs (adds ~13 torch "
                                              "kernels of harness per step)",
                                   "roofline": None if not train else {
                                       "bound": "mfma", "kernel": "PTF training step: every library kernel of the fold and its "
                                       "backward (match, GRU, state movement; ptf_gru_bwd_kernel, ptf_gru_dw_kernel, "
                                       "ptf_gru_inputs*, ptf_write_state_bwd)",
                                       "algorithmic_flops_per_step": 3 * 2 * 44928 * sum(c[1] for c in steps_counts[1:]),
                                       "kernel_ms_per_step": ms_tr_stage,
                                       "achieved": 3 * 2 * 44928 * sum(c[1] for c in steps_counts[1:]) / (ms_tr_stage * 1e-3) / 1e12,
                                       "peak": 157.3, "unit": "TFLOP/s",
                                       "frac": 3 * 2 * 44928 * sum(c[1] for c in steps_counts[1:]) / (ms_tr_stage * 1e-3) / 1e12 / 157.3,
                                       "flops_note": "the GRU's 44 928 MACs per fused pair, priced once for the forward and twice for "
                                                     "the backward (input and weight gradients); the step's HBM-bound kernels (state "
                                                     "movement, index lists, 264 MB of per-pair factors per 10^5 pairs) are in the "
                                                     "denominator -- per-kernel traffic: profiles/r4_ptf_hbm_traffic.json"},

                                   "what": "forward + backward of the fold w.r.t. latents, coords, densities, weights, "
                                           "depths and the GRU parameters; gradients checked against the oracle's "
                                           "autograd in tests/test_ptf_hip.py"},
                 "roofline": {"bound": "hbm", "kernel": "ptf fold (match + gru_inputs + gru + write_state, all steps)",
                              "achieved": alg / (kern_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                              "frac": alg / (kern_ms * 1e-3) / 8e12, "algorithmic_bytes_per_fold": alg,
                              "kernel_ms_per_fold": kern_ms, "kernel_ms_per_fold_event_timed": kern_ms_events,
                              "launches": cnt // max(steps, 1),
                              "traffic": _traffic(f"ptf_{V}_views")[0], "traffic_source": _traffic(f"ptf_{V}_views")[1]},
                 # the other ruler of the same fold (round 6): its GRU -- 44 928 MACs per fused pair on v_mfma_f32_16x16x4_f32 -- is three
                 # quarters of the fold's GPU time (profiles/r6_gru_bwd_waves_ab.txt) and sits on the fp32 matrix pipe, not on HBM
                 "roofline_mfma": {"bound": "mfma", "kernel": "ptf_gru16_kernel<true> inside the fold (the whole fold's kernel time in the denominator)",
                                   "algorithmic_flops_per_fold": 2 * 44928 * sum(c[1] for c in steps_counts[1:]),
                                   "achieved": 2 * 44928 * sum(c[1] for c in steps_counts[1:]) / (kern_ms * 1e-3) / 1e12,
                                   "peak": 157.3, "unit": "TFLOP/s",
                                   "frac": 2 * 44928 * sum(c[1] for c in steps_counts[1:]) / (kern_ms * 1e-3) / 1e12 / 157.3}}, **extra)


def bench_depth_tail(dev, steps, warmup, V=2, D=128, h2=192, w2=256):
    """Regression tail of the finest DepthDecoder scale (networks.py:130-152), forward and forward+backward,
    against the same op chain in torch on this GPU (rocm eager) and on the host (the oracle)."""
    from freesplat_amd.depth_tail import depth_regression_tail
    from oracle.depth_tail_oracle import depth_tail
    g = torch.Generator().manual_seed(2)
    logits = 3.0 * torch.randn(V, D, h2, w2, generator=g)
    cand = torch.log(torch.tensor(0.5)) + torch.linspace(0, 1, D) * torch.log(torch.tensor(30.0))
    lg, cd = logits.to(dev).requires_grad_(True), cand.to(dev)

    def both(fn):
        o = fn(lg, cd, True)
        (o["depth_map"].sum() + o["depth_weights"].sum() + o["depth"].sum()).backward()
        lg.grad = None

    with torch.no_grad():
        dt = timed(lambda: depth_regression_tail(lg, cd, True), steps, warmup)
        dt_eager = timed(lambda: depth_tail(lg, cd, True), steps, warmup)
    dt_fb = timed(lambda: both(depth_regression_tail), steps, warmup)
    dt_fb_eager = timed(lambda: both(depth_tail), steps, warmup)
    torch.set_num_threads(os.cpu_count() or 1)
    with torch.no_grad():
        got = depth_regression_tail(lg, cd, True)
        t0 = time.perf_counter()
        ref = depth_tail(logits, cand, True)
        t_cpu = time.perf_counter() - t0
    torch.set_num_threads(8)
    err = max(float((got[k].cpu() - ref[k]).abs().max()) for k in ref)
    nbytes = logits.numel() * 4 + 2 * V * 4 * h2 * w2 * 4       # read the logits once, write the two x2 maps
    return {"metric": f"depth-regression tails/sec, {V} views x {D} planes @ {h2}x{w2}", "value": V / dt, "unit": "views/s",
            "ms_per_call": dt * 1e3, "ms_fwd_bwd": dt_fb * 1e3, "torch_eager_ms": dt_eager * 1e3,
            "torch_eager_fwd_bwd_ms": dt_fb_eager * 1e3, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "depth_tail_native", "views": V, "planes": D},
            "roofline": {"bound": "hbm", "achieved": nbytes / dt / 1e9, "peak": 8000.0, "unit": "GB/s",
                         "frac": nbytes / dt / 8e12, "traffic": None, "note": "wall clock per call, 2 kernels"},
            "cpu_baseline": {"value": V / t_cpu, "unit": "views/s", "cores": os.cpu_count(), "kind": "port",
                             "sample": "1 call of oracle/depth_tail_oracle.py (torch CPU)"},
            "parity": {"max_abs_err_vs_oracle": err}}


def bench_gaussian_head(dev, steps, warmup, M=290044, cpu=True):
    """Latent -> Gaussian head (gaussian_adapter.py:151-172) on the fused set of the native 2-view scene: forward and
    forward + backward, against the oracle on the host."""
    from freesplat_amd.gaussian_adapter import _Head
    from oracle import adapter_oracle as ao
    g = torch.Generator().manual_seed(3)
    raw = torch.randn(M, 34, generator=g)
    dep = 1.0 + torch.rand(M, generator=g)
    E = torch.eye(4).repeat(M, 1, 1) + 0.1 * torch.randn(M, 4, 4, generator=g)
    mult = torch.tensor([0.0123])
    mask = torch.tensor([1.0, .025, .025, .025, .00625, .00625, .00625, .00625, .00625])
    d = lambda t: t.to(dev)
    rg, dg, eg = d(raw).requires_grad_(True), d(dep).requires_grad_(True), d(E).requires_grad_(True)
    mg, kg = d(mult), d(mask)
    gs = [d(torch.randn(M, 3, 3, generator=g)), d(torch.randn(M, 3, 9, generator=g)), d(torch.randn(M, 3, generator=g)),
          d(torch.randn(M, 4, generator=g))]

    def both():
        o = _Head.apply(rg, dg, eg, mg, kg, 0.5, 15.0)
        torch.autograd.backward(list(o), gs)
        rg.grad = dg.grad = eg.grad = None

    with torch.no_grad():
        dt = timed(lambda: _Head.apply(rg, dg, eg, mg, kg, 0.5, 15.0), steps, warmup)
        got = _Head.apply(rg, dg, eg, mg, kg, 0.5, 15.0)
    dt_fb = timed(both, steps, warmup)
    nbytes = M * (34 + 1 + 16 + 9 + 27 + 3 + 4) * 4        # read raw, depth, extrinsics; write cov, sh, scales, rotations
    extra = {}
    if cpu:
        cores = min(64, os.cpu_count() or 1)
        torch.set_num_threads(cores)
        with torch.no_grad():
            t0 = time.perf_counter()
            ref = ao.gaussian_head(raw, dep, E, mult[0], mask)
            t_cpu = time.perf_counter() - t0
        torch.set_num_threads(8)
        err = max(float((a.cpu() - b).abs().max() / (b.abs().max() + 1e-20)) for a, b in zip(got, ref))
        extra = {"cpu_baseline": {"value": 1.0 / t_cpu, "unit": "calls/s", "cores": cores, "kind": "port",
                                  "sample": "1 call of oracle/adapter_oracle.py:gaussian_head (torch CPU)"},
                 "parity": {"max_rel_err_vs_oracle": err}}
    return dict({"metric": f"Gaussian-head calls/sec, {M} Gaussians", "value": 1.0 / dt, "unit": "calls/s",
                 "ms_per_call": dt * 1e3, "ms_fwd_bwd": dt_fb * 1e3, "dtype": "f32", "data": "synthetic",
                 "config": {"workload": "gaussian_head_native", "gaussians": M},
                 "roofline": {"bound": "hbm", "achieved": nbytes / dt / 1e9, "peak": 8000.0, "unit": "GB/s",
                              "frac": nbytes / dt / 8e12, "traffic": None, "note": "wall clock per call, 1 kernel"}}, **extra)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    print(json.dumps(bench_cost_volume(dev, a.steps, a.warmup)), flush=True)
    print(json.dumps(bench_cost_volume(dev, a.steps, a.warmup, V=3, K=2, h4=242, w4=324)), flush=True)
    print(json.dumps(bench_ptf(dev, a.steps, a.warmup)), flush=True)
    print(json.dumps(bench_ptf(dev, max(2, a.steps // 4), 1, V=10)), flush=True)
    print(json.dumps(bench_depth_tail(dev, a.steps, a.warmup)), flush=True)
    print(json.dumps(bench_gaussian_head(dev, a.steps, a.warmup)), flush=True)

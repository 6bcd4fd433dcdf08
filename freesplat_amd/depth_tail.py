"""Fused depth-regression tail of the DepthDecoder.

`depth_regression_tail(logits, candidates, log_planes, upsample)` replaces the op chain of
/root/reference/src/model/encoder/modules/networks.py:130-152 (softmax over the D planes -> expected
log-depth -> exp; for the finest scale also the x2 align_corners bilinear of the expectation and the
max over planes of the x2-upsampled probabilities) with fs_depth_tail_forward/backward
(libfreesplat_hip.so): the softmax and its upsampled copy are never materialised.
`apply_to_depth_outputs` fills the reference's `depth_outputs` keys from per-scale logits.
"""
from __future__ import annotations

import torch
from torch import Tensor

from . import _lib


class _DepthTail(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, cand, log_planes, upsample):
        B, D, h2, w2 = logits.shape
        dev = logits.device
        stats = torch.empty(B, 2, h2, w2, device=dev)
        coarse = torch.empty(B, 1, h2, w2, device=dev)
        depth = torch.empty(B, 1, h2, w2, device=dev)
        dmap = torch.empty(B, 1, 2 * h2, 2 * w2, device=dev) if upsample else None
        dw = torch.empty(B, 1, 2 * h2, 2 * w2, device=dev) if upsample else None
        am = torch.empty(B, 2 * h2, 2 * w2, dtype=torch.int32, device=dev) if upsample else None
        p = _lib.ptr
        _lib.check(_lib.lib().fs_depth_tail_forward(B, D, h2, w2, p(logits), p(cand), int(log_planes), p(stats), p(coarse),
                                                    p(depth), p(dmap), p(dw), p(am), _lib.current_stream()),
                   "fs_depth_tail_forward")
        ctx.save_for_backward(logits, cand, stats, coarse, depth, dmap, am)
        ctx.cfg = (bool(log_planes), bool(upsample))
        ctx.set_materialize_grads(False)
        if upsample:
            return coarse, depth, dmap, dw
        return coarse, depth

    @staticmethod
    def backward(ctx, g_coarse, g_depth, g_map=None, g_w=None):
        logits, cand, stats, coarse, depth, dmap, am = ctx.saved_tensors
        log_planes, upsample = ctx.cfg
        B, D, h2, w2 = logits.shape
        dev = logits.device
        if all(g is None for g in (g_coarse, g_depth, g_map, g_w)):
            return None, None, None, None
        c = lambda t: None if t is None else t.contiguous()
        sE = sP = None        # (scratch of the round 2 - 3 scatter form; the gather form of round 4 keeps both in LDS)
        g_logits = torch.empty_like(logits)
        p = _lib.ptr
        gc_, gd_, gm_, gw_ = c(g_coarse), c(g_depth), c(g_map), c(g_w)   # (kept alive until the launch is queued)
        _lib.check(_lib.lib().fs_depth_tail_backward(B, D, h2, w2, p(logits), p(cand), int(log_planes), p(stats), p(coarse),
                                                     p(depth), p(dmap), p(am), p(gc_), p(gd_), p(gm_), p(gw_), p(sE), p(sP),
                                                     p(g_logits), _lib.current_stream()), "fs_depth_tail_backward")
        return g_logits, None, None, None


def depth_regression_tail(logits: Tensor, candidates: Tensor, log_planes: bool = True, upsample: bool = True) -> dict:
    """logits [B,D,h2,w2] = conv_depth output, candidates [D] = depth_candi_curr.  Returns coarse (the
    reference's `log_depth_pred`), depth, and with `upsample` depth_map / depth_weights at twice the size."""
    if logits.device.type != "cuda":
        raise RuntimeError(f"freesplat_amd depth tail: tensors must live on a HIP device (got {logits.device}); no CPU path")
    out = _DepthTail.apply(logits.float().contiguous(), candidates.reshape(-1).float().contiguous().to(logits.device),
                           log_planes, upsample)
    keys = ("coarse", "depth", "depth_map", "depth_weights")
    return dict(zip(keys, out))


def apply_to_depth_outputs(depth_outputs: dict, logits_per_scale: dict, candidates: Tensor, log_planes: bool = True) -> dict:
    """Fill `depth_pred_s{i}_b1hw`, `log_depth_pred_s{i}_b1hw` (i = 3..0), `depth_pred_s-1_b1hw` and
    `depth_weights` exactly as networks.py:130-152 does; logits_per_scale = {i: conv_depth[i](output_pred_s{i})}."""
    for i in sorted(logits_per_scale, reverse=True):
        r = depth_regression_tail(logits_per_scale[i], candidates, log_planes, upsample=(i == 0))
        depth_outputs[f"depth_pred_s{i}_b1hw"] = r["depth"]
        depth_outputs[f"log_depth_pred_s{i}_b1hw"] = r["coarse"]
        if i == 0:
            depth_outputs["depth_pred_s-1_b1hw"] = r["depth_map"]
            depth_outputs["depth_weights"] = r["depth_weights"]
    return depth_outputs


def _up2(x: Tensor) -> Tensor:
    """sr_utils.generic_utils.upsample (src/loss/utils/generic_utils.py:97-107): x2 bilinear, align_corners=False."""
    return torch.nn.functional.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)


def depth_decoder_forward(self, input_features):
    """Drop-in for `DepthDecoder.forward` (networks.py:108-154), bound by compat.patch_reference(): the UNet++
    convolution pyramid runs through the module's own `convs` / `conv_depth` / `conv_last` (MIOpen; outside
    the hot path), the regression tail of every scale through the fused HIP op.  Same `depth_outputs` keys."""
    prev_outputs = input_features
    outputs = []
    depth_outputs = {}
    for j in range(1, self.max_depth + 1):
        for i in range(self.max_depth - j, -1, -1):
            inputs = [self.convs[f"right_conv_{i}{j - 1}"](prev_outputs[i]),
                      _up2(self.convs[f"diag_conv_{i + 1}{j - 1}"](prev_outputs[i + 1]))]
            if i + j != self.max_depth:
                inputs.append(_up2(self.convs[f"up_conv_{i + 1}{j}"](outputs[-1])))
            output = self.convs[f"in_conv_{i}{j}"](torch.cat(inputs, dim=1))
            outputs.append(output)
            depth_outputs[f"output_pred_s{i}_b1hw"] = self.convs[f"output_{i}"](output)
        prev_outputs = outputs[::-1]
    logits = {i: self.conv_depth[f"{i}"](depth_outputs[f"output_pred_s{i}_b1hw"]) for i in range(self.max_depth)}
    apply_to_depth_outputs(depth_outputs, logits, self.depth_candi_curr, self.log_planes)
    depth_outputs["output_pred_s-1_b1hw"] = self.conv_last(_up2(depth_outputs["output_pred_s0_b1hw"]))
    return depth_outputs

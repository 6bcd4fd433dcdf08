"""CPU restatement (torch) of the depth-regression tail of FreeSplat's DepthDecoder.

TEST INFRASTRUCTURE ONLY.  Pinned against the reference itself: tests/golden/depth_tail_{log,inv}.npz hold
plane logits produced by the real DepthDecoder (imported from /root/reference by
tests/golden/make_golden.py) and the outputs of its tail ops; tests/test_depth_tail_oracle.py checks this
file against them (including the module's own log_depth / depth_map / depth_weights outputs).

Follows /root/reference/src/model/encoder/modules/networks.py:130-152.
"""
import torch
import torch.nn.functional as F
from torch import Tensor


def depth_tail(logits: Tensor, candidates: Tensor, log_planes: bool = True, upsample: bool = True) -> dict:
    """logits [B,D,h2,w2] (output of conv_depth), candidates [D] (depth_candi_curr, :79-93)."""
    planes = F.softmax(logits, dim=1)                                                        # :131
    coarse = (candidates.view(1, -1, 1, 1) * planes).sum(dim=1, keepdim=True)                # :132
    out = dict(coarse=coarse, depth=torch.exp(coarse) if log_planes else 1.0 / coarse)       # :133-137
    if upsample:
        fine = F.interpolate(coarse, scale_factor=2, mode="bilinear", align_corners=True)    # :139-144
        out["depth_map"] = torch.exp(fine) if log_planes else 1.0 / fine                     # :145
        out["depth_weights"] = F.interpolate(planes, scale_factor=2, mode="bilinear",
                                             align_corners=True).max(dim=1, keepdim=True)[0]  # :148-152
    return out

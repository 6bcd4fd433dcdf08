"""Pixel-wise Triplet Fusion: drop-in for EncoderFreeSplat.fuse_gaussians.

Mirrors /root/reference/src/model/encoder/encoder_freesplat.py:431-522 (same signature, same four
outputs in the same ORDER) together with the modules it needs: `GRU`
(src/model/encoder/modules/networks.py:188-214, parameter names mlp_{z,r,n}.{0,2}.{weight,bias} and
construction order kept so checkpoints and seeded inits carry over); the positional encodings
(encoder_freesplat.py:62-77) are computed inside the kernels (fs_ptf_gru_inputs).

The fold is HIP in both modes (b = 1, the only shape the reference's indexing supports):
  * forward (inference AND training): per view fs_ptf_fold_step = match (projection of the M global Gaussians,
    per-pixel z-buffer, depth-consistency mask, winner selection, the ORDERED index lists) -> GRU inputs (gather +
    positional encodings) -> GRU on the fp32 matrix cores -> next state, every data-dependent size device-resident;
    ONE host sync per fold (the reference: four per view);
  * backward (_PtfFold): per step, in reverse, fs_ptf_write_state_backward (density-weighted blends, keep / append
    copies) and fs_ptf_gru_inputs_backward (gather + positional encodings) are HIP kernels, and so is the GRU's own
    backward (fs_ptf_gru_backward: forward re-run + the six linear layers transposed, on the fp32 matrix cores, giving
    the input-row gradients); only the weight gradients dW = dY^T X -- sums over ~10^5 pairs -- are library GEMMs on
    the per-pair factors the kernel writes;
  * no torch-op formulation of the fold lives in the product: the op-by-op cross-check of the HIP backward is
    tests/ptf_torch_ref.py.  One scene per call, as the reference's caller does (encoder_freesplat.py:355-368).
"""
from __future__ import annotations

import ctypes as C
import math
import os
import weakref

import torch
from torch import Tensor, nn

from . import _lib


class _GruRows(torch.autograd.Function):
    """The GRU on materialised rows [hid(64) | he(24) | x(64) | xe(24)]: fs_ptf_gru_forward, and for the backward
    fs_ptf_gru_backward + fs_ptf_gru_weight_grads (gru_backward below).  What GRU.forward runs when the module is called on
    its own; the fold uses the gathering variants of the same kernels."""

    @staticmethod
    def forward(ctx, cat, gru, *params):
        n = cat.shape[0]
        tab = gru_tables(gru)
        out = torch.empty(n, 64, device=cat.device)
        if n:
            _lib.check(_lib.lib().fs_ptf_gru_forward(n, _lib.ptr(cat), _lib.ptr(tab), _lib.ptr(out), _lib.current_stream()),
                       "fs_ptf_gru_forward")
        ctx.gru = gru
        ctx.save_for_backward(cat)
        return out

    @staticmethod
    def backward(ctx, g):
        (cat,) = ctx.saved_tensors
        gru = ctx.gru
        dcat, grads = gru_backward(_gru_params(gru), gru_tables(gru), gru_operand_stream(gru), cat, g.float().contiguous())
        return (dcat, None) + tuple(grads)


class GRU(nn.Module):
    """networks.py:188-214: same constructor, parameter names and call signature.  `forward` runs on the HIP kernels (rows
    through fs_ptf_gru_forward / fs_ptf_gru_backward on the fp32 matrix cores); there is no CPU path (the
    reference-pinned restatement the tests compare against lives with the test infrastructure, not in this package)."""

    def __init__(self, input_channel=64, hidden_channel=64, weights_dim=24):
        super().__init__()
        if (input_channel, hidden_channel, weights_dim) != (64, 64, 24):
            raise ValueError("freesplat_amd GRU: the kernels are built for 64 / 64 / 24 channels (the reference's only use, "
                             "encoder_freesplat.py:163)")
        mk = lambda d: nn.Sequential(nn.Linear(d, hidden_channel), nn.ReLU(), nn.Linear(hidden_channel, hidden_channel))
        self.mlp_z = mk(hidden_channel + input_channel + 2 * weights_dim)
        self.mlp_r = mk(hidden_channel + input_channel + 2 * weights_dim)
        self.mlp_n = mk(hidden_channel + input_channel + 1 * weights_dim)

    def forward(self, input_feat, hidden_feat, input_weights_emb, hidden_weights_emb):
        if input_feat.device.type != "cuda":
            raise RuntimeError(f"freesplat_amd GRU: tensors must live on a HIP device (got {input_feat.device}); no CPU path")
        if hidden_feat is None:
            hidden_feat = torch.zeros_like(input_feat)
        lead = input_feat.shape[:-1]
        rows = [t.reshape(-1, t.shape[-1]).float() for t in (hidden_feat, hidden_weights_emb, input_feat, input_weights_emb)]
        cat = torch.cat(rows, dim=-1).contiguous()                 # [n, 176] = [hid | he | x | xe]
        return _GruRows.apply(cat, self, *_gru_params(self)).reshape(lead + (64,))


def match_view(xyz: Tensor, w2c: Tensor, kpix: Tensor, depth_i: Tensor, h: int, w: int, depth_thres: float = 0.1):
    """fs_ptf_match on device tensors: xyz [M,3], w2c [4,4], kpix [4], depth_i [h*w] -> ascending int64
    index tensors (keep_idx, fuse_idx, fuse_pix, append_pix).  One host sync (the three counts)."""
    if xyz.device.type != "cuda":
        raise RuntimeError(f"freesplat_amd PTF: tensors must live on a HIP device (got {xyz.device}); no CPU path")
    dev = xyz.device
    M, P = xyz.shape[0], h * w
    L = _lib.lib()
    xyz = xyz.detach().float().contiguous()
    scratch = torch.empty(L.fs_ptf_scratch_bytes(M, h, w), dtype=torch.uint8, device=dev)
    keep = torch.empty(max(M, 1), dtype=torch.int64, device=dev)
    fuse = torch.empty(max(M, 1), dtype=torch.int64, device=dev)
    fpix = torch.empty(max(M, 1), dtype=torch.int64, device=dev)
    app = torch.empty(P, dtype=torch.int64, device=dev)
    counts = torch.empty(4, dtype=torch.int32, device=dev)
    p = _lib.ptr
    w2c_, kpix_, depth_ = (t.detach().float().contiguous() for t in (w2c, kpix, depth_i))  # alive until the launch is queued
    _lib.check(L.fs_ptf_match(M, h, w, p(xyz), p(w2c_), p(kpix_), p(depth_),
                              C.c_float(depth_thres), p(scratch), p(keep), p(fuse), p(fpix), p(app), p(counts),
                              _lib.current_stream()), "fs_ptf_match")
    nk, nf, na, _ = counts.tolist()
    return keep[:nk], fuse[:nf], fpix[:nf], app[:na]


# operand tables per GRU module, valid while the parameters keep their storage and version; weak keys: a table must not
# outlive its module (a new module may get the same id(), the same parameter addresses and the same version counters)
_KEEP_BYTES_ENV = os.environ.get("FREESPLAT_PTF_KEEP_BYTES")   # see _PtfFold.forward


def _keep_bytes(device) -> int:
    """How many bytes of worst-case fold state a training fold may keep untrimmed until its backward: FREESPLAT_PTF_KEEP_BYTES if
    set, else 1/16 of the memory currently free on the device, capped at 8 GiB (18 GB on an idle MI355X -> 8 GiB; a card with
    16 GB free keeps 1 GB and trims above it -- ADVICE r5: a fixed 8 GiB could run smaller devices out of memory)."""
    if _KEEP_BYTES_ENV is not None:
        return int(_KEEP_BYTES_ENV)
    try:
        free, _total = torch.cuda.mem_get_info(device)
    except Exception:
        return 1 << 30
    return int(min(8 << 30, free // 16))
_table_cache: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()
LAST_FOLD_COUNTS = None   # device tensor [V,4] (kept, fused, appended, state rows) of the last fused fold: bench accounting


def gru_tables(gru: "GRU") -> Tensor:
    """The GRU's weights and biases as MFMA operands of csrc/ptf_gru.hip: rows of 64 lanes, lane l = (p = l & 31,
    hf = l >> 5), one row per MFMA in the order the kernel consumes them, then the bias rows.  Cached per parameter
    version."""
    params = _gru_params(gru)
    key = tuple((q.data_ptr(), q._version) for q in params)
    hit = _table_cache.get(gru)
    if hit is not None and hit[0] == key:
        return hit[1]
    dev = params[0].device
    if _lib.lib().fs_ptf_gru_table_layout() == 1:          # the 16-pair forward kernel's tables
        tab = _gru_operand_stream16(gru, forward_only=True)
        assert tab.shape[0] == _lib.lib().fs_ptf_gru_table_rows()
        _table_cache[gru] = (key, tab)
        return tab
    with torch.no_grad():
        Wr1, br1, Wr2, br2, Wz1, bz1, Wz2, bz2, Wn1, bn1, Wn2, bn2 = [q.detach().float() for q in params]
        lane = torch.arange(64, device=dev)
        pp, hf = lane & 31, lane >> 5
        acc_row = lambda q, h: (q & 3) + 8 * (q >> 2) + 4 * h
        unit = lambda s: acc_row(s[:, None] & 15, hf[None, :]) + 32 * (s[:, None] >> 4)      # [steps, 64]

        def l1(W):   # [64,176]: step s <-> input s + 88*hf
            s = torch.arange(88, device=dev)
            col = s[:, None] + 88 * hf[None, :]
            return torch.stack([W[32 * b + pp[None, :].expand_as(col), col] for b in range(2)])

        def l2(W):   # [64,64]: step s <-> hidden unit of accumulator register s & 15, block s >> 4
            col = unit(torch.arange(32, device=dev))
            return torch.stack([W[32 * b + pp[None, :].expand_as(col), col] for b in range(2)])

        def n1(W):   # [64,152]: steps 0..31 <-> r*hid units, 32..75 <-> x|xe input 64 + (s-32) + 44*hf
            s = torch.arange(44, device=dev)
            col = torch.cat([unit(torch.arange(32, device=dev)), 64 + s[:, None] + 44 * hf[None, :]])
            return torch.stack([W[32 * b + pp[None, :].expand_as(col), col] for b in range(2)])

        def bias(bv):  # [64] -> [2,16,64]
            q = torch.arange(16, device=dev)
            return torch.stack([bv[acc_row(q[:, None], hf[None, :]) + 32 * b] for b in range(2)])

        # one row per MFMA, in the order ptf_gru_kernel consumes them (the workgroup streams them through LDS once for
        # its four wavefronts): per k-step the row blocks of the matrices that share the step's B operand
        r1, z1, r2, z2, nn1, nn2 = l1(Wr1), l1(Wz1), l2(Wr2), l2(Wz2), n1(Wn1), l2(Wn2)       # [2 blocks, steps, 64]
        il = lambda *ms: torch.stack([m[b] for m in ms for b in range(2)], dim=1).reshape(-1, 64)   # [steps * 2 * len(ms), 64]
        ops = torch.cat([il(r1, z1), il(r2, z2), il(nn1), il(nn2)])
        rows_b = _lib.lib().fs_ptf_gru_table_rows() - 6 * 32
        assert ops.shape[0] == 696 <= rows_b
        tab = torch.cat([ops, ops.new_zeros(rows_b - ops.shape[0], 64)] +
                        [bias(bv).reshape(-1, 64) for bv in (br1, bz1, br2, bz2, bn1, bn2)]).contiguous()
    assert tab.shape[0] == _lib.lib().fs_ptf_gru_table_rows()
    _table_cache[gru] = (key, tab)
    return tab


_table_t_cache: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()


def gru_tables_t(gru: "GRU") -> Tensor:
    """The GRU's six weight matrices TRANSPOSED, in the MFMA operand order of csrc/ptf_gru.hip:ptf_gru_bwd_kernel:
    row (rb, s) of a matrix, lane l = (p = l & 31, hf = l >> 5), holds W[u(s, hf)][32 rb + p] -- u = the forward's
    accumulator unit map -- and 0 where 32 rb + p is not an input of that matrix.  mlp_n's first layer appears twice:
    its r*hid columns as two row blocks of their own, its x | xe columns placed at their positions among the 176
    features of the concatenated row (blocks 2..5), so that they accumulate straight into dcat.  Cached per parameter
    version."""
    params = _gru_params(gru)
    key = tuple((q.data_ptr(), q._version) for q in params)
    hit = _table_t_cache.get(gru)
    if hit is not None and hit[0] == key:
        return hit[1]
    dev = params[0].device
    with torch.no_grad():
        Wr1, _, Wr2, _, Wz1, _, Wz2, _, Wn1, _, Wn2, _ = [q.detach().float() for q in params]
        lane = torch.arange(64, device=dev)
        pp, hf = lane & 31, lane >> 5
        s = torch.arange(32, device=dev)
        unit = ((s[:, None] & 15) & 3) + 8 * ((s[:, None] & 15) >> 2) + 4 * hf[None, :] + 32 * (s[:, None] >> 4)   # [32, 64]

        def tr(W, blocks, col_of_feature=None):
            """[blocks*32 rows, 64 lanes]: W[unit(s, hf), col(32 rb + p)], zero where the column does not exist."""
            out = []
            Wp = torch.cat([W, torch.zeros(W.shape[0], 1, device=dev)], dim=1)       # last column = 0 (the "no input" slot)
            for rb in blocks:
                f = 32 * rb + pp                                                     # feature of this lane's row
                col = f if col_of_feature is None else col_of_feature(f)
                col = torch.where((col >= 0) & (col < W.shape[1]), col, torch.full_like(col, W.shape[1]))
                out.append(Wp[unit, col[None, :].expand_as(unit)])
            return torch.cat(out)

        n1_from_cat = lambda f: torch.where(f >= 88, f - 24, torch.full_like(f, -1))   # cat feature -> mlp_n input (x | xe)
        tab = torch.cat([tr(Wn2, range(2)), tr(Wn1, range(2)), tr(Wn1, range(2, 6), n1_from_cat), tr(Wr2, range(2)),
                         tr(Wz2, range(2)), tr(Wr1, range(6)), tr(Wz1, range(6))]).contiguous()
    assert tab.shape == (_lib.lib().fs_ptf_gru_table_t_rows(), 64)
    _table_t_cache[gru] = (key, tab)
    return tab


# first rows of the matrices in the transposed operand table (csrc/ptf_gru.hip)
_KTN2, _KTN1H, _KTN1C, _KTR2, _KTZ2, _KTR1, _KTZ1 = 0, 64, 128, 256, 320, 384, 576
_stream_cache: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()


_stream16_index_cache: dict = {}


def _gru_stream16_index(forward_only: bool, transposed_only: bool = False) -> "np.ndarray":
    """For every float of the 16-pair kernels' operand stream, its position in the concatenation of the GRU's 12 parameter
    tensors (flattened, in _gru_params order) followed by one zero -- the stream is ONE gather of that vector.  Rows of 64 lanes
    per v_mfma_f32_16x16x4_f32 in consumption order, lane l = (i = l & 15, kk = l >> 4) holding A[i][kk] --
      forward layer, output block ob:     W[16 ob + i][input of k-step s for quarter kk]
      transposed layer, feature block ob: W[unit of k-step s for quarter kk][feature 16 ob + i]      (the zero where there is none)
    where a 64-unit activation is consumed register by register of the accumulator layout: k-step s <-> units
    16 (s >> 2) + 4 kk + (s & 3).  696 forward + 704 transposed rows, padded to whole ring chunks and interleaved by quads of rows;
    then the six bias vectors (64 floats each).  forward_only: the first 696 rows, packaged the same way; transposed_only: the
    last 704 rows alone (exactly 11 chunks, no bias rows) -- the stream of fs_ptf_gru_backward_saved."""
    import numpy as np
    sizes = [int(np.prod(sh)) for sh in GRU_PARAM_SHAPES]
    off = np.concatenate([[0], np.cumsum(sizes)])
    zero = int(off[-1])
    R1, BR1, R2, BR2, Z1, BZ1, Z2, BZ2, N1, BN1, N2, BN2 = range(12)
    ncols = {m: GRU_PARAM_SHAPES[m][1] for m in (R1, R2, Z1, Z2, N1, N2)}
    lane = np.arange(64)
    i, kk = lane & 15, lane >> 4
    rows = []

    def at(m, r, col):
        ok = (col >= 0) & (col < ncols[m])
        return np.where(ok, off[m] + r * ncols[m] + np.where(ok, col, 0), zero)

    def fwd(m, ob, col):                      # col [64]: input index per lane (outside the matrix -> the zero)
        return at(m, 16 * ob + i, col)

    def acc(s):                               # units of k-step s, per lane
        return 16 * (s >> 2) + 4 * kk + (s & 3)

    def tr(m, ob, s, col_of_feature=None):    # out row = feature 16 ob + i, k = unit acc(s)
        f = 16 * ob + i
        return at(m, acc(s), f if col_of_feature is None else col_of_feature(f))

    for s in range(44):                       # layer 1 of r and z: feature 44 kk + s
        col = 44 * kk + s
        rows += [fwd(R1, ob, col) for ob in range(4)] + [fwd(Z1, ob, col) for ob in range(4)]
    for s in range(16):                       # layer 2 of r and z
        rows += [fwd(R2, ob, acc(s)) for ob in range(4)] + [fwd(Z2, ob, acc(s)) for ob in range(4)]
    for s in range(16):                       # mlp_n layer 1: r * hid ...
        rows += [fwd(N1, ob, acc(s)) for ob in range(4)]
    for s in range(22):                       # ... then x | xe: row feature 88 + 22 kk + s = mlp_n input 64 + 22 kk + s
        rows += [fwd(N1, ob, 64 + 22 * kk + s) for ob in range(4)]
    for s in range(16):                       # mlp_n layer 2
        rows += [fwd(N2, ob, acc(s)) for ob in range(4)]
    assert len(rows) == 696
    if not forward_only:
        n1_from_cat = lambda f: np.where(f >= 88, f - 24, -1)
        for s in range(16):
            rows += [tr(N2, ob, s) for ob in range(4)]
        for s in range(16):
            rows += [tr(N1, ob, s) for ob in range(4)] + [tr(N1, ob, s, n1_from_cat) for ob in range(5, 11)]
        for s in range(16):
            rows += [tr(R2, ob, s) for ob in range(4)] + [tr(Z2, ob, s) for ob in range(4)]
        for s in range(16):
            rows += [tr(R1, ob, s) for ob in range(11)] + [tr(Z1, ob, s) for ob in range(11)]
        assert len(rows) == 1400
    c = _lib.lib().fs_ptf_gru_stream_chunk_rows()
    n_rows = (_lib.lib().fs_ptf_gru_table_rows() if forward_only else _lib.lib().fs_ptf_gru_stream_rows()) - 6
    if transposed_only:
        rows, n_rows = rows[696:], _lib.lib().fs_ptf_gru_stream_t_rows()
    assert n_rows % c == 0 and n_rows >= len(rows) and c % 16 == 0
    ops = np.full((n_rows, 64), zero, dtype=np.int64)
    ops[:len(rows)] = np.stack(rows)
    ops = ops.reshape(n_rows // c, 4, c // 16, 4, 64).transpose(0, 1, 2, 4, 3).reshape(n_rows, 64)
    if transposed_only:
        return ops.astype(np.int64)
    bias = np.stack([off[m] + lane for m in (BR1, BZ1, BR2, BZ2, BN1, BN2)])
    return np.concatenate([ops, bias]).astype(np.int64)


def _gru_operand_stream16(gru: "GRU", forward_only: bool = False, transposed_only: bool = False) -> Tensor:
    """Operand stream of csrc/ptf_gru.hip:ptf_gru_bwd16_kernel (fs_ptf_gru_stream_layout() = 2) or, forward_only, the tables of
    ptf_gru16_kernel (fs_ptf_gru_table_layout() = 1): one concatenation + one gather of the parameters through the index
    _gru_stream16_index describes (built once per device: a training loop rebuilds the stream after every optimizer step)."""
    params = _gru_params(gru)
    dev = params[0].device
    key = (str(dev), bool(forward_only), bool(transposed_only))
    idx = _stream16_index_cache.get(key)
    if idx is None:
        idx = _stream16_index_cache[key] = torch.from_numpy(_gru_stream16_index(forward_only, transposed_only)).to(dev)
    with torch.no_grad():
        flat = torch.cat([q.detach().float().reshape(-1) for q in params] + [torch.zeros(1, device=dev)])
        return flat[idx]


def gru_operand_stream(gru: "GRU") -> Tensor:
    """The operand rows of ptf_gru_bwd_kernel in the order in which it consumes them: the 696 rows of the forward
    (gru_tables, already in consumption order), then the transposed layers' rows (gru_tables_t) -- mlp_n second layer,
    mlp_n first layer, r/z second layers, r/z first layers --, padded to whole LDS chunks."""
    if _lib.lib().fs_ptf_gru_stream_layout() == 2:          # the 16-pair backward kernel's stream (built from the parameters)
        tab = gru_tables(gru)                               # (cached per parameter version: a new table = new parameters)
        hit = _stream_cache.get(gru)
        if hit is not None and hit[0] is tab:
            return hit[2]
        stream = _gru_operand_stream16(gru)
        _stream_cache[gru] = (tab, None, stream)
        return stream
    tab, tab_t = gru_tables(gru), gru_tables_t(gru)
    hit = _stream_cache.get(gru)
    if hit is not None and hit[0] is tab and hit[1] is tab_t:
        return hit[2]
    o = []
    for s in range(32):
        o += [_KTN2 + s, _KTN2 + 32 + s]
    for s in range(32):
        o += [_KTN1H + s, _KTN1H + 32 + s] + [_KTN1C + 32 * j + s for j in range(4)]
    for s in range(32):
        o += [_KTR2 + s, _KTR2 + 32 + s, _KTZ2 + s, _KTZ2 + 32 + s]
    for s in range(32):
        for rb in range(6):
            o += [_KTR1 + 32 * rb + s, _KTZ1 + 32 * rb + s]
    rows = _lib.lib().fs_ptf_gru_stream_rows()
    assert 696 + len(o) == 1464 <= rows
    stream = tab.new_zeros(rows, 64)
    stream[:696] = tab[:696]
    stream[696: 1464] = tab_t[torch.tensor(o, device=tab.device)]
    if _lib.lib().fs_ptf_gru_stream_layout() == 1:
        # interleaved by quads of rows (include/freesplat_amd.h): [chunk][owner wavefront][quad][lane][row of the quad] -- a lane's
        # four consecutive operand rows are one float4 in memory and in the kernel's LDS ring
        c = _lib.lib().fs_ptf_gru_stream_chunk_rows()
        assert rows % c == 0 and c % 16 == 0
        stream = stream.view(rows // c, 4, c // 16, 4, 64).permute(0, 1, 2, 4, 3).contiguous().view(rows, 64)
    _stream_cache[gru] = (tab, tab_t, stream)
    return stream


_stream_t_cache: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()


def save_gru_activations(V: int = 2, P: int = 0, device=None) -> bool:
    """Training folds keep the GRU's hidden activations, gates and gathered input rows (fs_ptf_fold_step_save) so that the
    backward runs the transposed layers only (fs_ptf_gru_backward_saved: 704 instead of 1 400 MFMAs per 16 pairs) -- 4 KB per
    possible fused pair and step held until the backward: (V - 1) * P * 4 KB, 10 GB for config 3's three views at 968x1296.
    FREESPLAT_GRU_SAVE=1 / 0 forces it on / off (A/B); otherwise it is on when that fits a quarter of the memory currently free
    on the device (a 288 GB MI355X: always; a small card falls back to the re-running backward instead of running out of memory).
    Unavailable with the 32-pair kernels."""
    if _lib.lib().fs_ptf_gru_stream_t_rows() <= 0:
        return False
    e = os.environ.get("FREESPLAT_GRU_SAVE")
    if e is not None:
        return e != "0"
    if P <= 0 or device is None:
        return True
    need = (V - 1) * P * 4 * (_lib.lib().fs_ptf_gru_side_cols() + _lib.lib().fs_ptf_gru_act_cols() + 176)
    try:
        free, _total = torch.cuda.mem_get_info(device)
    except Exception:
        return False
    return need <= free // 4


def gru_operand_stream_t(gru: "GRU") -> Tensor:
    """The transposed layers' operand rows alone (fs_ptf_gru_stream_t_rows() rows): the stream of fs_ptf_gru_backward_saved."""
    tab = gru_tables(gru)                                   # (cached per parameter version: a new table = new parameters)
    hit = _stream_t_cache.get(gru)
    if hit is not None and hit[0] is tab:
        return hit[1]
    stream = _gru_operand_stream16(gru, transposed_only=True)
    assert stream.shape[0] == _lib.lib().fs_ptf_gru_stream_t_rows()
    _stream_t_cache[gru] = (tab, stream)
    return stream


GRU_PARAM_SHAPES = [(64, 176), (64,), (64, 64), (64,), (64, 176), (64,), (64, 64), (64,), (64, 152), (64,), (64, 64), (64,)]


def gru_grad_views(flat: Tensor) -> list:
    """The 12 parameter gradients (order of _gru_params) as views of the flat buffer fs_ptf_gru_weight_grads adds to."""
    out, o = [], 0
    for shp in GRU_PARAM_SHAPES:
        k = math.prod(shp)
        out.append(flat[o: o + k].view(shp))
        o += k
    assert o == flat.numel()
    return out


def gru_backward(params: list, tables: Tensor, operand_stream: Tensor, cat: Tensor, g_fused: Tensor, grads: Tensor = None,
                 saved: tuple = None):
    """Backward of the GRU over n materialised input rows: fs_ptf_gru_backward (forward re-run + the six transposed
    layers on the matrix cores) gives dcat [n,176] and the per-pair factors of the weight gradients (`side`);
    fs_ptf_gru_weight_grads contracts those over the n pairs -- dW = dY^T X and the bias sums, two launches -- ADDING to
    the flat buffer `grads` (fs_ptf_gru_grad_floats() floats; a zeroed one is made if not given).
    saved = (side, act, stream_t) of a fold step run by fs_ptf_fold_step_save: the forward is not re-run (fs_ptf_gru_backward_saved
    fills the remaining columns of THAT side buffer).
    Returns (dcat, [12 parameter gradients in the order of _gru_params], views of `grads`)."""
    L = _lib.lib()
    p = _lib.ptr
    n = cat.shape[0]
    dev = cat.device
    if grads is None:
        grads = torch.zeros(L.fs_ptf_gru_grad_floats(), dtype=torch.float32, device=dev)
    dcat = torch.empty(n, 176, dtype=torch.float32, device=dev)
    g_fused = g_fused.contiguous()
    st = _lib.current_stream()
    if saved is not None:
        side, act, stream_t = saved
        _lib.check(L.fs_ptf_gru_backward_saved(n, p(cat), p(stream_t), p(act), p(g_fused), p(dcat), p(side), st),
                   "fs_ptf_gru_backward_saved")
    else:
        side = torch.empty(n, L.fs_ptf_gru_side_cols(), dtype=torch.float32, device=dev)
        _lib.check(L.fs_ptf_gru_backward(n, p(cat), p(tables), p(operand_stream), p(g_fused), p(dcat), p(side), st),
                   "fs_ptf_gru_backward")
    ws = torch.empty(L.fs_ptf_gru_weight_grads_bytes(n), dtype=torch.uint8, device=dev)
    _lib.check(L.fs_ptf_gru_weight_grads(n, p(cat), p(side), p(grads), p(ws), st), "fs_ptf_gru_weight_grads")
    return dcat, gru_grad_views(grads)


_fold_scratch: dict = {}       # (device, stream, V, h, w) -> the inference fold's internal scratch (reuse is ordered by the
                               # stream; two folds in flight on different streams must not share it; at most 8 shapes are kept)


def world_to_camera(Es: Tensor) -> Tensor:
    """[V,16] (or [V,4,4]) camera-to-world extrinsics -> [V,16] world-to-camera matrices, one launch (fs_invert_4x4: double
    precision inside, rounded once).  The reference calls `extrinsic.inverse()` per view (encoder_freesplat.py:455);
    torch's batched LU inverse costs ~0.11 ms of HOST time per fold -- a third of a 2-view call, which is host-bound.
    A pixel's round-half-even decision can hinge on the last bit of this matrix (about one in 10^6 projections): any
    two inverses -- the reference's cuSOLVER LU, a host LAPACK, this one -- agree to an ulp, not to the bit, which is why
    the full-size parity tests hand the oracle THESE matrices (tests/test_configs_4_5.py)."""
    V = Es.shape[0]
    src = Es.reshape(V, 16).contiguous()
    out = torch.empty_like(src)
    _lib.check(_lib.lib().fs_invert_4x4(V, _lib.ptr(src), _lib.ptr(out), _lib.current_stream()), "fs_invert_4x4")
    return out


def _f32c(t: Tensor) -> Tensor:
    return t if (t.dtype is torch.float32 and t.is_contiguous()) else t.float().contiguous()


def _fold_inputs(gaussians, coords, densities, weight_emb, depths, extrinsics, V, P, detach):
    """The reference's tensors (encoder_freesplat.py:431-440: latents [1,V,P,64], coords [1,V,P,srf,spp,3], densities /
    weights [1,V,P,srf,spp], depths [V,1,h,w], extrinsics [1,V,4,4]) as the fold's arrays lat [V,P,64], xs [V,P,3],
    rho / om / dep [V,P], Es [V,16] -- views when they already are fp32 and contiguous.  With one surface and one sample
    per pixel (what FreeSplat uses) `[0, :, :, 0, 0]` is a reshape: the five-index slice costs 18 us of host time each."""
    d = (lambda t: t.detach()) if detach else (lambda t: t)
    c, rh, om = coords[0], densities, weight_emb
    if (c.numel() == 3 * V * P and rh.numel() == V * P and om.numel() == V * P and c.is_contiguous() and rh.is_contiguous()
            and om.is_contiguous()):
        xs, rho, om = c.view(V, P, 3), rh.view(V, P), om.view(V, P)
    else:
        xs, rho, om = c[0, :, :, 0, 0], rh[0, :, :, 0, 0], om[0, :, :, 0, 0]
    g0 = gaussians[0]
    lat = g0.reshape(V, P, g0.shape[-1]) if g0.shape[0] == 1 else g0[0]      # (a reshape's backward is a view; a select's fills + copies)
    return (_f32c(d(lat)), _f32c(d(xs)), _f32c(d(rho)), _f32c(d(om)), _f32c(d(depths.reshape(V, -1))),
            _f32c(extrinsics[0].detach()).reshape(V, 16))


def _fuse_gaussians_fused(gru, gaussians, coords, densities, weight_emb, depths, extrinsics, intrinsics, image_shape,
                          depth_thres):
    """Inference path (no autograd): ONE library call (fs_ptf_fold) folds all views -- per view match -> GRU inputs ->
    GRU on the fp32 matrix cores -> next state, with every data-dependent size kept on the device -- and the host
    syncs once afterwards, for the final number of Gaussians (the reference syncs four times per view).
    Same results and order as the differentiable path below."""
    L = _lib.lib()
    p = _lib.ptr
    h, w = image_shape
    V = gaussians[0].shape[1]
    P = h * w
    # (no detach: this path runs without autograd -- fuse_gaussians checked -- and hands raw pointers to the library)
    lat, xs, rho, om, dep, Es = _fold_inputs(gaussians, coords, densities, weight_emb, depths, extrinsics, V, P, detach=False)
    dev = lat.device
    if V == 1:
        return lat[:1].detach(), xs[:1].detach(), Es[0].reshape(1, 1, 4, 4).repeat(1, P, 1, 1), dep[:1].detach()
    tables = gru_tables(gru)
    # the state after view 0 is view 0 itself; fs_ptf_fold folds views 1 .. V-1 into it, writing the successive states
    # alternately into two sets of buffers -- one library call (camera constants included), one host sync afterwards
    Kn = _f32c(intrinsics[0]).reshape(V, 9)
    rows = 2 * P if V == 2 else V * P
    # The call is HOST-bound at two views (0.19 ms of kernels; profiles/r4_ptf_call_breakdown.txt: 85 us of Python ran
    # BEFORE the first launch): one allocation for the state buffers, the six arrays of each set addressed by pointer
    # arithmetic for the call -- the tensor views of the final state are made AFTER the launches, while the GPU works --
    # and the fold's internal scratch is kept per (device, stream, V, h, w): it is dead when the call returns (the count
    # read-back below waits for the fold).
    widths = (64, 3, 1, 1, 16, 1)
    n_sets = 1 if V == 2 else 2
    big = torch.empty(n_sets * rows * 86, device=dev)
    base = big.data_ptr()
    ptrs = []
    for k in range(n_sets):
        arr, off = (C.c_void_p * 6)(), k * rows * 86
        for q, n_ in enumerate(widths):
            arr[q] = base + 4 * off
            off += rows * n_
        ptrs.append(arr)
    counts = torch.empty(V, 4, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    key = (dev, stream, V, h, w)
    scratch = _fold_scratch.get(key)
    if scratch is None:
        while len(_fold_scratch) >= 8:                          # (oldest entry first: dicts keep insertion order)
            _fold_scratch.pop(next(iter(_fold_scratch)))
        scratch = _fold_scratch[key] = torch.empty(L.fs_ptf_fold_bytes(V, h, w), dtype=torch.uint8, device=dev)
    # (w2c = NULL: the library inverts the extrinsics itself, with the kernel world_to_camera() uses -- one call less)
    _lib.check(L.fs_ptf_fold(V, h, w, p(lat), p(xs), p(rho), p(om), p(dep), p(Es), None, p(Kn), C.c_float(depth_thres),
                             p(tables), p(scratch), ptrs[0], ptrs[-1], p(counts), C.c_void_p(stream)), "fs_ptf_fold")
    global LAST_FOLD_COUNTS
    LAST_FOLD_COUNTS = counts
    last = counts[V - 1, 3]
    off = 0 if ((V - 1) & 1 or V == 2) else rows * 86        # the set that holds the final state
    G = big[off: off + rows * 64].view(rows, 64)
    X = big[off + rows * 64: off + rows * 67].view(rows, 3)
    E = big[off + rows * 69: off + rows * 85].view(rows, 4, 4)
    D = big[off + rows * 85: off + rows * 86]
    n = int(last.item())                           # the only host sync of the fold
    return G[None, :n], X[None, :n], E[None, :n], D[None, :n]


def _gru_params(gru: "GRU") -> list:
    """[mlp_r[0].weight, .bias, mlp_r[2].weight, .bias, mlp_z .., mlp_n ..] -- through the modules' dicts: as attribute
    chains (gru.mlp_r[0].weight: Module.__getattr__ + Sequential.__getitem__) the twelve look-ups cost 24 us of a 2-view
    call that is host-bound."""
    out = []
    for name in ("mlp_r", "mlp_z", "mlp_n"):
        layers = gru._modules[name]._modules
        for k in ("0", "2"):
            pr = layers[k]._parameters
            out.append(pr["weight"])
            out.append(pr["bias"])
    return out


class _PtfFold(torch.autograd.Function):
    """Differentiable fold of V views (b = 1) on the HIP kernels.  Inputs: lat [V,P,64], xs [V,P,3], rho / om / dep
    [V,P], Es [V,16] and Kn [V,9] (no gradient), then the 12 GRU parameters.  Outputs: G [n,64], X [n,3], E [n,16],
    D [n]."""

    @staticmethod
    def forward(ctx, lat, xs, rho, om, dep, Es, Kn, h, w, depth_thres, tables, operand_stream, stream_t, *params):
        L = _lib.lib()
        p = _lib.ptr
        V, P = lat.shape[0], lat.shape[1]
        dev = lat.device
        w2c = world_to_camera(Es)
        kpix = torch.empty(V, 4, dtype=torch.float32, device=dev)
        E0 = torch.empty(P, 16, dtype=torch.float32, device=dev)
        _lib.check(L.fs_ptf_cameras(V, h, w, p(Es), p(Kn), p(kpix), p(E0), _lib.current_stream()), "fs_ptf_cameras")
        counts = torch.empty(V, 4, dtype=torch.int32, device=dev)
        state = (lat[0], xs[0], rho[0], om[0], E0, dep[0])         # G, X, R, O, E, D of the state after view 0
        states, scratches, saves = [state], [None], [None]
        for i in range(1, V):
            M_max = i * P
            rows = M_max + P
            out = tuple(torch.empty(rows, n, device=dev) for n in (64, 3, 1, 1, 16, 1))
            scratch = torch.empty(L.fs_ptf_fold_scratch_bytes(M_max, h, w), dtype=torch.uint8, device=dev)
            G, X, R, O, E, D = state
            args = (M_max, None if i == 1 else p(counts[i - 1, 3:]), h, w, p(G), p(X), p(R), p(O), p(E), p(D),
                    p(lat[i]), p(xs[i]), p(rho[i]), p(om[i]), p(dep[i]), p(Es[i]), p(w2c[i]), p(kpix[i]),
                    C.c_float(depth_thres), p(tables), p(scratch), *[p(t) for t in out], p(counts[i]))
            if stream_t is not None:
                # the GRU leaves its hidden activations (side columns 6 .. 9) and gates for the backward: one row per fused pair,
                # at most min(M_max, P) of them (the step's fuse list has that many slots)
                nf_max = min(M_max, P)
                side = torch.empty(nf_max, L.fs_ptf_gru_side_cols(), dtype=torch.float32, device=dev)
                act = torch.empty(-(-nf_max // 16) * 16, L.fs_ptf_gru_act_cols(), dtype=torch.float32, device=dev)   # (whole 16-pair groups)
                cat = torch.empty(nf_max, 176, dtype=torch.float32, device=dev)
                _lib.check(L.fs_ptf_fold_step_save(*args, p(side), p(act), p(cat), _lib.current_stream()), "fs_ptf_fold_step_save")
                saves.append((side, act, cat))
            else:
                _lib.check(L.fs_ptf_fold_step(*args, _lib.current_stream()), "fs_ptf_fold_step")
                saves.append(None)
            state = out
            states.append(out)
            scratches.append(scratch)
        global LAST_FOLD_COUNTS
        LAST_FOLD_COUNTS = counts
        cnt = counts.tolist()                          # the only host sync of the fold
        cnt[0] = [P, 0, 0, P]
        n = cnt[V - 1][3]
        # The steps were queued into WORST-CASE buffers ((i + 1) P rows of 86 floats each: O(V^2 P) in total, 2.4 GB at
        # V = 8 and 384x512) because their sizes were still on the device.  Now that the counts are known, keep only the
        # rows that exist (one copy of sum_i n_i rows, ~0.3 ms at 10 views): what the backward holds on to is O(V n).
        # The LAST state is only returned (the backward reads states 0 .. V-2): its G, X, E, D stay views of the worst-case
        # buffer unless that would pin more than 256 MB of rows that do not exist.
        # ... unless the worst-case buffers are small next to the device's memory (_keep_bytes: FREESPLAT_PTF_KEEP_BYTES, default 1/16 of the free device memory up to 8 GiB: config 3's
        # three views at 968x1296 queue 2.2 GB): then nothing is copied -- the trims were 0.8 ms of rocclr copies per config-3
        # training step (profiles/r5_c3_step_glue.json) for memory a 288 GB device does not miss.
        worst = sum((i + 1) * P for i in range(1, V)) * 86 * 4
        trim = worst > _keep_bytes(lat.device)
        for i in range(1, V - 1):
            states[i] = tuple((t[: cnt[i][3]].clone() if trim else t[: cnt[i][3]]) for t in states[i])
        G, X, R, O, E, D = states[V - 1]
        if trim and (V * P - n) * 84 * 4 > (256 << 20):
            G, X, E, D = (t[:n].clone() for t in (G, X, E, D))
        states[V - 1] = None
        ctx.cnt, ctx.hw, ctx.states, ctx.scratches = cnt, (h, w), states, scratches
        ctx.tables, ctx.operand_stream, ctx.stream_t, ctx.saves = tables, operand_stream, stream_t, saves
        ctx.save_for_backward(lat, xs, rho, om, dep, Es, *params)
        return G[:n], X[:n], E[:n], D[:n, 0]

    @staticmethod
    def backward(ctx, gG, gX, gE, gD):
        lat, xs, rho, om, dep, Es, *params = ctx.saved_tensors
        L = _lib.lib()
        p = _lib.ptr
        h, w = ctx.hw
        V, P = lat.shape[0], lat.shape[1]
        dev = lat.device
        cnt = ctx.cnt
        n = cnt[V - 1][3]
        z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=dev)
        # the five view gradients are accumulated into: one zero fill for all of them
        seg = [-(-V * P * k // 64) * 64 for k in (64, 3, 1, 1, 1)]          # 256-byte aligned segments
        flat = z(sum(seg)).split(seg)
        g_lat, g_xs = flat[0][: V * P * 64].view(V, P, 64), flat[1][: V * P * 3].view(V, P, 3)
        g_rho, g_om, g_dep = (t[: V * P].view(V, P) for t in flat[2:])
        g_params, g_flat = [None] * len(params), None       # the fold steps' weight gradients accumulate in g_flat
        # gradient of the current out state: G, X, R, O, E, D (None = zero)
        c = lambda t: None if t is None else t.contiguous()
        g_out = [c(gG), c(gX), None, None, c(gE), None if gD is None else gD.contiguous().view(n, 1)]
        vp = lambda ts: (C.c_void_p * 6)(*[None if t is None else t.data_ptr() for t in ts])
        for i in range(V - 1, 0, -1):
            nk, nf, na, _ = cnt[i]
            M_in = cnt[i - 1][3]
            G, X, R, O, E, D = ctx.states[i - 1]
            lists = (C.c_void_p * 4)()
            _lib.check(L.fs_ptf_fold_step_lists(i * P, h, w, p(ctx.scratches[i]), lists), "fs_ptf_fold_step_lists")
            keep, fuse, fpix, app = (C.c_void_p(v) for v in lists)
            if i == 1:
                # the state before view 1 IS view 0 (every row of it kept or fused): its gradient is written straight
                # into view 0's slices (the extrinsics' gradient goes nowhere)
                g_in = [g_lat[0], g_xs[0], g_rho[0].view(P, 1), g_om[0].view(P, 1),
                        torch.empty(M_in, 16, dtype=torch.float32, device=dev), g_dep[0].view(P, 1)]
            else:
                g_in = [torch.empty(M_in, k, dtype=torch.float32, device=dev) for k in (64, 3, 1, 1, 16, 1)]
            _lib.check(L.fs_ptf_write_state_backward(
                nk, nf, na, keep, fuse, fpix, app, p(X), p(R), p(E), p(D), p(xs[i]), p(rho[i]), p(dep[i]), p(Es[i]),
                vp(g_out), vp(g_in), p(g_lat[i]), p(g_xs[i]), p(g_rho[i]), p(g_om[i]), p(g_dep[i]),
                _lib.current_stream()), "fs_ptf_write_state_backward")
            if nf > 0:
                # the GRU rows: re-gather their inputs (HIP), GRU backward on the matrix cores (+ the weight-gradient GEMMs)
                sv = ctx.saves[i]
                if sv is not None:
                    cat = sv[2][:nf]                       # (the forward kept the gathered + encoded rows)
                else:
                    cat = torch.empty(nf, 176, dtype=torch.float32, device=dev)
                    _lib.check(L.fs_ptf_gru_inputs(nf, fuse, fpix, p(G), p(R), p(O), p(lat[i]), p(rho[i]), p(om[i]), p(cat),
                                                   _lib.current_stream()), "fs_ptf_gru_inputs")
                g_fused = g_out[0][nk: nk + nf] if g_out[0] is not None else z(nf, 64)
                if g_flat is None:
                    g_flat = z(L.fs_ptf_gru_grad_floats())
                dcat, g_params = gru_backward(params, ctx.tables, ctx.operand_stream, cat, g_fused, g_flat,
                                              saved=None if sv is None else (sv[0], sv[1], ctx.stream_t))
                ctx.saves[i] = None
                _lib.check(L.fs_ptf_gru_inputs_backward(nf, fuse, fpix, p(R), p(O), p(rho[i]), p(om[i]), p(dcat),
                                                        p(g_in[0]), p(g_in[2]), p(g_in[3]), p(g_lat[i]), p(g_rho[i]),
                                                        p(g_om[i]), _lib.current_stream()), "fs_ptf_gru_inputs_backward")
            g_out = g_in
        need = ctx.needs_input_grad
        pick = lambda k, t: t if need[k] else None
        return (pick(0, g_lat), pick(1, g_xs), pick(2, g_rho), pick(3, g_om), pick(4, g_dep), None, None, None, None,
                None, None, None, None) + tuple(g if need[13 + k] else None for k, g in enumerate(g_params))


def _fuse_gaussians_train(gru, gaussians, coords, densities, weight_emb, depths, extrinsics, intrinsics, image_shape,
                          depth_thres):
    """Training path (autograd): the same HIP fold as inference, with a HIP backward (_PtfFold)."""
    h, w = image_shape
    V = gaussians[0].shape[1]
    f32 = _f32c
    P = h * w
    lat, xs, rho, om, dep, Es = _fold_inputs(gaussians, coords, densities, weight_emb, depths, extrinsics, V, P, detach=False)
    if V == 1:
        return lat[:1], xs[:1], Es[0].reshape(1, 1, 4, 4).repeat(1, P, 1, 1), dep[:1]
    Kn = f32(intrinsics[0].detach()).reshape(V, 9)
    G, X, E, D = _PtfFold.apply(lat, xs, rho, om, dep, Es, Kn, h, w, float(depth_thres), gru_tables(gru),
                                gru_operand_stream(gru), gru_operand_stream_t(gru) if save_gru_activations(V, P, lat.device) else None,
                                *_gru_params(gru))
    n = G.shape[0]
    return G[None], X[None], E.view(1, n, 4, 4), D[None]


def fuse_gaussians(self, gaussians, coords, densities, weight_emb, depths, extrinsics, intrinsics, image_shape,
                   depth_thres=0.1):
    """Same contract as EncoderFreeSplat.fuse_gaussians (encoder_freesplat.py:431-522); `self` only
    needs a `.gru` attribute.  gaussians = [latents [1,V,P,64]], coords = [[1,V,P,1,1,3]],
    densities / weight_emb [1,V,P,1,1], depths [V,1,h,w], extrinsics [1,V,4,4], intrinsics [1,V,3,3].
    Returns (latents [1,M,64], xyz [1,M,3], extrinsics [1,M,4,4], depths [1,M]).
    The fold runs on the HIP kernels with or without autograd (module docstring)."""
    needs_grad = torch.is_grad_enabled() and (
        any(t.requires_grad for t in (gaussians[0], coords[0], densities, weight_emb, depths))
        or any(q.requires_grad for q in self.gru.parameters()))
    if gaussians[0].shape[0] != 1:
        # the reference folds one scene per call (encoder_freesplat.py:355-368 slices x[b:b+1] for every b); with b > 1
        # its body would apply scene 0's match lists to every scene's rows -- no caller does that
        raise NotImplementedError("fuse_gaussians folds ONE scene per call (pass x[b:b+1] slices, as "
                                  "EncoderFreeSplat.forward does); got b = %d" % gaussians[0].shape[0])
    fn = _fuse_gaussians_train if needs_grad else _fuse_gaussians_fused
    return fn(self.gru, gaussians, coords, densities, weight_emb, depths, extrinsics, intrinsics, image_shape, depth_thres)


class PixelwiseTripletFusion(nn.Module):
    """Owner of the `gru` sub-module with the reference's `fuse_gaussians` method bound to it, so that
    `encoder.gru.*` state-dict keys and `encoder.fuse_gaussians(...)` call sites carry over."""

    def __init__(self):
        super().__init__()
        self.gru = GRU()

    fuse_gaussians = fuse_gaussians

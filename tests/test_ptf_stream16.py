"""CPU model of how csrc/ptf_gru.hip's 16-pair kernels (ptf_gru16_kernel, ptf_gru_bwd16_kernel) consume the operand stream that
freesplat_amd/ptf.py builds for fs_ptf_gru_stream_layout() = 2 / fs_ptf_gru_table_layout() = 1: one row of 64 lanes per
v_mfma_f32_16x16x4_f32, lane (i = l & 15, kk = l >> 4) = A[i][kk]; the B operand of k-step s is register s of the consumer's
accumulator layout (unit 16 (s >> 2) + 4 kk + (s & 3) of quarter kk; for the input row: feature 44 kk + s, for mlp_n's x | xe part
feature 88 + 22 kk + s).  Walking the stream in order with numpy must give the reference GRU's linear layers (networks.py:188-214)
and, in the second half, their transposes.  The GPU tests check the kernels' values; this one the table bookkeeping, and that the
stream is ONE gather of the parameters (a training loop rebuilds it after every optimizer step)."""
import numpy as np
import torch

from freesplat_amd import _lib
from freesplat_amd import ptf as P


def _rows(stream, n_rows):
    """undo the quad interleave: [chunk][owner][quad][lane][row of quad] -> rows of 64 lanes"""
    c = _lib.lib().fs_ptf_gru_stream_chunk_rows()
    body = stream[:n_rows].reshape(n_rows // c, 4, c // 16, 64, 4).transpose(0, 1, 2, 4, 3).reshape(n_rows, 64)
    return body, stream[n_rows:]


def _mfma(acc, a_row, b):
    """acc[16 outputs] += A[i][kk] * B[kk] for one pair: a_row [64] lane-ordered, b [4] per quarter"""
    return acc + (a_row.reshape(4, 16) * b[:, None]).sum(0)


def test_stream16_rows_are_the_layers_in_consumption_order():
    torch.manual_seed(3)
    gru = P.GRU()
    with torch.no_grad():
        for q in gru.parameters():
            q.normal_()
    Wr1, br1, Wr2, br2, Wz1, bz1, Wz2, bz2, Wn1, bn1, Wn2, bn2 = [q.detach().double().numpy() for q in P._gru_params(gru)]
    lib = _lib.lib()
    assert lib.fs_ptf_gru_stream_layout() == 2 and lib.fs_ptf_gru_table_layout() == 1
    stream = P._gru_operand_stream16(gru).double().numpy()
    table = P._gru_operand_stream16(gru, forward_only=True).double().numpy()
    assert stream.shape == (lib.fs_ptf_gru_stream_rows(), 64) and table.shape == (lib.fs_ptf_gru_table_rows(), 64)
    rows, bias = _rows(stream, lib.fs_ptf_gru_stream_rows() - 6)
    trow, tbias = _rows(table, lib.fs_ptf_gru_table_rows() - 6)
    assert np.array_equal(rows[:696], trow[:696]) and np.array_equal(bias, tbias)
    assert np.array_equal(bias, np.stack([br1, bz1, br2, bz2, bn1, bn2]))
    assert not rows[1400:].any() and not trow[696:].any()
    # the transposed rows alone (fs_ptf_gru_backward_saved: the training forward kept its activations): exactly 11 chunks, no bias rows
    tt = P._gru_operand_stream16(gru, transposed_only=True).double().numpy()
    assert tt.shape == (lib.fs_ptf_gru_stream_t_rows(), 64) == (704, 64)
    assert np.array_equal(_rows(tt, 704)[0], rows[696:1400])

    rng = np.random.default_rng(0)
    kk = np.arange(4)
    unit = lambda s: 16 * (s >> 2) + 4 * kk + (s & 3)
    x = rng.standard_normal(176)            # one pair's input row: hid | (x | xe) ...
    pos = 0

    def layer(n_steps, n_blocks_per_step, b_of_step, take):
        """n_steps k-steps of n_blocks_per_step rows each; returns the blocks `take` picks, each [16 outputs] accumulated"""
        nonlocal pos
        out = np.zeros((n_blocks_per_step, 16))
        for s in range(n_steps):
            b = b_of_step(s)
            for j in range(n_blocks_per_step):
                out[j] = _mfma(out[j], rows[pos], b)
                pos += 1
        return out[take].reshape(-1) if take is not None else out

    # forward: layer 1 of r and z (8 rows per k-step: 4 blocks of r, 4 of z), B = feature 44 kk + s
    o = layer(44, 8, lambda s: x[44 * kk + s], None)
    assert np.allclose(o[:4].reshape(-1), Wr1 @ x) and np.allclose(o[4:].reshape(-1), Wz1 @ x)
    h = rng.standard_normal(64)
    o = layer(16, 8, lambda s: h[unit(s)], None)
    assert np.allclose(o[:4].reshape(-1), Wr2 @ h) and np.allclose(o[4:].reshape(-1), Wz2 @ h)
    # mlp_n layer 1: 16 k-steps over r * hid, then 22 over x | xe = features 88 .. 175 of the row
    o1 = layer(16, 4, lambda s: h[unit(s)], None)
    o2 = layer(22, 4, lambda s: x[88 + 22 * kk + s], None)
    assert np.allclose((o1 + o2).reshape(-1), Wn1 @ np.concatenate([h, x[88:]]))
    assert np.allclose(layer(16, 4, lambda s: h[unit(s)], None).reshape(-1), Wn2 @ h)
    assert pos == 696
    # transposed layers: B = dY's register s, outputs = feature blocks
    d = rng.standard_normal(64)
    assert np.allclose(layer(16, 4, lambda s: d[unit(s)], None).reshape(-1), Wn2.T @ d)
    o = layer(16, 10, lambda s: d[unit(s)], None)
    full = Wn1.T @ d                                          # [152]: r * hid units, then x | xe
    assert np.allclose(o[:4].reshape(-1), full[:64])
    got = o[4:].reshape(-1)                                   # dcat feature blocks 5 .. 10 = features 80 .. 175
    want = np.concatenate([np.zeros(8), full[64:]])           # (features 80 .. 87 are not inputs of mlp_n)
    assert np.allclose(got, want)
    o = layer(16, 8, lambda s: d[unit(s)], None)
    assert np.allclose(o[:4].reshape(-1), Wr2.T @ d) and np.allclose(o[4:].reshape(-1), Wz2.T @ d)
    o = layer(16, 22, lambda s: d[unit(s)], None)
    assert np.allclose(o[:11].reshape(-1), Wr1.T @ d) and np.allclose(o[11:].reshape(-1), Wz1.T @ d)
    assert pos == 1400


def test_stream16_is_one_gather_of_the_parameters():
    gru = P.GRU()
    P._gru_operand_stream16(gru)                               # (index built and cached)
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        P._gru_operand_stream16(gru)
    ops = [e.key for e in prof.key_averages() if e.key in ("aten::index", "aten::cat")]
    n_index = sum(e.count for e in prof.key_averages() if e.key == "aten::index")
    assert n_index == 1 and "aten::cat" in ops

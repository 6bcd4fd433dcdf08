"""CPU model of the index map of fs_ptf_gru_weight_grads (csrc/ptf_gru.hip: DwJob* tables, dw_out_index, dw_bias_index): the
46 MFMA tiles a workgroup accumulates -- with the interleaved column maps its wide loads give them -- and its 12 bias rows must
cover every one of the 44 928 parameter-gradient floats EXACTLY ONCE, and every tile element must pair the dY column and the X
column that the parameter's (unit, feature) says.  The tables below restate the kernel's (segment, tile) descriptors; the GPU
tests check the values, this one the bookkeeping."""
import numpy as np

# parameter layout (fs_ptf_gru_grad_floats): r1.W [64,176], r1.b, r2.W [64,64], r2.b, z1.W, z1.b, z2.W, z2.b, n1.W [64,152], n1.b, n2.W, n2.b
SHAPES = [(64, 176), (64,), (64, 64), (64,), (64, 176), (64,), (64, 64), (64,), (64, 152), (64,), (64, 64), (64,)]
OFF = np.cumsum([0] + [int(np.prod(s)) for s in SHAPES])
R1W, R1B, R2W, R2B, Z1W, Z1B, Z2W, Z2B, N1W, N1B, N2W, N2B = OFF[:12]

# side blocks (64 columns each): 0 dr1, 1 dz1, 2 dR, 3 dZ, 4 dn1, 5 dN, 6 relu(r1), 7 relu(z1), 8 relu(n1), 9 r*hid; cat: 176 columns
# a load segment: (base, col, width, lim); a B tile: (register, out col, out mul, lim); a product: (a0, [tiles], W offset, ldw, bias offset)
JOBS = {
    "R1": dict(seg=[("side", 0, 2, 32), ("cat", 0, 4, 32), ("cat", 128, 2, 24)],
               prod=[(0, [(2, 0, 4, 32), (3, 1, 4, 32), (4, 2, 4, 32), (5, 3, 4, 32), (6, 128, 2, 24), (7, 129, 2, 24)], R1W, 176, R1B)]),
    "Z1": dict(seg=[("side", 64, 2, 32), ("cat", 0, 4, 32), ("cat", 128, 2, 24)],
               prod=[(0, [(2, 0, 4, 32), (3, 1, 4, 32), (4, 2, 4, 32), (5, 3, 4, 32), (6, 128, 2, 24), (7, 129, 2, 24)], Z1W, 176, Z1B)]),
    "N1": dict(seg=[("side", 256, 2, 32), ("side", 576, 2, 32), ("cat", 88, 2, 32), ("cat", 152, 1, 24)],
               prod=[(0, [(2, 0, 2, 32), (3, 1, 2, 32), (4, 64, 2, 32), (5, 65, 2, 32), (6, 128, 1, 24)], N1W, 152, N1B)]),
    "L2": dict(seg=[("side", 128, 2, 32), ("side", 384, 2, 32), ("side", 192, 2, 32), ("side", 448, 2, 32), ("side", 320, 2, 32),
                    ("side", 512, 2, 32)],
               prod=[(0, [(2, 0, 2, 32), (3, 1, 2, 32)], R2W, 64, R2B), (4, [(6, 0, 2, 32), (7, 1, 2, 32)], Z2W, 64, Z2B),
                     (8, [(10, 0, 2, 32), (11, 1, 2, 32)], N2W, 64, N2B)]),
}
# which (dY block, X source) a parameter's gradient contracts: W[unit, f] = sum_pairs dY[unit] * X[f]
EXPECT = {R1W: (0, lambda f: ("cat", f)), Z1W: (1, lambda f: ("cat", f)),
          N1W: (4, lambda f: ("side", 576 + f) if f < 64 else ("cat", 88 + f - 64)),
          R2W: (2, lambda f: ("side", 384 + f)), Z2W: (3, lambda f: ("side", 448 + f)), N2W: (5, lambda f: ("side", 512 + f))}


def _register_columns(seg):
    """register index -> function lane j -> (base, column) of the k-step's value set (None where the lane's column does not exist)"""
    regs = []
    for base, col, width, lim in seg:
        for e in range(width):
            regs.append((base, col, width, lim, e))
    return regs


def test_tiles_and_bias_rows_cover_every_parameter_gradient_once():
    hits = np.zeros(OFF[-1], dtype=np.int32)
    tiles = 0
    for name, job in JOBS.items():
        regs = _register_columns(job["seg"])
        for a0, btiles, woff, ldw, boff in job["prod"]:
            blk, xsrc = EXPECT[woff]
            for ia in range(2):
                abase, acol, awidth, alim, ae = regs[a0 + ia]
                assert abase == "side" and awidth == 2 and ae == ia and acol == 64 * blk, name
                # bias row: lane j < 32 holds unit 2 j + ia
                for j in range(32):
                    hits[boff + 2 * j + ia] += 1
                for (vreg, ocol, omul, lim) in btiles:
                    tiles += 1
                    bbase, bcol, bwidth, blim, be = regs[vreg]
                    assert blim == lim and omul == bwidth, (name, vreg)
                    for q in range(16):
                        for lane in range(64):
                            j, hf = lane & 31, lane >> 5
                            if j >= lim:
                                continue
                            i = 8 * (q >> 2) + 4 * hf + (q & 3)            # MFMA accumulator row of (q, lane)
                            unit = 2 * i + ia                               # A tile ia holds dY columns 2 i + ia
                            f_out = ocol + omul * j
                            hits[woff + unit * ldw + f_out] += 1
                            # the X column lane j feeds into this tile must be the feature the parameter pairs with `f_out`
                            assert (bbase, bcol + bwidth * j + be) == xsrc(f_out), (name, vreg, j)
    assert tiles == 46
    assert hits.min() == 1 and hits.max() == 1, (int((hits == 0).sum()), int((hits > 1).sum()))

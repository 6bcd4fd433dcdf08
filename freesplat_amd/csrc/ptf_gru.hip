// ptf_gru.hip -- the GRU of Pixel-wise Triplet Fusion on the fp32 matrix cores (gfx950).
//
// Replaces, on the inference path, the three 2-layer MLPs + gates of
// src/model/encoder/modules/networks.py:188-214 applied to the fused pairs of one fold step
// (src/model/encoder/encoder_freesplat.py:487-490): 44.5 kMAC per pair, i.e. the only dense
// contraction of PTF.  Input = the concatenated rows built by fs_ptf_gru_inputs,
//     cat[t] = [hid(64) | he(24) | x(64) | xe(24)]            (176 floats),
// output fused[t] = (1 - z) * hid + z * tanh(mlp_n([r * hid | x | xe])),  r, z = sigmoid(mlp_{r,z}(cat)).
//
// One wavefront owns 32 pairs; lane l = (pair l & 31, half l >> 5).  Everything is computed TRANSPOSED
// (H^T = W X^T) with v_mfma_f32_32x32x2_f32 so that a lane always holds values of ITS OWN pair:
//   * layer-1 of r and z share the B operand (the pair's input), 4 accumulators (2 row blocks x {r, z});
//     the K order is permuted so that half h of the lanes consumes the contiguous half row
//     cat[88h .. 88h+88) -- 22 dwordx4 loads instead of 88 strided scalars;
//   * every later layer takes its B operand straight from the previous accumulator registers (k order =
//     accumulator row map), so there is no cross-lane traffic anywhere;
//   * A operands come from weight tables pre-arranged in operand order ([step][64 lanes], one coalesced
//     256-byte load per MFMA, L2-resident: 178 KB for all six matrices).
// 696 MFMAs (exact fp32) per 32 pairs.
#include "fs_common.h"

namespace fs {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// table layout (floats, each row = 64 lanes):
//   [r1: 2*88][z1: 2*88][r2: 2*32][z2: 2*32][n1: 2*76][n2: 2*32] rows of A operands, then bias rows
//   [b_r1: 2*16][b_z1: 2*16][b_r2: 2*16][b_z2: 2*16][b_n1: 2*16][b_n2: 2*16]
constexpr int kR1 = 0, kZ1 = kR1 + 2 * 88, kR2 = kZ1 + 2 * 88, kZ2 = kR2 + 2 * 32, kN1 = kZ2 + 2 * 32,
              kN2 = kN1 + 2 * 76, kBias = kN2 + 2 * 32, kRows = kBias + 6 * 32;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

#define FS_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// GATHER: the input row [hid | he | x | xe] of pair t is not read from a materialised `cat` array but assembled here:
// hid = G[fuse_idx[t]], x = g_i[fuse_pix[t]] (the half of the lanes that owns them loads them), he / xe = the positional
// encodings of the densities and weights (encoder_freesplat.py:485-486) -- 24 sin/cos per lane, every lane busy, instead
// of a separate kernel in which 2 lanes of 16 did them and 141 MB of rows went to HBM and back per 10^5 pairs.
struct GruGather {
    const long long *fuse_idx, *fuse_pix;
    const float *G, *R, *O, *g_i, *rho_i, *om_i;
};
template <bool GATHER>
__global__ __launch_bounds__(256) void ptf_gru_kernel(int n, const int32_t* __restrict__ counts,
                                                      const float* __restrict__ cat, GruGather ga,
                                                      const float* __restrict__ tab, float* __restrict__ fused)
{
    if (counts) n = counts[1];  // (device-resident pair count: fs_ptf_fold_step)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int grp = blockIdx.x * 4 + wave;
    if (grp * 32 >= n) return;
    const int p = lane & 31, hf = lane >> 5;
    const int t = grp * 32 + p;
    const bool live = t < n;
    const float* T = tab + lane;  // T[r * 64] = row r of the tables for this lane
    // sources of this pair: `row` = a materialised row, or (GATHER) the state latent / the view latent
    const float* row = GATHER ? nullptr : cat + (size_t)(live ? t : 0) * 176;
    const long long gm = GATHER ? ga.fuse_idx[live ? t : 0] : 0, gp = GATHER ? ga.fuse_pix[live ? t : 0] : 0;
    const float* hrow = GATHER ? ga.G + gm * 64 : row;          // hid: 64 floats
    const float* xrow = GATHER ? ga.g_i + gp * 64 : row + 88;   // x: 64 floats (xe follows only in a materialised row)

    // this half's contiguous half row, and hid in accumulator-row order (units acc rows of this half)
    float xh[88];
    if (GATHER) {
        const float* src = hf ? xrow : hrow;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float4 v = ((const float4*)src)[k];
            xh[4 * k] = v.x; xh[4 * k + 1] = v.y; xh[4 * k + 2] = v.z; xh[4 * k + 3] = v.w;
        }
        // half 0: he = PE(rho_i[p], O[m]); half 1: xe = PE(R[m], om_i[p])
        pos_enc2(hf ? ga.R[gm] : ga.rho_i[gp], hf ? ga.om_i[gp] : ga.O[gm], xh + 64);
    } else {
#pragma unroll
        for (int k = 0; k < 22; ++k) {
            const float4 v = ((const float4*)(row + 88 * hf))[k];
            xh[4 * k] = v.x; xh[4 * k + 1] = v.y; xh[4 * k + 2] = v.z; xh[4 * k + 3] = v.w;
        }
    }
    float hid[32];  // hid[16*blk + q] = unit (q&3) + 8*(q>>2) + 4*hf + 32*blk
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4 v = *(const float4*)(hrow + 32 * blk + 8 * g4 + 4 * hf);
            hid[16 * blk + 4 * g4] = v.x; hid[16 * blk + 4 * g4 + 1] = v.y;
            hid[16 * blk + 4 * g4 + 2] = v.z; hid[16 * blk + 4 * g4 + 3] = v.w;
        }

    f32x16 r0, r1, z0, z1;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        r0[q] = T[(kBias + 0 * 32 + q) * 64]; r1[q] = T[(kBias + 0 * 32 + 16 + q) * 64];
        z0[q] = T[(kBias + 1 * 32 + q) * 64]; z1[q] = T[(kBias + 1 * 32 + 16 + q) * 64];
    }
    // ---- layer 1 of r and z: 88 k-steps, one shared B operand ----
#pragma unroll
    for (int s = 0; s < 88; ++s) {
        const float b = xh[s];
        r0 = FS_MFMA(T[(kR1 + s) * 64], b, r0);
        r1 = FS_MFMA(T[(kR1 + 88 + s) * 64], b, r1);
        z0 = FS_MFMA(T[(kZ1 + s) * 64], b, z0);
        z1 = FS_MFMA(T[(kZ1 + 88 + s) * 64], b, z1);
    }
    // ---- layer 2 of r and z: k-step s <-> hidden unit held as register (s & 15) of block (s >> 4) ----
    f32x16 R0, R1, Z0, Z1;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        R0[q] = T[(kBias + 2 * 32 + q) * 64]; R1[q] = T[(kBias + 2 * 32 + 16 + q) * 64];
        Z0[q] = T[(kBias + 3 * 32 + q) * 64]; Z1[q] = T[(kBias + 3 * 32 + 16 + q) * 64];
    }
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        const float br = fmaxf(s < 16 ? r0[s & 15] : r1[s & 15], 0.0f);
        const float bz = fmaxf(s < 16 ? z0[s & 15] : z1[s & 15], 0.0f);
        R0 = FS_MFMA(T[(kR2 + s) * 64], br, R0);
        R1 = FS_MFMA(T[(kR2 + 32 + s) * 64], br, R1);
        Z0 = FS_MFMA(T[(kZ2 + s) * 64], bz, Z0);
        Z1 = FS_MFMA(T[(kZ2 + 32 + s) * 64], bz, Z1);
    }
    // ---- mlp_n layer 1: [r * hid (64) | x (64) | xe (24)] = 32 + 44 k-steps ----
    f32x16 n0, n1;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        n0[q] = T[(kBias + 4 * 32 + q) * 64]; n1[q] = T[(kBias + 4 * 32 + 16 + q) * 64];
    }
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        const float rr = sigmoidf_(s < 16 ? R0[s & 15] : R1[s & 15]);
        const float b = rr * hid[s];
        n0 = FS_MFMA(T[(kN1 + s) * 64], b, n0);
        n1 = FS_MFMA(T[(kN1 + 76 + s) * 64], b, n1);
    }
    {   // tail inputs cat[88 + s + 44*hf], s < 44: half 1 already holds them (xh[44..88)), half 0 loads them
        float xt[44];
        if (hf) {
#pragma unroll
            for (int k = 0; k < 44; ++k) xt[k] = xh[44 + k];
        } else {
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const float4 v = ((const float4*)xrow)[k];       // x[0..44)
                xt[4 * k] = v.x; xt[4 * k + 1] = v.y; xt[4 * k + 2] = v.z; xt[4 * k + 3] = v.w;
            }
        }
#pragma unroll
        for (int s = 0; s < 44; ++s) {
            n0 = FS_MFMA(T[(kN1 + 32 + s) * 64], xt[s], n0);
            n1 = FS_MFMA(T[(kN1 + 76 + 32 + s) * 64], xt[s], n1);
        }
    }
    // ---- mlp_n layer 2 ----
    f32x16 N0, N1;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        N0[q] = T[(kBias + 5 * 32 + q) * 64]; N1[q] = T[(kBias + 5 * 32 + 16 + q) * 64];
    }
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        const float b = fmaxf(s < 16 ? n0[s & 15] : n1[s & 15], 0.0f);
        N0 = FS_MFMA(T[(kN2 + s) * 64], b, N0);
        N1 = FS_MFMA(T[(kN2 + 32 + s) * 64], b, N1);
    }
    // ---- gates: out = (1 - z) * hid + z * tanh(q), lane holds 32 units of its pair ----
    if (live) {
        float* o = fused + (size_t)t * 64;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int q = 4 * g4 + e;
                    const float zz = sigmoidf_(blk ? Z1[q] : Z0[q]);
                    const float qq = tanhf(blk ? N1[q] : N0[q]);
                    const float h = hid[16 * blk + q];
                    v[e] = (1.0f - zz) * h + zz * qq;
                }
                *(float4*)(o + 32 * blk + 8 * g4 + 4 * hf) = make_float4(v[0], v[1], v[2], v[3]);
            }
    }
}

// n pairs, or (counts != NULL) at most n_max with the actual number in counts[1] on the device
int launch_ptf_gru(int n_max, const int32_t* counts, const float* cat, const float* tables, float* fused, hipStream_t st)
{
    if (n_max <= 0) return FS_OK;
    const int groups = (n_max + 31) / 32;
    hipLaunchKernelGGL(ptf_gru_kernel<false>, dim3((groups + 3) / 4), dim3(256), 0, st, n_max, counts, cat, GruGather{}, tables,
                       fused);
    FS_CHECK_LAUNCH("ptf_gru_forward");
    return FS_OK;
}

int launch_ptf_gru_gather(int n_max, const int32_t* counts, const long long* fuse_idx, const long long* fuse_pix,
                          const float* G, const float* R, const float* O, const float* g_i, const float* rho_i,
                          const float* om_i, const float* tables, float* fused, hipStream_t st)
{
    if (n_max <= 0) return FS_OK;
    const int groups = (n_max + 31) / 32;
    const GruGather ga{fuse_idx, fuse_pix, G, R, O, g_i, rho_i, om_i};
    hipLaunchKernelGGL(ptf_gru_kernel<true>, dim3((groups + 3) / 4), dim3(256), 0, st, n_max, counts, (const float*)nullptr, ga,
                       tables, fused);
    FS_CHECK_LAUNCH("ptf_gru_gather");
    return FS_OK;
}

}  // namespace fs

using namespace fs;

FS_API int32_t fs_ptf_gru_table_rows(void) { return kRows; }

FS_API int fs_ptf_gru_forward(int32_t n, const float* cat, const float* tables, float* fused, void* stream_)
{
    if (n < 0) return FS_ERR_INVALID_ARG;
    if (n == 0) return FS_OK;
    if (!cat || !tables || !fused) return FS_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream_;
    ScopedStage prof_(kStPtf, st);
    return launch_ptf_gru(n, nullptr, cat, tables, fused, st);
}

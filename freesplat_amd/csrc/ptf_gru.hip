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
#include <algorithm>
#include <cstdlib>

namespace fs {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// table layout (floats, each row = 64 lanes): the A-operand rows of the six matrices IN THE ORDER THE KERNEL CONSUMES
// THEM (one row per MFMA), then the bias rows:
//   [r1/z1: 88 steps x (r block 0, r block 1, z block 0, z block 1)][r2/z2: 32 x 4][n1: 76 x 2][n2: 32 x 2]   696 rows
//   [pad to whole LDS chunks: 704][b_r1: 2*16][b_z1][b_r2][b_z2][b_n1][b_n2]                                   896 rows
// All four wavefronts of a workgroup consume the same rows in the same order, so they reach the matrix pipe through a
// two-chunk LDS ring: every wavefront fetches a quarter of the NEXT chunk (kCh rows) from L2 into kCh/4 registers while
// the current chunk is being multiplied -- a prefetch distance of kCh MFMAs (4 k cycles), one barrier per chunk, a
// quarter of the L2 traffic.  (With each wavefront streaming its own rows from L2 -- one 256-byte load per MFMA, issued
// a few MFMAs ahead -- the matrix pipe waited half of the time or more: backward 749 -> 324 us per 157 k pairs.)
#ifndef FS_GRU_CH
#define FS_GRU_CH 64     // (16 / 32 / 64 measured: both kernels indifferent -- forward 117 - 119 us, backward 309 - 310 us per 103 k pairs; profiles/r4_gru_chunk_ab.txt)
#endif
constexpr int kCh = FS_GRU_CH;
constexpr int kP0 = 0, kP1 = kP0 + 4 * 88, kP2 = kP1 + 4 * 32, kP3 = kP2 + 2 * 76, kFwdUsed = kP3 + 2 * 32,
              kFwdChunks = (kFwdUsed + kCh - 1) / kCh;
constexpr int kBias = kFwdChunks * kCh, kRows = kBias + 6 * 32;
static_assert(kFwdUsed == 696 && kBias == 704, "operand table layout");

// The ring: G = this wavefront's rows of the chunk after the current one.  FS_AOP(pos) = operand row `pos` of the stream
// (pos a compile-time constant after unrolling; the switch to a new chunk folds away everywhere else).
#define FS_RING_SETUP(STREAM, NCHUNKS, TOTAL, LA, QUAD)                                                                  \
    __shared__ __attribute__((aligned(16))) float s_ring[2 * kCh * 64];                                                  \
    constexpr int kRingTotal = (TOTAL), kRingLA = (LA);                                                                  \
    constexpr bool kRingQuad = (QUAD);                                                                                   \
    float Q[8];     /* LA > 0: operand rows pos .. pos + LA - 1 already read (slot pos % 8) */                          \
    float4 Q4[2];   /* QUAD: the quad of rows in use and the next one */                                                 \
    float G[kCh / 4];                                                                                                    \
    auto load_chunk = [&](int c) {                                                                                       \
        if constexpr (kRingQuad) {                                                                                       \
            _Pragma("unroll") for (int q = 0; q < kCh / 16; ++q) {                                                       \
                const float4 v = ((const float4*)(STREAM))[((size_t)(c * 4 + wave) * (kCh / 16) + q) * 64 + lane];       \
                G[4 * q] = v.x; G[4 * q + 1] = v.y; G[4 * q + 2] = v.z; G[4 * q + 3] = v.w;                              \
            }                                                                                                            \
        } else {                                                                                                         \
            _Pragma("unroll") for (int j = 0; j < kCh / 4; ++j)                                                          \
                G[j] = (STREAM)[(size_t)(c * kCh + wave * (kCh / 4) + j) * 64 + lane];                                   \
        }                                                                                                                \
    };                                                                                                                   \
    auto switch_chunk = [&](int c) {   /* before the first operand of chunk c is read; G holds this wavefront's rows of it */ \
        if constexpr (kRingQuad) {                                                                                       \
            _Pragma("unroll") for (int q = 0; q < kCh / 16; ++q)                                                         \
                ((float4*)s_ring)[((c & 1) * (kCh / 4) + wave * (kCh / 16) + q) * 64 + lane] =                           \
                    make_float4(G[4 * q], G[4 * q + 1], G[4 * q + 2], G[4 * q + 3]);                                     \
        } else {                                                                                                         \
            _Pragma("unroll") for (int j = 0; j < kCh / 4; ++j)                                                          \
                s_ring[((c & 1) * kCh + wave * (kCh / 4) + j) * 64 + lane] = G[j];                                       \
        }                                                                                                                \
        if (c + 1 < (NCHUNKS)) load_chunk(c + 1);                                                                        \
        __syncthreads();                                                                                                 \
    };                                                                                                                   \
    load_chunk(0);                                                                                                       \
    if constexpr (kRingQuad) { FS_RING_FETCH4(0); }                                                                      \
    else { _Pragma("unroll") for (int q_ = 0; q_ < kRingLA; ++q_) FS_RING_FETCH(q_); }
// row `pos` of the stream from the ring (switching to its chunk first where a chunk starts)
#define FS_RING_ROW(pos) (((pos) % kCh == 0 ? switch_chunk((pos) / kCh) : (void)0), \
                          s_ring[((((pos) / kCh) & 1) * kCh + (pos) % kCh) * 64 + lane])
#define FS_RING_FETCH(pos) ((pos) < kRingTotal ? (void)(Q[(pos) & 7] = FS_RING_ROW(pos)) : (void)0)
// QUAD: the stream is stored interleaved -- [chunk][owner wavefront][quad of 4 rows][lane][4 rows], i.e. a lane's four consecutive
// operand rows are ONE float4 in memory and in the ring -- so a chunk's fill is kCh/16 global_load_dwordx4 + kCh/16
// ds_write_b128 per wavefront (instead of kCh/4 dword loads + kCh/8 ds_write2st64), and the operands of four MFMAs come
// back with ONE ds_read_b128, issued a whole quad (four MFMAs) ahead of its first use.
#define FS_RING_ROW4(qp) (((qp) % (kCh / 4) == 0 ? switch_chunk((qp) / (kCh / 4)) : (void)0), \
                          ((const float4*)s_ring)[((((qp) / (kCh / 4)) & 1) * (kCh / 4) + (qp) % (kCh / 4)) * 64 + lane])
#define FS_RING_FETCH4(qp) ((qp) * 4 < kRingTotal ? (void)(Q4[(qp) & 1] = FS_RING_ROW4(qp)) : (void)0)
#define FS_Q4_COMP(v, e) ((e) == 0 ? (v).x : ((e) == 1 ? (v).y : ((e) == 2 ? (v).z : (v).w)))
// The operand of MFMA `pos`.  LA = 0: read where it is used (the forward: two wavefronts per SIMD hide the LDS round trip).  LA > 0
// (the backward: ONE wavefront per SIMD): row pos + LA is read HERE, LA MFMAs ahead of its use, and a scheduling barrier keeps
// the read in front of this MFMA -- left alone the scheduler issues a ds_read at most two MFMAs before its use and the single
// wavefront waits out the LDS round trip in front of every other MFMA pair.  The stream is consumed strictly in order.
#define FS_AOP(pos) (kRingQuad ? ((((pos) & 3) == 0 ? (FS_RING_FETCH4((pos) / 4 + 1), __builtin_amdgcn_sched_barrier(0)) : (void)0), \
                                  FS_Q4_COMP(Q4[((pos) / 4) & 1], (pos) & 3))                                                          \
                     : (kRingLA == 0 ? FS_RING_ROW(pos)                                                                                 \
                                     : (FS_RING_FETCH((pos) + kRingLA), __builtin_amdgcn_sched_barrier(0), Q[(pos) & 7])))

// gates on the hardware transcendentals (v_exp_f32 / v_rcp_f32, 1 ulp each; |error| of a gate ~2e-7, the fold's bar is
// 1e-4): the libm forms are ~30 VALU instructions each, 96 per lane and 32 pairs, and fp32 VALU work does not overlap
// fp32 MFMA on this chip (-4 % kernel time)
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177792681f * x)); }

#define FS_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// The six bias vectors (192 table rows, one per accumulator register and block) initialise the accumulators at the six layer
// starts: T[r * 64] = row r of the tables for this lane.  (Staging them in LDS once per workgroup -- 32 ds_reads instead of 32 L2
// round trips in front of a layer's first MFMA -- measured no different: profiles/r4_gru_bias_lds_ab.txt.)
#define FS_BIAS_SETUP(TAB) const float* T = (TAB) + lane;
#define FS_BIAS(r) T[(kBias + (r)) * 64]

// GATHER: the input row [hid | he | x | xe] of pair t is not read from a materialised `cat` array but assembled here:
// hid = G[fuse_idx[t]], x = g_i[fuse_pix[t]] (the half of the lanes that owns them loads them), he / xe = the positional
// encodings of the densities and weights (encoder_freesplat.py:485-486) -- 24 sin/cos per lane, every lane busy, instead
// of a separate kernel in which 2 lanes of 16 did them and 141 MB of rows went to HBM and back per 10^5 pairs.
struct GruGather {
    const long long *fuse_idx, *fuse_pix;
    const float *G, *R, *O, *g_i, *rho_i, *om_i;
    float *side = nullptr, *act = nullptr, *cat_out = nullptr;   // ptf_gru16_kernel<true, true>: where the training forward leaves what the backward needs
};
constexpr int kAct = 192;   // floats per pair the saving forward keeps beside the `side` columns: r, z (gates), q = tanh(.)
// (two workgroups per CU: at one -- 407 registers if the compiler is left alone -- the fold is 7 % slower)
template <bool GATHER>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void ptf_gru_kernel(int n, const int32_t* __restrict__ counts,
                                                      const float* __restrict__ cat, GruGather ga,
                                                      const float* __restrict__ tab, float* __restrict__ fused,
                                                      int out_after_keep)
{
    // out_after_keep: `fused` is the step's OUT state G and pair t goes to row counts[0] + t (behind the kept rows), where
    // fs_ptf_write_state would have copied it from a scratch array: 52 MB less traffic per 10^5 pairs
    const size_t out_row0 = (out_after_keep && counts) ? (size_t)counts[0] : 0;
    if (counts) n = counts[1];  // (device-resident pair count: fs_ptf_fold_step)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int grp = blockIdx.x * 4 + wave;
    if (blockIdx.x * 128 >= n) return;   // whole workgroup beyond n (a single wavefront beyond n stays for the barriers)
    const int p = lane & 31, hf = lane >> 5;
    const int t = grp * 32 + p;
    const bool live = t < n;
    FS_BIAS_SETUP(tab)
    FS_RING_SETUP(tab, kFwdChunks, kFwdUsed, 0, false)
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
        r0[q] = FS_BIAS(0 * 32 + q); r1[q] = FS_BIAS(0 * 32 + 16 + q);
        z0[q] = FS_BIAS(1 * 32 + q); z1[q] = FS_BIAS(1 * 32 + 16 + q);
    }
    // ---- layer 1 of r and z: 88 k-steps, one shared B operand ----
#pragma unroll
    for (int s = 0; s < 88; ++s) {
        const float b = xh[s];
        r0 = FS_MFMA(FS_AOP(kP0 + 4 * s), b, r0);
        r1 = FS_MFMA(FS_AOP(kP0 + 4 * s + 1), b, r1);
        z0 = FS_MFMA(FS_AOP(kP0 + 4 * s + 2), b, z0);
        z1 = FS_MFMA(FS_AOP(kP0 + 4 * s + 3), b, z1);
    }
    // ---- layer 2 of r and z: k-step s <-> hidden unit held as register (s & 15) of block (s >> 4) ----
    f32x16 R0, R1, Z0, Z1;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        R0[q] = FS_BIAS(2 * 32 + q); R1[q] = FS_BIAS(2 * 32 + 16 + q);
        Z0[q] = FS_BIAS(3 * 32 + q); Z1[q] = FS_BIAS(3 * 32 + 16 + q);
    }
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        const float br = fmaxf(s < 16 ? r0[s & 15] : r1[s & 15], 0.0f);
        const float bz = fmaxf(s < 16 ? z0[s & 15] : z1[s & 15], 0.0f);
        R0 = FS_MFMA(FS_AOP(kP1 + 4 * s), br, R0);
        R1 = FS_MFMA(FS_AOP(kP1 + 4 * s + 1), br, R1);
        Z0 = FS_MFMA(FS_AOP(kP1 + 4 * s + 2), bz, Z0);
        Z1 = FS_MFMA(FS_AOP(kP1 + 4 * s + 3), bz, Z1);
    }
    // ---- mlp_n layer 1: [r * hid (64) | x (64) | xe (24)] = 32 + 44 k-steps ----
    f32x16 n0, n1;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        n0[q] = FS_BIAS(4 * 32 + q); n1[q] = FS_BIAS(4 * 32 + 16 + q);
    }
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        const float rr = sigmoidf_(s < 16 ? R0[s & 15] : R1[s & 15]);
        const float b = rr * hid[s];
        n0 = FS_MFMA(FS_AOP(kP2 + 2 * s), b, n0);
        n1 = FS_MFMA(FS_AOP(kP2 + 2 * s + 1), b, n1);
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
            n0 = FS_MFMA(FS_AOP(kP2 + 64 + 2 * s), xt[s], n0);
            n1 = FS_MFMA(FS_AOP(kP2 + 64 + 2 * s + 1), xt[s], n1);
        }
    }
    // ---- mlp_n layer 2 ----
    f32x16 N0, N1;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        N0[q] = FS_BIAS(5 * 32 + q); N1[q] = FS_BIAS(5 * 32 + 16 + q);
    }
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        const float b = fmaxf(s < 16 ? n0[s & 15] : n1[s & 15], 0.0f);
        N0 = FS_MFMA(FS_AOP(kP3 + 2 * s), b, N0);
        N1 = FS_MFMA(FS_AOP(kP3 + 2 * s + 1), b, N1);
    }
    // ---- gates: out = (1 - z) * hid + z * tanh(q), lane holds 32 units of its pair ----
    if (live) {
        float* o = fused + (out_row0 + (size_t)t) * 64;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int q = 4 * g4 + e;
                    const float zz = sigmoidf_(blk ? Z1[q] : Z0[q]);
                    const float qq = tanhf_(blk ? N1[q] : N0[q]);
                    const float h = hid[16 * blk + q];
                    v[e] = (1.0f - zz) * h + zz * qq;
                }
                *(float4*)(o + 32 * blk + 8 * g4 + 4 * hf) = make_float4(v[0], v[1], v[2], v[3]);
            }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Backward of the GRU, same organisation: one wavefront owns 32 pairs, lane = (pair, half), every value of a pair
// stays in its own lanes.  The forward is re-run from the materialised input rows (keeping r, z, q, the ReLU masks as
// bit fields), then every linear layer runs TRANSPOSED once more -- dX^T = W^T dY^T: the A operand is the table of
// W^T in operand order, the B operand is dY straight from registers with the k order = accumulator row map, exactly
// the trick of the forward's second layers -- so the input gradient dcat[n,176] needs no cross-lane traffic either:
// 696 (forward) + 768 (transposed) MFMAs per 32 pairs.  The weight gradients are sums over ALL pairs of outer products
// dY (x) X (44 928 accumulators -- 702 registers per lane if a wavefront kept them), so the kernel writes the six
// pre-activation gradients and the four hidden activations they pair with into `side` [n,640]
//   [dr1 | dz1 | dR | dZ | dn1 | dN | relu(r1) | relu(z1) | relu(n1) | r*hid]            (64 floats each)
// and the caller forms dW = dY^T X with six library GEMMs whose contraction runs over n (ptf.py:_PtfFold.backward).
//
// transposed tables (rows of 64 lanes; lane (p, hf) of row (rb, s) holds W[u(s, hf)][32 rb + p], u = the forward's
// accumulator unit map; 0 outside the matrix):
//   [n2T: 2*32][n1T, r*hid rows: 2*32][n1T in cat-feature rows, blocks 2..5: 4*32][r2T: 2*32][z2T: 2*32][r1T: 6*32][z1T: 6*32]
constexpr int kTN2 = 0, kTN1H = kTN2 + 64, kTN1C = kTN1H + 64, kTR2 = kTN1C + 128, kTZ2 = kTR2 + 64, kTR1 = kTZ2 + 64,
              kTZ1 = kTR1 + 192, kRowsT = kTZ1 + 192;
constexpr int kSide = 640;
// consumption order of the backward kernel's 1464 A-operand rows: the forward's 696 (re-run), then the transposed layers;
// `stream` holds them in this order (the last chunk padded).  Positions of the transposed layers' first rows:
constexpr int kP4 = kFwdUsed, kP5 = kP4 + 2 * 32, kP6 = kP5 + 6 * 32, kP7 = kP6 + 4 * 32, kStreamUsed = kP7 + 12 * 32,
              kStreamChunks = (kStreamUsed + kCh - 1) / kCh;
static_assert(kStreamUsed == 1464, "operand stream layout");

__device__ __forceinline__ void store_acc(float* __restrict__ dst, int hf, const f32x16& a)   // units acc rows of one block
{
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
        *(float4*)(dst + 8 * g4 + 4 * hf) = make_float4(a[4 * g4], a[4 * g4 + 1], a[4 * g4 + 2], a[4 * g4 + 3]);
}

#ifndef FS_GRU_BWD_QUAD
#define FS_GRU_BWD_QUAD 1   // the backward's operand stream interleaved by quads of rows (fs_ptf_gru_stream_layout() tells the host)
#endif
#ifndef FS_GRU_BWD_LA
#define FS_GRU_BWD_LA 4    // operand rows read this many MFMAs ahead in the backward (0: A/B, the scheduler's own placement)
#endif
#ifdef FS_GRU_BWD_WAVES    // (A/B builds: make VARIANT=gru2 EXTRA=-DFS_GRU_BWD_WAVES=2 -- 256 registers, ~155 values through scratch)
#define FS_GRU_BWD_OCC __attribute__((amdgpu_waves_per_eu(FS_GRU_BWD_WAVES, FS_GRU_BWD_WAVES)))
#else
#define FS_GRU_BWD_OCC
#endif
__global__ __launch_bounds__(256) FS_GRU_BWD_OCC void ptf_gru_bwd_kernel(int n, const float* __restrict__ cat, const float* __restrict__ tab,
                                                          const float* __restrict__ stream, const float* __restrict__ g_fused,
                                                          float* __restrict__ dcat, float* __restrict__ side)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int grp = blockIdx.x * 4 + wave;
    const int p = lane & 31, hf = lane >> 5;
    const int t = grp * 32 + p;
    const bool live = t < n;             // (a wavefront beyond n stays for the barriers, computes on row 0, stores nothing)
    const size_t tr = (size_t)(live ? t : 0);
    FS_BIAS_SETUP(tab)
    FS_RING_SETUP(stream, kStreamChunks, kStreamUsed, FS_GRU_BWD_LA, FS_GRU_BWD_QUAD != 0)
    const float* row = cat + tr * 176;
    float* sd = side + tr * kSide;       // (dead pairs compute on row 0 and store nothing)

    float hid[32];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4 v = *(const float4*)(row + 32 * blk + 8 * g4 + 4 * hf);
            hid[16 * blk + 4 * g4] = v.x; hid[16 * blk + 4 * g4 + 1] = v.y;
            hid[16 * blk + 4 * g4 + 2] = v.z; hid[16 * blk + 4 * g4 + 3] = v.w;
        }
    // ================= forward, keeping what the backward needs =================
    uint32_t mr = 0, mz = 0, mn = 0;     // ReLU masks of the three first layers, bit 16*blk + q
    float rr[32], zz[32];
    f32x16 n0, n1;
    {
        f32x16 r0, r1, z0, z1;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            r0[q] = FS_BIAS(0 * 32 + q); r1[q] = FS_BIAS(0 * 32 + 16 + q);
            z0[q] = FS_BIAS(1 * 32 + q); z1[q] = FS_BIAS(1 * 32 + 16 + q);
        }
        {
            float xh[88];
#pragma unroll
            for (int k = 0; k < 22; ++k) {
                const float4 v = ((const float4*)(row + 88 * hf))[k];
                xh[4 * k] = v.x; xh[4 * k + 1] = v.y; xh[4 * k + 2] = v.z; xh[4 * k + 3] = v.w;
            }
#pragma unroll
            for (int s = 0; s < 88; ++s) {
                const float b = xh[s];
                r0 = FS_MFMA(FS_AOP(kP0 + 4 * s), b, r0);
                r1 = FS_MFMA(FS_AOP(kP0 + 4 * s + 1), b, r1);
                z0 = FS_MFMA(FS_AOP(kP0 + 4 * s + 2), b, z0);
                z1 = FS_MFMA(FS_AOP(kP0 + 4 * s + 3), b, z1);
            }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            mr |= (r0[q] > 0.0f ? 1u : 0u) << q;  mr |= (r1[q] > 0.0f ? 1u : 0u) << (16 + q);
            mz |= (z0[q] > 0.0f ? 1u : 0u) << q;  mz |= (z1[q] > 0.0f ? 1u : 0u) << (16 + q);
            r0[q] = fmaxf(r0[q], 0.0f); r1[q] = fmaxf(r1[q], 0.0f);
            z0[q] = fmaxf(z0[q], 0.0f); z1[q] = fmaxf(z1[q], 0.0f);
        }
        if (live) {
            store_acc(sd + 6 * 64, hf, r0); store_acc(sd + 6 * 64 + 32, hf, r1);
            store_acc(sd + 7 * 64, hf, z0); store_acc(sd + 7 * 64 + 32, hf, z1);
        }
        f32x16 R0, R1, Z0, Z1;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            R0[q] = FS_BIAS(2 * 32 + q); R1[q] = FS_BIAS(2 * 32 + 16 + q);
            Z0[q] = FS_BIAS(3 * 32 + q); Z1[q] = FS_BIAS(3 * 32 + 16 + q);
        }
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const float br = s < 16 ? r0[s & 15] : r1[s & 15];
            const float bz = s < 16 ? z0[s & 15] : z1[s & 15];
            R0 = FS_MFMA(FS_AOP(kP1 + 4 * s), br, R0);
            R1 = FS_MFMA(FS_AOP(kP1 + 4 * s + 1), br, R1);
            Z0 = FS_MFMA(FS_AOP(kP1 + 4 * s + 2), bz, Z0);
            Z1 = FS_MFMA(FS_AOP(kP1 + 4 * s + 3), bz, Z1);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            rr[q] = sigmoidf_(R0[q]); rr[16 + q] = sigmoidf_(R1[q]);
            zz[q] = sigmoidf_(Z0[q]); zz[16 + q] = sigmoidf_(Z1[q]);
        }
    }
    {   // mlp_n layer 1
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            n0[q] = FS_BIAS(4 * 32 + q); n1[q] = FS_BIAS(4 * 32 + 16 + q);
        }
        f32x16 h0, h1;   // r * hid, stored for the weight gradient of mlp_n layer 1
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const float b = rr[s] * hid[s];
            if (s < 16) h0[s & 15] = b; else h1[s & 15] = b;
            n0 = FS_MFMA(FS_AOP(kP2 + 2 * s), b, n0);
            n1 = FS_MFMA(FS_AOP(kP2 + 2 * s + 1), b, n1);
        }
        if (live) { store_acc(sd + 9 * 64, hf, h0); store_acc(sd + 9 * 64 + 32, hf, h1); }
        float xt[44];
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float4 v = ((const float4*)(row + 88 + 44 * hf))[k];
            xt[4 * k] = v.x; xt[4 * k + 1] = v.y; xt[4 * k + 2] = v.z; xt[4 * k + 3] = v.w;
        }
#pragma unroll
        for (int s = 0; s < 44; ++s) {
            n0 = FS_MFMA(FS_AOP(kP2 + 64 + 2 * s), xt[s], n0);
            n1 = FS_MFMA(FS_AOP(kP2 + 64 + 2 * s + 1), xt[s], n1);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            mn |= (n0[q] > 0.0f ? 1u : 0u) << q;  mn |= (n1[q] > 0.0f ? 1u : 0u) << (16 + q);
            n0[q] = fmaxf(n0[q], 0.0f); n1[q] = fmaxf(n1[q], 0.0f);
        }
        if (live) { store_acc(sd + 8 * 64, hf, n0); store_acc(sd + 8 * 64 + 32, hf, n1); }
    }
    f32x16 dN0, dN1, dZ0, dZ1;   // pre-activation gradients of the two output layers
    float dh[32];                // gradient of hid through the gate (1 - z) * hid
    {
        f32x16 N0, N1;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            N0[q] = FS_BIAS(5 * 32 + q); N1[q] = FS_BIAS(5 * 32 + 16 + q);
        }
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const float b = s < 16 ? n0[s & 15] : n1[s & 15];
            N0 = FS_MFMA(FS_AOP(kP3 + 2 * s), b, N0);
            N1 = FS_MFMA(FS_AOP(kP3 + 2 * s + 1), b, N1);
        }
        const float* go = g_fused + tr * 64;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 gv = *(const float4*)(go + 32 * blk + 8 * g4 + 4 * hf);
                const float gq[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int q = 4 * g4 + e;
                    const float g = live ? gq[e] : 0.0f;
                    const float z = zz[16 * blk + q], h = hid[16 * blk + q];
                    const float qq = tanhf_(blk ? N1[q] : N0[q]);
                    const float dn = g * z * (1.0f - qq * qq);
                    const float dz = g * (qq - h) * z * (1.0f - z);
                    if (blk) { dN1[q] = dn; dZ1[q] = dz; } else { dN0[q] = dn; dZ0[q] = dz; }
                    dh[16 * blk + q] = g * (1.0f - z);
                }
            }
    }
    if (live) {
        store_acc(sd + 5 * 64, hf, dN0); store_acc(sd + 5 * 64 + 32, hf, dN1);
        store_acc(sd + 3 * 64, hf, dZ0); store_acc(sd + 3 * 64 + 32, hf, dZ1);
    }
    // ================= transposed layers =================
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 C2 = zero16, C3 = zero16, C4 = zero16, C5 = zero16;   // dcat feature blocks 2..5 (x | xe of mlp_n first)
    f32x16 dR0, dR1;
    {
        // mlp_n layer 2: d relu(n1) = Wn2^T dN, masked -> dn1
        f32x16 a0 = zero16, a1 = zero16;
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const float b = s < 16 ? dN0[s & 15] : dN1[s & 15];
            a0 = FS_MFMA(FS_AOP(kP4 + 2 * s), b, a0);
            a1 = FS_MFMA(FS_AOP(kP4 + 2 * s + 1), b, a1);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            a0[q] = (mn >> q) & 1u ? a0[q] : 0.0f;
            a1[q] = (mn >> (16 + q)) & 1u ? a1[q] : 0.0f;
        }
        if (live) { store_acc(sd + 4 * 64, hf, a0); store_acc(sd + 4 * 64 + 32, hf, a1); }
        // mlp_n layer 1: d(r*hid) (2 blocks) and the x | xe part straight into its dcat blocks
        f32x16 H0 = zero16, H1 = zero16;
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const float b = s < 16 ? a0[s & 15] : a1[s & 15];
            H0 = FS_MFMA(FS_AOP(kP5 + 6 * s), b, H0);
            H1 = FS_MFMA(FS_AOP(kP5 + 6 * s + 1), b, H1);
            C2 = FS_MFMA(FS_AOP(kP5 + 6 * s + 2), b, C2);
            C3 = FS_MFMA(FS_AOP(kP5 + 6 * s + 3), b, C3);
            C4 = FS_MFMA(FS_AOP(kP5 + 6 * s + 4), b, C4);
            C5 = FS_MFMA(FS_AOP(kP5 + 6 * s + 5), b, C5);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float r_0 = rr[q], r_1 = rr[16 + q];
            dR0[q] = H0[q] * hid[q] * r_0 * (1.0f - r_0);
            dR1[q] = H1[q] * hid[16 + q] * r_1 * (1.0f - r_1);
            dh[q] += H0[q] * r_0;
            dh[16 + q] += H1[q] * r_1;
        }
    }
    if (live) { store_acc(sd + 2 * 64, hf, dR0); store_acc(sd + 2 * 64 + 32, hf, dR1); }
    f32x16 e0 = zero16, e1 = zero16, f0 = zero16, f1 = zero16;   // dr1, dz1 (first-layer pre-activation gradients)
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        const float br = s < 16 ? dR0[s & 15] : dR1[s & 15];
        const float bz = s < 16 ? dZ0[s & 15] : dZ1[s & 15];
        e0 = FS_MFMA(FS_AOP(kP6 + 4 * s), br, e0);
        e1 = FS_MFMA(FS_AOP(kP6 + 4 * s + 1), br, e1);
        f0 = FS_MFMA(FS_AOP(kP6 + 4 * s + 2), bz, f0);
        f1 = FS_MFMA(FS_AOP(kP6 + 4 * s + 3), bz, f1);
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        e0[q] = (mr >> q) & 1u ? e0[q] : 0.0f;  e1[q] = (mr >> (16 + q)) & 1u ? e1[q] : 0.0f;
        f0[q] = (mz >> q) & 1u ? f0[q] : 0.0f;  f1[q] = (mz >> (16 + q)) & 1u ? f1[q] : 0.0f;
    }
    if (live) {
        store_acc(sd + 0 * 64, hf, e0); store_acc(sd + 0 * 64 + 32, hf, e1);
        store_acc(sd + 1 * 64, hf, f0); store_acc(sd + 1 * 64 + 32, hf, f1);
    }
    // first layers of r and z: all six feature blocks of dcat
    f32x16 C0, C1;
#pragma unroll
    for (int q = 0; q < 16; ++q) { C0[q] = dh[q]; C1[q] = dh[16 + q]; }
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        const float br = s < 16 ? e0[s & 15] : e1[s & 15];
        const float bz = s < 16 ? f0[s & 15] : f1[s & 15];
        C0 = FS_MFMA(FS_AOP(kP7 + 12 * s + 0), br, C0);  C0 = FS_MFMA(FS_AOP(kP7 + 12 * s + 1), bz, C0);
        C1 = FS_MFMA(FS_AOP(kP7 + 12 * s + 2), br, C1);  C1 = FS_MFMA(FS_AOP(kP7 + 12 * s + 3), bz, C1);
        C2 = FS_MFMA(FS_AOP(kP7 + 12 * s + 4), br, C2);  C2 = FS_MFMA(FS_AOP(kP7 + 12 * s + 5), bz, C2);
        C3 = FS_MFMA(FS_AOP(kP7 + 12 * s + 6), br, C3);  C3 = FS_MFMA(FS_AOP(kP7 + 12 * s + 7), bz, C3);
        C4 = FS_MFMA(FS_AOP(kP7 + 12 * s + 8), br, C4);  C4 = FS_MFMA(FS_AOP(kP7 + 12 * s + 9), bz, C4);
        C5 = FS_MFMA(FS_AOP(kP7 + 12 * s + 10), br, C5);  C5 = FS_MFMA(FS_AOP(kP7 + 12 * s + 11), bz, C5);
    }
    if (live) {
        float* dc = dcat + tr * 176;
        store_acc(dc, hf, C0); store_acc(dc + 32, hf, C1); store_acc(dc + 64, hf, C2);
        store_acc(dc + 96, hf, C3); store_acc(dc + 128, hf, C4);
#pragma unroll
        for (int g4 = 0; g4 < 2; ++g4)   // features 160 .. 175 of the last block
            *(float4*)(dc + 160 + 8 * g4 + 4 * hf) = make_float4(C5[4 * g4], C5[4 * g4 + 1], C5[4 * g4 + 2], C5[4 * g4 + 3]);
    }
}

// FS_GRU_BWD16=0: the 32-pair backward kernel of rounds 4 - 5 (A/B; the operand stream's layout follows: fs_ptf_gru_stream_layout())
static bool gru_bwd16()
{
    static const bool on = [] { const char* e = getenv("FS_GRU_BWD16"); return !(e && atoi(e) == 0); }();
    return on;
}

// ------------------------------------------------------------------------------------------------------------
// Backward of the GRU on 16-PAIR wavefronts (round 6): the same computation as ptf_gru_bwd_kernel on v_mfma_f32_16x16x4_f32, lane =
// (pair n = lane & 15, quarter g = lane >> 4).  An activation matrix [64 units x 16 pairs] is 4 blocks x 4 registers per lane
// (register (blk, r) = unit 16 blk + 4 g + r of pair n) instead of 2 x 16, the input row 44 instead of 88 registers: the kernel
// fits 256 registers and TWO wavefronts share a SIMD -- with one, 45 % of the 32-pair kernel's time was gate math, LDS / memory
// issue, waits and layer-boundary dependencies that nothing overlapped (profiles/r6_gru_bwd_waves_ab.txt).  As everywhere in this
// layout an accumulator IS the next layer's B operand (k-step s of a 64-unit input = register (s >> 2, s & 3): units
// 16 (s >> 2) + 4 kk + (s & 3) over the quarters kk), forward and transposed.  Operand rows: 696 (forward re-run) + 704
// (transposed layers; the 176 features of dcat are 11 blocks of 16) = 1 400, consumed strictly in order from the quad-interleaved
// stream freesplat_amd/ptf.py:gru_operand_stream builds for fs_ptf_gru_stream_layout() = 2; the six bias vectors follow it.
// ------------------------------------------------------------------------------------------------------------
constexpr int kS16Used = 696 + 704, kS16Chunks = (kS16Used + kCh - 1) / kCh, kT16Chunks = (704 + kCh - 1) / kCh;
#define FS_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

#ifndef FS_GRU_BWD16_WAVES
#define FS_GRU_BWD16_WAVES 2
#endif
// SAVED: the training forward (ptf_gru16_kernel<true, true>) left relu(r1), relu(z1), relu(n1), r * hid in `side` (columns 6 .. 9) and the
// gates r, z, q in `act`: no re-run of the forward, `stream` = the 704 transposed rows alone (kT16Chunks chunks).
template <bool SAVED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FS_GRU_BWD16_WAVES, FS_GRU_BWD16_WAVES))) void ptf_gru_bwd16_kernel(
    int n, const float* __restrict__ cat, const float* __restrict__ stream, const float* __restrict__ g_fused,
    float* __restrict__ dcat, float* __restrict__ side, const float* __restrict__ act)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int grp = blockIdx.x * 4 + wave;
    const int pn = lane & 15, g = lane >> 4;
    const int t = grp * 16 + pn;
    const bool live = t < n;             // (a wavefront beyond n stays for the barriers, computes on row 0, stores nothing)
    const size_t tr = (size_t)(live ? t : 0);
    FS_RING_SETUP(stream, (SAVED ? kT16Chunks : kS16Chunks), (SAVED ? kS16Used - 696 : kS16Used), 0, true)
    const float* const bias = stream + (size_t)kS16Chunks * kCh * 64;     // [6][64]: br1, bz1, br2, bz2, bn1, bn2 (!SAVED)
    // (the forward's lane-native layout; a wavefront beyond n reads group 0, the last group's dead lanes rows the caller padded to 16)
    const float* const ac = SAVED ? act + (size_t)(grp * 16 < n ? grp : 0) * (16 * kAct) + 4 * lane : nullptr;
    const float* row = cat + tr * 176;
    float* sd = side + tr * kSide;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    auto ld4 = [&](const float* p) __attribute__((always_inline)) { const float4 v = *(const float4*)p; return f32x4{v.x, v.y, v.z, v.w}; };
    auto st4 = [&](float* p, const f32x4 v) __attribute__((always_inline)) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); };
    const int ao = 4 * g;                                  // accumulator block blk, registers 0..3 = units / features 16 blk + ao + 0..3
    int pos = 0;                                           // (compile-time after unrolling: every loop below is fully unrolled)

    f32x4 hid[4];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) hid[blk] = ld4(row + 16 * blk + ao);
    // ================= forward, keeping what the backward needs =================
    uint32_t mr = 0, mz = 0, mn = 0;     // ReLU masks of the three first layers, bit 4 blk + r
    f32x4 rr[4], zz[4], n1[4];
    constexpr int p2 = 44 * 8 + 16 * 8;
    if constexpr (SAVED) {
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) { rr[blk] = ld4(ac + (0 * 4 + blk) * 256); zz[blk] = ld4(ac + (1 * 4 + blk) * 256); }
    } else {
    {
        f32x4 r1[4], z1[4];
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) { r1[blk] = ld4(bias + 0 * 64 + 16 * blk + ao); z1[blk] = ld4(bias + 1 * 64 + 16 * blk + ao); }
        {
            float xh[44];       // this quarter's 44 features of the input row: k-step s, quarter g <-> feature 44 g + s
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const float4 v = ((const float4*)(row + 44 * g))[k];
                xh[4 * k] = v.x; xh[4 * k + 1] = v.y; xh[4 * k + 2] = v.z; xh[4 * k + 3] = v.w;
            }
#pragma unroll
            for (int s = 0; s < 44; ++s) {
#pragma unroll
                for (int ob = 0; ob < 4; ++ob) r1[ob] = FS_MFMA16(FS_AOP(s * 8 + ob), xh[s], r1[ob]);
#pragma unroll
                for (int ob = 0; ob < 4; ++ob) z1[ob] = FS_MFMA16(FS_AOP(s * 8 + 4 + ob), xh[s], z1[ob]);
            }
        }
        constexpr int p1 = 44 * 8;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                mr |= (r1[blk][r] > 0.0f ? 1u : 0u) << (4 * blk + r);
                mz |= (z1[blk][r] > 0.0f ? 1u : 0u) << (4 * blk + r);
                r1[blk][r] = fmaxf(r1[blk][r], 0.0f); z1[blk][r] = fmaxf(z1[blk][r], 0.0f);
            }
        if (live) {
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) { st4(sd + 6 * 64 + 16 * blk + ao, r1[blk]); st4(sd + 7 * 64 + 16 * blk + ao, z1[blk]); }
        }
        f32x4 R[4], Z[4];
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) { R[blk] = ld4(bias + 2 * 64 + 16 * blk + ao); Z[blk] = ld4(bias + 3 * 64 + 16 * blk + ao); }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float br = r1[s >> 2][s & 3], bz = z1[s >> 2][s & 3];
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) R[ob] = FS_MFMA16(FS_AOP(p1 + s * 8 + ob), br, R[ob]);
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) Z[ob] = FS_MFMA16(FS_AOP(p1 + s * 8 + 4 + ob), bz, Z[ob]);
        }
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) { rr[blk][r] = sigmoidf_(R[blk][r]); zz[blk][r] = sigmoidf_(Z[blk][r]); }
    }
    {   // mlp_n layer 1: [r * hid (64) | x (64) | xe (24)]
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) n1[blk] = ld4(bias + 4 * 64 + 16 * blk + ao);
        f32x4 h0[4];   // r * hid, stored for the weight gradient of mlp_n layer 1
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) h0[blk][r] = rr[blk][r] * hid[blk][r];
        if (live) {
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) st4(sd + 9 * 64 + 16 * blk + ao, h0[blk]);
        }
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) n1[ob] = FS_MFMA16(FS_AOP(p2 + s * 4 + ob), h0[s >> 2][s & 3], n1[ob]);
        float xt[22];   // k-step s, quarter g <-> feature 88 + 22 g + s of the row (x | xe)
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float2 v = ((const float2*)(row + 88 + 22 * g))[k];
            xt[2 * k] = v.x; xt[2 * k + 1] = v.y;
        }
#pragma unroll
        for (int s = 0; s < 22; ++s)
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) n1[ob] = FS_MFMA16(FS_AOP(p2 + 64 + s * 4 + ob), xt[s], n1[ob]);
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                mn |= (n1[blk][r] > 0.0f ? 1u : 0u) << (4 * blk + r);
                n1[blk][r] = fmaxf(n1[blk][r], 0.0f);
            }
        if (live) {
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) st4(sd + 8 * 64 + 16 * blk + ao, n1[blk]);
        }
    }
    }
    constexpr int p3 = p2 + 64 + 88;
    f32x4 dN[4], dZ[4], dh[4];   // pre-activation gradients of the two output layers; gradient of hid through (1 - z) * hid
    {
        f32x4 N[4];
        if constexpr (SAVED) {
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) N[blk] = ld4(ac + (2 * 4 + blk) * 256);       // q = tanh(.) itself
        } else {
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) N[blk] = ld4(bias + 5 * 64 + 16 * blk + ao);
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int ob = 0; ob < 4; ++ob) N[ob] = FS_MFMA16(FS_AOP(p3 + s * 4 + ob), n1[s >> 2][s & 3], N[ob]);
        }
        const float* go = g_fused + tr * 64;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            const f32x4 gv = ld4(go + 16 * blk + ao);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float gq = live ? gv[r] : 0.0f;
                const float z = zz[blk][r], h = hid[blk][r];
                const float qq = SAVED ? N[blk][r] : tanhf_(N[blk][r]);
                dN[blk][r] = gq * z * (1.0f - qq * qq);
                dZ[blk][r] = gq * (qq - h) * z * (1.0f - z);
                dh[blk][r] = gq * (1.0f - z);
            }
        }
    }
    if (live) {
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) { st4(sd + 5 * 64 + 16 * blk + ao, dN[blk]); st4(sd + 3 * 64 + 16 * blk + ao, dZ[blk]); }
    }
    static_assert(p3 + 64 == 696, "forward operand rows");
    // ================= transposed layers =================
    constexpr int q0 = SAVED ? 0 : 696;
    const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    f32x4 C[11];                 // dcat: feature block ob = features 16 ob + 4 g + r
    f32x4 dR[4];
    {
        f32x4 a[4] = {zero4, zero4, zero4, zero4};      // mlp_n layer 2: d relu(n1) = Wn2^T dN, masked -> dn1
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) a[ob] = FS_MFMA16(FS_AOP(q0 + s * 4 + ob), dN[s >> 2][s & 3], a[ob]);
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) a[blk][r] = (SAVED || ((mn >> (4 * blk + r)) & 1u)) ? a[blk][r] : 0.0f;
        if constexpr (SAVED) {      // (the masks from the kept post-ReLU activations: relu(x) > 0 <=> x > 0)
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) {
                const f32x4 kept = ld4(sd + 8 * 64 + 16 * blk + ao);
#pragma unroll
                for (int r = 0; r < 4; ++r) a[blk][r] = kept[r] > 0.0f ? a[blk][r] : 0.0f;
            }
        }
        if (live) {
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) st4(sd + 4 * 64 + 16 * blk + ao, a[blk]);
        }
        // mlp_n layer 1: d(r * hid) (4 blocks) and the x | xe part straight into dcat's blocks 5 .. 10
        f32x4 H[4] = {zero4, zero4, zero4, zero4};
#pragma unroll
        for (int ob = 5; ob < 11; ++ob) C[ob] = zero4;
        constexpr int q1 = q0 + 64;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float b = a[s >> 2][s & 3];
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) H[ob] = FS_MFMA16(FS_AOP(q1 + s * 10 + ob), b, H[ob]);
#pragma unroll
            for (int j = 0; j < 6; ++j) C[5 + j] = FS_MFMA16(FS_AOP(q1 + s * 10 + 4 + j), b, C[5 + j]);
        }
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float r_ = rr[blk][r];
                dR[blk][r] = H[blk][r] * hid[blk][r] * r_ * (1.0f - r_);
                dh[blk][r] += H[blk][r] * r_;
            }
    }
    if (live) {
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) st4(sd + 2 * 64 + 16 * blk + ao, dR[blk]);
    }
    constexpr int q2 = q0 + 64 + 160;
    f32x4 e[4] = {zero4, zero4, zero4, zero4}, f[4] = {zero4, zero4, zero4, zero4};   // dr1, dz1
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const float br = dR[s >> 2][s & 3], bz = dZ[s >> 2][s & 3];
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) e[ob] = FS_MFMA16(FS_AOP(q2 + s * 8 + ob), br, e[ob]);
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) f[ob] = FS_MFMA16(FS_AOP(q2 + s * 8 + 4 + ob), bz, f[ob]);
    }
#pragma unroll
    for (int blk = 0; blk < 4; ++blk)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            e[blk][r] = (SAVED || ((mr >> (4 * blk + r)) & 1u)) ? e[blk][r] : 0.0f;
            f[blk][r] = (SAVED || ((mz >> (4 * blk + r)) & 1u)) ? f[blk][r] : 0.0f;
        }
    if constexpr (SAVED) {
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            const f32x4 kr = ld4(sd + 6 * 64 + 16 * blk + ao), kz = ld4(sd + 7 * 64 + 16 * blk + ao);
#pragma unroll
            for (int r = 0; r < 4; ++r) { e[blk][r] = kr[r] > 0.0f ? e[blk][r] : 0.0f; f[blk][r] = kz[r] > 0.0f ? f[blk][r] : 0.0f; }
        }
    }
    if (live) {
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) { st4(sd + 0 * 64 + 16 * blk + ao, e[blk]); st4(sd + 1 * 64 + 16 * blk + ao, f[blk]); }
    }
    // first layers of r and z: all eleven feature blocks of dcat (blocks 0..3 start from the gate's d hid, 5..10 hold mlp_n's part)
    constexpr int q3 = q2 + 128;
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) C[blk] = dh[blk];
    C[4] = zero4;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const float br = e[s >> 2][s & 3], bz = f[s >> 2][s & 3];
#pragma unroll
        for (int ob = 0; ob < 11; ++ob) C[ob] = FS_MFMA16(FS_AOP(q3 + s * 22 + ob), br, C[ob]);
#pragma unroll
        for (int ob = 0; ob < 11; ++ob) C[ob] = FS_MFMA16(FS_AOP(q3 + s * 22 + 11 + ob), bz, C[ob]);
    }
    static_assert(q3 + 16 * 22 == (SAVED ? kS16Used - 696 : kS16Used), "operand rows");
    (void)pos;
    if (live) {
        float* dc = dcat + tr * 176;
#pragma unroll
        for (int ob = 0; ob < 11; ++ob) st4(dc + 16 * ob + ao, C[ob]);
    }
}

// ------------------------------------------------------------------------------------------------------------
// Forward of the GRU on 16-pair wavefronts (round 6; the layout of ptf_gru_bwd16_kernel): 696 v_mfma_f32_16x16x4_f32 per 16 pairs,
// ~120 registers -- FS_GRU_FWD16_WAVES wavefronts per SIMD instead of the 32-pair kernel's two.  `tab` = the forward part of the
// 16-pair operand stream (696 rows padded to whole chunks, quad-interleaved) followed by the six bias vectors
// (fs_ptf_gru_table_layout() = 1; freesplat_amd/ptf.py:gru_tables builds it).
// GATHER: k-step s of layer 1, quarter g <-> feature 44 g + s of the virtual row [hid (64) | he (24) | x (64) | xe (24)]: quarter 0 = hid[0:44],
// 1 = hid[44:64] | he, 2 = x[0:44], 3 = x[44:64] | xe; every lane computes ONE positional encoding -- he for quarters 0 - 1,
// xe for 2 - 3 (a wavefront executes both anyway) -- and mlp_n's x | xe steps (feature 88 + 22 g + s) take x from memory and xe from it.
// ------------------------------------------------------------------------------------------------------------
constexpr int kF16Chunks = (696 + kCh - 1) / kCh;
#ifndef FS_GRU_FWD16_WAVES
#define FS_GRU_FWD16_WAVES 3
#endif
template <bool GATHER, bool SAVE = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FS_GRU_FWD16_WAVES, FS_GRU_FWD16_WAVES))) void ptf_gru16_kernel(
    int n, const int32_t* __restrict__ counts, const float* __restrict__ cat, GruGather ga, const float* __restrict__ tab,
    float* __restrict__ fused, int out_after_keep)
{
    const size_t out_row0 = (out_after_keep && counts) ? (size_t)counts[0] : 0;
    if (counts) n = counts[1];  // (device-resident pair count: fs_ptf_fold_step)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int grp = blockIdx.x * 4 + wave;
    if (blockIdx.x * 64 >= n) return;   // whole workgroup beyond n (a single wavefront beyond n stays for the barriers)
    const int pn = lane & 15, g = lane >> 4;
    const int t = grp * 16 + pn;
    const bool live = t < n;
    FS_RING_SETUP(tab, kF16Chunks, 696, 0, true)
    const float* const bias = tab + (size_t)kF16Chunks * kCh * 64;     // [6][64]: br1, bz1, br2, bz2, bn1, bn2
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    auto ld4 = [&](const float* p) __attribute__((always_inline)) { const float4 v = *(const float4*)p; return f32x4{v.x, v.y, v.z, v.w}; };
    const int ao = 4 * g;
    const float* row = GATHER ? nullptr : cat + (size_t)(live ? t : 0) * 176;
    const long long gm = GATHER ? ga.fuse_idx[live ? t : 0] : 0, gp = GATHER ? ga.fuse_pix[live ? t : 0] : 0;
    const float* hrow = GATHER ? ga.G + gm * 64 : row;          // hid: 64 floats
    const float* xrow = GATHER ? ga.g_i + gp * 64 : row + 88;   // x: 64 floats (xe follows only in a materialised row)
    // SAVE (the training fold): the hidden activations the weight gradients pair with go straight into the backward's `side` rows
    // (columns 6 .. 9: relu(r1), relu(z1), relu(n1), r * hid) and the gates r, z, q into `act` -- ptf_gru_bwd16_kernel<true> then
    // runs the transposed layers only (704 instead of 1 400 MFMAs per 16 pairs)
    float* const sd = SAVE ? ga.side + (size_t)(live ? t : 0) * kSide : nullptr;
    // act is private to the two 16-pair kernels, which share their lane map: [group of 16 pairs][r, z, q][blk][lane] float4 -- every
    // store / load instruction moves 1 KB of consecutive bytes (a pair-major row would be 64-byte pieces 768 bytes apart)
    float* const ac = SAVE ? ga.act + (size_t)grp * (16 * kAct) + 4 * lane : nullptr;
    auto st4 = [&](float* p, const f32x4 v) __attribute__((always_inline)) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); };

    float xh[44];      // layer-1 inputs of this quarter: feature 44 g + s of the (virtual) row
    float pe[24];      // GATHER: he (quarters 0, 1) or xe (quarters 2, 3)
    if (GATHER) {
        pos_enc2(g < 2 ? ga.rho_i[gp] : ga.R[gm], g < 2 ? ga.O[gm] : ga.om_i[gp], pe);
        const float* src = (g < 2 ? hrow : xrow) + 44 * (g & 1);       // even quarters: 44 floats; odd: 20 floats, then the encoding
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float4 v = *(const float4*)((k < 5 || !(g & 1)) ? src + 4 * k : src);     // (no read past the 64-float row)
            xh[4 * k] = v.x; xh[4 * k + 1] = v.y; xh[4 * k + 2] = v.z; xh[4 * k + 3] = v.w;
        }
#pragma unroll
        for (int k = 0; k < 24; ++k) xh[20 + k] = (g & 1) ? pe[k] : xh[20 + k];
    } else {
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float4 v = ((const float4*)(row + 44 * g))[k];
            xh[4 * k] = v.x; xh[4 * k + 1] = v.y; xh[4 * k + 2] = v.z; xh[4 * k + 3] = v.w;
        }
    }
    f32x4 hid[4];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) hid[blk] = ld4(hrow + 16 * blk + ao);
    if constexpr (SAVE && GATHER) {
        // the pair's gathered + encoded input row [hid | he | x | xe] -- this quarter's 44 features ARE xh -- for the backward and the
        // weight gradients (saves the backward its re-gather, fs_ptf_gru_inputs: 0.21 ms per step at 968x1296)
        if (live) {
            float* cr = ga.cat_out + (size_t)t * 176 + 44 * g;
#pragma unroll
            for (int k = 0; k < 11; ++k) *(float4*)(cr + 4 * k) = make_float4(xh[4 * k], xh[4 * k + 1], xh[4 * k + 2], xh[4 * k + 3]);
        }
    }

    f32x4 r1[4], z1[4];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) { r1[blk] = ld4(bias + 0 * 64 + 16 * blk + ao); z1[blk] = ld4(bias + 1 * 64 + 16 * blk + ao); }
#pragma unroll
    for (int s = 0; s < 44; ++s) {
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) r1[ob] = FS_MFMA16(FS_AOP(s * 8 + ob), xh[s], r1[ob]);
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) z1[ob] = FS_MFMA16(FS_AOP(s * 8 + 4 + ob), xh[s], z1[ob]);
    }
    constexpr int p1 = 44 * 8;
    if constexpr (SAVE) {
        if (live) {
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) {
                f32x4 a, b;
#pragma unroll
                for (int r = 0; r < 4; ++r) { a[r] = fmaxf(r1[blk][r], 0.0f); b[r] = fmaxf(z1[blk][r], 0.0f); }
                st4(sd + 6 * 64 + 16 * blk + ao, a); st4(sd + 7 * 64 + 16 * blk + ao, b);
            }
        }
    }
    f32x4 R[4], Z[4];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) { R[blk] = ld4(bias + 2 * 64 + 16 * blk + ao); Z[blk] = ld4(bias + 3 * 64 + 16 * blk + ao); }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const float br = fmaxf(r1[s >> 2][s & 3], 0.0f), bz = fmaxf(z1[s >> 2][s & 3], 0.0f);
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) R[ob] = FS_MFMA16(FS_AOP(p1 + s * 8 + ob), br, R[ob]);
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) Z[ob] = FS_MFMA16(FS_AOP(p1 + s * 8 + 4 + ob), bz, Z[ob]);
    }
    constexpr int p2 = p1 + 16 * 8;
    f32x4 n1[4];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) n1[blk] = ld4(bias + 4 * 64 + 16 * blk + ao);
    if constexpr (SAVE) {
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            f32x4 h0;
#pragma unroll
            for (int r = 0; r < 4; ++r) { R[blk][r] = sigmoidf_(R[blk][r]); h0[r] = R[blk][r] * hid[blk][r]; }
            if (live) {
                st4(ac + (0 * 4 + blk) * 256, R[blk]);
                st4(sd + 9 * 64 + 16 * blk + ao, h0);
            }
        }
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const float b = (SAVE ? R[s >> 2][s & 3] : sigmoidf_(R[s >> 2][s & 3])) * hid[s >> 2][s & 3];
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) n1[ob] = FS_MFMA16(FS_AOP(p2 + s * 4 + ob), b, n1[ob]);
    }
    {
        float xt[22];   // k-step s, quarter g <-> feature 88 + 22 g + s = entry 22 g + s of x | xe
        if (GATHER) {
            // quarters 0, 1: x[22 g : 22 g + 22]; quarter 2: x[44:64] | xe[0:2]; quarter 3: xe[2:24] (its own encoding IS xe)
            const float* src = xrow + 22 * (g < 3 ? g : 0);
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const float2 v = *(const float2*)((g < 2 || (g == 2 && k < 10)) ? src + 2 * k : xrow);
                xt[2 * k] = v.x; xt[2 * k + 1] = v.y;
            }
#pragma unroll
            for (int s = 0; s < 22; ++s) xt[s] = g == 3 ? pe[2 + s] : ((g == 2 && s >= 20) ? pe[s - 20] : xt[s]);
        } else {
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const float2 v = ((const float2*)(row + 88 + 22 * g))[k];
                xt[2 * k] = v.x; xt[2 * k + 1] = v.y;
            }
        }
#pragma unroll
        for (int s = 0; s < 22; ++s)
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) n1[ob] = FS_MFMA16(FS_AOP(p2 + 64 + s * 4 + ob), xt[s], n1[ob]);
    }
    constexpr int p3 = p2 + 64 + 88;
    if constexpr (SAVE) {
        if (live) {
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) {
                f32x4 a;
#pragma unroll
                for (int r = 0; r < 4; ++r) a[r] = fmaxf(n1[blk][r], 0.0f);
                st4(sd + 8 * 64 + 16 * blk + ao, a);
            }
        }
    }
    f32x4 N[4];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) N[blk] = ld4(bias + 5 * 64 + 16 * blk + ao);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const float b = fmaxf(n1[s >> 2][s & 3], 0.0f);
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) N[ob] = FS_MFMA16(FS_AOP(p3 + s * 4 + ob), b, N[ob]);
    }
    static_assert(p3 + 64 == 696, "forward operand rows");
    // ---- gates: out = (1 - z) * hid + z * tanh(q), lane holds 16 units of its pair ----
    if (live) {
        float* o = fused + (out_row0 + (size_t)t) * 64;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            float v[4];
            f32x4 zs, qs;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float zz = sigmoidf_(Z[blk][r]);
                const float qq = tanhf_(N[blk][r]);
                v[r] = (1.0f - zz) * hid[blk][r] + zz * qq;
                zs[r] = zz; qs[r] = qq;
            }
            *(float4*)(o + 16 * blk + ao) = make_float4(v[0], v[1], v[2], v[3]);
            if constexpr (SAVE) { st4(ac + (1 * 4 + blk) * 256, zs); st4(ac + (2 * 4 + blk) * 256, qs); }
        }
    }
}

// FS_GRU_FWD16=0 (or FS_GRU_BWD16=0): the 32-pair forward kernel and its tables
static bool gru_fwd16()
{
    static const bool on = [] { const char* e = getenv("FS_GRU_FWD16"); return !(e && atoi(e) == 0); }() && gru_bwd16();
    return on;
}

// n pairs, or (counts != NULL) at most n_max with the actual number in counts[1] on the device
int launch_ptf_gru(int n_max, const int32_t* counts, const float* cat, const float* tables, float* fused, hipStream_t st)
{
    if (n_max <= 0) return FS_OK;
    if (gru_fwd16()) {
        hipLaunchKernelGGL(ptf_gru16_kernel<false>, dim3((n_max + 63) / 64), dim3(256), 0, st, n_max, counts, cat, GruGather{}, tables,
                           fused, 0);
    } else {
        const int groups = (n_max + 31) / 32;
        hipLaunchKernelGGL(ptf_gru_kernel<false>, dim3((groups + 3) / 4), dim3(256), 0, st, n_max, counts, cat, GruGather{}, tables,
                           fused, 0);
    }
    FS_CHECK_LAUNCH("ptf_gru_forward");
    return FS_OK;
}

int launch_ptf_gru_gather(int n_max, const int32_t* counts, const long long* fuse_idx, const long long* fuse_pix,
                          const float* G, const float* R, const float* O, const float* g_i, const float* rho_i,
                          const float* om_i, const float* tables, float* fused, bool out_after_keep, hipStream_t st,
                          float* save_side, float* save_act, float* save_cat)
{
    if (n_max <= 0) return FS_OK;
    if ((save_side != nullptr) != (save_act != nullptr) || (save_side != nullptr) != (save_cat != nullptr) || (save_side && !gru_fwd16()))
        return FS_ERR_INVALID_ARG;
    const GruGather ga{fuse_idx, fuse_pix, G, R, O, g_i, rho_i, om_i, save_side, save_act, save_cat};
    if (gru_fwd16() && save_side) {
        hipLaunchKernelGGL((ptf_gru16_kernel<true, true>), dim3((n_max + 63) / 64), dim3(256), 0, st, n_max, counts, (const float*)nullptr,
                           ga, tables, fused, out_after_keep ? 1 : 0);
    } else if (gru_fwd16()) {
        hipLaunchKernelGGL(ptf_gru16_kernel<true>, dim3((n_max + 63) / 64), dim3(256), 0, st, n_max, counts, (const float*)nullptr, ga,
                           tables, fused, out_after_keep ? 1 : 0);
    } else {
        const int groups = (n_max + 31) / 32;
        hipLaunchKernelGGL(ptf_gru_kernel<true>, dim3((groups + 3) / 4), dim3(256), 0, st, n_max, counts, (const float*)nullptr, ga,
                           tables, fused, out_after_keep ? 1 : 0);
    }
    FS_CHECK_LAUNCH("ptf_gru_gather");
    return FS_OK;
}


int launch_ptf_gru_bwd(int n, const float* cat, const float* tables, const float* stream, const float* g_fused,
                       float* dcat, float* side, hipStream_t st, const float* act = nullptr)
{
    if (n <= 0) return FS_OK;
    if (act && !gru_fwd16()) return FS_ERR_INVALID_ARG;
    if (gru_bwd16()) {      // 16 pairs per wavefront, 64 per workgroup
        const int groups = (n + 15) / 16;
        if (act)
            hipLaunchKernelGGL(ptf_gru_bwd16_kernel<true>, dim3((groups + 3) / 4), dim3(256), 0, st, n, cat, stream, g_fused, dcat, side, act);
        else
            hipLaunchKernelGGL(ptf_gru_bwd16_kernel<false>, dim3((groups + 3) / 4), dim3(256), 0, st, n, cat, stream, g_fused, dcat, side,
                               (const float*)nullptr);
    } else {
        const int groups = (n + 31) / 32;
        hipLaunchKernelGGL(ptf_gru_bwd_kernel, dim3((groups + 3) / 4), dim3(256), 0, st, n, cat, tables, stream, g_fused, dcat, side);
    }
    FS_CHECK_LAUNCH("ptf_gru_backward");
    return FS_OK;
}


// ------------------------------------------------------------------------------------------------------------
// Weight gradients of the GRU: dW = dY^T X, a contraction over ALL n pairs of the rows the backward kernel left in
// `side` and of the input rows `cat`.  Both are row-major, so with v_mfma_f32_32x32x2_f32 taking k = 2 ROWS per step the
// A operand is "lane i reads its columns of the dY block" and the B operand "lane j reads its columns of the X block":
// coalesced loads straight from L2 / HBM, no LDS, no transposition.  One wavefront owns one quarter of the 46 output tiles
//   wave 0: dr1 (2 tiles of 32 units) x cat (6 tiles of 32 features, the last one 16 wide)   -> W_r1 [64,176]
//   wave 1: dz1 x cat                                                                       -> W_z1 [64,176]
//   wave 2: dn1 x [r*hid (2) | cat[88:176] (3, the last 24 wide)]                             -> W_n1 [64,152]
//   wave 3: dR x relu(r1), dZ x relu(z1), dN x relu(n1) (2 x 2 tiles each)                    -> W_r2, W_z2, W_n2
// (192 accumulator registers) and the four wavefronts of a workgroup walk the SAME block of rows, so the rows of `cat`
// are fetched from L2 once per workgroup and found in the L1 by the other two.  The bias gradients -- column sums of the
// six dY blocks -- ride along: every A operand is added to a per-lane sum.  Each workgroup (one per CU) writes its partial
// matrices to the workspace and a second kernel adds them up, in a fixed order, into `grads`: 44 928 floats (parameter
// order, below), which the caller zeroes once per backward -- the fold steps of a scene accumulate in place.
// 445 MB of operands per 10^5 pairs: HBM-bound (~95 us + ~20 us for the partial sums).
// Replaces five split-K batched GEMMs + five chunk sums + a column sum + a concatenation of the host glue
// (13 launches, ~0.33 ms per 10^5 pairs) by two launches.
constexpr int kGr1W = 0, kGr1b = kGr1W + 64 * 176, kGr2W = kGr1b + 64, kGr2b = kGr2W + 64 * 64, kGz1W = kGr2b + 64,
              kGz1b = kGz1W + 64 * 176, kGz2W = kGz1b + 64, kGz2b = kGz2W + 64 * 64, kGn1W = kGz2b + 64,
              kGn1b = kGn1W + 64 * 152, kGn2W = kGn1b + 64, kGn2b = kGn2W + 64 * 64, kGradFloats = kGn2b + 64;
static_assert(kGradFloats == 44928, "parameter gradient layout");
constexpr int kDwU = 4;          // k-steps (2 rows each) per load group; three groups rotate
constexpr int kDwTiles = 46, kDwRaw = kDwTiles * 1024 + 12 * 32;   // floats of one workgroup's partial sums

// What one wavefront multiplies, as compile-time tables.  A load segment: lane j (= lane & 31) of row-half kk loads
// `width` consecutive floats at column col + width * j of its row (j < lim) -- so the 64 columns of a dY block arrive
// as ONE dwordx2 per lane and 128 columns of `cat` as one dwordx4, and MFMA tile e of a segment holds the columns
// col + width * j + e: the tiles interleave instead of tiling, which only the flush has to know.  (With one dword per
// tile and lane -- 8 load instructions per k-step -- the 6-bit vmcnt caps a wavefront at 63 loads = 16 KB in flight:
// too little for one wavefront per SIMD to cover HBM latency.)
struct DwSeg { int base, col, width, lim, v0; };          // base 0 = side, 1 = cat; v0 = first register of the k-step's set
struct DwB { int v, col, mul, lim; };                     // B tile: register, output columns col + mul * j for j < lim
struct DwProd { int a0, nb, t0, w, ldw, bias; DwB b[6]; };   // A tiles = registers a0, a0 + 1 (units 2 i + e); t0 = first accumulator

struct DwJobR1 {   // dr1 x cat -> W_r1 [64,176]
    static constexpr int NSEG = 3, NV = 8, NP = 1, NT = 12, T0 = 0, B0 = 0;
    static constexpr DwSeg seg[3] = {{0, 0, 2, 32, 0}, {1, 0, 4, 32, 2}, {1, 128, 2, 24, 6}};
    static constexpr DwProd prod[1] = {{0, 6, 0, kGr1W, 176, kGr1b,
                                        {{2, 0, 4, 32}, {3, 1, 4, 32}, {4, 2, 4, 32}, {5, 3, 4, 32}, {6, 128, 2, 24}, {7, 129, 2, 24}}}};
};
struct DwJobZ1 {   // dz1 x cat -> W_z1
    static constexpr int NSEG = 3, NV = 8, NP = 1, NT = 12, T0 = 12, B0 = 2;
    static constexpr DwSeg seg[3] = {{0, 64, 2, 32, 0}, {1, 0, 4, 32, 2}, {1, 128, 2, 24, 6}};
    static constexpr DwProd prod[1] = {{0, 6, 0, kGz1W, 176, kGz1b,
                                        {{2, 0, 4, 32}, {3, 1, 4, 32}, {4, 2, 4, 32}, {5, 3, 4, 32}, {6, 128, 2, 24}, {7, 129, 2, 24}}}};
};
struct DwJobN1 {   // dn1 x [r*hid | x | xe] -> W_n1 [64,152]
    static constexpr int NSEG = 4, NV = 7, NP = 1, NT = 10, T0 = 24, B0 = 4;
    static constexpr DwSeg seg[4] = {{0, 256, 2, 32, 0}, {0, 576, 2, 32, 2}, {1, 88, 2, 32, 4}, {1, 152, 1, 24, 6}};
    static constexpr DwProd prod[1] = {{0, 5, 0, kGn1W, 152, kGn1b,
                                        {{2, 0, 2, 32}, {3, 1, 2, 32}, {4, 64, 2, 32}, {5, 65, 2, 32}, {6, 128, 1, 24}, {0, 0, 0, 0}}}};
};
struct DwJob2 {    // the three second layers: dR x relu(r1), dZ x relu(z1), dN x relu(n1) -> W_r2, W_z2, W_n2 [64,64]
    static constexpr int NSEG = 6, NV = 12, NP = 3, NT = 12, T0 = 34, B0 = 6;
    static constexpr DwSeg seg[6] = {{0, 128, 2, 32, 0}, {0, 384, 2, 32, 2}, {0, 192, 2, 32, 4}, {0, 448, 2, 32, 6},
                                     {0, 320, 2, 32, 8}, {0, 512, 2, 32, 10}};
    static constexpr DwProd prod[3] = {{0, 2, 0, kGr2W, 64, kGr2b, {{2, 0, 2, 32}, {3, 1, 2, 32}}},
                                       {4, 2, 4, kGz2W, 64, kGz2b, {{6, 0, 2, 32}, {7, 1, 2, 32}}},
                                       {8, 2, 8, kGn2W, 64, kGn2b, {{10, 0, 2, 32}, {11, 1, 2, 32}}}};
};

template <class J>
__device__ __forceinline__ void dw_wave(int r0, int r1, int lane, const float* __restrict__ side,
                                        const float* __restrict__ cat, float* __restrict__ partial)
{
    const int j = lane & 31, kk = lane >> 5;
    const float* ptr[J::NSEG];
#pragma unroll
    for (int s = 0; s < J::NSEG; ++s)
        ptr[s] = (J::seg[s].base ? cat : side) + J::seg[s].col + (j < J::seg[s].lim ? J::seg[s].width * j : 0);
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 acc[J::NT];
#pragma unroll
    for (int t = 0; t < J::NT; ++t) acc[t] = zero16;
    float bs[J::NP][2];
#pragma unroll
    for (int p = 0; p < J::NP; ++p) { bs[p][0] = 0.0f; bs[p][1] = 0.0f; }

    // three load groups rotate: while one is multiplied, two are in flight (one wavefront per SIMD -- 192 accumulator
    // registers -- has nothing else to hide the HBM latency with)
    float V0[kDwU][J::NV], V1[kDwU][J::NV], V2[kDwU][J::NV];
    // Loads are never masked (a select right behind a load makes the scheduler wait for it there and the prefetch is
    // gone): rows past r1 read row r1 - 1 again and their A operand is zeroed at the MULTIPLY (0 x finite = 0); lanes
    // whose column does not exist read lane 0's and fill accumulator columns the flush never looks at.
    auto load = [&](int r, float (&V)[kDwU][J::NV]) {
#pragma unroll
        for (int u = 0; u < kDwU; ++u) {
            const size_t rr = (size_t)min(r + 2 * u + kk, r1 - 1);
#pragma unroll
            for (int s = 0; s < J::NSEG; ++s) {
                const float* q = ptr[s] + rr * (J::seg[s].base ? 176 : kSide);
                if (J::seg[s].width == 4) {
                    const float4 x = *(const float4*)q;
                    V[u][J::seg[s].v0] = x.x; V[u][J::seg[s].v0 + 1] = x.y;
                    V[u][J::seg[s].v0 + 2] = x.z; V[u][J::seg[s].v0 + 3] = x.w;
                } else if (J::seg[s].width == 2) {
                    const float2 x = *(const float2*)q;
                    V[u][J::seg[s].v0] = x.x; V[u][J::seg[s].v0 + 1] = x.y;
                } else {
                    V[u][J::seg[s].v0] = *q;
                }
            }
        }
    };
    auto mma = [&](int r, const float (&V)[kDwU][J::NV]) {
#pragma unroll
        for (int u = 0; u < kDwU; ++u) {
            const bool ok = r + 2 * u + kk < r1;
#pragma unroll
            for (int p = 0; p < J::NP; ++p)
#pragma unroll
                for (int ia = 0; ia < 2; ++ia) {
                    const float a = ok ? V[u][J::prod[p].a0 + ia] : 0.0f;
                    bs[p][ia] += a;
#pragma unroll
                    for (int t = 0; t < J::prod[p].nb; ++t)
                        acc[J::prod[p].t0 + ia * J::prod[p].nb + t] =
                            FS_MFMA(a, V[u][J::prod[p].b[t].v], acc[J::prod[p].t0 + ia * J::prod[p].nb + t]);
                }
        }
    };
    constexpr int g = 2 * kDwU;      // rows per group
    load(r0, V0);
    load(r0 + g, V1);
    __builtin_amdgcn_sched_barrier(0);
    for (int r = r0; r < r1; r += 3 * g) {
        // (no branches in here; the launch makes a workgroup's block a multiple of 3 g rows, so only the last
        // workgroup multiplies zero rows)
        // (the scheduling barriers keep the compiler from sinking each group's loads down to their first use -- it
        // minimises register pressure that way, and waits for every load right after issuing it)
        load(r + 2 * g, V2);
        __builtin_amdgcn_sched_barrier(0);
        mma(r, V0);
        __builtin_amdgcn_sched_barrier(0);
        load(r + 3 * g, V0);
        __builtin_amdgcn_sched_barrier(0);
        mma(r + g, V1);
        __builtin_amdgcn_sched_barrier(0);
        load(r + 4 * g, V1);
        __builtin_amdgcn_sched_barrier(0);
        mma(r + 2 * g, V2);
        __builtin_amdgcn_sched_barrier(0);
    }
    // The workgroup's partial sums go out as they sit in the registers (256-byte stores): [tile][q][lane], then the bias
    // sums [block][32].  ptf_gru_dw_reduce_kernel adds the workgroups' partials up, in a fixed order.
    // (Float atomics straight into `grads` from here -- 11.6 M of them from 253 workgroups -- took 146 us of this
    // kernel's 240; profiles/r4_ptf_dw_trace.txt.)
    float* P = partial + (size_t)blockIdx.x * kDwRaw;
#pragma unroll
    for (int t = 0; t < J::NT; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) P[((J::T0 + t) * 16 + q) * 64 + lane] = acc[t][q];
#pragma unroll
    for (int p = 0; p < J::NP; ++p)
#pragma unroll
        for (int ia = 0; ia < 2; ++ia) {
            const float sum = bs[p][ia] + __shfl_xor(bs[p][ia], 32);
            if (kk == 0) P[kDwTiles * 1024 + (J::B0 + 2 * p + ia) * 32 + j] = sum;
        }
}

// raw position (local tile of job J, accumulator q, lane) -> index into `grads`, or -1: accumulator q of lane (j, hf) is
// D[i = 8 (q / 4) + 4 hf + q % 4][j]; unit = 2 i + ia, column by the tile's map
template <class J>
__device__ __forceinline__ int dw_out_index(int local, int q, int lane)
{
    const int j = lane & 31, hf = lane >> 5;
    const int unit2 = 2 * (8 * (q >> 2) + 4 * hf + (q & 3));
    int out = -1;
#pragma unroll
    for (int p = 0; p < J::NP; ++p)
#pragma unroll
        for (int ia = 0; ia < 2; ++ia)
#pragma unroll
            for (int t = 0; t < J::prod[p].nb; ++t)
                if (local == J::prod[p].t0 + ia * J::prod[p].nb + t && j < J::prod[p].b[t].lim)
                    out = J::prod[p].w + (unit2 + ia) * J::prod[p].ldw + J::prod[p].b[t].col + J::prod[p].b[t].mul * j;
    return out;
}
template <class J>
__device__ __forceinline__ int dw_bias_index(int local, int j)      // local = 2 p + ia
{
    int out = -1;
#pragma unroll
    for (int p = 0; p < J::NP; ++p)
#pragma unroll
        for (int ia = 0; ia < 2; ++ia)
            if (local == 2 * p + ia) out = J::prod[p].bias + 2 * j + ia;
    return out;
}

// grads += the sum of the `wgs` partial sets.  A workgroup = 64 raw positions x 4 slices of the workgroup list.
__global__ __launch_bounds__(256) void ptf_gru_dw_reduce_kernel(int wgs, const float* __restrict__ partial,
                                                                float* __restrict__ grads)
{
    __shared__ float s_sum[4][64];
    const int xl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + xl;                 // < kDwRaw (a multiple of 64)
    float sum = 0.0f;
#pragma unroll 8
    for (int g = sl; g < wgs; g += 4) sum += partial[(size_t)g * kDwRaw + x];
    s_sum[sl][xl] = sum;
    __syncthreads();
    if (sl != 0) return;
    sum = (s_sum[0][xl] + s_sum[1][xl]) + (s_sum[2][xl] + s_sum[3][xl]);
    int out;
    if (x < kDwTiles * 1024) {
        const int tile = x >> 10, q = (x >> 6) & 15, lane = x & 63;
        out = tile < DwJobZ1::T0 ? dw_out_index<DwJobR1>(tile, q, lane)
            : tile < DwJobN1::T0 ? dw_out_index<DwJobZ1>(tile - DwJobZ1::T0, q, lane)
            : tile < DwJob2::T0 ? dw_out_index<DwJobN1>(tile - DwJobN1::T0, q, lane)
                                : dw_out_index<DwJob2>(tile - DwJob2::T0, q, lane);
    } else {
        const int y = x - kDwTiles * 1024, b = y >> 5, j = y & 31;
        out = b < DwJobZ1::B0 ? dw_bias_index<DwJobR1>(b, j)
            : b < DwJobN1::B0 ? dw_bias_index<DwJobZ1>(b - DwJobZ1::B0, j)
            : b < DwJob2::B0 ? dw_bias_index<DwJobN1>(b - DwJobN1::B0, j)
                             : dw_bias_index<DwJob2>(b - DwJob2::B0, j);
    }
    if (out >= 0) grads[out] += sum;
}

__global__ __launch_bounds__(256) void ptf_gru_dw_kernel(int n, int rows_per_wg, const float* __restrict__ cat,
                                                         const float* __restrict__ side, float* __restrict__ partial)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r0 = blockIdx.x * rows_per_wg;
    const int r1 = min(n, r0 + rows_per_wg);              // (r0 < n: the grid is ceil(n / rows_per_wg))
    if (wave == 0) dw_wave<DwJobR1>(r0, r1, lane, side, cat, partial);
    else if (wave == 1) dw_wave<DwJobZ1>(r0, r1, lane, side, cat, partial);
    else if (wave == 2) dw_wave<DwJobN1>(r0, r1, lane, side, cat, partial);
    else dw_wave<DwJob2>(r0, r1, lane, side, cat, partial);
}

void dw_grid(int n, int& wgs, int& rows)
{
    const int want = std::min(256, (n + 63) / 64);       // one wavefront per SIMD (~390 registers): one round of workgroups
    rows = ((n + want - 1) / want + 6 * kDwU - 1) / (6 * kDwU) * (6 * kDwU);   // whole turns of the three load groups
    wgs = (n + rows - 1) / rows;
}

int launch_ptf_gru_dw(int n, const float* cat, const float* side, float* grads, float* partial, hipStream_t st)
{
    if (n <= 0) return FS_OK;
    int wgs, rows;
    dw_grid(n, wgs, rows);
    hipLaunchKernelGGL(ptf_gru_dw_kernel, dim3(wgs), dim3(256), 0, st, n, rows, cat, side, partial);
    FS_CHECK_LAUNCH("ptf_gru_weight_grads");
    static_assert(kDwRaw % 64 == 0, "reduce grid");
    hipLaunchKernelGGL(ptf_gru_dw_reduce_kernel, dim3(kDwRaw / 64), dim3(256), 0, st, wgs, partial, grads);
    FS_CHECK_LAUNCH("ptf_gru_weight_grads_reduce");
    return FS_OK;
}

}  // namespace fs

using namespace fs;

FS_API int32_t fs_ptf_gru_table_rows(void) { return gru_fwd16() ? kF16Chunks * kCh + 6 : kRows; }
// 0: the 32-pair kernels' tables (operand rows of ptf_gru_kernel, then 192 bias rows); 1: the 16-pair forward's -- its 696 operand rows
// padded to whole chunks and interleaved by quads (as operand-stream layout 2), then six rows = the bias vectors
FS_API int32_t fs_ptf_gru_table_layout(void) { return gru_fwd16() ? 1 : 0; }

FS_API int fs_ptf_gru_forward(int32_t n, const float* cat, const float* tables, float* fused, void* stream_)
{
    if (n < 0) return FS_ERR_INVALID_ARG;
    if (n == 0) return FS_OK;
    if (!cat || !tables || !fused) return FS_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream_;
    ScopedStage prof_(kStPtf, st);
    return launch_ptf_gru(n, nullptr, cat, tables, fused, st);
}

FS_API int32_t fs_ptf_gru_table_t_rows(void) { return kRowsT; }
FS_API int32_t fs_ptf_gru_side_cols(void) { return kSide; }

FS_API int32_t fs_ptf_gru_stream_rows(void) { return gru_bwd16() ? kS16Chunks * kCh + 6 : kStreamChunks * kCh; }
// 0: row r of the operand stream is 64 consecutive floats; 1: interleaved -- [chunk of kCh rows][owner wavefront (4)][quad][lane (64)][4 rows]
// 2: the 16-pair backward's stream -- its 1 400 operand rows (padded to whole chunks) interleaved as in layout 1, then six rows = the
// bias vectors br1, bz1, br2, bz2, bn1, bn2 (64 floats each)
FS_API int32_t fs_ptf_gru_stream_layout(void) { return gru_bwd16() ? 2 : (FS_GRU_BWD_QUAD != 0 ? 1 : 0); }
FS_API int32_t fs_ptf_gru_stream_chunk_rows(void) { return kCh; }

FS_API int fs_ptf_gru_backward(int32_t n, const float* cat, const float* tables, const float* operand_stream,
                               const float* g_fused, float* dcat, float* side, void* stream_)
{
    if (n < 0) return FS_ERR_INVALID_ARG;
    if (n == 0) return FS_OK;
    if (!cat || !tables || !operand_stream || !g_fused || !dcat || !side) return FS_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream_;
    ScopedStage prof_(kStPtf, st);
    return launch_ptf_gru_bwd(n, cat, tables, operand_stream, g_fused, dcat, side, st);
}

// The saving training forward + the backward that re-runs nothing (round 6): fs_ptf_fold_step_save leaves, per fused pair, the `side`
// columns 6 .. 9 and fs_ptf_gru_act_cols() floats of gates; fs_ptf_gru_backward_saved then takes `stream_t` = the
// fs_ptf_gru_stream_t_rows() transposed operand rows (quad-interleaved like layout 2, no bias rows) and fills side's columns 0 .. 5.
// 0 rows = not available in this build / mode (FS_GRU_FWD16=0 or FS_GRU_BWD16=0): use fs_ptf_gru_backward.
FS_API int32_t fs_ptf_gru_act_cols(void) { return kAct; }
FS_API int32_t fs_ptf_gru_stream_t_rows(void) { return gru_fwd16() ? kT16Chunks * kCh : 0; }
FS_API int fs_ptf_gru_backward_saved(int32_t n, const float* cat, const float* stream_t, const float* act, const float* g_fused,
                                     float* dcat, float* side, void* stream_)
{
    if (n < 0 || !gru_fwd16()) return FS_ERR_INVALID_ARG;
    if (n == 0) return FS_OK;
    if (!cat || !stream_t || !act || !g_fused || !dcat || !side) return FS_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream_;
    ScopedStage prof_(kStPtf, st);
    return launch_ptf_gru_bwd(n, cat, nullptr, stream_t, g_fused, dcat, side, st, act);
}

FS_API int32_t fs_ptf_gru_grad_floats(void) { return kGradFloats; }

// grads [fs_ptf_gru_grad_floats()] += the gradients of the 12 GRU parameters over the n pairs whose rows
// fs_ptf_gru_backward left in `side` (and whose inputs are `cat`), concatenated in the order
// mlp_r[0].weight [64,176], .bias, mlp_r[2].weight [64,64], .bias, mlp_z[0] .., mlp_z[2] .., mlp_n[0].weight [64,152],
// .bias, mlp_n[2].weight, .bias.  ADDED to `grads` (zero it before the first fold step of a backward); the summation
// order is fixed, so equal inputs give equal bits.  workspace: fs_ptf_gru_weight_grads_bytes(n).
FS_API size_t fs_ptf_gru_weight_grads_bytes(int32_t n)
{
    if (n <= 0) return 0;
    int wgs, rows;
    dw_grid(n, wgs, rows);
    return (size_t)wgs * kDwRaw * sizeof(float);
}

FS_API int fs_ptf_gru_weight_grads(int32_t n, const float* cat, const float* side, float* grads, void* workspace,
                                   void* stream_)
{
    if (n < 0) return FS_ERR_INVALID_ARG;
    if (n == 0) return FS_OK;
    if (!cat || !side || !grads || !workspace) return FS_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream_;
    ScopedStage prof_(kStPtf, st);
    return launch_ptf_gru_dw(n, cat, side, grads, (float*)workspace, st);
}

// cost_volume.hip -- fused plane-sweep cost volume for MI355X (gfx950).
//
// Replaces AVGFeatureVolumeManager.build_cost_volume
// (src/model/encoder/modules/cost_volume.py:429-619 with sr_utils/geometry_utils.py:22-89 and
// the MLP of src/model/encoder/modules/networks.py:218-236): the reference runs a Python loop
// over the D planes with ~15 torch kernels each and re-materialises the warped [B*K,C,h,w]
// features every iteration.  Here one kernel does everything:
//
//   * per (pixel, source) the plane-induced homography  q(d) = d * (P[:, :3] r) + P[:, 3]  is
//     3 FMAs per plane;
//   * the averaged warped features feed the 49->32 layer on the matrix cores computed transposed (H1^T = W1 X^T), so the
//     accumulator holds, per lane, hidden units of ITS pixel -- which is directly the B operand of the 32->32 layer with
//     the k order permuted to the accumulator's row map; the final 32->1 layer is a few FMAs and a cross-lane add.  Exact
//     fp32 (= fmaf chains); the bias of layer 1 rides in the padding column of the K dimension (feature 49 := 1).
//   * three sweeps share that scheme:
//     - K = 1 (two-view configurations), cost_volume_proj_kernel: the first layer's feature block is applied once per
//       SOURCE TEXEL by the re-layout (it is linear, and so is the bilinear warp) and the sweep blends the result: 16 MFMAs
//       (32x32x2) per 32 pixels per plane instead of 41; lane = (pixel of 32, channel parity), slot-major source records;
//     - K >= 2, cost_volume16_kernel: 16-pixel wavefronts, lane = (pixel, channel quarter) with the four lanes of a pixel
//       adjacent, so that a tap load reads 64 contiguous bytes per quad (the sweep is bound by the delivery of its taps:
//       1.7x fewer cycles per load instruction than with (pixel, parity) lanes), v_mfma_f32_16x16x4_f32;
//     - backward (round 4: two passes, no global float atomics on the source maps): pass 1 -- cost_volume16_bwd_kernel (round 6:
//       the forward's 16-pixel lane orders, v_mfma_f32_16x16x4_f32, natural-order records; FS_CV_BWD16=0 or a saved-activation
//       call: cost_volume_bwd_kernel, 32 pixels x parity on texel-major [y][x][parity][C/2] records) -- recomputes the forward per
//       plane and forms the six MLP gradients on the matrix cores, d cur, and one record per (pixel, plane) point -- then
//       cv_src_grad_kernel, whose single-wavefront workgroups
//       own 8 x 8 tiles of SOURCE texels and collect, plane by plane, from the pixels whose taps cover them (found through
//       the inverse plane homography), accumulating in LDS.  (Per-pixel plane depths or K > 16: the round-3 one-kernel
//       form, which scatters with 192-byte atomic records.)
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "fs_common.h"

#ifndef FS_BWD16_SAVED_PREFETCH
#define FS_BWD16_SAVED_PREFETCH 1
#endif
#ifndef FS_REC_PIXEL_MAJOR
#define FS_REC_PIXEL_MAJOR 1     // records of the 16-pixel backward pass 1: [view, plane][pixel][C/4 float4 chunks] (0, A/B: chunk-planar)
#endif

namespace fs {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- in-kernel phase timing of the sweep (debug builds: make EXTRA=-DFS_CV_TRACE; profiles/cv_phase_trace.py) ----
// Per wavefront: shader cycles (s_memtime) between the top of a plane iteration and the point where the averaged
// features are final (gather: depth, projection, taps, reduction), and from there to the plane's output value
// (the two MFMA layers + glue), summed over the wavefront's planes.
#ifdef FS_CV_TRACE
constexpr int kCvTraceWaves = 16384;
__device__ unsigned long long g_cv_trace[kCvTraceWaves * 4];
__device__ __forceinline__ unsigned long long cv_stamp(float dep)
{
    unsigned long long t;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(dep) : "memory");
    return t;
}
#define FS_CV_T(var, dep) const unsigned long long var = cv_stamp(dep)
// backward: per wavefront the shader cycles of [0] forward recompute, [1] MLP backward, [2] weight-gradient outer
// products (LDS transposes + 48 MFMAs), [3] feature gradients (re-gather for K > 1, scatter), [4] planes, [5] total
__device__ unsigned long long g_cvb_trace[kCvTraceWaves * 6];
#else
#define FS_CV_T(var, dep) do {} while (0)
#endif

// ---- feature re-layouts between the caller's [C, h*w] maps and pixel-major [h*w][C] records, through an LDS tile ----
// One workgroup moves 64 pixels x C channels: rows of 256 contiguous bytes on the channel-major side, 64 x C x 4
// contiguous bytes on the pixel-major side (the one-thread-per-element versions wrote 4 bytes at a 4 C-byte stride:
// 0.7 TB/s, 0.25 ms per call at config-3 scale; these: profiles/r4_cv_*).  PARITY: record position of channel c is
// (c & 1) * C/2 + (c >> 1) (the backward's layout), else c (the forward's).  BACK: pixel-major -> channel-major.
template <int C, bool PARITY, bool BACK>
__global__ __launch_bounds__(256) void cv_relayout_tiled_kernel(const float* __restrict__ src, float* __restrict__ dst, int hw,
                                                                 int n_maps)
{
    constexpr int CS = C + 1;
    __shared__ float tile[64 * CS];                 // [pixel][C + 1] (odd stride: both access directions conflict-free)
    const int tiles = (hw + 63) / 64;
    for (long long blk = blockIdx.x; blk < (long long)n_maps * tiles; blk += gridDim.x) {
        const int map = (int)(blk / tiles), p0 = (int)(blk - (long long)map * tiles) * 64;
        const int np = min(64, hw - p0);
        const float* cm = (BACK ? dst : src) + (size_t)map * C * hw;          // channel-major side
        const float* pm = (BACK ? src : dst) + ((size_t)map * hw + p0) * C;    // pixel-major side
        if (!BACK) {
            for (int e = threadIdx.x; e < C * 64; e += 256) {
                const int c = e >> 6, px = e & 63;
                if (px < np) tile[px * CS + (PARITY ? (c & 1) * (C / 2) + (c >> 1) : c)] = cm[(size_t)c * hw + p0 + px];
            }
            __syncthreads();
            float* out = const_cast<float*>(pm);
            for (int e = threadIdx.x; e < np * C; e += 256) out[e] = tile[(e / C) * CS + (e % C)];
        } else {
            for (int e = threadIdx.x; e < np * C; e += 256) tile[(e / C) * CS + (e % C)] = pm[e];
            __syncthreads();
            float* out = const_cast<float*>(cm);
            for (int e = threadIdx.x; e < C * 64; e += 256) {
                const int c = e >> 6, px = e & 63;
                if (px < np) out[(size_t)c * hw + p0 + px] = tile[px * CS + (PARITY ? (c & 1) * (C / 2) + (c >> 1) : c)];
            }
        }
        __syncthreads();
    }
}
static inline void cv_relayout(bool parity, bool back, const float* src, float* dst, int C, int hw, int n_maps, hipStream_t st)
{
    const long long blocks = (long long)n_maps * ((hw + 63) / 64);
    const dim3 grid((unsigned)std::min<long long>(blocks, 65536));
    auto go = [&](auto kernel) { hipLaunchKernelGGL(kernel, grid, dim3(256), 0, st, src, dst, hw, n_maps); };
    if (C == 48) {
        if (parity) { if (back) go(cv_relayout_tiled_kernel<48, true, true>); else go(cv_relayout_tiled_kernel<48, true, false>); }
        else { if (back) go(cv_relayout_tiled_kernel<48, false, true>); else go(cv_relayout_tiled_kernel<48, false, false>); }
    } else {   // (C == 16: the entry points accept nothing else)
        if (parity) { if (back) go(cv_relayout_tiled_kernel<16, true, true>); else go(cv_relayout_tiled_kernel<16, true, false>); }
        else { if (back) go(cv_relayout_tiled_kernel<16, false, true>); else go(cv_relayout_tiled_kernel<16, false, false>); }
    }
}

// accumulator row held by (reg r, half hf) of a 32x32 MFMA result (guide, "Fragment layout")
__device__ __forceinline__ constexpr int acc_row(int r, int hf) { return (r & 3) + 8 * (r >> 2) + 4 * hf; }

// ---- source re-layout for the forward sweep, with the feature block of the MLP's first layer applied PER TEXEL ----
// The first layer is linear in the averaged warped features, and those are linear in the source texels:
//   W1f (sum_k s_k sum_tap w_tap v_tap) = sum_k s_k sum_tap w_tap (W1f v_tap),
// so U = W1f v is formed once per source texel here (48 x 32 MACs x h*w texels: nothing) instead of once per
// (pixel, plane) on the matrix cores (25 of the sweep's 41 MFMAs), and the sweep blends U with the bilinear weights it
// computes anyway.  Per texel and lane half hf: REC = C/2 + 16 floats -- the C/2 features of parity hf, then the 16
// units of U that the lane half holds as layer-2 operands (accumulator rows acc_row(r, hf)) -- stored SLOT-MAJOR:
//   dst[map][y][s][x][hf][4],  s < REC/4  (float4 slot s of the record),
// so that the 64 lanes of ONE tap load (32 neighbouring pixels x 2 halves, the same slot) read one contiguous KB when the
// source is sampled at about its own resolution: 8 cache lines per instruction.  (Texel-major records, 160 B per half,
// put every lane's 16 bytes in a line of its own -- 48 lines per instruction, re-walked by each of the 10 slot loads:
// that version of the sweep was bound by the L1 and 1.6x SLOWER at config-3 scale than the one it replaced.)
constexpr int kCvU = 16;
// one thread per (map, texel, half, group of 4 units): the texel's C channel values are read by its 8 threads
// (coalesced along the pixel index, L1 hits after the first), 4 hidden units are C FMAs each with the weights
// arriving as wave-uniform scalar loads, and the record's float4 slots are shared out over the 8 threads
template <int C>
__global__ __launch_bounds__(256) void cv_relayout_project_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                                  const float* __restrict__ w1, int h, int w, int n_maps)
{
    constexpr int HC = C / 2, REC = HC + kCvU, NS = REC / 4;
    const int hw = h * w;
    const long long total = (long long)n_maps * 8 * hw;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int pix = (int)(e % hw);
        const long long r0 = e / hw;
        const int part = (int)(r0 % 8), hf = part >> 2, u4 = part & 3;     // (wave-uniform: hw is a multiple of 64 or the
        const long long map = r0 / 8;                                      //  branchy tail below is just divergent)
        const float* sp = src + (map * C) * hw + pix;
        float v[C];
#pragma unroll
        for (int c = 0; c < C; ++c) v[c] = sp[(size_t)c * hw];
        const int y = pix / w, x = pix % w;
        float* const drow = dst + (((size_t)map * h + y) * NS * w + x) * 8 + hf * 4;   // slot s at + s * w * 8
#pragma unroll
        for (int s4 = 0; s4 < HC / 4; ++s4)
            if ((s4 & 3) == u4)
                *(float4*)(drow + (size_t)s4 * w * 8) =
                    make_float4(hf ? v[8 * s4 + 1] : v[8 * s4], hf ? v[8 * s4 + 3] : v[8 * s4 + 2],
                                hf ? v[8 * s4 + 5] : v[8 * s4 + 4], hf ? v[8 * s4 + 7] : v[8 * s4 + 6]);
        float o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float* wr = w1 + (size_t)acc_row(4 * u4 + q, hf) * (C + 1);
            float a = 0.0f;
#pragma unroll
            for (int c = 0; c < C; ++c) a = fmaf(wr[c], v[c], a);
            o[q] = a;
        }
        *(float4*)(drow + (size_t)(HC / 4 + u4) * w * 8) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// P = (K_src @ T_src<-cur)[:3, :] per (b, k)  (geometry_utils.py:78-80): computed once per call
__global__ void cv_proj_kernel(int n, const float* __restrict__ src_Ks, const float* __restrict__ src_extrinsics,
                               float* __restrict__ P)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * 12) return;
    const int m = e / 12, i = (e % 12) / 4, j = e % 4;
    const float* Ks = src_Ks + (size_t)m * 16;
    const float* Tx = src_extrinsics + (size_t)m * 16;
    P[e] = Ks[4 * i] * Tx[j] + Ks[4 * i + 1] * Tx[4 + j] + Ks[4 * i + 2] * Tx[8 + j] + Ks[4 * i + 3] * Tx[12 + j];
}

// Workgroup -> (batch row, 32-pixel group, plane slice), XCD-aware.  Workgroup i runs on XCD i % 8 (hardware round
// robin), and every XCD has its own 4 MB L2: with the plain (row * groups + group, slice) grid, neighbouring pixel groups
// -- which sample the SAME rows of the source maps -- landed on eight different XCDs, every L2 had to hold all K source
// maps of a view (10 views, K = 8: 19 MB) and the sweep re-fetched them from memory over and over: 16.7 GB of fetches per
// call against 0.27 GB of algorithmic bytes (profiles/r3_traffic.json, cv_fvt10_K8).  Here XCD x owns the x-th horizontal
// BAND of pixel groups of every view and walks it in row-major order, plane slice by plane slice: the workgroups in
// flight on an XCD share a few source rows per map.
struct CvBlock { int b, grp, slice; bool ok; };
__device__ __forceinline__ CvBlock cv_block(int B, int groups, int slices)
{
#ifdef FS_CV_LINEAR_GRID   // (A/B builds: round 2's order -- group fastest, then batch row, then slice)
    {
        CvBlock o;
        const int id = (int)blockIdx.x, per = B * groups;
        o.slice = id / per;
        o.b = (id - o.slice * per) / groups;
        o.grp = id - o.slice * per - o.b * groups;
        o.ok = o.slice < slices;
        return o;
    }
#endif
    const int xcd = (int)(blockIdx.x & 7u), j = (int)(blockIdx.x >> 3);
    const int gb = (groups + 7) >> 3;          // pixel groups per band
    const int per_b = slices * gb;
    CvBlock o;
    o.b = j / per_b;
    const int r = j - o.b * per_b;
    // (plane slice fastest instead -- the slices of one pixel group side by side, ~0.6 MB of source rows in flight per XCD
    //  instead of ~3.3 MB at K = 8 -- measured the same: the sweep is not bound by L2 misses)
    o.slice = r / gb;
    o.grp = xcd * gb + (r - o.slice * gb);
    o.ok = o.b < B && o.grp < groups;
    return o;
}
static inline unsigned cv_grid(int B, int groups, int slices) { return 8u * (unsigned)B * (unsigned)slices * (unsigned)((groups + 7) >> 3); }

// LeakyReLU(0.01) = 0.505 x + 0.495 |x|: two VALU operations (|x| is a free source modifier).  fmaxf(x, 0.01 x) costs four
// here (the multiply, the max and two NaN-canonicalising v_max that IEEE mode puts in front of it).
__device__ __forceinline__ float lrelu(float x) { return fmaf(0.495f, fabsf(x), 0.505f * x); }

// ==========================================================================================
// The K >= 2 sweep on 16-pixel wavefronts: lane = (pixel j of 16, QUARTER c of its channels), the four lanes of a pixel
// ADJACENT (lane = 4 j + c), source records texel-major in natural channel order.  One tap load instruction then reads,
// per pixel, 64 contiguous bytes with the four lanes of one quad (bytes [64 s + 16 c, +16) of the texel's record, s < C/16)
// -- the texture path coalesces a quad's lanes, not the 16-byte chunks of lanes 32 apart: profiles/tools/tap_pattern_rate.hip
// measures 8.2 ns per wave load instruction per CU for this pattern against 13.9 ns for (pixel of 32, channel parity) lanes
// on [parity][C/2] records -- the sweep this one replaced: 10 views, K = 8 at 96x128: 5.60 -> 4.07 ms; config-3 scale,
// K = 2: 3.6 -> 3.45 ms -- at any sampling scale (a chunk-planar map, which coalesces NEIGHBOURING pixels' chunks instead,
// is as fast at <= 1 texel per pixel and slower than either from 2 texels per pixel on: the round-2 slot-major experiment).
// The gather, the matching score (a quad reduction: two DPP adds) and the average over the sources all happen in that lane
// order; once per plane the C/4 + 1 averaged values of a lane move to the operand order of v_mfma_f32_16x16x4_f32
// (lane = pixel n + 16 k, k = the quarter) with one ds_bpermute each -- a fixed rotation of the lane index bits.
//   layer 1: H1^T[32 units][16 px] = W1 X^T: 2 row blocks x (C/4 + 1) k-steps of 4 channels (step t, quarter k -> channel
//            16 (t / 4) + 4 k + t % 4; last step: dot, 1, 0, 0);  accumulator: lane (n, g) holds units 16 blk + 4 g + r
//   layer 2: those registers ARE the B operands of W2 (k-step (blk, r), quarter g -> unit 16 blk + 4 g + r): 2 x 8 steps
//   layer 3: 8 FMAs per lane + the sum over the four quarters.
// ==========================================================================================
typedef float f32x4 __attribute__((ext_vector_type(4)));


// SAVE (training forward): the MLP's input of every (pixel, plane) point -- the averaged warped features x = favg / cnt,
// the averaged score and the sources' (valid, in-front) bits -- is kept for the backward, which then starts from it
// instead of gathering K x 4 taps again (chunk-planar, natural channel order: cost_volume16_bwd_kernel<C, true>).
// Two shapes of the sweep (round 6, profiles/r6_cv_fwd_taps_ab.txt).  Both read the W1 / W2 operands from per-lane LDS images where
// their MFMA is (42 registers freed).  TAPS = 1: a source's four bilinear taps one after the other (each behind its branch), 124
// registers, FOUR wavefronts per SIMD -- best at K <= 2, where a wavefront's chain of 8 dependent tap round trips per plane is what
// the other wavefronts hide.  TAPS = 4: all four taps' loads in flight before the first is blended, 152 registers, three wavefronts --
// best from K = 3 up, where the L1 / texture path is the bound and a deeper queue per wavefront feeds it better.
#ifndef FS_FWD16_W_LDS
#define FS_FWD16_W_LDS 2       // 0: W1 / W2 operands in registers (42); 1: W2's 16 from an LDS image; 2: both from LDS images
#endif
template <int C, bool SAVE, int TAPS, int WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void cost_volume16_kernel(
    int B, int K, int h, int w, int D, int slices, const float* __restrict__ curN, const float* __restrict__ srcN,
    const float* __restrict__ Pmat,
    const float* __restrict__ cur_invK, const float* __restrict__ planes, long long ps_b,
    long long ps_d, long long ps_p, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ w2, const float* __restrict__ b2, const float* __restrict__ w3,
    const float* __restrict__ b3, float* __restrict__ out, float* __restrict__ xs, float2* __restrict__ xm,
    uint32_t* __restrict__ xhdr)
{
    constexpr int NS = C / 16;          // tap load instructions (float4 per lane each)
    constexpr int NR = C / 4;           // channels per lane
    constexpr int NT = NR + 1;          // k-steps of layer 1 (the last one: dot, 1, 0, 0)
    const int hw = h * w;
    const int groups = (hw + 15) / 16;
    const CvBlock blk_ = cv_block(B, groups, slices);
    if (!blk_.ok) return;   // (workgroup-uniform)
    const int b = blk_.b, grp = blk_.grp;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // gather order
    const int j = lane >> 2, c = lane & 3;
    const int pix = grp * 16 + j;
    const bool live = pix < hw;
    const int pu = live ? pix % w : 0, pv = live ? pix / w : 0;
    // operand order
    const int n = lane & 15, g = lane >> 4;
    const int pix_m = grp * 16 + n;
    const int pull = (4 * n + g) * 4;   // ds_bpermute address: operand lane (n, g) takes gather lane 4 n + g

    // ---- MLP weights in registers, in MFMA A-operand order (lane = (row n of the block, k = g)) ----
    // (round 6: FS_FWD16_W_LDS >= 1 keeps W2's -- 2: also W1's -- operands as per-lane LDS images shared by the four wavefronts, one
    //  ds_read_b32 where the MFMA is: the registers go to tap loads in flight, profiles/r6_cv_fwd_taps_ab.txt)
    constexpr bool kW2Lds = FS_FWD16_W_LDS >= 1, kW1Lds = FS_FWD16_W_LDS >= 2;
    __shared__ float s_w[(kW2Lds ? 16 : 0) * 64 + (kW1Lds ? 2 * NT : 0) * 64 + 64];
    float* const sA2 = s_w + lane;                          // [(blk * 8 + t) * 64]
    float* const sA1 = s_w + (kW2Lds ? 16 * 64 : 0) + lane; // [(blk * NT + t) * 64]
    float a1[kW1Lds ? 1 : 2][kW1Lds ? 1 : NT], a2[kW2Lds ? 1 : 2][kW2Lds ? 1 : 8], w3v[2][4], b2v[2][4];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const int u = 16 * blk + n;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float v = t < NR ? w1[u * (C + 1) + 16 * (t >> 2) + 4 * g + (t & 3)] : (g == 0 ? w1[u * (C + 1) + C] : (g == 1 ? b1[u] : 0.0f));
            if constexpr (kW1Lds) { if (wave == 0) sA1[(blk * NT + t) * 64] = v; } else a1[blk][t] = v;
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float v = w2[u * 32 + 16 * (t >> 2) + 4 * g + (t & 3)];
            if constexpr (kW2Lds) { if (wave == 0) sA2[(blk * 8 + t) * 64] = v; } else a2[blk][t] = v;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { w3v[blk][r] = w3[16 * blk + 4 * g + r]; b2v[blk][r] = b2[16 * blk + 4 * g + r]; }
    }
    const float b3v = b3[0];
    if constexpr (kW2Lds) __syncthreads();

    // ---- current-view feature: this lane's quarter of its pixel's channels ----
    float cur[NR];
    {
        const float4* q = (const float4*)(curN + ((size_t)b * hw + (live ? pix : 0)) * C) + c;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float4 v = q[4 * s];
            cur[4 * s] = v.x; cur[4 * s + 1] = v.y; cur[4 * s + 2] = v.z; cur[4 * s + 3] = v.w;
        }
    }
    // ---- ray r = invK[:3,:3] (u+.5, v+.5, 1) ----
    const float* iK = cur_invK + (size_t)b * 16;
    const float ux = (float)pu + 0.5f, vy = (float)pv + 0.5f;
    const float rx = iK[0] * ux + iK[1] * vy + iK[2];
    const float ry = iK[4] * ux + iK[5] * vy + iK[6];
    const float rz = iK[8] * ux + iK[9] * vy + iK[10];

    const float inv_w = (float)(1.0 / (double)w), inv_h = (float)(1.0 / (double)h);
    // planes [d0, d1) of this wavefront: slices * 4 wavefronts share the D planes of a pixel group
    const int dchunk = (D + slices * 4 - 1) / (slices * 4);
    const int d0 = min(D, (blk_.slice * 4 + wave) * dchunk), d1 = min(D, d0 + dchunk);

    // Per plane the sweep would pay three dependent memory round trips before its first MFMA: the plane's depth, the
    // projection rows, then the taps.  The depth is fetched one plane ahead and the projection rows of the first group of
    // four sources stay in registers for the whole sweep (one source per lane of the quad; the shipped configs have K <= 4).
    const float* pl = planes + b * ps_b + (live ? pix : 0) * ps_p;
    float depth_next = d0 < d1 ? pl[d0 * ps_d] : 0.0f;
    float Pq[12];      // the rows of source min(c, K - 1): what this lane projects for the first group of four sources
#pragma unroll
    for (int e = 0; e < 12; ++e) Pq[e] = Pmat[((size_t)b * K + min(c, K - 1)) * 12 + e];
#ifdef FS_CV_TRACE
    unsigned long long tr_g = 0, tr_m = 0;
    const unsigned long long tr_c0 = cv_stamp(rx), tr_w0 = wall_clock64();
#endif
    for (int d = d0; d < d1; ++d) {
        FS_CV_T(t_top, rx);
        const float depth = depth_next;
        depth_next = pl[min(d + 1, d1 - 1) * ps_d];
        // (one LDS store per plane into the image array's padding, through an index the compiler cannot relate to the operand reads:
        //  without it the loop-invariant ds_reads are hoisted out of the plane loop -- back into the registers they were to free)
        if constexpr (kW2Lds) s_w[(kW2Lds ? 16 : 0) * 64 + (kW1Lds ? 2 * NT : 0) * 64 + (lane ^ 1)] = depth;
        float favg[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) favg[r] = 0.0f;
        float dot_sum = 0.0f, cnt = 0.0f;
        uint32_t flags = 0;   // SAVE: bit 2k = source k valid (dot != 0), bit 2k+1 = in front of it (z > 0)
        // One source's sampling position: base texel offset, bilinear fractions, tap validity + in-front bits.
        struct Proj { uint32_t off; float tx, ty; uint32_t bits; };   // bits: xin0 | xin1 << 1 | yin0 << 2 | yin1 << 3 | (z > 0) << 4
        auto project = [&](const float* P) __attribute__((always_inline)) -> Proj {
            // world point = depth * r (homogeneous 1): geometry_utils.py:56-58
            const float X = depth * rx, Y = depth * ry, Z = depth * rz;
            const float qx = P[0] * X + P[1] * Y + P[2] * Z + P[3];
            const float qy = P[4] * X + P[5] * Y + P[6] * Z + P[7];
            const float qz = P[8] * X + P[9] * Y + P[10] * Z + P[11];
            const float zz = qz + 1e-8f;                                  // :84
            const float sc = (fabsf(qz) > 1e-8f) ? 1.0f / zz : 1.0f;       // :83,85
            // cost_volume.py:536: uv = 2 * pix * (1/size) - 1, then grid_sample(align_corners=False)
            // un-normalises with ((uv + 1) * size - 1) / 2 (= pix - 0.5 in exact arithmetic).  The
            // reference's rounding sequence is mirrored op by op (no contraction): whether a tap is just
            // inside or outside the source image decides `dot != 0`, i.e. the validity count.
            const float uvx = __fsub_rn(__fmul_rn(__fmul_rn(2.0f, __fmul_rn(qx, sc)), inv_w), 1.0f);
            const float uvy = __fsub_rn(__fmul_rn(__fmul_rn(2.0f, __fmul_rn(qy, sc)), inv_h), 1.0f);
            const float ix = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(uvx, 1.0f), (float)w), 1.0f), 0.5f);
            const float iy = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(uvy, 1.0f), (float)h), 1.0f), 0.5f);
            const float fx0 = floorf(ix), fy0 = floorf(iy);
            // NaN/inf coordinates sample nothing (comparisons false)
            const bool xin0 = fx0 >= 0.0f && fx0 <= (float)(w - 1), xin1 = fx0 >= -1.0f && fx0 <= (float)(w - 2);
            const bool yin0 = fy0 >= 0.0f && fy0 <= (float)(h - 1), yin1 = fy0 >= -1.0f && fy0 <= (float)(h - 2);
            const int x0 = xin0 || xin1 ? (int)fx0 : 0, y0 = yin0 || yin1 ? (int)fy0 : 0;
            Proj pr;
            pr.off = (uint32_t)((y0 * w + x0) * C) * 4u;      // (byte offset of the texel record; the lane adds its quarter)
            pr.tx = ix - fx0; pr.ty = iy - fy0;
            pr.bits = (xin0 ? 1u : 0u) | (xin1 ? 2u : 0u) | (yin0 ? 4u : 0u) | (yin1 ? 8u : 0u) | (zz > 0.0f ? 16u : 0u);
            return pr;
        };
        auto gather = [&](int k, const Proj pr) __attribute__((always_inline)) {
            const bool xin0 = pr.bits & 1u, xin1 = pr.bits & 2u, yin0 = pr.bits & 4u, yin1 = pr.bits & 8u, front = pr.bits & 16u;
            const float tx = pr.tx, ty = pr.ty;
            float wv[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) wv[r] = 0.0f;
            // wave-uniform map base (SGPR pair) + a 32-bit per-lane byte offset: one address add per tap instead of
            // 64-bit multiply-adds (a source map is far below 4 GB)
            const char* base = (const char*)(srcN + (((size_t)b * K + k) * hw) * C);
            const uint32_t off0 = pr.off + 16u * (uint32_t)c;
            // (tap by tap behind a branch each: with three wavefronts per SIMD that beats having a source's taps in flight together --
            //  the backward's form, 2 wavefronts per SIMD -- which costs the third wavefront: profiles/r6_cv_fwd_taps_ab.txt)
            if constexpr (TAPS == 1) {
#pragma unroll
            for (int tap = 0; tap < 4; ++tap) {
                const int ox = tap & 1, oy = tap >> 1;
                const bool ok = live && (ox ? xin1 : xin0) && (oy ? yin1 : yin0);
                const float wt = (ox ? tx : 1.0f - tx) * (oy ? ty : 1.0f - ty);
                if (ok) {
                    const float4* q = (const float4*)(base + (off0 + (uint32_t)((oy * w + ox) * C) * 4u));
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        const float4 v = q[4 * s];
                        wv[4 * s] += wt * v.x; wv[4 * s + 1] += wt * v.y;
                        wv[4 * s + 2] += wt * v.z; wv[4 * s + 3] += wt * v.w;
                    }
                }
            }
            } else {
            // TAPS taps' loads are issued before the first of them is blended (a tap outside the image loads nothing: its
            // registers are zero and so is its weight's contribution) -- 4 / TAPS dependent round trips per source instead of 4
#pragma unroll
            for (int t0 = 0; t0 < 4; t0 += TAPS) {
                float4 v[TAPS][NS];
                float wt[TAPS];
#pragma unroll
                for (int e = 0; e < TAPS; ++e) {
                    const int tap = t0 + e, ox = tap & 1, oy = tap >> 1;
                    const bool ok = live && (ox ? xin1 : xin0) && (oy ? yin1 : yin0);
                    wt[e] = (ox ? tx : 1.0f - tx) * (oy ? ty : 1.0f - ty);
#pragma unroll
                    for (int s = 0; s < NS; ++s) v[e][s] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    if (ok) {
                        const float4* q = (const float4*)(base + (off0 + (uint32_t)((oy * w + ox) * C) * 4u));
#pragma unroll
                        for (int s = 0; s < NS; ++s) v[e][s] = q[4 * s];
                    }
                }
#pragma unroll
                for (int e = 0; e < TAPS; ++e)
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        wv[4 * s] += wt[e] * v[e][s].x; wv[4 * s + 1] += wt[e] * v[e][s].y;
                        wv[4 * s + 2] += wt[e] * v[e][s].z; wv[4 * s + 3] += wt[e] * v[e][s].w;
                    }
            }
            }
            float part = 0.0f;
#pragma unroll
            for (int r = 0; r < NR; ++r) part += wv[r] * cur[r];
            // sum over the pixel's four lanes (one quad)
            part += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(part), 0xB1, 0xF, 0xF, true));   // lane ^ 1
            part += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(part), 0x4E, 0xF, 0xF, true));   // lane ^ 2
            const float dotk = front ? part : 0.0f;                       // cost_volume.py:571-572,589-593
            if (SAVE) {
                flags |= (front ? 2u : 0u) << (2 * k);
                // in front, some tap inside the source image, and still an exactly zero score (all-zero features): the
                // backward must re-gather such a source (cost_volume_bwd_kernel); flagged once per call, practically never
                if (live && front && dotk == 0.0f && (xin0 || xin1) && (yin0 || yin1) && c == 0) atomicOr(xhdr, 1u);
            }
            if (dotk != 0.0f) {                                           // :595 (exact zero test)
                if (SAVE) flags |= 1u << (2 * k);
                cnt += 1.0f;
                dot_sum += dotk;
#pragma unroll
                for (int r = 0; r < NR; ++r) favg[r] += wv[r];
            }
        };
        // The four lanes of a pixel's quad used to compute the SAME projection for every source (~50 of the ~125 VALU
        // operations a source costs).  Now lane c of the quad projects source k0 + c of a group of four and the four results go
        // round the quad with DPP broadcasts (4 moves per source): config-3 scale K = 2 2.86 -> 2.60 ms, 5 views K = 4 1.19 -> 1.07,
        // 10 views K = 8 3.76 -> 3.46 (profiles/r4_cv_quadproj_ab.txt).
        for (int k0 = 0; k0 < K; k0 += 4) {
            float Pk[12];      // this lane's source of the group: the first group's rows stay in registers for the whole sweep
#pragma unroll
            for (int e = 0; e < 12; ++e) Pk[e] = k0 == 0 ? Pq[e] : Pmat[((size_t)b * K + min(k0 + c, K - 1)) * 12 + e];
            const Proj mine = project(Pk);
            auto from = [&](auto sel) __attribute__((always_inline)) {
                constexpr int q = decltype(sel)::value, ctl = q * 0x55;       // quad_perm: every lane reads lane q of its quad
                Proj pr;
                pr.off = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine.off, ctl, 0xF, 0xF, true);
                pr.tx = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mine.tx), ctl, 0xF, 0xF, true));
                pr.ty = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mine.ty), ctl, 0xF, 0xF, true));
                pr.bits = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine.bits, ctl, 0xF, 0xF, true);
                return pr;
            };
            gather(k0, from(std::integral_constant<int, 0>{}));
            if (k0 + 1 < K) gather(k0 + 1, from(std::integral_constant<int, 1>{}));
            if (k0 + 2 < K) gather(k0 + 2, from(std::integral_constant<int, 2>{}));
            if (k0 + 3 < K) gather(k0 + 3, from(std::integral_constant<int, 3>{}));
        }
        const float inv = 1.0f / (cnt + 1e-8f);                          // :595-598
        FS_CV_T(t_gath, favg[0] + favg[NR - 1] + inv + dot_sum);
        if (SAVE && live) {
            // natural channel order, chunk-planar: this lane's channels 16 s + 4 c + 0..3 are float4 chunk 4 s + c of the point's
            // C/4 chunks ([view, plane][chunk][pixel]: 16 pixels of a chunk are 256 contiguous bytes) -- what a lane (n, g) of
            // cost_volume16_bwd_kernel<C, true> takes as its layer-1 operands, one float4 per 16-channel block
            const size_t pl = (size_t)b * D + d;
#pragma unroll
            for (int s = 0; s < NS; ++s)
                ((float4*)xs)[(pl * (C / 4) + 4 * s + c) * hw + pix] =
                    make_float4(favg[4 * s] * inv, favg[4 * s + 1] * inv, favg[4 * s + 2] * inv, favg[4 * s + 3] * inv);
            if (c == 0) xm[pl * hw + pix] = make_float2(dot_sum * inv, __uint_as_float(flags));
        }
        // ---- to the operand order: lane (n, g) takes quarter g of pixel n ----
        float xop[NT];
#pragma unroll
        for (int r = 0; r < NR; ++r) xop[r] = __int_as_float(__builtin_amdgcn_ds_bpermute(pull, __float_as_int(favg[r] * inv)));
        xop[NR] = __int_as_float(__builtin_amdgcn_ds_bpermute(pull, __float_as_int(c == 0 ? dot_sum * inv : (c == 1 ? 1.0f : 0.0f))));
        // ---- layer 1 ----
        f32x4 h1[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            h1[blk] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int t = 0; t < NT; ++t)
                h1[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(kW1Lds ? sA1[(blk * NT + t) * 64] : a1[kW1Lds ? 0 : blk][kW1Lds ? 0 : t], xop[t], h1[blk], 0, 0, 0);
        }
        // ---- layer 2 ----
        f32x4 h2[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            h2[blk] = f32x4{b2v[blk][0], b2v[blk][1], b2v[blk][2], b2v[blk][3]};
#pragma unroll
            for (int t = 0; t < 8; ++t)
                h2[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(kW2Lds ? sA2[(blk * 8 + t) * 64] : a2[kW2Lds ? 0 : blk][kW2Lds ? 0 : t], lrelu(h1[t >> 2][t & 3]), h2[blk], 0, 0, 0);
        }
        // ---- layer 3 ----
        float o = 0.0f;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) o += w3v[blk][r] * lrelu(h2[blk][r]);
        o += __shfl_xor(o, 16, 64);
        o += __shfl_xor(o, 32, 64);
        if (g == 0 && pix_m < hw) out[((size_t)b * D + d) * hw + pix_m] = o + b3v;
#ifdef FS_CV_TRACE
        FS_CV_T(t_end, o);
        tr_g += t_gath - t_top;
        tr_m += t_end - t_gath;
#endif
    }
#ifdef FS_CV_TRACE
    {
        const int wid = (int)blockIdx.x * 4 + wave;
        if (lane == 0 && wid < kCvTraceWaves) {
            g_cv_trace[4 * wid] = tr_g; g_cv_trace[4 * wid + 1] = tr_m;
            // [3]: shader ticks (s_memtime) << 32 | 100 MHz wall ticks of this wavefront's whole sweep -> effective clock
            const unsigned long long dc = cv_stamp(rx) - tr_c0, dw = wall_clock64() - tr_w0;
            g_cv_trace[4 * wid + 2] = (unsigned long long)(d1 - d0); g_cv_trace[4 * wid + 3] = (dc << 32) | (dw & 0xffffffffull);
        }
    }
#endif
}

// The sweep with the first layer's feature block folded into the source records (cv_relayout_project_kernel): 16 MFMAs
// per (32-pixel group, plane) instead of 41, 160 instead of 96 bytes per tap and lane.  Used for K = 1 (the reference's
// two-view configurations): with more sources per view the extra tap bytes outweigh the 25 MFMAs saved per plane
// (config-3 scale, K = 2: on par; 10 views, K = 8: 18 % slower than the 32-pixel texel-major sweep of round 2, which the
// 16-pixel sweep above has since beaten by another 27 %).
template <int HC>  // HC = C/2 channels per lane
__global__ __launch_bounds__(256) void cost_volume_proj_kernel(
    int B, int K, int h, int w, int D, int slices, const float* __restrict__ cur_feats, const float* __restrict__ srcT,
    const float* __restrict__ src_Ks, const float* __restrict__ src_extrinsics,
    const float* __restrict__ cur_invK, const float* __restrict__ planes, long long ps_b,
    long long ps_d, long long ps_p, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ w2, const float* __restrict__ b2, const float* __restrict__ w3,
    const float* __restrict__ b3, float* __restrict__ out)
{
    constexpr int C = 2 * HC;
    const int hw = h * w;
    const int groups = (hw + 31) / 32;
    const CvBlock blk_ = cv_block(B, groups, slices);
    if (!blk_.ok) return;   // (workgroup-uniform)
    const int b = blk_.b, grp = blk_.grp;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int p = lane & 31, hf = lane >> 5;
    const int pix = grp * 32 + p;
    const bool live = pix < hw;
    const int pu = live ? pix % w : 0, pv = live ? pix / w : 0;

    // ---- MLP weights in registers: layer 2 in MFMA A-operand order; of layer 1 only the dot column and the bias (its
    //      feature block is already folded into the source records, cv_relayout_project_kernel) ----
    float a2[16], w3r[16], b2r[16], w1d[16], b1r[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        a2[s] = w2[p * 32 + acc_row(s, hf)];  // W2[i=p][k = unit held as reg s by half hf]
        w3r[s] = w3[acc_row(s, hf)];
        b2r[s] = b2[acc_row(s, hf)];
        w1d[s] = w1[acc_row(s, hf) * (C + 1) + C];
        b1r[s] = b1[acc_row(s, hf)];
    }
    const float b3v = b3[0];

    // ---- current-view feature (this lane's parity), straight from the caller's [C, h, w] map: read once per
    //      wavefront, the 32 pixels of a channel are one 128-byte segment -- no re-laid-out copy of the current view ----
    float cur[HC];
    {
        const float* q = cur_feats + ((size_t)b * C + hf) * hw + (live ? pix : 0);
#pragma unroll
        for (int s = 0; s < HC; ++s) cur[s] = q[(size_t)(2 * s) * hw];
    }
    // ---- ray r = invK[:3,:3] (u+.5, v+.5, 1) ----
    const float* iK = cur_invK + (size_t)b * 16;
    const float ux = (float)pu + 0.5f, vy = (float)pv + 0.5f;
    const float rx = iK[0] * ux + iK[1] * vy + iK[2];
    const float ry = iK[4] * ux + iK[5] * vy + iK[6];
    const float rz = iK[8] * ux + iK[9] * vy + iK[10];

    const float inv_w = (float)(1.0 / (double)w), inv_h = (float)(1.0 / (double)h);
    // planes [d0, d1) of this wavefront: slices * 4 wavefronts share the D planes of a pixel group
    const int dchunk = (D + slices * 4 - 1) / (slices * 4);
    const int d0 = min(D, (blk_.slice * 4 + wave) * dchunk), d1 = min(D, d0 + dchunk);

    // Per plane the sweep used to pay three dependent memory round trips before its first MFMA: the plane's depth,
    // the projection rows (scalar loads), then the taps.  The depth is now fetched one plane ahead and the projection
    // rows of the first two sources stay in SGPRs for the whole sweep (the shipped configs have K <= 2 except the
    // 9-nearest selection of config 4).
    const float* pl = planes + b * ps_b + (live ? pix : 0) * ps_p;
    float depth_next = d0 < d1 ? pl[d0 * ps_d] : 0.0f;
    // P = (K_src @ T_src<-cur)[:3, :] (geometry_utils.py:78-80; the expression of cv_proj_kernel) formed here: 48 FMAs
    // per source and wavefront instead of a launch of its own
    auto proj_row = [&](int k, float* P) __attribute__((always_inline)) {
        const float* Ks = src_Ks + ((size_t)b * K + k) * 16;
        const float* Tx = src_extrinsics + ((size_t)b * K + k) * 16;
#pragma unroll
        for (int e = 0; e < 12; ++e) {
            const int i = e / 4, j = e % 4;
            const float v = Ks[4 * i] * Tx[j] + Ks[4 * i + 1] * Tx[4 + j] + Ks[4 * i + 2] * Tx[8 + j] + Ks[4 * i + 3] * Tx[12 + j];
            P[e] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));   // wave-uniform: keep it in an SGPR
        }
    };
    float P0[12];      // (only the first source's rows stay resident: this sweep is the K = 1 path)
    proj_row(0, P0);
#ifdef FS_CV_TRACE
    unsigned long long tr_g = 0, tr_m = 0;
    const unsigned long long tr_c0 = cv_stamp(rx), tr_w0 = wall_clock64();
#endif
    for (int d = d0; d < d1; ++d) {
        FS_CV_T(t_top, rx);
        const float depth = depth_next;
        depth_next = pl[min(d + 1, d1 - 1) * ps_d];
        float uavg[kCvU];   // sum over the valid sources of the bilinearly blended U (this half's 16 hidden units)
#pragma unroll
        for (int r = 0; r < kCvU; ++r) uavg[r] = 0.0f;
        float dot_sum = 0.0f, cnt = 0.0f;
        auto one_source = [&](int k, const float* P) __attribute__((always_inline)) {

            // world point = depth * r (homogeneous 1): geometry_utils.py:56-58
            const float X = depth * rx, Y = depth * ry, Z = depth * rz;
            const float qx = P[0] * X + P[1] * Y + P[2] * Z + P[3];
            const float qy = P[4] * X + P[5] * Y + P[6] * Z + P[7];
            const float qz = P[8] * X + P[9] * Y + P[10] * Z + P[11];
            const float zz = qz + 1e-8f;                                  // :84
            const float sc = (fabsf(qz) > 1e-8f) ? 1.0f / zz : 1.0f;       // :83,85
            // cost_volume.py:536: uv = 2 * pix * (1/size) - 1, then grid_sample(align_corners=False)
            // un-normalises with ((uv + 1) * size - 1) / 2 (= pix - 0.5 in exact arithmetic).  The
            // reference's rounding sequence is mirrored op by op (no contraction): whether a tap is just
            // inside or outside the source image decides `dot != 0`, i.e. the validity count.
            const float uvx = __fsub_rn(__fmul_rn(__fmul_rn(2.0f, __fmul_rn(qx, sc)), inv_w), 1.0f);
            const float uvy = __fsub_rn(__fmul_rn(__fmul_rn(2.0f, __fmul_rn(qy, sc)), inv_h), 1.0f);
            const float ix = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(uvx, 1.0f), (float)w), 1.0f), 0.5f);
            const float iy = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(uvy, 1.0f), (float)h), 1.0f), 0.5f);
            const float fx0 = floorf(ix), fy0 = floorf(iy);
            const float tx = ix - fx0, ty = iy - fy0;
            // NaN/inf coordinates sample nothing (comparisons false)
            const bool xin0 = fx0 >= 0.0f && fx0 <= (float)(w - 1), xin1 = fx0 >= -1.0f && fx0 <= (float)(w - 2);
            const bool yin0 = fy0 >= 0.0f && fy0 <= (float)(h - 1), yin1 = fy0 >= -1.0f && fy0 <= (float)(h - 2);
            const int x0 = xin0 || xin1 ? (int)fx0 : 0, y0 = yin0 || yin1 ? (int)fy0 : 0;
            // wave-uniform map base (SGPR pair) + a 32-bit per-lane byte offset: one address add per tap instead of
            // 64-bit multiply-adds (a source map is far below 4 GB)
            // slot-major records (cv_relayout_project_kernel): float4 slot s of texel (x, y), half hf, at
            //   ((y * NS + s) * w + x) * 32 + hf * 16 bytes: the slot term is wave-uniform (scalar base per slot)
            constexpr int REC = HC + kCvU, NS = REC / 4;
            const char* base = (const char*)(srcT + (((size_t)b * K + k) * hw) * (2 * REC));
            const uint32_t off0 = (uint32_t)((y0 * NS * w + x0) * 8 + hf * 4) * 4u;
            const uint32_t slotb = (uint32_t)w * 32u;                        // bytes between consecutive slots of a row
            // dot_k = warped . cur = sum_tap w_tap (v_tap . cur): the warped features themselves are never formed
            float part = 0.0f, ub[kCvU];
#pragma unroll
            for (int r = 0; r < kCvU; ++r) ub[r] = 0.0f;
#pragma unroll
            for (int tap = 0; tap < 4; ++tap) {
                const int ox = tap & 1, oy = tap >> 1;
                const bool ok = live && (ox ? xin1 : xin0) && (oy ? yin1 : yin0);
                const float wt = (ox ? tx : 1.0f - tx) * (oy ? ty : 1.0f - ty);
                if (ok) {
                    const uint32_t offt = off0 + (uint32_t)((oy * NS * w + ox) * 8) * 4u;
                    float td = 0.0f;
#pragma unroll
                    for (int s = 0; s < HC / 4; ++s) {
                        const float4 v = *(const float4*)(base + (size_t)s * slotb + offt);
                        td += v.x * cur[4 * s]; td += v.y * cur[4 * s + 1];
                        td += v.z * cur[4 * s + 2]; td += v.w * cur[4 * s + 3];
                    }
                    part += wt * td;
#pragma unroll
                    for (int s = 0; s < kCvU / 4; ++s) {
                        const float4 u = *(const float4*)(base + (size_t)(HC / 4 + s) * slotb + offt);
                        ub[4 * s] += wt * u.x; ub[4 * s + 1] += wt * u.y;
                        ub[4 * s + 2] += wt * u.z; ub[4 * s + 3] += wt * u.w;
                    }
                }
            }
            float dotk = part + __shfl_xor(part, 32, 64);
            dotk = (zz > 0.0f) ? dotk : 0.0f;                             // cost_volume.py:571-572,589-593
            const bool valid = dotk != 0.0f;                              // :595 (exact zero test)
            if (valid) {
                cnt += 1.0f;
                dot_sum += dotk;
#pragma unroll
                for (int r = 0; r < kCvU; ++r) uavg[r] += ub[r];
            }
        
        };
        one_source(0, P0);
        for (int k = 1; k < K; ++k) {
            float Pk[12];
            proj_row(k, Pk);
            one_source(k, Pk);
        }
        const float inv = 1.0f / (cnt + 1e-8f);                          // :595-598
        FS_CV_T(t_gath, uavg[0] + uavg[kCvU - 1] + inv + dot_sum);
        // ---- layer 1: z1 = W1f favg + w1d * dot_avg + b1 with W1f favg = blended U / cnt; this half's 16 units ----
        const float dbar = dot_sum * inv;
        // ---- layer 2: H2^T = W2 lrelu(H1)^T + b2, k order = accumulator row map ----
        f32x16 acc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[r] = b2r[r];
#pragma unroll
        for (int s = 0; s < 16; ++s)
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[s], lrelu(fmaf(uavg[s], inv, fmaf(w1d[s], dbar, b1r[s]))), acc2, 0, 0, 0);
        // ---- layer 3 ----
        float o = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) o += w3r[r] * lrelu(acc2[r]);
        o += __shfl_xor(o, 32, 64);
        if (live && hf == 0) out[((size_t)b * D + d) * hw + pix] = o + b3v;
#ifdef FS_CV_TRACE
        FS_CV_T(t_end, o);
        tr_g += t_gath - t_top;
        tr_m += t_end - t_gath;
#endif
    }
#ifdef FS_CV_TRACE
    {
        const int wid = (int)blockIdx.x * 4 + wave;
        if (lane == 0 && wid < kCvTraceWaves) {
            g_cv_trace[4 * wid] = tr_g; g_cv_trace[4 * wid + 1] = tr_m;
            // [3]: shader ticks (s_memtime) << 32 | 100 MHz wall ticks of this wavefront's whole sweep -> effective clock
            const unsigned long long dc = cv_stamp(rx) - tr_c0, dw = wall_clock64() - tr_w0;
            g_cv_trace[4 * wid + 2] = (unsigned long long)(d1 - d0); g_cv_trace[4 * wid + 3] = (dc << 32) | (dw & 0xffffffffull);
        }
    }
#endif
}


// ==========================================================================================
// Backward.  Same work decomposition as the forward (a wavefront = 32 pixels x a chunk of planes,
// lane = (pixel, channel parity)); the forward is recomputed per plane, then
//   dz2 = g * w3 * lrelu'(z2);  dh1^T = W2^T dz2^T (16 MFMA);  dz1 = dh1 * lrelu'(z1);
//   dx^T = W1^T dz1^T with the rows of W1^T permuted so that every lane receives the gradient of
//   exactly the channels it owns (2 x 16 MFMA) -- again no cross-lane traffic;
//   d warped_k = valid_k/cnt * df + m_k/cnt * ddot * cur   -> scattered to the source maps with
//   float atomics through the same 4 bilinear taps;  d cur += m_k/cnt * ddot * warped_k.
// The MLP's weight gradients are sums of outer products over ALL (pixel, plane) points,
//   dW1 = sum_pt dz1[pt] (x) x[pt],   dW2 = sum_pt dz2[pt] (x) h1[pt],   db = sum_pt dz,
// i.e. GEMMs whose contraction index is the point -- which in this kernel's layout runs along the LANES.  Per plane
// the wavefront transposes its 32 points' factors through a private LDS tile (padded rows: conflict-free both ways)
// and accumulates the products with 48 more MFMAs into 48 registers that live across the plane loop; the workgroup's
// four wavefronts are summed in LDS and leave as one atomic per weight and workgroup.  (The first version wrote the
// factors to HBM for rocBLAS: 584 B per point -- 1.8 GB per call at the native size, 17.6 GB at config 3's.)
// ==========================================================================================
// One source's bilinear gather for a (pixel, parity) lane: the 4 tap weights, validity and 32-bit texel indices (the scatter
// of the gradient needs exactly those) and the blended half-record.
template <int HC>
struct SrcWarp {
    float wv[HC];
    float wt[4];
    uint32_t tex[4];
    bool ok[4];
    float zz;
};

template <int HC>
__device__ __forceinline__ void warp_source(SrcWarp<HC>& W, const float* __restrict__ map, int w, int h, int hf, bool live,
                                             float depth, float rx, float ry, float rz, const float* __restrict__ P,
                                             float inv_w, float inv_h)
{
    constexpr int C = 2 * HC;
    const float X = depth * rx, Y = depth * ry, Z = depth * rz;
    const float qx = P[0] * X + P[1] * Y + P[2] * Z + P[3];
    const float qy = P[4] * X + P[5] * Y + P[6] * Z + P[7];
    const float qz = P[8] * X + P[9] * Y + P[10] * Z + P[11];
    W.zz = qz + 1e-8f;
    const float sc = (fabsf(qz) > 1e-8f) ? 1.0f / W.zz : 1.0f;
    const float uvx = __fsub_rn(__fmul_rn(__fmul_rn(2.0f, __fmul_rn(qx, sc)), inv_w), 1.0f);
    const float uvy = __fsub_rn(__fmul_rn(__fmul_rn(2.0f, __fmul_rn(qy, sc)), inv_h), 1.0f);
    const float ix = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(uvx, 1.0f), (float)w), 1.0f), 0.5f);
    const float iy = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(uvy, 1.0f), (float)h), 1.0f), 0.5f);
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const float tx = ix - fx0, ty = iy - fy0;
    const bool xin0 = fx0 >= 0.0f && fx0 <= (float)(w - 1), xin1 = fx0 >= -1.0f && fx0 <= (float)(w - 2);
    const bool yin0 = fy0 >= 0.0f && fy0 <= (float)(h - 1), yin1 = fy0 >= -1.0f && fy0 <= (float)(h - 2);
    const int x0 = xin0 || xin1 ? (int)fx0 : 0, y0 = yin0 || yin1 ? (int)fy0 : 0;
#pragma unroll
    for (int s = 0; s < HC; ++s) W.wv[s] = 0.0f;
#pragma unroll
    for (int tap = 0; tap < 4; ++tap) {
        const int ox = tap & 1, oy = tap >> 1;
        W.ok[tap] = live && (ox ? xin1 : xin0) && (oy ? yin1 : yin0);
        W.wt[tap] = (ox ? tx : 1.0f - tx) * (oy ? ty : 1.0f - ty);
        W.tex[tap] = W.ok[tap] ? (uint32_t)((y0 + oy) * w + (x0 + ox)) : 0u;
        if (W.ok[tap]) {
            const float4* q = (const float4*)(map + (size_t)W.tex[tap] * C + hf * HC);
#pragma unroll
            for (int s = 0; s < HC / 4; ++s) {
                const float4 v = q[s];
                W.wv[4 * s] += W.wt[tap] * v.x; W.wv[4 * s + 1] += W.wt[tap] * v.y;
                W.wv[4 * s + 2] += W.wt[tap] * v.z; W.wv[4 * s + 3] += W.wt[tap] * v.w;
            }
        }
    }
}

__device__ __forceinline__ float dlrelu(float z) { return z > 0.0f ? 1.0f : 0.01f; }

// channel slot (16*blk + r) held by row `i` of the permuted W1^T block: row i belongs to lane half
// (i>>2)&1 as accumulator register r = (i&3) + 4*(i>>3)
__device__ __forceinline__ constexpr int row_reg(int i) { return (i & 3) + 4 * (i >> 3); }
__device__ __forceinline__ constexpr int row_half(int i) { return (i >> 2) & 1; }

// The backward sweep.  TWO wavefronts per SIMD (<= 256 registers, <= 80 KB of LDS per workgroup; the first version held
// every MFMA operand in registers -- 484 of them, one wavefront per SIMD -- and left the SIMD idle during each wavefront's
// gather latency and atomic walk: native fwd+bwd 2.30 -> 1.73 ms, the 10-view K = 8 shape 65 -> 47 ms):
//   * the MLP's weights are not held as per-lane operand registers (121 of them) but read from an LDS copy as each MFMA
//     needs its A operand: W1 as [32 units][XS] (columns 0..C-1 channels, C the dot feature, C+1 = b1, C+2 = 0; XS odd),
//     W2 as [32][33] -- both row- and column-wise reads are conflict-free -- w3 and b2 as vectors;
//   * b2's and w3's gradients are column sums over the points: of the dz2 tile that passes through LDS anyway (each lane
//     adds the 16 entries of its unit that it reads as MFMA operands) and of a half-height tile of g * h2 (neighbouring
//     pixels pre-added with one DPP step) -- two accumulators instead of 32, and z2 is dead as soon as it is computed;
//   * dW2 and dW1 are accumulated in two phases that share the dz tile (dz2, then dz1); lrelu'(z1) is kept as a bit mask.
template <int HC, bool SPLIT>
__global__ __launch_bounds__(256, 2) void cost_volume_bwd_kernel(
    int B, int K, int h, int w, int D, int slices, const float* __restrict__ curT, const float* __restrict__ srcT,
    const float* __restrict__ Pmat,
    const float* __restrict__ cur_invK, const float* __restrict__ planes, long long ps_b,
    long long ps_d, long long ps_p, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ w2, const float* __restrict__ b2, const float* __restrict__ w3,
    const float* __restrict__ g_out, float* __restrict__ d_curT, float* __restrict__ d_srcT,
    float* __restrict__ gw1, float* __restrict__ gb1, float* __restrict__ gw2, float* __restrict__ gb2,
    float* __restrict__ gw3, float* __restrict__ gb3, float4* __restrict__ recS, float2* __restrict__ recM)
{
    constexpr int C = 2 * HC;
    constexpr int NBLK = (HC + 1 + 15) / 16;  // row blocks of the permuted W1^T
    constexpr int XW = 2 * (HC + 1);          // features of a point: C channels, dot, 1
    constexpr int XS = XW | 1;                // odd row stride of the feature tile and of the W1 copy (column XW: zero)
    constexpr int kTile1 = 32 * 33, kTileH = 16 * 33, kTileX = 32 * XS, kStage = 2 * kTile1 + kTileH + kTileX;
    constexpr int NCB = (XW + 31) / 32;       // column blocks of dW1
    constexpr int kW1 = 0, kW2 = 32 * XS, kW3 = kW2 + 32 * 33, kB2 = kW3 + 32, kWts = kB2 + 32;
    static_assert((kWts + 4 * kStage) * 4 <= 80 * 1024, "two workgroups per CU need <= 80 KB of LDS each");
    static_assert((1 + NCB) * 16 * 64 <= kStage, "final weight-gradient staging exceeds the wavefront's LDS tile");
    __shared__ float s_all[kWts + 4 * kStage];
    const int hw = h * w;
    const int groups = (hw + 31) / 32;
    const CvBlock blk_ = cv_block(B, groups, slices);
    if (!blk_.ok) return;   // (workgroup-uniform)
    for (int e = threadIdx.x; e < 32 * XS; e += 256) {
        const int u = e / XS, f = e - u * XS;
        s_all[kW1 + e] = f <= C ? w1[u * (C + 1) + f] : (f == C + 1 ? b1[u] : 0.0f);
    }
    for (int e = threadIdx.x; e < 32 * 33; e += 256) {
        const int u = e / 33, c = e - u * 33;
        s_all[kW2 + e] = c < 32 ? w2[u * 32 + c] : 0.0f;
    }
    if (threadIdx.x < 32) { s_all[kW3 + threadIdx.x] = w3[threadIdx.x]; s_all[kB2 + threadIdx.x] = b2[threadIdx.x]; }
    __syncthreads();
    const float* const sW1 = s_all + kW1, * const sW2 = s_all + kW2, * const sW3 = s_all + kW3, * const sB2 = s_all + kB2;
    const int b = blk_.b, grp = blk_.grp;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int p = lane & 31, hf = lane >> 5;
    float* const stage = s_all + kWts + wave * kStage;
    const int pix = grp * 32 + p;
    const bool live = pix < hw;
    const int pu = live ? pix % w : 0, pv = live ? pix / w : 0;
    // column of the W1 copy that row p of the permuted W1^T block `blk` reads (the zero column where it has no feature)
    int tcol[NBLK];
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {
        const int sl = 16 * blk + row_reg(p), ph = row_half(p);
        tcol[blk] = sl < HC ? 2 * sl + ph : (sl == HC && ph == 0 ? C : XW);
    }

    float cur[HC], dcur[HC];
    {
        const float4* q = (const float4*)(curT + ((size_t)b * hw + (live ? pix : 0)) * C + (size_t)hf * HC);
#pragma unroll
        for (int s = 0; s < HC / 4; ++s) {
            const float4 v = q[s];
            cur[4 * s] = v.x; cur[4 * s + 1] = v.y; cur[4 * s + 2] = v.z; cur[4 * s + 3] = v.w;
        }
#pragma unroll
        for (int s = 0; s < HC; ++s) dcur[s] = 0.0f;
    }
    const float* iK = cur_invK + (size_t)b * 16;
    const float ux = (float)pu + 0.5f, vy = (float)pv + 0.5f;
    const float rx = iK[0] * ux + iK[1] * vy + iK[2];
    const float ry = iK[4] * ux + iK[5] * vy + iK[6];
    const float rz = iK[8] * ux + iK[9] * vy + iK[10];
    const float inv_w = (float)(1.0 / (double)w), inv_h = (float)(1.0 / (double)h);
    const int dchunk = (D + slices * 4 - 1) / (slices * 4);
    const int d0 = min(D, (blk_.slice * 4 + wave) * dchunk), d1 = min(D, d0 + dchunk);
    float gw3a = 0.0f, gb2a = 0.0f;   // this lane's 16 points of unit p's sums (the other half holds the other 16)
    float gb3r = 0.0f;
    f32x16 gW2, gW1[NCB];   // dW2[unit acc_row(r,hf)][col p],  dW1[unit acc_row(r,hf)][feature 32 cb + p]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        gW2[r] = 0.0f;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) gW1[cb][r] = 0.0f;
    }
    SrcWarp<HC> W;
#ifdef FS_CV_TRACE
    unsigned long long tb0 = 0, tb1 = 0, tb2 = 0, tb3 = 0;
    const unsigned long long tb_start = cv_stamp(rx);
#endif

    for (int d = d0; d < d1; ++d) {
        FS_CV_T(tq0, rx);
#ifdef FS_CV_TRACE
        unsigned long long tq1 = 0;
#endif
        const float depth = planes[b * ps_b + d * ps_d + (live ? pix : 0) * ps_p];
        const size_t pt = ((size_t)b * D + d) * hw + (live ? pix : 0);
        const float go = live ? g_out[pt] : 0.0f;
        float* const tA = stage, * const tB = tA + kTile1, * const tC = tB + kTile1, * const tx = tC + kTileH;
        // ---- forward recompute ----
        f32x16 z1;
        float inv, xlast;
        uint32_t flags = 0, rare = 0;   // SPLIT: bit 2k = source k valid (dot != 0), bit 2k+1 = in front of it (z > 0); rare: bit 2k
        {
        float favg[HC];
#pragma unroll
        for (int s = 0; s < HC; ++s) favg[s] = 0.0f;
        float dot_sum = 0.0f, cnt = 0.0f;
        for (int k = 0; k < K; ++k) {
            warp_source<HC>(W, srcT + (((size_t)b * K + k) * hw) * C, w, h, hf, live, depth, rx, ry, rz,
                             Pmat + ((size_t)b * K + k) * 12, inv_w, inv_h);
            float part = 0.0f;
#pragma unroll
            for (int s = 0; s < HC; ++s) part += W.wv[s] * cur[s];
            float dotk = part + __shfl_xor(part, 32, 64);
            dotk = (W.zz > 0.0f) ? dotk : 0.0f;
            if (SPLIT) {
                flags |= (W.zz > 0.0f ? 2u : 0u) << (2 * k);
                // in front, some tap inside the source image, and still an exactly zero score (all-zero features): the
                // score's gradient reaches the current feature although the source is not averaged -- see below
                if (W.zz > 0.0f && dotk == 0.0f && (W.ok[0] || W.ok[1] || W.ok[2] || W.ok[3])) rare |= 1u << (2 * k);
            }
            if (dotk != 0.0f) {
                if (SPLIT) flags |= 1u << (2 * k);
                cnt += 1.0f;
                dot_sum += dotk;
#pragma unroll
                for (int s = 0; s < HC; ++s) favg[s] += W.wv[s];
            }
        }
        inv = 1.0f / (cnt + 1e-8f);
        xlast = hf ? 1.0f : dot_sum * inv;
#pragma unroll
        for (int r = 0; r < 16; ++r) z1[r] = 0.0f;
#pragma unroll
        for (int s = 0; s < HC; ++s) {
            const float x = favg[s] * inv;
            tx[p * XS + 2 * s + hf] = x;      // the point's features for the dW1 products below
            z1 = __builtin_amdgcn_mfma_f32_32x32x2f32(sW1[p * XS + 2 * s + hf], x, z1, 0, 0, 0);
        }
        tx[p * XS + C + hf] = xlast;
        z1 = __builtin_amdgcn_mfma_f32_32x32x2f32(sW1[p * XS + C + hf], xlast, z1, 0, 0, 0);
        }
        // h1 replaces z1 (lrelu' survives as a bit per unit) and goes to its tile at once
        uint32_t pos1 = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            pos1 |= (z1[r] > 0.0f ? 1u : 0u) << r;
            z1[r] = lrelu(z1[r]);
            tB[p * 33 + acc_row(r, hf)] = z1[r];
        }
        const float lv = live ? 1.0f : 0.0f;   // points past the image contribute nothing
        f32x16 dz2;
        {
            f32x16 z2;
#pragma unroll
            for (int r = 0; r < 16; ++r) z2[r] = sB2[acc_row(r, hf)];
#pragma unroll
            for (int s = 0; s < 16; ++s) z2 = __builtin_amdgcn_mfma_f32_32x32x2f32(sW2[p * 33 + acc_row(s, hf)], z1[s], z2, 0, 0, 0);
            FS_CV_T(tq1_, z2[0] + z2[15]);
#ifdef FS_CV_TRACE
            tq1 = tq1_;
#endif
            // ---- backward through the MLP; dW2 += dz2 (x) h1 (phase 1 of the LDS tiles) ----
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int u = acc_row(r, hf);
                dz2[r] = go * sW3[u] * dlrelu(z2[r]);
                tA[p * 33 + u] = dz2[r] * lv;
                const float g = go * lrelu(z2[r]);   // (go = 0 past the image)
                const float g2 = g + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(g), 0xB1, 0xF, 0xF, true));   // + pixel p ^ 1
                if ((p & 1) == 0) tC[(p >> 1) * 33 + u] = g2;
            }
        }
        if (hf == 0) gb3r += go;
        wave_lds_sync();
        // k-step m contracts points 2m and 2m+1: A = dz[pt][unit = lane&31], B = x / h1 [pt][column = lane&31]
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const int q = 2 * m + hf;
            const float az2 = tA[q * 33 + p];
            gb2a += az2;
            gW2 = __builtin_amdgcn_mfma_f32_32x32x2f32(az2, tB[q * 33 + p], gW2, 0, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) gw3a += tC[(8 * hf + m) * 33 + p];
        f32x16 dz1;
        {
            f32x16 dh1;
#pragma unroll
            for (int r = 0; r < 16; ++r) dh1[r] = 0.0f;
#pragma unroll
            for (int s = 0; s < 16; ++s) dh1 = __builtin_amdgcn_mfma_f32_32x32x2f32(sW2[acc_row(s, hf) * 33 + p], dz2[s], dh1, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) dz1[r] = dh1[r] * ((pos1 >> r) & 1u ? 1.0f : 0.01f);
        }
        wave_lds_sync();   // (phase 1's reads are done)
        // ---- phase 2: dW1 += dz1 (x) x ----
#pragma unroll
        for (int r = 0; r < 16; ++r) tA[p * 33 + acc_row(r, hf)] = dz1[r] * lv;
        wave_lds_sync();
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const int q = 2 * m + hf;
            const float az1 = tA[q * 33 + p];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const float bx = (32 * cb + p < XW) ? tx[q * XS + 32 * cb + p] : 0.0f;
                gW1[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(az1, bx, gW1[cb], 0, 0, 0);
            }
        }
        float dfavg[HC];
        float ddot = 0.0f;
#pragma unroll
        for (int blk = 0; blk < NBLK; ++blk) {
            f32x16 dx;
#pragma unroll
            for (int r = 0; r < 16; ++r) dx[r] = 0.0f;
#pragma unroll
            for (int s = 0; s < 16; ++s) dx = __builtin_amdgcn_mfma_f32_32x32x2f32(sW1[acc_row(s, hf) * XS + tcol[blk]], dz1[s], dx, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (16 * blk + r < HC) dfavg[16 * blk + r] = dx[r];
                else if (16 * blk + r == HC) ddot = dx[r];  // meaningful on the hf == 0 lane of the pixel
            }
        }
        ddot = __shfl(ddot, p, 64);  // both parities of the pixel need it
        FS_CV_T(tq2, ddot + dfavg[0]);
        if (SPLIT) {
            // Two-pass form (round 4): the source-feature gradient is NOT scattered from here.  What a source that counts
            // receives from this point through each bilinear tap -- S = d favg / cnt + d dot / cnt * cur, this lane's
            // channels -- goes to memory once, chunk-planar ([view, plane][float4 chunk][pixel]: neighbouring pixels are
            // neighbours in memory for the writer and for the reader), with d dot / cnt and the sources' (valid, in-front)
            // bits beside it; cv_src_grad_kernel, whose workgroups own TILES OF SOURCE TEXELS, walks the pixels that
            // sample its tile plane by plane and accumulates in LDS: no global float atomics (259 M 192-byte atomic
            // records per 10-view K = 8 call before: 38 of the backward's 41.5 ms).
            if (live) {
                const size_t pl = (size_t)b * D + d;
                float4* rp = recS + (pl * (C / 4) + (size_t)hf * (HC / 4)) * hw + pix;
                const float di = ddot * inv;
#pragma unroll
                for (int s = 0; s < HC / 4; ++s)
                    rp[(size_t)s * hw] = make_float4(fmaf(di, cur[4 * s], dfavg[4 * s] * inv), fmaf(di, cur[4 * s + 1], dfavg[4 * s + 1] * inv),
                                                     fmaf(di, cur[4 * s + 2], dfavg[4 * s + 2] * inv), fmaf(di, cur[4 * s + 3], dfavg[4 * s + 3] * inv));
                if (hf == 0) recM[pl * hw + pix] = make_float2(di, __uint_as_float(flags));
            }
            // d cur = d dot / cnt * sum_k [z_k > 0] warped_k.  For every source that counts, [z_k > 0] = valid_k, so the
            // sum is favg = cnt * x, and x is still in this lane's entries of the feature tile.
#pragma unroll
            for (int s = 0; s < HC; ++s) dcur[s] = fmaf(ddot, tx[p * XS + 2 * s + hf], dcur[s]);
            // (a source in front whose score is EXACTLY zero with taps inside its image -- all-zero features -- is not
            //  averaged but still passes d dot / cnt on: re-gather those, wave-uniformly; never taken on real data)
            if (__builtin_amdgcn_ballot_w64(rare != 0u) != 0ull) {
                for (int k = 0; k < K; ++k) {
                    if (__builtin_amdgcn_ballot_w64(((rare >> (2 * k)) & 1u) != 0u) == 0ull) continue;
                    warp_source<HC>(W, srcT + (((size_t)b * K + k) * hw) * C, w, h, hf, live, depth, rx, ry, rz,
                                     Pmat + ((size_t)b * K + k) * 12, inv_w, inv_h);
                    const float cd = ((rare >> (2 * k)) & 1u) ? inv * ddot : 0.0f;
#pragma unroll
                    for (int s = 0; s < HC; ++s) dcur[s] = fmaf(cd, W.wv[s], dcur[s]);
                }
            }
        }
        wave_lds_sync();   // (the scatter staging below overwrites the tiles)
        FS_CV_T(tq3, gW2[0] + gW1[0][0]);
        // ---- back to the features ----
        for (int k = 0; !SPLIT && k < K; ++k) {
            // (K = 1: W still holds this source from the forward recompute above -- no second gather)
            if (K > 1) warp_source<HC>(W, srcT + (((size_t)b * K + k) * hw) * C, w, h, hf, live, depth, rx, ry, rz,
                                        Pmat + ((size_t)b * K + k) * 12, inv_w, inv_h);
            float part = 0.0f;
#pragma unroll
            for (int s = 0; s < HC; ++s) part += W.wv[s] * cur[s];
            const float dotk = part + __shfl_xor(part, 32, 64);
            const bool m = W.zz > 0.0f;
            const bool valid = m && dotk != 0.0f;
            const float cf = valid ? inv : 0.0f, cd = m ? inv * ddot : 0.0f;
            // Scatter to the source map, TRANSPOSED through the wavefront's LDS tile: with lane = pixel one atomic
            // instruction touched 64 different texel records (~40 cache lines, 4 bytes each); with lane = channel it
            // covers ONE texel's contiguous record (2 lines).  The L2 executes atomics per 64-byte request, so the
            // request count is what the 6e8 float atomics of a native call cost: 96 instructions x ~40 lines before,
            // 128 x 2 now, per (32-pixel group, plane, source).
            {
                float* tD = stage;                                  // [32 pixels][C slots]  (slot = parity * HC + s)
                float* tW = tD + 32 * C;                            // [32][4] tap weights (0: tap unused)
                uint32_t* tO = (uint32_t*)(tW + 32 * 4);            // [32][4] texel index of the tap in the source map
                static_assert(32 * C + 2 * 32 * 4 <= kStage, "scatter staging exceeds the wavefront's LDS tile");
#pragma unroll
                for (int s = 0; s < HC; ++s) {
                    tD[p * C + hf * HC + s] = cf * dfavg[s] + cd * cur[s];
                    dcur[s] += cd * W.wv[s];
                }
                // (pixel, tap) records in WALK order: texel row (tap >> 1), then pixel, then column (tap & 1) -- equal texels
                // of a row are neighbours (pixel p's right tap is pixel p+1's left one when the source is sampled at
                // about its own resolution)
                if (hf == 0) {
#pragma unroll
                    for (int tap = 0; tap < 4; ++tap) {
                        tW[(tap >> 1) * 64 + 2 * p + (tap & 1)] = W.ok[tap] ? W.wt[tap] : 0.0f;
                        tO[(tap >> 1) * 64 + 2 * p + (tap & 1)] = W.tex[tap];
                    }
                }
                wave_lds_sync();
                float* const dmap = d_srcT + (((size_t)b * K + k) * hw) * C;
                // Everything the walk needs comes back from LDS ONCE: the 128 weights and texel indices as two registers
                // each (record e in lane e & 63, handed out with v_readlane) and this lane's channel of the pixels, 16
                // registers at a time.  Runs of the same texel are summed in a register and leave as ONE atomic instruction
                // (~66 instead of 128 per (group, plane, source)); where a run starts is decided for all records at once
                // (two ballots), so a step of the walk is a scalar bit test, a v_readlane and an FMA -- the first version
                // read its value from LDS and compared its index inside every step: ~190 cycles of latency per step,
                // 80 % of the kernel's time at K = 8.
                const float w_lo = tW[lane], w_hi = tW[64 + lane];
                const uint32_t o_lo = tO[lane], o_hi = tO[64 + lane];
                const unsigned long long act_lo = __builtin_amdgcn_ballot_w64(w_lo != 0.0f), act_hi = __builtin_amdgcn_ballot_w64(w_hi != 0.0f);
                const unsigned long long same_lo = __builtin_amdgcn_ballot_w64(o_lo == (uint32_t)__shfl_up((int)o_lo, 1, 64));
                const unsigned long long same_hi = __builtin_amdgcn_ballot_w64(o_hi == (uint32_t)__shfl_up((int)o_hi, 1, 64));
                // a record starts a run unless its predecessor is in use and names the same texel (record 64 always does)
                const unsigned long long head_lo = act_lo & ~((act_lo << 1) & same_lo), head_hi = act_hi & ~((act_hi << 1) & same_hi);
                uint32_t run_idx = 0;
                bool have = false;
                float run_acc = 0.0f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {                       // texel row q >> 1, pixels 16 (q & 1) ..
                    float tv[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) tv[i] = tD[(16 * (q & 1) + i) * C + lane];
#pragma unroll
                    for (int e2 = 32 * q; e2 < 32 * q + 32; ++e2) { // texel row e2 >> 6, pixel (e2 >> 1) & 31, column e2 & 1
                        const int i = (e2 >> 1) & 15, ln = e2 & 63;
                        const float wt = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e2 < 64 ? w_lo : w_hi), ln));
                        if (((e2 < 64 ? head_lo : head_hi) >> ln) & 1ull) {
                            if (have && lane < C) atomicAdd(dmap + (size_t)run_idx * C + lane, run_acc);
                            run_idx = (uint32_t)__builtin_amdgcn_readlane((int)(e2 < 64 ? o_lo : o_hi), ln);
                            have = true;
                            run_acc = wt * tv[i];
                        } else {
                            run_acc = fmaf(wt, tv[i], run_acc);     // (an unused record has weight 0)
                        }
                    }
                }
                if (have && lane < C) atomicAdd(dmap + (size_t)run_idx * C + lane, run_acc);
                wave_lds_sync();
            }
        }
#ifdef FS_CV_TRACE
        {
            FS_CV_T(tq4, dcur[0]);
            tb0 += tq1 - tq0; tb1 += tq2 - tq1; tb2 += tq3 - tq2; tb3 += tq4 - tq3;
        }
#endif
    }
#ifdef FS_CV_TRACE
    {
        const int wid = (int)blockIdx.x * 4 + wave;
        if (lane == 0 && wid < kCvTraceWaves) {
            unsigned long long* o = g_cvb_trace + 6 * (size_t)wid;
            o[0] = tb0; o[1] = tb1; o[2] = tb2; o[3] = tb3; o[4] = (unsigned long long)(d1 - d0); o[5] = cv_stamp(rx) - tb_start;
        }
    }
#endif
    {   // d cur: same transposition (one atomic instruction per pixel record instead of 24 over 64 scattered records)
        float* tD = stage;
#pragma unroll
        for (int s = 0; s < HC; ++s) tD[p * C + hf * HC + s] = live ? dcur[s] : 0.0f;
        wave_lds_sync();
        const int npx = min(32, hw - grp * 32);
        float* const dst = d_curT + ((size_t)b * hw + (size_t)grp * 32) * C;
        for (int j = 0; j < npx; ++j)
            if (lane < C) atomicAdd(dst + (size_t)j * C + lane, tD[j * C + lane]);
        wave_lds_sync();
    }
    // w3 / b2: the two halves of the wavefront hold 16 points each of unit p's sums;  b3: sum over the pixels
    {
        const float v3 = gw3a + __shfl_xor(gw3a, 32, 64), v2 = gb2a + __shfl_xor(gb2a, 32, 64);
        if (hf == 0) { atomicAdd(&gw3[p], v3); atomicAdd(&gb2[p], v2); }
        float v = gb3r;
#pragma unroll
        for (int sft = 16; sft >= 1; sft >>= 1) v += __shfl_xor(v, sft, 64);
        if (lane == 0) atomicAdd(gb3, v);
    }
    // W1 (+ b1 = its "1" column) and W2: sum the workgroup's four wavefronts in LDS, one atomic per weight
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        stage[r * 64 + lane] = gW2[r];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) stage[(16 * (1 + cb) + r) * 64 + lane] = gW1[cb][r];
    }
    __syncthreads();
    const float* const st0 = s_all + kWts;
    for (int e = threadIdx.x; e < (1 + NCB) * 16 * 64; e += 256) {
        const float v = st0[e] + st0[kStage + e] + st0[2 * kStage + e] + st0[3 * kStage + e];
        const int blk = e >> 10, r = (e >> 6) & 15, l = e & 63;
        const int unit = acc_row(r, l >> 5), col = l & 31;
        if (blk == 0) {
            atomicAdd(&gw2[unit * 32 + col], v);
        } else {
            const int f = 32 * (blk - 1) + col;        // feature: channels 0..C-1, dot, 1
            if (f <= C) atomicAdd(&gw1[unit * (C + 1) + f], v);
            else if (f == C + 1) atomicAdd(&gb1[unit], v);
        }
    }
}


// ==========================================================================================
// Backward pass 1 on 16-pixel wavefronts (round 6; replaces cost_volume_bwd_kernel<., true, false> for the two-pass form).
// Same lane orders as the K >= 2 forward sweep above: the gather runs with lane = (pixel j, channel quarter c), the matrix
// cores with lane = (pixel n, quarter g), and v_mfma_f32_16x16x4_f32 throughout.  What that buys over the 32-pixel kernel:
//   * FOUR of the six products need no transposition at all: an accumulator of layer l (lane (n, g), register r = unit
//     16 blk + 4 g + r of pixel n) IS the B operand of the next product in either direction --
//        z1 = W1 x,  z2 = W2 h1,  dh1 = W2^T dz2,  dx = W1^T dz1
//     (k-step (blk, r), quarter g  <->  unit 16 blk + 4 g + r), and dx comes out in the operand order of x itself: register
//     (rb, r) = channel 16 rb + 4 g + r, i.e. one float4 CHUNK of the record in natural channel order;
//   * the two weight gradients contract over the POINTS, which must lie along k: the factors (dz2, h1, dz1: [32 units][16 px],
//     x: [C channels][16 px]) pass through wavefront-private LDS tiles laid out so that ONE ds_read_b128 returns the operands
//     of all four k-steps of a 16 x 16 block (pixel 4 kk + s -> lane quarter kk, k-step s), conflict-free both ways: float
//     index of (row i of the block, pixel p) = 64 (p >> 2) + 4 (i ^ (p >> 2)) + (p & 3);
//   * exact block cuts: the C channels are C/16 blocks of 16 (dW1: 2 x C/16 x 4, dx: C/16 x 8 MFMAs), the score and bias
//     columns of dW1, db2, dw3, db3 are per-lane running sums in the accumulator layout (reduced over the pixels once, at the
//     end) -- 122 MFMAs of 32 cycles per 16 pixels = 7.8 k matrix-pipe cycles per 32 points against 8.8 k, no column-sum tiles;
//   * every weight operand (W1, W2 and their transposes) is read from an LDS image in MFMA A-operand order -- one lane-consecutive
//     ds_read_b32 where its MFMA is; in registers (82 of them), or staged a phase ahead, the kernel spilled at two wavefronts per
//     SIMD and every scratch reload inside the plane loop is a memory round trip in front of an MFMA;
//   * all taps of a source are loaded back to back behind wave-uniform tests (two at a time), then blended: at two wavefronts per
//     SIMD the forward's tap-by-tap form left the taps' latency exposed (9.6 -> 8.1 ms at config-3 scale, 11.9 -> 7.1 ms at K = 8);
//   * the record's stores are issued in front of dW1's MFMAs, not at the end of the plane: the memory counter is in-order, so the
//     next plane's first tap wait also waited for them.
// Measured against the 32-pixel kernel (profiles/r6_cv_bwd16.txt): config-3 scale 7.0 - 7.3 -> 6.7 ms, 10 views K = 8 7.5 -> 6.1 ms.
// A workgroup covers one 32-pixel group (the same XCD-aware order as before) as two 16-pixel halves in turn; its four wavefronts
// share the planes.  Records leave in NATURAL channel order ([view, plane][chunk of 4 channels][pixel]); cv_src_grad_kernel<C, true>
// reads them.  d cur is summed over the workgroup's wavefronts in LDS and leaves as plain stores when the planes are not split
// over workgroups.
// ==========================================================================================
// SAVED: the training forward kept the MLP's inputs (cost_volume16_kernel<C, true>: x, the averaged score, the validity bits) --
// no gather here at all: three float4 loads per lane and plane instead of 4 K taps.
template <int C, bool SAVED>
#ifdef FS_BWD16_ONE_WAVE     // (A/B build: one wavefront per SIMD with the whole 512-register file)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void cost_volume16_bwd_kernel(
#else
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void cost_volume16_bwd_kernel(
#endif
    int B, int K, int h, int w, int D, int slices, const float* __restrict__ curN, const float* __restrict__ srcN,
    const float* __restrict__ Pmat, const float* __restrict__ cur_invK, const float* __restrict__ planes, long long ps_b,
    long long ps_d, const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
    const float* __restrict__ b2, const float* __restrict__ w3, const float* __restrict__ g_out,
    float* __restrict__ d_curN, float* __restrict__ gw1, float* __restrict__ gb1, float* __restrict__ gw2,
    float* __restrict__ gb2, float* __restrict__ gw3, float* __restrict__ gb3, float4* __restrict__ recS,
    float2* __restrict__ recM, const float4* __restrict__ xs, const float2* __restrict__ xm, const uint32_t* __restrict__ xhdr)
{
    constexpr int NS = C / 16;          // tap load instructions per tap = 16-channel blocks
    constexpr int NR = C / 4;           // channels per lane
    constexpr int NT = NR + 1;          // k-steps of layer 1 (the last one: dot, 1, 0, 0)
    constexpr int RB = C / 16;          // channel blocks of dx / dW1
    // LDS (floats): operand images of W2^T and W1^T, the small vectors, then one region per wavefront
    constexpr int kA2T = 0, kA1T = 16 * 64, kA2 = kA1T + RB * 8 * 64, kA1 = kA2 + 16 * 64, kVec = kA1 + 2 * NT * 64, kWts = kVec + 160;
    // (the dz1 tile takes the dz2 tile's place: dW2's reads of dz2 are issued before dz1 is written, and a wavefront's LDS operations
    //  execute in order)
    constexpr int kTD2 = 0, kTD1 = 0, kTH1 = 512, kTX = 1024, kCur = kTX + RB * 256, kCurG = kCur + NR * 64, kWave = kCurG + NR * 64;
    constexpr int kAccRegs = 16 + 2 * RB * 4;   // dW2 + dW1 accumulator registers
    static_assert((kWts + 4 * kWave) * 4 <= 80 * 1024, "two workgroups per CU need <= 80 KB of LDS each");
    static_assert(kAccRegs * 64 + 33 * 4 <= kWave && 16 * C <= kWave, "final staging exceeds the wavefront's tiles");
    __shared__ __attribute__((aligned(16))) float s_all[kWts + 4 * kWave];
    const int hw = h * w;
    const int groups = (hw + 31) / 32;
    const CvBlock blk_ = cv_block(B, groups, slices);
    if (!blk_.ok) return;   // (workgroup-uniform)
    for (int e = threadIdx.x; e < 16 * 64; e += 256) {        // A operand of dh1: row n of block bo, k = unit u(t, g)
        const int l = e & 63, t = (e >> 6) & 7, bo = e >> 9, nn = l & 15, gg = l >> 4;
        s_all[kA2T + e] = w2[(16 * (t >> 2) + 4 * gg + (t & 3)) * 32 + 16 * bo + nn];
    }
    for (int e = threadIdx.x; e < 16 * 64; e += 256) {        // A operand of layer 2: row = unit 16 blk + n, k = unit u(t, g)
        const int l = e & 63, t = (e >> 6) & 7, blk = e >> 9, nn = l & 15, gg = l >> 4;
        s_all[kA2 + e] = w2[(16 * blk + nn) * 32 + 16 * (t >> 2) + 4 * gg + (t & 3)];
    }
    for (int e = threadIdx.x; e < 2 * NT * 64; e += 256) {    // A operand of layer 1: row = unit 16 blk + n, k = feature (t, g)
        const int l = e & 63, q = e >> 6, blk = q / NT, t = q - blk * NT, nn = l & 15, gg = l >> 4, u = 16 * blk + nn;
        s_all[kA1 + e] = t < NR ? w1[u * (C + 1) + 16 * (t >> 2) + 4 * gg + (t & 3)] : (gg == 0 ? w1[u * (C + 1) + C] : (gg == 1 ? b1[u] : 0.0f));
    }
    for (int e = threadIdx.x; e < RB * 8 * 64; e += 256) {    // A operand of dx: row = channel 16 rb + n, k = unit u(t, g)
        const int l = e & 63, t = (e >> 6) & 7, rb = e >> 9, nn = l & 15, gg = l >> 4;
        s_all[kA1T + e] = w1[(16 * (t >> 2) + 4 * gg + (t & 3)) * (C + 1) + 16 * rb + nn];
    }
    if (threadIdx.x < 32) {
        s_all[kVec + threadIdx.x] = w3[threadIdx.x];
        s_all[kVec + 32 + threadIdx.x] = b2[threadIdx.x];
        s_all[kVec + 64 + threadIdx.x] = w1[threadIdx.x * (C + 1) + C];
    }
    const int b = blk_.b;
    // the projection rows of the first group of four sources (lane c of a quad projects source min(c, K - 1)): [4][12]
    if (threadIdx.x >= 64 && threadIdx.x < 64 + 48) {
        const int e = threadIdx.x - 64, q = e / 12;
        s_all[kVec + 96 + e] = Pmat[((size_t)b * K + min(q, K - 1)) * 12 + (e - 12 * q)];
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = lane >> 2, c = lane & 3;      // gather order
    const int n = lane & 15, g = lane >> 4;     // operand order
    const int pull = (4 * n + g) * 4;           // operand lane (n, g) takes gather lane 4 n + g
    const int pull0 = (4 * n) * 4;              // ... or a quad-uniform value of pixel n
    float* const wv_ = s_all + kWts + wave * kWave;
    float* const tD2 = wv_ + kTD2, * const tH1 = wv_ + kTH1, * const tD1 = wv_ + kTD1, * const tX = wv_ + kTX;
    float4* const sCur = (float4*)(wv_ + kCur), * const sCurG = (float4*)(wv_ + kCurG);
    const float* const sA2T = s_all + kA2T + lane, * const sA1T = s_all + kA1T + lane, * const sA2 = s_all + kA2 + lane,
                * const sA1 = s_all + kA1 + lane;
    // tile addresses: accumulator register (blk, r) of lane (n, g) = row 4 g + r of block blk, pixel n
    const int kq = n >> 2;
    int wr_at[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) wr_at[r] = 64 * kq + 16 * g + 4 * (r ^ kq) + (n & 3);
    const int rd_at = 64 * g + ((n ^ g) << 2);                    // b128: row n of a block, pixels 4 g .. 4 g + 3
    const int xw_at = 64 * (j >> 2) + 16 * c + (j & 3), xw_k = j >> 2;   // x tile from the gather order: + 256 blk + 4 ((r & 3) ^ xw_k)

    // (every MFMA's weight operand comes from its LDS image: one lane-consecutive ds_read_b32 each; in registers -- 82 of them --
    //  the kernel spilled at two wavefronts per SIMD)
    const float4* const sW3 = (const float4*)(s_all + kVec) + g, * const sB2 = (const float4*)(s_all + kVec + 32) + g,
                * const sW1d = (const float4*)(s_all + kVec + 64) + g;        // [blk]: + 4 blk  (units 16 blk + 4 g + 0..3)

    const float* iK = cur_invK + (size_t)b * 16;
    const float inv_w = (float)(1.0 / (double)w), inv_h = (float)(1.0 / (double)h);
    const int dchunk = (D + slices * 4 - 1) / (slices * 4);
    const int d0 = min(D, (blk_.slice * 4 + wave) * dchunk), d1 = min(D, d0 + dchunk);

    // running sums that live across both halves and all planes
    f32x4 acc2[2][2], acc1[2][RB];      // dW2[16 rb + 4 g + r][16 cb + n],  dW1[16 rb + 4 g + r][channel 16 cb + n]
    float s_dot[2][4], s_b1[2][4], s_b2[2][4], s_w3[2][4], s_b3 = 0.0f;   // unit 16 blk + 4 g + r, this lane's pixels
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc2[rb][cb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int cb = 0; cb < RB; ++cb) acc1[rb][cb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int r = 0; r < 4; ++r) { s_dot[rb][r] = 0.0f; s_b1[rb][r] = 0.0f; s_b2[rb][r] = 0.0f; s_w3[rb][r] = 0.0f; }
    }
    const float4* const sPq = (const float4*)(s_all + kVec + 96) + 3 * c;

    float dcur[NR];    // d cur of the half in flight, operand order (channel 16 (t >> 2) + 4 g + (t & 3) of pixel n)
#ifdef FS_CV_TRACE   // [0] gather, [1] x hand-over + layer 1, [2] layer 2 .. dh1 (+ dW2), [3] dz1, dx, dW1, [4] planes, [5] total
    unsigned long long tb0 = 0, tb1 = 0, tb2 = 0, tb3 = 0, tbn = 0;
    const unsigned long long tb_start = cv_stamp((float)lane);
#endif
    for (int half = 0; half < 2; ++half) {
        const int pix0 = blk_.grp * 32 + 16 * half;
        const int pix = pix0 + j, pix_m = pix0 + n;
        const bool live = pix < hw, live_m = pix_m < hw;
        const int pu = live ? pix % w : 0, pv = live ? pix / w : 0;
        {   // the current feature, parked in LDS in both lane orders (24 registers less across the sweep)
            const float4* q = (const float4*)(curN + ((size_t)b * hw + (live ? pix : 0)) * C) + c;   // gather order: quarter c of pixel j
#pragma unroll
            for (int s = 0; s < NS; ++s) sCurG[64 * s + lane] = q[4 * s];
            const float4* qm = (const float4*)(curN + ((size_t)b * hw + (live_m ? pix_m : 0)) * C) + g;   // operand order, parked in LDS
#pragma unroll
            for (int s = 0; s < NS; ++s) sCur[64 * s + lane] = qm[4 * s];
        }
#pragma unroll
        for (int t = 0; t < NR; ++t) dcur[t] = 0.0f;
        const float ux = (float)pu + 0.5f, vy = (float)pv + 0.5f;
        const float rx = iK[0] * ux + iK[1] * vy + iK[2];
        const float ry = iK[4] * ux + iK[5] * vy + iK[6];
        const float rz = iK[8] * ux + iK[9] * vy + iK[10];
        const float* const gp = g_out + (size_t)b * D * hw + (live_m ? pix_m : 0);
        float go_next = (d0 < d1 && live_m) ? gp[(size_t)d0 * hw] : 0.0f;
        float depth_next = planes[b * ps_b + min(d0, D - 1) * ps_d];     // (two-pass form: one depth per plane, wave-uniform; one plane ahead)
        // SAVED: the kept inputs of the NEXT plane are loaded while this plane's matrix part runs (both wavefronts of a SIMD otherwise
        // wait for them together at the top of every plane)
        float4 xs_next[RB];
        float2 xm_next = make_float2(0.0f, 0.0f);
        auto load_saved = [&](int dd) __attribute__((always_inline)) {
            const size_t pl = (size_t)b * D + min(dd, D - 1);
            xm_next = live_m ? xm[pl * hw + pix_m] : make_float2(0.0f, 0.0f);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
                xs_next[rb] = live_m ? xs[(pl * (C / 4) + 4 * rb + g) * hw + pix_m] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        };
        if (SAVED) load_saved(d0);

        for (int d = d0; d < d1; ++d) {
            const float depth = depth_next;
            depth_next = planes[b * ps_b + min(d + 1, d1 - 1) * ps_d];
            FS_CV_T(tq0, depth);
            const float go = go_next;
            go_next = live_m ? gp[(size_t)min(d + 1, d1 - 1) * hw] : 0.0f;
#if !FS_BWD16_SAVED_PREFETCH
            if (SAVED && d > d0) load_saved(d);
#endif
            // ---------------- forward recompute: gather (the forward sweep's code, plus the backward's bits) ----------------
            float favg[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) favg[r] = 0.0f;
            float dot_sum = 0.0f, cnt = 0.0f;
            uint32_t flags = 0, rare = 0;   // bit 2k = source k valid (dot != 0), bit 2k+1 = in front (z > 0); rare: bit 2k
            struct Proj { uint32_t off; float tx, ty; uint32_t bits; };
            auto project = [&](const float* P) __attribute__((always_inline)) -> Proj {
                const float X = depth * rx, Y = depth * ry, Z = depth * rz;
                const float qx = P[0] * X + P[1] * Y + P[2] * Z + P[3];
                const float qy = P[4] * X + P[5] * Y + P[6] * Z + P[7];
                const float qz = P[8] * X + P[9] * Y + P[10] * Z + P[11];
                const float zz = qz + 1e-8f;
                const float sc = (fabsf(qz) > 1e-8f) ? 1.0f / zz : 1.0f;
                const float uvx = __fsub_rn(__fmul_rn(__fmul_rn(2.0f, __fmul_rn(qx, sc)), inv_w), 1.0f);
                const float uvy = __fsub_rn(__fmul_rn(__fmul_rn(2.0f, __fmul_rn(qy, sc)), inv_h), 1.0f);
                const float ix = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(uvx, 1.0f), (float)w), 1.0f), 0.5f);
                const float iy = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(uvy, 1.0f), (float)h), 1.0f), 0.5f);
                const float fx0 = floorf(ix), fy0 = floorf(iy);
                const bool xin0 = fx0 >= 0.0f && fx0 <= (float)(w - 1), xin1 = fx0 >= -1.0f && fx0 <= (float)(w - 2);
                const bool yin0 = fy0 >= 0.0f && fy0 <= (float)(h - 1), yin1 = fy0 >= -1.0f && fy0 <= (float)(h - 2);
                const int x0 = xin0 || xin1 ? (int)fx0 : 0, y0 = yin0 || yin1 ? (int)fy0 : 0;
                Proj pr;
                pr.off = (uint32_t)((y0 * w + x0) * C) * 4u;
                pr.tx = ix - fx0; pr.ty = iy - fy0;
                pr.bits = (xin0 ? 1u : 0u) | (xin1 ? 2u : 0u) | (yin0 ? 4u : 0u) | (yin1 ? 8u : 0u) | (zz > 0.0f ? 16u : 0u);
                return pr;
            };
            auto taps = [&](int k, const Proj pr, float (&wv)[NR]) __attribute__((always_inline)) {
                const bool xin0 = pr.bits & 1u, xin1 = pr.bits & 2u, yin0 = pr.bits & 4u, yin1 = pr.bits & 8u;
#pragma unroll
                for (int r = 0; r < NR; ++r) wv[r] = 0.0f;
                const char* base = (const char*)(srcN + (((size_t)b * K + k) * hw) * C);
                const uint32_t off0 = pr.off + 16u * (uint32_t)c;
                // The taps of a source in flight together, two at a time: wave-uniform tests (a source, or a tap, that no pixel of the
                // wavefront sees costs no load instruction), the loads issued back to back -- a tap outside the image reads texel 0
                // with weight 0 --, then the blends.  (Tap by tap, each behind its own branch and wait, the two wavefronts of a SIMD
                // spent most of a plane's time in 4 K serial memory round trips.)
                const bool any_ok = live && (xin0 || xin1) && (yin0 || yin1);
#ifdef FS_BWD16_NO_TAPS   // (timing-only build, WRONG results: the sweep without its tap loads)
                if (false) {
#else
                if (__builtin_amdgcn_ballot_w64(any_ok) != 0ull) {
#endif
#ifndef FS_BWD16_TAPS_IN_FLIGHT
#define FS_BWD16_TAPS_IN_FLIGHT 2      // taps of a source loaded together (4: measured the same, 5 more spilled registers)
#endif
                    constexpr int TF = FS_BWD16_TAPS_IN_FLIGHT;
#pragma unroll
                    for (int t0 = 0; t0 < 4; t0 += TF) {
                        float4 v[TF][NS];
                        float wt[TF];
#pragma unroll
                        for (int tt = 0; tt < TF; ++tt) {
                            const int tap = t0 + tt, ox = tap & 1, oy = tap >> 1;
                            const bool ok = live && (ox ? xin1 : xin0) && (oy ? yin1 : yin0);
                            wt[tt] = ok ? (ox ? pr.tx : 1.0f - pr.tx) * (oy ? pr.ty : 1.0f - pr.ty) : 0.0f;
#pragma unroll
                            for (int s = 0; s < NS; ++s) v[tt][s] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                            const float4* q = (const float4*)(base + (ok ? off0 + (uint32_t)((oy * w + ox) * C) * 4u : 16u * (uint32_t)c));
                            if (__builtin_amdgcn_ballot_w64(ok) != 0ull) {     // (a tap no pixel of the wavefront has: no load instructions)
#pragma unroll
                                for (int s = 0; s < NS; ++s) v[tt][s] = q[4 * s];
                            }
                        }
#pragma unroll
                        for (int tt = 0; tt < TF; ++tt)
#pragma unroll
                            for (int s = 0; s < NS; ++s) {
                                wv[4 * s] = fmaf(wt[tt], v[tt][s].x, wv[4 * s]); wv[4 * s + 1] = fmaf(wt[tt], v[tt][s].y, wv[4 * s + 1]);
                                wv[4 * s + 2] = fmaf(wt[tt], v[tt][s].z, wv[4 * s + 2]); wv[4 * s + 3] = fmaf(wt[tt], v[tt][s].w, wv[4 * s + 3]);
                            }
                    }
                }
            };
            auto gather = [&](int k, const Proj pr) __attribute__((always_inline)) {
                float wv[NR];
                taps(k, pr, wv);
                const bool front = pr.bits & 16u;
                float part = 0.0f;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const float4 cu = sCurG[64 * s + lane];
                    part += wv[4 * s] * cu.x; part += wv[4 * s + 1] * cu.y; part += wv[4 * s + 2] * cu.z; part += wv[4 * s + 3] * cu.w;
                }
                part += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(part), 0xB1, 0xF, 0xF, true));   // lane ^ 1
                part += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(part), 0x4E, 0xF, 0xF, true));   // lane ^ 2
                const float dotk = front ? part : 0.0f;
                flags |= (front ? 2u : 0u) << (2 * k);
                // in front, some tap inside the source image, and still an exactly zero score (all-zero features): the score's
                // gradient reaches the current feature although the source is not averaged -- re-gathered below
                if (live && front && dotk == 0.0f && (pr.bits & 3u) && (pr.bits & 12u)) rare |= 1u << (2 * k);
                if (dotk != 0.0f) {
                    flags |= 1u << (2 * k);
                    cnt += 1.0f;
                    dot_sum += dotk;
#pragma unroll
                    for (int r = 0; r < NR; ++r) favg[r] += wv[r];
                }
            };
            for (int k0 = 0; !SAVED && k0 < K; k0 += 4) {
                float Pk[12];
                if (k0 == 0) {
                    const float4 p0 = sPq[0], p1 = sPq[1], p2 = sPq[2];
                    Pk[0] = p0.x; Pk[1] = p0.y; Pk[2] = p0.z; Pk[3] = p0.w; Pk[4] = p1.x; Pk[5] = p1.y; Pk[6] = p1.z; Pk[7] = p1.w;
                    Pk[8] = p2.x; Pk[9] = p2.y; Pk[10] = p2.z; Pk[11] = p2.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 12; ++e) Pk[e] = Pmat[((size_t)b * K + min(k0 + c, K - 1)) * 12 + e];
                }
                const Proj mine = project(Pk);
                auto from = [&](auto sel) __attribute__((always_inline)) {
                    constexpr int q = decltype(sel)::value, ctl = q * 0x55;       // quad_perm: every lane reads lane q of its quad
                    Proj pr;
                    pr.off = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine.off, ctl, 0xF, 0xF, true);
                    pr.tx = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mine.tx), ctl, 0xF, 0xF, true));
                    pr.ty = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mine.ty), ctl, 0xF, 0xF, true));
                    pr.bits = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine.bits, ctl, 0xF, 0xF, true);
                    return pr;
                };
                gather(k0, from(std::integral_constant<int, 0>{}));
                if (k0 + 1 < K) gather(k0 + 1, from(std::integral_constant<int, 1>{}));
                if (k0 + 2 < K) gather(k0 + 2, from(std::integral_constant<int, 2>{}));
                if (k0 + 3 < K) gather(k0 + 3, from(std::integral_constant<int, 3>{}));
            }
            const float inv_g = 1.0f / (cnt + 1e-8f);
            const float dot_g = dot_sum * inv_g;
            FS_CV_T(tq1, dot_g + favg[0] + favg[NR - 1]);
            // The matrix part below is laid out as PHASES separated by scheduling barriers: every phase first issues the LDS reads
            // of the NEXT phase's operands (weight images, tile rows), then runs its MFMAs -- left to itself the scheduler put each
            // ds_read / ds_bpermute directly in front of the MFMA that uses it (s_waitcnt lgkmcnt(0) before almost every MFMA).
#define FS_PHASE() __builtin_amdgcn_sched_barrier(0)
            __builtin_amdgcn_s_setprio(0);
            // ---- phase 0: x = favg / cnt into its tile (B operand of dW1) and, by ds_bpermute, to the operand order; W1 operands ----
            float xop[NT];
            uint32_t flags_m;
            float dot_m;
            if (SAVED) {
                // from the training forward: lane (n, g)'s operand-order x is chunk 4 rb + g of the point, one float4 per block
                dot_m = xm_next.x;
                flags_m = __float_as_uint(xm_next.y);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    const float4 v4 = xs_next[rb];
                    xop[4 * rb] = v4.x; xop[4 * rb + 1] = v4.y; xop[4 * rb + 2] = v4.z; xop[4 * rb + 3] = v4.w;
                }
#if FS_BWD16_SAVED_PREFETCH
                if (d + 1 < d1) load_saved(d + 1);
#endif
#pragma unroll
                for (int t = 0; t < NR; ++t) tX[256 * (t >> 2) + wr_at[t & 3]] = xop[t];
                xop[NR] = g == 0 ? dot_m : (g == 1 ? 1.0f : 0.0f);
                // a source in front that is not averaged MAY have had an exactly zero score with taps inside (the forward raised the
                // header flag if any did): re-gathered below; one whose taps are all outside contributes zeros there
                if (xhdr[0] != 0u) rare = __builtin_amdgcn_ds_bpermute((lane >> 2) * 4, (int)((flags_m >> 1) & ~flags_m & 0x55555555u));
            } else {
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const float x = favg[r] * inv_g;
                    xop[r] = __int_as_float(__builtin_amdgcn_ds_bpermute(pull, __float_as_int(x)));
                    tX[256 * (r >> 2) + xw_at + 4 * ((r & 3) ^ xw_k)] = x;
                }
                xop[NR] = __int_as_float(__builtin_amdgcn_ds_bpermute(pull, __float_as_int(c == 0 ? dot_g : (c == 1 ? 1.0f : 0.0f))));
                flags_m = (uint32_t)__builtin_amdgcn_ds_bpermute(pull0, (int)flags);
                dot_m = __int_as_float(__builtin_amdgcn_ds_bpermute(pull0, __float_as_int(dot_g)));
            }
            FS_PHASE();
            // ---- phase 1: W2 operands, b2, w3 on their way; layer 1 ----
            const float4 b2A = sB2[0], b2B = sB2[4], w3A = sW3[0], w3B = sW3[4];
            FS_PHASE();
            f32x4 z1[2] = {f32x4{0.0f, 0.0f, 0.0f, 0.0f}, f32x4{0.0f, 0.0f, 0.0f, 0.0f}};
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) z1[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(sA1[(NT * blk + t) * 64], xop[t], z1[blk], 0, 0, 0);
            FS_PHASE();
            FS_CV_T(tq2, z1[0][0] + z1[1][3]);
            // ---- phase 2: h1 (to its tile), W2^T operands on their way; layer 2 ----
            const float inv = 1.0f / ((float)__builtin_popcount(flags_m & 0x55555555u) + 1e-8f);
            uint32_t pos1 = 0;      // lrelu'(z1) as a bit per (blk, r)
            f32x4 h1[2];
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pos1 |= (z1[blk][r] > 0.0f ? 1u : 0u) << (4 * blk + r);
                    h1[blk][r] = lrelu(z1[blk][r]);
                    tH1[256 * blk + wr_at[r]] = h1[blk][r];
                }
            FS_PHASE();
            f32x4 z2[2] = {f32x4{b2A.x, b2A.y, b2A.z, b2A.w}, f32x4{b2B.x, b2B.y, b2B.z, b2B.w}};
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) z2[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(sA2[(8 * blk + t) * 64], h1[t >> 2][t & 3], z2[blk], 0, 0, 0);
            FS_PHASE();
            // ---- phase 3: dz2 (to its tile) and the sums over it; the h1 / dz2 tile rows (dW2 operands) on their way; dh1 ----
            f32x4 dz2[2];
            {
                const float w3v[2][4] = {{w3A.x, w3A.y, w3A.z, w3A.w}, {w3B.x, w3B.y, w3B.z, w3B.w}};
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float z = z2[blk][r];
                        dz2[blk][r] = go * w3v[blk][r] * dlrelu(z);
                        s_w3[blk][r] = fmaf(go, lrelu(z), s_w3[blk][r]);
                        s_b2[blk][r] += dz2[blk][r];
                        tD2[256 * blk + wr_at[r]] = dz2[blk][r];
                    }
            }
            if (g == 0) s_b3 += go;
            wave_lds_sync();
            const float4 H0 = *(const float4*)(tH1 + rd_at), H1 = *(const float4*)(tH1 + 256 + rd_at);
            const float4 E0 = *(const float4*)(tD2 + rd_at), E1 = *(const float4*)(tD2 + 256 + rd_at);
            FS_PHASE();
            f32x4 dz1[2] = {f32x4{0.0f, 0.0f, 0.0f, 0.0f}, f32x4{0.0f, 0.0f, 0.0f, 0.0f}};
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int bo = 0; bo < 2; ++bo) dz1[bo] = __builtin_amdgcn_mfma_f32_16x16x4f32(sA2T[(8 * bo + t) * 64], dz2[t >> 2][t & 3], dz1[bo], 0, 0, 0);
            FS_PHASE();
            FS_CV_T(tq3, dz1[0][0] + dz1[1][3]);
            // ---- phase 4: W1^T operands on their way; dW2 (independent of dz1: the matrix pipe runs while dh1 drains) ----
            const float4 w1dA = sW1d[0], w1dB = sW1d[4];
            FS_PHASE();
            {
                const float av[2][4] = {{E0.x, E0.y, E0.z, E0.w}, {E1.x, E1.y, E1.z, E1.w}};
                const float hv[2][4] = {{H0.x, H0.y, H0.z, H0.w}, {H1.x, H1.y, H1.z, H1.w}};
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb)
#ifdef FS_BWD16_NO_DW     // (timing-only build, WRONG weight gradients: the sweep without its two outer-product phases)
                            acc2[rb][cb][s] += av[rb][s] + hv[cb][s];
#else
                            acc2[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rb][s], hv[cb][s], acc2[rb][cb], 0, 0, 0);
#endif
            }
            FS_PHASE();
            // ---- phase 5: dz1 (to its tile) and the sums over it; its tile rows and x's (dW1 operands) on their way; dx ----
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dz1[blk][r] *= ((pos1 >> (4 * blk + r)) & 1u) ? 1.0f : 0.01f;
                    s_b1[blk][r] += dz1[blk][r];
                    s_dot[blk][r] = fmaf(dz1[blk][r], dot_m, s_dot[blk][r]);
                    tD1[256 * blk + wr_at[r]] = dz1[blk][r];
                }
            wave_lds_sync();
            const float4 F0 = *(const float4*)(tD1 + rd_at), F1 = *(const float4*)(tD1 + 256 + rd_at);
            float4 X4[RB];
#pragma unroll
            for (int cb = 0; cb < RB; ++cb) X4[cb] = *(const float4*)(tX + 256 * cb + rd_at);
            FS_PHASE();
            // dx = W1^T dz1: register (rb, r) = channel 16 rb + 4 g + r of pixel n
            f32x4 dx[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) dx[rb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) dx[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(sA1T[(8 * rb + t) * 64], dz1[t >> 2][t & 3], dx[rb], 0, 0, 0);
            FS_PHASE();
            // ---- phase 6: the current feature (record) on its way, d dot; dW1 ----
            float4 cv4[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) cv4[rb] = sCur[64 * rb + lane];
            float ddot;
            {
                float a = w1dA.x * dz1[0][0];
                a = fmaf(w1dA.y, dz1[0][1], a); a = fmaf(w1dA.z, dz1[0][2], a); a = fmaf(w1dA.w, dz1[0][3], a);
                a = fmaf(w1dB.x, dz1[1][0], a); a = fmaf(w1dB.y, dz1[1][1], a); a = fmaf(w1dB.z, dz1[1][2], a); a = fmaf(w1dB.w, dz1[1][3], a);
                a += __shfl_xor(a, 16, 64);
                ddot = a + __shfl_xor(a, 32, 64);
            }
            // ---- the record of the point leaves HERE, in front of dW1's MFMAs: at the end of the plane its stores were still in flight
            //      at the next plane's first tap wait (the memory counter is in-order), ~1.5 k cycles per plane ----
            const float di = ddot * inv;
            if (live_m) {
                const size_t pl = (size_t)b * D + d;
#if FS_REC_PIXEL_MAJOR
                // pixel-major records ([view, plane][pixel][C/4 chunks]: 4 C contiguous bytes per point): pass 2 reads the pixels of
                // a ~10-pixel-wide box row by row, and with chunk-planar records a row was 160 useful bytes of every 256 fetched
                float4* rp = recS + (pl * hw + pix_m) * (C / 4) + g;
#else
                float4* rp = recS + (pl * (C / 4) + g) * hw + pix_m;
#endif
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    const float4 cv = cv4[rb];
                    rp[FS_REC_PIXEL_MAJOR ? (size_t)(4 * rb) : (size_t)(4 * rb) * hw] = make_float4(fmaf(di, cv.x, dx[rb][0] * inv), fmaf(di, cv.y, dx[rb][1] * inv),
                                                            fmaf(di, cv.z, dx[rb][2] * inv), fmaf(di, cv.w, dx[rb][3] * inv));
                }
                if (g == 0) recM[pl * hw + pix_m] = make_float2(di, __uint_as_float(flags_m));
            }
            FS_PHASE();
            {
                const float av[2][4] = {{F0.x, F0.y, F0.z, F0.w}, {F1.x, F1.y, F1.z, F1.w}};
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                        for (int cb = 0; cb < RB; ++cb) {
                            const float xv = s == 0 ? X4[cb].x : (s == 1 ? X4[cb].y : (s == 2 ? X4[cb].z : X4[cb].w));
#ifdef FS_BWD16_NO_DW
                            acc1[rb][cb][s] += av[rb][s] + xv;
#else
                            acc1[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rb][s], xv, acc1[rb][cb], 0, 0, 0);
#endif
                        }
            }
            FS_PHASE();
#undef FS_PHASE
            wave_lds_sync();   // (the next plane overwrites the tiles)
            __builtin_amdgcn_s_setprio(1);
#ifdef FS_CV_TRACE
            {
                FS_CV_T(tq4, acc1[0][0][0] + acc1[1][RB - 1][3] + dx[0][0]);
                tb0 += tq1 - tq0; tb1 += tq2 - tq1; tb2 += tq3 - tq2; tb3 += tq4 - tq3; tbn += 1;
            }
#endif
            // d cur = d dot / cnt * sum_k [z_k > 0] warped_k; for every source that counts [z_k > 0] = valid_k: the sum is cnt * x
            // (x comes back from its tile -- lane (n, g)'s operand-order values sit at the accumulator-order addresses -- instead of
            //  staying in 12 registers through the whole matrix part)
#pragma unroll
            for (int t = 0; t < NR; ++t) dcur[t] = fmaf(ddot, tX[256 * (t >> 2) + wr_at[t & 3]], dcur[t]);
            if (__builtin_amdgcn_ballot_w64(rare != 0u) != 0ull) {      // (never taken on real data; wave-uniform)
                // (gather order: pixel j's d dot / cnt from operand lane j)
                const float di_g = __int_as_float(__builtin_amdgcn_ds_bpermute((lane >> 2) * 4, __float_as_int(di)));
                for (int k = 0; k < K; ++k) {
                    if (__builtin_amdgcn_ballot_w64(((rare >> (2 * k)) & 1u) != 0u) == 0ull) continue;
                    float Pk[12];
#pragma unroll
                    for (int e = 0; e < 12; ++e) Pk[e] = Pmat[((size_t)b * K + k) * 12 + e];
                    float wv[NR];
                    taps(k, project(Pk), wv);
                    const float cd = ((rare >> (2 * k)) & 1u) ? di_g : 0.0f;
#pragma unroll
                    for (int r = 0; r < NR; ++r)
                        dcur[r] += __int_as_float(__builtin_amdgcn_ds_bpermute(pull, __float_as_int(cd * wv[r])));
                }
            }
        }
        // ---- this half's d cur: the four wavefronts' planes summed through the (now free) tiles ([pixel][channel]); plain stores
        //      when no other workgroup shares the pixels ----
        wave_lds_sync();
#pragma unroll
        for (int t = 0; t < NR; ++t) wv_[n * C + 16 * (t >> 2) + 4 * g + (t & 3)] = dcur[t];
        __syncthreads();
        {
            const int npx = max(0, min(16, hw - pix0));
            float* const dst = d_curN + ((size_t)b * hw + pix0) * C;
            for (int e = threadIdx.x; e < npx * C; e += 256) {
                const float v = s_all[kWts + e] + s_all[kWts + kWave + e] + s_all[kWts + 2 * kWave + e] + s_all[kWts + 3 * kWave + e];
                if (slices > 1) atomicAdd(dst + e, v); else dst[e] = v;
            }
        }
        __syncthreads();
    }
#ifdef FS_CV_TRACE
    {
        const int wid = (int)blockIdx.x * 4 + wave;
        if (lane == 0 && wid < kCvTraceWaves) {
            unsigned long long* o = g_cvb_trace + 6 * (size_t)wid;
            o[0] = tb0; o[1] = tb1; o[2] = tb2; o[3] = tb3; o[4] = tbn; o[5] = cv_stamp((float)lane) - tb_start;
        }
    }
#endif
    // ---- the six MLP gradients: per wavefront into its region, summed over the wavefronts, one atomic per weight ----
    {
        float* const stg = wv_;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) stg[((2 * rb + cb) * 4 + r) * 64 + lane] = acc2[rb][cb][r];
#pragma unroll
            for (int cb = 0; cb < RB; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) stg[(16 + (RB * rb + cb) * 4 + r) * 64 + lane] = acc1[rb][cb][r];
        }
        // the per-lane sums: over the 16 pixels of the lane's row (lanes n + 16 g: xor 1, 2, 4, 8), then lane n == 0 of each g
        auto row_sum = [&](float v) __attribute__((always_inline)) {
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            return v;
        };
        float* const sums = stg + kAccRegs * 64;     // [kind 0..3][unit 32], then b3
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v0 = row_sum(s_dot[blk][r]), v1 = row_sum(s_b1[blk][r]), v2 = row_sum(s_b2[blk][r]), v3 = row_sum(s_w3[blk][r]);
                if (n == 0) {
                    const int u = 16 * blk + 4 * g + r;
                    sums[u] = v0; sums[32 + u] = v1; sums[64 + u] = v2; sums[96 + u] = v3;
                }
            }
        {
            float v = row_sum(s_b3);     // (only the g == 0 row counted)
            if (lane == 0) sums[128] = v;
        }
    }
    __syncthreads();
    {
        const float* const st0 = s_all + kWts;
        for (int e = threadIdx.x; e < kAccRegs * 64; e += 256) {
            const float v = st0[e] + st0[kWave + e] + st0[2 * kWave + e] + st0[3 * kWave + e];
            const int idx = e >> 6, l = e & 63, r = idx & 3, row4 = l >> 4, col = l & 15;
            if (idx < 16) {
                const int rb = idx >> 3, cb = (idx >> 2) & 1;
                atomicAdd(&gw2[(16 * rb + 4 * row4 + r) * 32 + 16 * cb + col], v);
            } else {
                const int q = (idx - 16) >> 2, rb = q / RB, cb = q - rb * RB;
                atomicAdd(&gw1[(16 * rb + 4 * row4 + r) * (C + 1) + 16 * cb + col], v);
            }
        }
        if (threadIdx.x < 129) {
            const int e = kAccRegs * 64 + threadIdx.x;
            const float v = st0[e] + st0[kWave + e] + st0[2 * kWave + e] + st0[3 * kWave + e];
            const int kind = threadIdx.x >> 5, u = threadIdx.x & 31;
            if (threadIdx.x == 128) atomicAdd(gb3, v);
            else if (kind == 0) atomicAdd(&gw1[u * (C + 1) + C], v);
            else if (kind == 1) atomicAdd(&gb1[u], v);
            else if (kind == 2) atomicAdd(&gb2[u], v);
            else atomicAdd(&gw3[u], v);
        }
    }
}

// ==========================================================================================
// Backward, second pass (round 4): the source-feature gradient WITHOUT global atomics.
//   d src_k[t] = sum over (pixel p, plane d) whose bilinear taps cover texel t of
//                w_tap * (valid_k * dfavg'(p, d) + [z_k > 0] * ddot'(p, d) * cur(p))
// with dfavg' = d favg / cnt, ddot' = d dot / cnt and the two bits per source from the first pass's records.  A workgroup
// (ONE wavefront) owns a TILE of 8 x 8 source texels of one (view, source) -- its gradient lives in LDS, texel-major rows of
// C + 4 floats, for the whole sweep -- and walks the planes: for plane d the current pixels that can touch the tile are the preimage of the tile
// rectangle (grown by the bilinear footprint) under the plane-induced homography, a convex quadrilateral whose bounding
// box comes from the four corners through the INVERSE homography (cv_bwd_prep_kernel; profiles/tools/cv_tile_box_count.py
// checks on the CPU that no pixel with a tap in a tile falls outside its box, and counts 1.0 - 1.1 visited pixels per
// (pixel, plane, source)).  Every pixel of the box is projected exactly as the first pass did, and the taps that fall
// into the tile are added into LDS (see below how); the tile leaves once, as plain stores in the caller's [C, h, w] layout.
// Tiles whose corners are all behind the source (c < 0 <=> z_k < 0) skip the plane; a tile that straddles the plane's
// horizon walks the whole image (never seen with forward-looking cameras; exercised by the tests' turned-round source).
// ==========================================================================================
// One launch for the backward's small preparations (seven hipMemsetAsync + two kernels before): the projection rows P of every
// (view, source), the inverse plane homographies of the two-pass form (each thread forms the P it needs itself), and zeros in the
// six MLP-gradient tensors, which the sweep accumulates into with atomics.  At the native size a training step is ~1.25 ms of
// kernels, and every launch is ~5 us of it.
__global__ __launch_bounds__(256) void cv_bwd_prep_kernel(
    int n_maps, int K, int D, int C, int with_ginv, const float* __restrict__ src_Ks, const float* __restrict__ src_extrinsics,
    const float* __restrict__ cur_invK, const float* __restrict__ planes, long long ps_b, long long ps_d, float* __restrict__ Pmat,
    float* __restrict__ Ginv, float* __restrict__ gw1, float* __restrict__ gb1, float* __restrict__ gw2, float* __restrict__ gb2,
    float* __restrict__ gw3, float* __restrict__ gb3)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    auto p_entry = [&](int m, int i, int j) {
        const float* Ks = src_Ks + (size_t)m * 16;
        const float* Tx = src_extrinsics + (size_t)m * 16;
        return Ks[4 * i] * Tx[j] + Ks[4 * i + 1] * Tx[4 + j] + Ks[4 * i + 2] * Tx[8 + j] + Ks[4 * i + 3] * Tx[12 + j];   // (= cv_proj_kernel)
    };
    if (e < n_maps * 12) Pmat[e] = p_entry(e / 12, (e % 12) / 4, e % 4);
    const int nw1 = 32 * (C + 1);
    if (e < nw1) gw1[e] = 0.0f;
    if (e < 32 * 32) gw2[e] = 0.0f;
    if (e < 32) { gb1[e] = 0.0f; gb2[e] = 0.0f; gw3[e] = 0.0f; }
    if (e == 0) gb3[0] = 0.0f;
    if (with_ginv && e < n_maps * D) {
        const int m = e / D, d = e - m * D, b = m / K;
        const double depth = (double)planes[b * ps_b + d * ps_d];
        const float* iK = cur_invK + (size_t)b * 16;
        float P[12];
#pragma unroll
        for (int q = 0; q < 12; ++q) P[q] = p_entry(m, q / 4, q % 4);
        double G[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double a = 0.0;
                for (int l = 0; l < 3; ++l) a += (double)P[4 * i + l] * (double)iK[4 * l + j];
                G[3 * i + j] = depth * a + (j == 2 ? (double)P[4 * i + 3] : 0.0);
            }
        const double c00 = G[4] * G[8] - G[5] * G[7], c01 = G[5] * G[6] - G[3] * G[8], c02 = G[3] * G[7] - G[4] * G[6];
        const double det = G[0] * c00 + G[1] * c01 + G[2] * c02;
        const double id = 1.0 / det;
        const double nan_ = __longlong_as_double(0x7ff8000000000000ll);
        const bool bad = !(fabs(det) > 0.0) || !(fabs(id) < 1e300);
        const double r[9] = {c00, G[2] * G[7] - G[1] * G[8], G[1] * G[5] - G[2] * G[4],
                             c01, G[0] * G[8] - G[2] * G[6], G[2] * G[3] - G[0] * G[5],
                             c02, G[1] * G[6] - G[0] * G[7], G[0] * G[4] - G[1] * G[3]};
        float* o = Ginv + (size_t)e * 9;
        for (int i = 0; i < 9; ++i) o[i] = (float)(bad ? nan_ : r[i] * id);
    }
}

#ifndef FS_SG_TH
#define FS_SG_TH 8            // (A/B builds: -DFS_SG_TH=4 halves the tile and its LDS: twice the wavefronts per CU, more halo)
#endif
constexpr int kSgTW = 8, kSgTH = FS_SG_TH, kSgG = 4;
#ifdef FS_CV_SG_STATS   // debug build: [0] wave iterations, [1] pixels with a tap in the tile, [2] (tile, plane) cells walked,
__device__ unsigned long long g_sg_stats[8];   // [3] whole-image fallbacks, [4] cells skipped (behind), [5] box pixels, [6] claim rounds
#define FS_SG_COUNT(i, n) do { atomicAdd(&g_sg_stats[i], (unsigned long long)(n)); } while (0)
#else
#define FS_SG_COUNT(i, n) do {} while (0)
#endif

// How the taps are ADDED.  LDS float atomics are not an option on this chip: ds_add_f32 costs 193 cycles per wave
// instruction per CU -- ds_add_u32 4.3, a ds_read_b32 + ds_write_b32 pair 11.7 (profiles/tools/lds_atomic_rate.hip,
// profiles/r4_lds_atomic_rate.txt); the first build of this kernel, (pixel, quarter) lanes adding with ds_add_f32, ran
// 116 ms on the 10-view K = 8 shape, 3x slower than the global atomics it was meant to replace.  So the adds are plain
// read-modify-writes, made safe by construction: a workgroup is ONE wavefront with a tile of 8 x 8 texels (no other
// wavefront ever touches its accumulators, no barriers), lane = pixel with all C channels (texel-major accumulator rows
// of C + 4 floats: a tap is C/4 ds_read_b128 + C/4 ds_write_b128, and the 208-byte row stride keeps neighbouring
// texels conflict-free); inside the wavefront two pixels collide iff they have the same tap base (x0, y0) (the taps
// of one pixel are distinct texels, and the four taps are four instruction groups, executed in order): each pending
// lane writes its id to claim[base], reads it back, the survivors add their four taps, the others go round again -- one
// round at >= 1 texel per pixel, n rounds where n pixels share a base.  (The second build split the CHANNELS over the
// four wavefronts of a 16 x 16 tile: every wavefront repeated the projection, and a serial chain of record loads, claim
// and four dependent read-modify-writes per 64 pixels left it latency-bound -- 9.9 ms on the 10-view K = 8 shape, of which
// the adds were hidden entirely: profiles/r4_cv_sg_v2_variants.txt.)
//   A tile's box is ~10 x 10 pixels, so kSgG = 4 planes (a quarter of the sweep apart) are walked together as one list of
// pixels (plane by lane);
// the four boxes are computed by 16 lanes at once (lane 4 g + q: corner q of plane g, quad reductions).  The loads of the
// next 64 pixels are issued before the taps of the current 64 are added.
// NAT: the records (and the current map of the rare path) are in natural channel order -- cost_volume16_bwd_kernel's -- instead of
// the 32-pixel kernel's [parity][C/2] order.
// FORM 1 (round 6): the tile's gradient lives in REGISTERS -- lane = texel of the 8 x 8 tile, C accumulators each -- and a batch's taps
// reach it through LDS once instead of four read-modify-writes: every pixel lane writes its record to a staging row (C/4
// ds_write_b128) and appends (pixel lane, weight) to the list of each texel its taps cover (a ds_add_rtn_u32 slot counter per
// texel, kSgCap entries; a tap that finds its texel's list full goes round again), then every texel lane walks its list and blends
// the staged records it names (C/4 ds_read_b128 per entry).  No claim rounds, no accumulator writes: per 64 visited pixels
// 12 b128 writes + ~4.5 x 12 b128 reads instead of 48 + 48 (a ds_write_b128 costs 13 LDS cycles, a read 4).
#ifndef FS_SG_CAP
#define FS_SG_CAP 8
#endif
constexpr int kSgCap = FS_SG_CAP;
template <int C, bool NAT, int FORM>
__global__ __launch_bounds__(64) void cv_src_grad_kernel(
    int B, int K, int h, int w, int D, int chunks, int tiles_x, int tiles_y, const float* __restrict__ curT,
    const float4* __restrict__ recS, const float2* __restrict__ recM, const float* __restrict__ Pmat,
    const float* __restrict__ Ginv, const float* __restrict__ cur_invK, const float* __restrict__ planes, long long ps_b,
    long long ps_d, float* __restrict__ d_src)
{
    constexpr int TW = kSgTW, TH = kSgTH, NT = TW * TH, NV = C / 4, ST = C + 4, HC = C / 2, G = kSgG;
    constexpr int kClaim = (TW + 1) * (TH + 1) + 3;   // bases (lx, ly) in [-1, TW - 1] x [-1, TH - 1]
    static_assert(FORM == 0 || NT == 64, "the register form keeps one texel per lane");
    __shared__ __attribute__((aligned(16))) float acc[NT * ST];          // FORM 0: the accumulators; FORM 1: the batch's staged records
    __shared__ uint32_t claim_[FORM == 0 ? kClaim : 64];                 // FORM 1: entries in each texel's list
    __shared__ uint2 lst_[FORM == 0 ? 1 : 64 * kSgCap];                  // FORM 1: (pixel lane, weight bits)
    float4 accr[NV];                                                     // FORM 1: this lane's texel
    const int hw = h * w, T = tiles_x * tiles_y;
    // XCD x (= workgroup id % 8) owns the x-th contiguous range of tiles of every view: the workgroups in flight on an XCD
    // -- the same tiles of all K sources, which read the same records -- share one L2
    const int xcd = (int)(blockIdx.x & 7u), jb = (int)(blockIdx.x >> 3);
    const int gb = (T + 7) >> 3, per_b = gb * K * chunks;
    const int b = jb / per_b;
    const int r0 = jb - b * per_b, tl = r0 / (K * chunks), r1 = r0 - tl * (K * chunks);
    const int k = r1 / chunks, chunk = r1 - k * chunks;
    const int tile = xcd * gb + tl;
    if (b >= B || tile >= T) return;   // (workgroup-uniform)
    const int tx0 = (tile % tiles_x) * TW, ty0 = (tile / tiles_x) * TH;
    const int tw_ = min(TW, w - tx0), th_ = min(TH, h - ty0);
    const int lane = threadIdx.x;
    if (FORM == 0) {
        for (int e = lane; e < NT * ST / 4; e += 64) ((float4*)acc)[e] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    } else {
        claim_[lane] = 0u;
#pragma unroll
        for (int s = 0; s < NV; ++s) accr[s] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    wave_lds_sync();
    volatile uint32_t* const claim = claim_;
    const float* P = Pmat + ((size_t)b * K + k) * 12;
    const float* iK = cur_invK + (size_t)b * 16;
    const float inv_w = (float)(1.0 / (double)w), inv_h = (float)(1.0 / (double)h);
    const int dchunk = (D + chunks - 1) / chunks;
    const int d0 = min(D, chunk * dchunk), d1 = min(D, d0 + dchunk);
    const float X0 = (float)tx0 - 0.55f, X1 = (float)(tx0 + tw_) + 0.55f;     // qx / qz = ix + 0.5, taps at floor(ix), + 1
    const float Y0 = (float)ty0 - 0.55f, Y1 = (float)(ty0 + th_) + 0.55f;
    const uint32_t kbit = 2u * (uint32_t)k;

    struct Geo { int t00, cid; float tx, ty; uint32_t okm; bool pend; };

    // The four planes of a group lie a quarter of the sweep apart (dg + g * gstride): walked together, the same pixel in
    // two of them samples texels many disparity steps apart -- with neighbouring planes (~0.56 texel apart at the native
    // size) every batch that straddles two planes had lanes with the same tap base, i.e. a second claim round.
    const int gstride = (d1 - d0 + G - 1) / G;
    for (int dg = d0; dg < d0 + gstride; ++dg) {
        // ---- the tile's preimage boxes in the current view, four planes at once: lane 4 g + q = corner q of its plane ----
        int bx0s[G], by0s[G], Wbs[G], ns[G];
        float rcps[G], deps[G];
        {
            const int g = (lane >> 2) & 3, q = lane & 3, d = dg + g * gstride;
            const float* Gi = Ginv + (((size_t)b * K + k) * D + min(d, D - 1)) * 9;
            const float X = (q & 1) ? X1 : X0, Y = (q & 2) ? Y1 : Y0;
            const float cu = fmaf(Gi[0], X, fmaf(Gi[1], Y, Gi[2]));
            const float cv_ = fmaf(Gi[3], X, fmaf(Gi[4], Y, Gi[5]));
            const float cc = fmaf(Gi[6], X, fmaf(Gi[7], Y, Gi[8]));
            auto qx1 = [](float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true)); };   // lane ^ 1
            auto qx2 = [](float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true)); };   // lane ^ 2
            auto qmin = [&](float v) { v = fminf(v, qx1(v)); return fminf(v, qx2(v)); };
            auto qmax = [&](float v) { v = fmaxf(v, qx1(v)); return fmaxf(v, qx2(v)); };
            float csum = cc + qx1(cc);
            csum += qx2(csum);
            const float cmin = qmin(cc), cmax = qmax(cc);
            const float amax = fmaxf(fabsf(cmin), fabsf(cmax));
            const bool finite = csum * 0.0f == 0.0f;                               // (NaN / inf in the inverse: whole image)
            const bool behind = finite && cmax < -1e-3f * amax;                    // tile entirely behind the source: z_k < 0
            const bool front = finite && cmin > 1e-3f * amax;
            const float u = cu / cc - 0.5f, v = cv_ / cc - 0.5f;
            const float umin = qmin(u), umax = qmax(u), vmin = qmin(v), vmax = qmax(v);
            int bx0 = 0, bx1 = w - 1, by0 = 0, by1 = h - 1;
            const bool boxed = front && (umin + umax + vmin + vmax) * 0.0f == 0.0f;
            if (boxed) {
                bx0 = (int)fminf(fmaxf(ceilf(umin - 0.05f), 0.0f), (float)w);
                bx1 = (int)fmaxf(fminf(floorf(umax + 0.05f), (float)(w - 1)), -1.0f);
                by0 = (int)fminf(fmaxf(ceilf(vmin - 0.05f), 0.0f), (float)h);
                by1 = (int)fmaxf(fminf(floorf(vmax + 0.05f), (float)(h - 1)), -1.0f);
            }
            int Wb = max(0, bx1 - bx0 + 1), Hb = max(0, by1 - by0 + 1);
            if (behind || d >= d1) Wb = 0;
            const int n = Wb * Hb;
            const float rcp = 1.0f / (float)max(Wb, 1);
            const float dep = planes[b * ps_b + (long long)min(d, D - 1) * ps_d];
#ifdef FS_CV_SG_STATS
            if (lane < 16 && q == 0 && d < d1) {
                if (behind) FS_SG_COUNT(4, 1); else if (n > 0) { FS_SG_COUNT(2, 1); FS_SG_COUNT(5, n); if (!boxed) FS_SG_COUNT(3, 1); }
            }
#endif
#pragma unroll
            for (int gg = 0; gg < G; ++gg) {
                bx0s[gg] = __builtin_amdgcn_readlane(bx0, 4 * gg);
                by0s[gg] = __builtin_amdgcn_readlane(by0, 4 * gg);
                Wbs[gg] = __builtin_amdgcn_readlane(Wb, 4 * gg);
                ns[gg] = __builtin_amdgcn_readlane(n, 4 * gg);
                rcps[gg] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rcp), 4 * gg));
                deps[gg] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dep), 4 * gg));
            }
        }
        const int o1 = ns[0], o2 = o1 + ns[1], o3 = o2 + ns[2], N = o3 + ns[3];
        if (N == 0) continue;

        // ---- one batch of 64 pixels: geometry (the first pass's projection, op by op: warp_source), record loads ----
        auto stage = [&](int i0, Geo& ge, float4 (&S)[NV]) __attribute__((always_inline)) {
            const int i = i0 + lane;
            const bool act = i < N;
            const int gs = (i >= o1 ? 1 : 0) + (i >= o2 ? 1 : 0) + (i >= o3 ? 1 : 0);
            auto sel = [&](auto a0, auto a1, auto a2, auto a3) { return gs == 0 ? a0 : (gs == 1 ? a1 : (gs == 2 ? a2 : a3)); };
            const int loc = i - sel(0, o1, o2, o3);
            const int Wb = sel(Wbs[0], Wbs[1], Wbs[2], Wbs[3]);
            const float rcpW = sel(rcps[0], rcps[1], rcps[2], rcps[3]);
            const float depth = sel(deps[0], deps[1], deps[2], deps[3]);
            int row = (int)(((float)loc + 0.5f) * rcpW), col = loc - row * Wb;
            if (col < 0) { --row; col += Wb; } else if (col >= Wb) { ++row; col -= Wb; }
            const int pu = act ? sel(bx0s[0], bx0s[1], bx0s[2], bx0s[3]) + col : 0;
            const int pv = act ? sel(by0s[0], by0s[1], by0s[2], by0s[3]) + row : 0;
            const int pix = pv * w + pu;
            const float ux = (float)pu + 0.5f, vy = (float)pv + 0.5f;
            const float rx = iK[0] * ux + iK[1] * vy + iK[2];
            const float ry = iK[4] * ux + iK[5] * vy + iK[6];
            const float rz = iK[8] * ux + iK[9] * vy + iK[10];
            const float Xw = depth * rx, Yw = depth * ry, Zw = depth * rz;
            const float qx = P[0] * Xw + P[1] * Yw + P[2] * Zw + P[3];
            const float qy = P[4] * Xw + P[5] * Yw + P[6] * Zw + P[7];
            const float qz = P[8] * Xw + P[9] * Yw + P[10] * Zw + P[11];
            const float zz = qz + 1e-8f;
            const float sc = (fabsf(qz) > 1e-8f) ? 1.0f / zz : 1.0f;
            const float uvx = __fsub_rn(__fmul_rn(__fmul_rn(2.0f, __fmul_rn(qx, sc)), inv_w), 1.0f);
            const float uvy = __fsub_rn(__fmul_rn(__fmul_rn(2.0f, __fmul_rn(qy, sc)), inv_h), 1.0f);
            const float ix = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(uvx, 1.0f), (float)w), 1.0f), 0.5f);
            const float iy = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(uvy, 1.0f), (float)h), 1.0f), 0.5f);
            const float fx0 = floorf(ix), fy0 = floorf(iy);
            ge.tx = ix - fx0; ge.ty = iy - fy0;
            // tap (ox, oy) = texel (fx0 + ox, fy0 + oy): inside the source image AND inside this tile
            const float lx = fx0 - (float)tx0, ly = fy0 - (float)ty0;
            const bool okx0 = lx >= 0.0f && lx <= (float)(tw_ - 1), okx1 = lx >= -1.0f && lx <= (float)(tw_ - 2);
            const bool oky0 = ly >= 0.0f && ly <= (float)(th_ - 1), oky1 = ly >= -1.0f && ly <= (float)(th_ - 2);
            ge.okm = (okx0 && oky0 ? 1u : 0u) | (okx1 && oky0 ? 2u : 0u) | (okx0 && oky1 ? 4u : 0u) | (okx1 && oky1 ? 8u : 0u);
            bool pend = act && zz > 0.0f && ge.okm != 0u;
            ge.t00 = pend ? (int)ly * TW + (int)lx : 0;       // tile-local index of tap (0, 0), >= -(TW + 1)
            ge.cid = pend ? ((int)ly + 1) * (TW + 1) + (int)lx + 1 : 0;   // the base's own slot (t00 wraps: (7, y) = (-1, y + 1))
            if (pend) {
#ifdef FS_SG_FLOOR   // timing-only build (WRONG results): every plane group reads the records of the FIRST one -- 4 planes per view stay
                     // cache-resident, i.e. the sweep with its record bytes taken out: what the projection + claim + LDS adds cost alone
                const size_t pl = (size_t)b * D + (size_t)(d0 + gs * gstride);
#else
                const size_t pl = (size_t)b * D + (size_t)(dg + gs * gstride);
#endif
                const float2 mt = recM[pl * hw + pix];
                const uint32_t fl = __float_as_uint(mt.y) >> kbit;
                pend = (fl & 2u) != 0u;                       // (the first pass's own z_k > 0)
                const bool pm = NAT && FS_REC_PIXEL_MAJOR;     // (the 16-pixel pass 1's records are pixel-major)
                const float4* sp = pm ? recS + (pl * hw + pix) * NV : recS + pl * NV * hw + pix;
#pragma unroll
                for (int s = 0; s < NV; ++s) S[s] = sp[pm ? (size_t)s : (size_t)s * hw];
                if (pend && !(fl & 1u)) {
                    // in front, not averaged (an exactly zero score): only d dot / cnt * cur reaches this source
                    const float4* c4 = (const float4*)(curT + ((size_t)b * hw + pix) * C);
#pragma unroll
                    for (int s = 0; s < NV; ++s) {
                        const float4 cv4 = c4[s];
                        S[s] = make_float4(mt.x * cv4.x, mt.x * cv4.y, mt.x * cv4.z, mt.x * cv4.w);
                    }
                }
            }
            ge.pend = pend;
#ifdef FS_CV_SG_STATS
            {
                const unsigned long long hit = __builtin_amdgcn_ballot_w64(pend);
                if (lane == 0) { FS_SG_COUNT(0, 1); FS_SG_COUNT(1, __builtin_popcountll(hit)); }
            }
#endif
        };
        // ---- FORM 1: records to the staging rows, taps to their texels' lists; then every texel lane blends its list ----
        struct Taps { int t00; float tx, ty; uint32_t todo; };
        auto place_taps = [&](Taps& tp) __attribute__((always_inline)) {
#pragma unroll
            for (int tap = 0; tap < 4; ++tap) {
                const int ox = tap & 1, oy = tap >> 1;
                if (tp.todo & (1u << tap)) {
                    const int tt = tp.t00 + oy * TW + ox;
                    const uint32_t slot = atomicAdd(&claim_[tt], 1u);
                    if (slot < (uint32_t)kSgCap) {
                        const float wt = (ox ? tp.tx : 1.0f - tp.tx) * (oy ? tp.ty : 1.0f - tp.ty);
                        lst_[tt * kSgCap + slot] = make_uint2((uint32_t)lane, __float_as_uint(wt));
                        tp.todo &= ~(1u << tap);
                    }
                }
            }
        };
        auto put_records = [&](const Geo& ge, const float4 (&S)[NV]) __attribute__((always_inline)) -> Taps {
            if (ge.pend) {
#pragma unroll
                for (int s = 0; s < NV; ++s) ((float4*)(acc + lane * ST))[s] = S[s];
            }
            Taps tp{ge.t00, ge.tx, ge.ty, ge.pend ? ge.okm : 0u};
            place_taps(tp);
            return tp;
        };
        auto blend_lists = [&](Taps tp) __attribute__((always_inline)) {
            for (;;) {
                wave_lds_sync();
                const int nl = (int)min(claim[lane], (uint32_t)kSgCap);
#pragma unroll 1
                for (int e = 0; e < kSgCap; ++e) {
                    if (__builtin_amdgcn_ballot_w64(e < nl) == 0ull) break;
                    if (e < nl) {
                        const uint2 en = lst_[lane * kSgCap + e];
                        const float wt = __uint_as_float(en.y);
                        const float4* sp = (const float4*)(acc + en.x * ST);
#pragma unroll
                        for (int s = 0; s < NV; ++s) {
                            const float4 v = sp[s];
                            accr[s].x = fmaf(wt, v.x, accr[s].x); accr[s].y = fmaf(wt, v.y, accr[s].y);
                            accr[s].z = fmaf(wt, v.z, accr[s].z); accr[s].w = fmaf(wt, v.w, accr[s].w);
                        }
                    }
                }
                claim[lane] = 0u;
                wave_lds_sync();
                if (__builtin_amdgcn_ballot_w64(tp.todo != 0u) == 0ull) break;
                place_taps(tp);          // (taps that found their texel's list full: the staged records are still there)
            }
        };
        // ---- FORM 0: the adds: claim rounds, then four taps of C/4 float4 read-modify-writes ----
        auto process = [&](const Geo& ge, const float4 (&S)[NV]) __attribute__((always_inline)) {
            bool pend = ge.pend;
            while (__builtin_amdgcn_ballot_w64(pend) != 0ull) {
#ifdef FS_CV_SG_STATS
                if (lane == 0) FS_SG_COUNT(6, 1);
#endif
                if (pend) claim[ge.cid] = (uint32_t)lane;
                wave_lds_sync();
                const bool win = pend && claim[ge.cid] == (uint32_t)lane;
                // Two winners never share a tap BASE, but tap 1 of base (0, 0) is tap 0 of base (1, 0): the four tap groups
                // must reach the LDS in program order (all lanes' tap t before any lane's tap t + 1).  wave_lds_sync() -- a
                // compiler-only fence, no instruction -- keeps the scheduler from hoisting a later group's ds_reads above an
                // earlier group's ds_writes, which are provably distinct addresses within ONE lane.
#pragma unroll
                for (int tap = 0; tap < 4; ++tap) {
                    const int ox = tap & 1, oy = tap >> 1;
                    if (win && (ge.okm & (1u << tap))) {
                        const float wt = (ox ? ge.tx : 1.0f - ge.tx) * (oy ? ge.ty : 1.0f - ge.ty);
                        float4* a = (float4*)(acc + (ge.t00 + oy * TW + ox) * ST);
#pragma unroll
                        for (int s = 0; s < NV; ++s) {
                            float4 v = a[s];
                            v.x = fmaf(wt, S[s].x, v.x); v.y = fmaf(wt, S[s].y, v.y);
                            v.z = fmaf(wt, S[s].z, v.z); v.w = fmaf(wt, S[s].w, v.w);
                            a[s] = v;
                        }
                    }
                    wave_lds_sync();
                }
                pend = pend && !win;
            }
        };
        if constexpr (FORM == 1) {
            // one record buffer: the next batch's loads are issued as soon as this batch's records sit in their staging rows, and
            // are in flight while the lists are blended (a second register buffer costs the third wavefront per SIMD: 3.0 vs 2.8 ms)
            Geo gA;
            float4 SA[NV];
            stage(0, gA, SA);
            for (int i0 = 0; i0 < N; i0 += 64) {
                const Taps tp = put_records(gA, SA);
                if (i0 + 64 < N) stage(i0 + 64, gA, SA);
                blend_lists(tp);
            }
            continue;
        } else {
        Geo gA, gB;
        float4 SA[NV], SB[NV];
        stage(0, gA, SA);
        for (int i0 = 0; i0 < N; i0 += 128) {
            const bool moreB = i0 + 64 < N;
            if (moreB) stage(i0 + 64, gB, SB);
            process(gA, SA);
            if (!moreB) break;
            if (i0 + 128 < N) stage(i0 + 128, gA, SA);
            process(gB, SB);
        }
        }
    }
    wave_lds_sync();
    // ---- the tile leaves once, in the caller's [C, h, w] layout (slot q = parity * C/2 + s  ->  channel 2 s + parity; NAT: q) ----
    {
        float* const dmap = d_src + (((size_t)b * K + k) * C) * hw;
        const int ty = lane / TW, tx = lane % TW;
        const bool inside = lane < NT && tx < tw_ && ty < th_;
        float* const dpx = dmap + (size_t)(ty0 + ty) * w + (tx0 + tx);
#pragma unroll
        for (int s = 0; s < NV; ++s) {
            const float4 v = FORM == 1 ? accr[s] : ((const float4*)(acc + (lane < NT ? lane : 0) * ST))[s];
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int q = 4 * s + e, ch = NAT ? q : 2 * (q % HC) + q / HC;
                if (inside) { if (chunks > 1) atomicAdd(dpx + (size_t)ch * hw, vv[e]); else dpx[(size_t)ch * hw] = vv[e]; }
            }
        }
    }
}

}  // namespace fs

#ifdef FS_CV_SG_STATS
extern "C" __attribute__((visibility("default"))) int fs_debug_cv_sg_stats(unsigned long long* dst, int reset)
{
    static unsigned long long z[8];
    if (reset) return (int)hipMemcpyToSymbol(HIP_SYMBOL(fs::g_sg_stats), z, sizeof(z));
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(fs::g_sg_stats), sizeof(z));
}
#endif
#ifdef FS_CV_TRACE
extern "C" __attribute__((visibility("default"))) int fs_debug_cv_trace(unsigned long long* dst, int reset)
{
    static unsigned long long z[fs::kCvTraceWaves * 4];
    if (reset) return (int)hipMemcpyToSymbol(HIP_SYMBOL(fs::g_cv_trace), z, sizeof(z));
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(fs::g_cv_trace), sizeof(z));
}
extern "C" __attribute__((visibility("default"))) int fs_debug_cvb_trace(unsigned long long* dst, int reset)
{
    static unsigned long long z[fs::kCvTraceWaves * 6];
    if (reset) return (int)hipMemcpyToSymbol(HIP_SYMBOL(fs::g_cvb_trace), z, sizeof(z));
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(fs::g_cvb_trace), sizeof(z));
}
#endif

using namespace fs;

// Number of plane slices (grid.y): enough wavefronts for ~6+ rounds over the chip's 1024 SIMDs x 2 slots,
// but at least 4 planes per wavefront so the per-wavefront weight loads stay amortised.
// which forward sweep: the projected one (first layer per source texel) pays for K = 1 only (see cost_volume_proj_kernel);
// FS_CV_PROJECTED=0/1 forces either (parity tests run both)
static bool cv_use_projected(int K)
{
    const char* e = getenv("FS_CV_PROJECTED");
    if (e && *e) return atoi(e) != 0;
    return K == 1;
}

static int cv_plane_split(int B, int groups, int D)
{
    static const int forced = getenv("FS_CV_SPLIT") ? atoi(getenv("FS_CV_SPLIT")) : 0;   // (tuning knob)
    if (forced > 0) return forced;
    int split = 1;
    while (split * 4 * 4 < D && (long long)B * groups * 4 * split < 12288) split *= 2;
    return split;
}

// Backward: two workgroups per CU, and every workgroup ends with 3 k weight atomics, so only as many plane slices as it
// takes to give each of the chip's 256 CUs a few rounds of workgroups (and at least 8 planes per wavefront).
static int cv_bwd_plane_split(int B, int groups, int D)
{
    int split = 1;
    while (split * 4 * 8 < D && (long long)B * groups * split < 2048) split *= 2;
    return split;
}

FS_API size_t fs_cost_volume_workspace_bytes(int32_t B, int32_t K, int32_t C, int32_t h, int32_t w)
{
    if (B < 0 || K < 0 || C <= 0 || h <= 0 || w <= 0) return 0;
    // current features [B, h*w, C] + source records [B*K, h*w, C + 32] (features + the projected first-layer block) + P
    return align_up(((size_t)B * C + (size_t)B * K * (C + 2 * kCvU)) * h * w * sizeof(float), 256) + align_up((size_t)B * K * 12 * 4, 256);
}

// layout of the training forward's `saved` buffer: a 256-byte header (word 0: some source had an exactly zero score with
// taps inside its image), then x = favg / cnt as [B*D][C/4 float4 chunks][h*w] (chunk q = channels 4 q .. 4 q + 3: natural order,
// what cost_volume16_bwd_kernel<C, true> reads), then (averaged score, validity bits) as [B*D][h*w] float2
static inline size_t cv_saved_xs_bytes(int B, int C, int h, int w, int D) { return align_up((size_t)B * D * h * w * C * sizeof(float), 256); }
FS_API size_t fs_cost_volume_saved_bytes(int32_t B, int32_t C, int32_t h, int32_t w, int32_t D)
{
    if (B <= 0 || C <= 0 || h <= 0 || w <= 0 || D <= 0) return 0;
    return 256 + cv_saved_xs_bytes(B, C, h, w, D) + align_up((size_t)B * D * h * w * 2 * sizeof(float), 256);
}

static int cv_forward_impl(int32_t B, int32_t K, int32_t C, int32_t h, int32_t w, int32_t D,
                           const float* cur_feats, const float* src_feats,
                           const float* src_extrinsics, const float* src_Ks,
                           const float* cur_invK, const float* planes, int64_t plane_stride_b,
                           int64_t plane_stride_d, int64_t plane_stride_pix, const float* w1,
                           const float* b1, const float* w2, const float* b2, const float* w3,
                           const float* b3, void* workspace, float* out, void* saved, int layout, void* stream_)
{
    if (B <= 0 || K <= 0 || h <= 0 || w <= 0 || D <= 0) return FS_ERR_INVALID_ARG;
    if (layout < 0 || layout > 3) return FS_ERR_INVALID_ARG;
    if (!cur_feats || !src_feats || !src_extrinsics || !src_Ks || !cur_invK || !planes || !w1 || !b1 ||
        !w2 || !b2 || !w3 || !b3 || !workspace || !out)
        return FS_ERR_INVALID_ARG;
    if (C != 48 && C != 16) return FS_ERR_UNSUPPORTED;  // matching_dim_size of FreeSplat (48) / SimpleRecon (16)
    // (the sweeps address a tap as a wave-uniform map base + a 32-bit per-lane byte offset)
    if ((unsigned long long)h * w * (C + 2 * kCvU) * sizeof(float) >= (1ull << 32)) return FS_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream_;
    const int hw = h * w;
    float* curT = (float*)workspace;
    float* srcT = curT + (size_t)B * hw * C;
    float* Pmat = (float*)((char*)workspace + align_up(((size_t)B * C + (size_t)B * K * (C + 2 * kCvU)) * hw * sizeof(float), 256));
    {
        ScopedStage prof_(kStCostVolume, st);
        const int groups = (hw + 31) / 32;
        const int slices = cv_plane_split(B, groups, D);
        const dim3 grid(cv_grid(B, groups, slices));
        if (!saved && layout == 0 && cv_use_projected(K)) {
            // K = 1: first layer's feature block applied per source texel, 16 MFMAs per (group, plane); two launches
            // (the sweep reads the current view from the caller's map and forms its projection rows itself)
            const unsigned gproj = (unsigned)std::min<long long>(((long long)B * K * hw * 8 + 255) / 256, 65536);
            if (C == 48) {
                hipLaunchKernelGGL(cv_relayout_project_kernel<48>, dim3(gproj), dim3(256), 0, st, src_feats, srcT, w1, h, w, B * K);
                hipLaunchKernelGGL(cost_volume_proj_kernel<24>, grid, dim3(256), 0, st, B, K, h, w, D, slices, cur_feats, srcT, src_Ks,
                                   src_extrinsics, cur_invK, planes, (long long)plane_stride_b, (long long)plane_stride_d,
                                   (long long)plane_stride_pix, w1, b1, w2, b2, w3, b3, out);
            } else {
                hipLaunchKernelGGL(cv_relayout_project_kernel<16>, dim3(gproj), dim3(256), 0, st, src_feats, srcT, w1, h, w, B * K);
                hipLaunchKernelGGL(cost_volume_proj_kernel<8>, grid, dim3(256), 0, st, B, K, h, w, D, slices, cur_feats, srcT, src_Ks,
                                   src_extrinsics, cur_invK, planes, (long long)plane_stride_b, (long long)plane_stride_d,
                                   (long long)plane_stride_pix, w1, b1, w2, b2, w3, b3, out);
            }
        } else {
            hipLaunchKernelGGL(cv_proj_kernel, dim3((B * K * 12 + 255) / 256), dim3(256), 0, st, B * K, src_Ks, src_extrinsics,
                               Pmat);
            const int groups16 = (hw + 15) / 16;
            const int slices16 = cv_plane_split(B, groups16, D);
            const dim3 grid16(cv_grid(B, groups16, slices16));
            // (layout bit 0 / 1: the caller's current / source maps already ARE pixel-major [h*w][C] records -- channels_last
            //  tensors --, which is what the 16-pixel sweep gathers from: no re-layout pass, the sweep reads them in place)
            const float* curN = cur_feats;
            const float* srcN = src_feats;
            if (!(layout & 1)) { cv_relayout(false, false, cur_feats, curT, C, hw, B, st); curN = curT; }
            if (!(layout & 2)) { cv_relayout(false, false, src_feats, srcT, C, hw, B * K, st); srcN = srcT; }
            // (training: the general sweep for every K -- it forms the averaged features the backward wants to keep; the
            //  K = 1 projected sweep never does)
            uint32_t* xhdr = (uint32_t*)saved;
            float* xs = saved ? (float*)((char*)saved + 256) : nullptr;
            float2* xm = saved ? (float2*)((char*)saved + 256 + cv_saved_xs_bytes(B, C, h, w, D)) : nullptr;
            if (saved && hipMemsetAsync(saved, 0, 256, st) != hipSuccess) {
                set_last_error("cost volume saved header", hipGetLastError());
                return FS_ERR_LAUNCH;
            }
            auto sweep16 = [&](auto kernel) {
                hipLaunchKernelGGL(kernel, grid16, dim3(256), 0, st, B, K, h, w, D, slices16, curN, srcN, Pmat,
                                   cur_invK, planes, (long long)plane_stride_b, (long long)plane_stride_d,
                                   (long long)plane_stride_pix, w1, b1, w2, b2, w3, b3, out, xs, xm, xhdr);
            };
            // (K <= 2: tap by tap at four wavefronts per SIMD; K >= 3: a source's four taps in flight at three -- FS_CV_FWD_SHAPE=1 / 4 forces one)
            static const int force_shape = [] { const char* e = getenv("FS_CV_FWD_SHAPE"); return e ? atoi(e) : 0; }();
            const bool deep = force_shape == 4 || (force_shape != 1 && K >= 3);
            if (C == 48) {
                if (deep) { if (saved) sweep16(cost_volume16_kernel<48, true, 4, 3>); else sweep16(cost_volume16_kernel<48, false, 4, 3>); }
                else { if (saved) sweep16(cost_volume16_kernel<48, true, 1, 4>); else sweep16(cost_volume16_kernel<48, false, 1, 4>); }
            } else {
                if (deep) { if (saved) sweep16(cost_volume16_kernel<16, true, 4, 3>); else sweep16(cost_volume16_kernel<16, false, 4, 3>); }
                else { if (saved) sweep16(cost_volume16_kernel<16, true, 1, 4>); else sweep16(cost_volume16_kernel<16, false, 1, 4>); }
            }
        }
    }
    FS_CHECK_LAUNCH("cost_volume");
    return FS_OK;
}

FS_API int fs_cost_volume_forward(int32_t B, int32_t K, int32_t C, int32_t h, int32_t w, int32_t D,
                                  const float* cur_feats, const float* src_feats,
                                  const float* src_extrinsics, const float* src_Ks,
                                  const float* cur_invK, const float* planes, int64_t plane_stride_b,
                                  int64_t plane_stride_d, int64_t plane_stride_pix, const float* w1,
                                  const float* b1, const float* w2, const float* b2, const float* w3,
                                  const float* b3, void* workspace, float* out, void* stream_)
{
    return cv_forward_impl(B, K, C, h, w, D, cur_feats, src_feats, src_extrinsics, src_Ks, cur_invK, planes, plane_stride_b,
                           plane_stride_d, plane_stride_pix, w1, b1, w2, b2, w3, b3, workspace, out, nullptr, 0, stream_);
}

FS_API int fs_cost_volume_forward_layout(int32_t B, int32_t K, int32_t C, int32_t h, int32_t w, int32_t D,
                                         const float* cur_feats, const float* src_feats,
                                         const float* src_extrinsics, const float* src_Ks,
                                         const float* cur_invK, const float* planes, int64_t plane_stride_b,
                                         int64_t plane_stride_d, int64_t plane_stride_pix, const float* w1,
                                         const float* b1, const float* w2, const float* b2, const float* w3,
                                         const float* b3, void* workspace, float* out, int32_t layout, void* stream_)
{
    return cv_forward_impl(B, K, C, h, w, D, cur_feats, src_feats, src_extrinsics, src_Ks, cur_invK, planes, plane_stride_b,
                           plane_stride_d, plane_stride_pix, w1, b1, w2, b2, w3, b3, workspace, out, nullptr, layout, stream_);
}

FS_API int fs_cost_volume_forward_train(int32_t B, int32_t K, int32_t C, int32_t h, int32_t w, int32_t D,
                                        const float* cur_feats, const float* src_feats,
                                        const float* src_extrinsics, const float* src_Ks,
                                        const float* cur_invK, const float* planes, int64_t plane_stride_b,
                                        int64_t plane_stride_d, int64_t plane_stride_pix, const float* w1,
                                        const float* b1, const float* w2, const float* b2, const float* w3,
                                        const float* b3, void* workspace, float* out, void* saved, void* stream_)
{
    if (!saved) return FS_ERR_INVALID_ARG;
    return cv_forward_impl(B, K, C, h, w, D, cur_feats, src_feats, src_extrinsics, src_Ks, cur_invK, planes, plane_stride_b,
                           plane_stride_d, plane_stride_pix, w1, b1, w2, b2, w3, b3, workspace, out, saved, 0, stream_);
}

// The two-pass backward (records + source-tile sweep) needs plane depths that do not vary per pixel (a plane-induced
// homography) and the sources' flag bits in one word; FS_CV_BWD_ATOMIC=1 forces the one-kernel scatter form (A/B runs)
static bool cv_bwd_two_pass(int K, int64_t plane_stride_pix)
{
    const char* e = getenv("FS_CV_BWD_ATOMIC");
    if (e && *e && atoi(e) != 0) return false;
    return plane_stride_pix == 0 && K <= 16;
}

static size_t cv_bwd_workspace_bytes(int32_t B, int32_t K, int32_t C, int32_t h, int32_t w, int32_t D, bool two_pass)
{
    if (B <= 0 || K <= 0 || C <= 0 || h <= 0 || w <= 0 || D <= 0) return 0;
    const size_t hw = (size_t)h * w;
    // pixel-major copies curT, srcT and their gradients d_curT, d_srcT; the projection rows; and for the two-pass form
    // (constant planes, K <= 16) one record of C + 2 floats per (view, plane, pixel) and the inverse plane homographies
    size_t n = align_up((size_t)B * (1 + (size_t)K) * C * hw * 2 * sizeof(float), 256) + align_up((size_t)B * K * 12 * 4, 256);
    if (two_pass)
        n += align_up((size_t)B * D * hw * C * sizeof(float), 256) + align_up((size_t)B * D * hw * 2 * sizeof(float), 256) +
             align_up((size_t)B * K * D * 9 * sizeof(float), 256);
    return n;
}
// Upper bound for any call of these dimensions (the two-pass form's records included: B * D * h * w * (C + 2) floats, ~3 GB for
// 10 views at 96 x 128 with D = 128) ...
FS_API size_t fs_cost_volume_backward_workspace_bytes(int32_t B, int32_t K, int32_t C, int32_t h, int32_t w,
                                                      int32_t D)
{
    return cv_bwd_workspace_bytes(B, K, C, h, w, D, true);
}
// ... and what THIS call needs: without the record and inverse-homography regions when the one-kernel scatter form will run
// (per-pixel planes, K > 16, FS_CV_BWD_ATOMIC=1), which never touches them.
FS_API size_t fs_cost_volume_backward_workspace_bytes_for(int32_t B, int32_t K, int32_t C, int32_t h, int32_t w,
                                                          int32_t D, int64_t plane_stride_pix)
{
    return cv_bwd_workspace_bytes(B, K, C, h, w, D, cv_bwd_two_pass(K, plane_stride_pix));
}

static int cv_backward_impl(int32_t B, int32_t K, int32_t C, int32_t h, int32_t w, int32_t D,
                            const float* cur_feats, const float* src_feats,
                            const float* src_extrinsics, const float* src_Ks,
                            const float* cur_invK, const float* planes, int64_t plane_stride_b,
                            int64_t plane_stride_d, int64_t plane_stride_pix, const float* w1,
                            const float* b1, const float* w2, const float* b2, const float* w3,
                            const float* grad_out, void* workspace, float* d_cur_feats,
                            float* d_src_feats, float* d_w1, float* d_b1, float* d_w2, float* d_b2,
                            float* d_w3, float* d_b3, const void* saved, void* stream_)
{
    if (B <= 0 || K <= 0 || h <= 0 || w <= 0 || D <= 0) return FS_ERR_INVALID_ARG;
    if (!cur_feats || !src_feats || !src_extrinsics || !src_Ks || !cur_invK || !planes || !w1 || !b1 || !w2 ||
        !b2 || !w3 || !grad_out || !workspace || !d_cur_feats || !d_src_feats || !d_w1 || !d_b1 || !d_w2 || !d_b2 ||
        !d_w3 || !d_b3)
        return FS_ERR_INVALID_ARG;
    if (C != 48 && C != 16) return FS_ERR_UNSUPPORTED;
    if ((unsigned long long)h * w * (C + 2 * kCvU) * sizeof(float) >= (1ull << 32)) return FS_ERR_UNSUPPORTED;   // (32-bit texel offsets)
    hipStream_t st = (hipStream_t)stream_;
    const int hw = h * w;
    const size_t n_cur = (size_t)B * hw * C, n_src = n_cur * K;
    float* curT = (float*)workspace;
    float* srcT = curT + n_cur;
    float* d_curT = srcT + n_src;
    float* d_srcT = d_curT + n_cur;
    float* Pmat = (float*)((char*)workspace + align_up((n_cur + n_src) * 2 * sizeof(float), 256));
    float4* recS = (float4*)((char*)Pmat + align_up((size_t)B * K * 12 * 4, 256));
    float2* recM = (float2*)((char*)recS + align_up((size_t)B * D * hw * C * sizeof(float), 256));
    float* Ginv = (float*)((char*)recM + align_up((size_t)B * D * hw * 2 * sizeof(float), 256));
    const bool two_pass = cv_bwd_two_pass(K, plane_stride_pix);
    if (!two_pass) saved = nullptr;   // (the one-kernel scatter form needs every source's taps: it recomputes the forward)
    const int tiles_x = (w + kSgTW - 1) / kSgTW, tiles_y = (h + kSgTH - 1) / kSgTH, tiles = tiles_x * tiles_y;
    // plane chunks of the source-tile sweep: one (plain stores) when there are enough tiles to fill the chip (256 CUs x
    // 11 single-wavefront workgroups), else enough chunks for one full round (their tiles then leave through atomics
    // into a zeroed map)
    int chunks = 1;
    while ((long long)B * K * tiles * chunks < 2816 && D / (chunks * 2) >= kSgG) chunks *= 2;
    if (const char* e = getenv("FS_CV_SG_CHUNKS")) {   // (tests: force the plain-store form, chunks = 1, at small sizes, or any split)
        const int f = atoi(e);
        if (f >= 1 && f <= D) chunks = f;
    }
    ScopedStage prof_(kStCostVolume, st);
    {
        const int n_thr = std::max(std::max(B * K * 12, 32 * (C + 1)), std::max(32 * 32, two_pass ? B * K * D : 0));
        hipLaunchKernelGGL(cv_bwd_prep_kernel, dim3((n_thr + 255) / 256), dim3(256), 0, st, B * K, K, D, C, two_pass ? 1 : 0, src_Ks,
                           src_extrinsics, cur_invK, planes, (long long)plane_stride_b, (long long)plane_stride_d, Pmat, Ginv, d_w1, d_b1,
                           d_w2, d_b2, d_w3, d_b3);
    }
    if ((two_pass && chunks > 1 && hipMemsetAsync(d_src_feats, 0, n_src * sizeof(float), st) != hipSuccess) ||
        hipMemsetAsync(d_curT, 0, (two_pass ? n_cur : n_cur + n_src) * sizeof(float), st) != hipSuccess) {
        set_last_error("cost volume backward memset", hipGetLastError());
        return FS_ERR_LAUNCH;
    }
    // pass 1 of the two-pass form: the 16-pixel kernel on natural-order maps (round 6), FS_CV_BWD16=0 or a saved-activation
    // call: the 32-pixel kernel on [parity][C/2] records
    bool bwd16 = two_pass;
    if (const char* e = getenv("FS_CV_BWD16")) bwd16 = bwd16 && atoi(e) != 0;
    if (!bwd16) saved = nullptr;   // (the saved activations are in the 16-pixel kernel's chunk order: the other forms recompute)
    const uint32_t* xhdr = (const uint32_t*)saved;
    const float4* xs = saved ? (const float4*)((const char*)saved + 256) : nullptr;
    const float2* xm = saved ? (const float2*)((const char*)saved + 256 + cv_saved_xs_bytes(B, C, h, w, D)) : nullptr;
    cv_relayout(!bwd16, false, cur_feats, curT, C, hw, B, st);
    cv_relayout(!bwd16, false, src_feats, srcT, C, hw, B * K, st);
    const int groups = (hw + 31) / 32;
    const int bslices = cv_bwd_plane_split(B, groups, D);
    auto sweep = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(cv_grid(B, groups, bslices)), dim3(256), 0, st, B, K, h, w, D, bslices, curT, srcT,
                           Pmat, cur_invK, planes, (long long)plane_stride_b,
                           (long long)plane_stride_d, (long long)plane_stride_pix, w1, b1, w2, b2, w3, grad_out,
                           d_curT, d_srcT, d_w1, d_b1, d_w2, d_b2, d_w3, d_b3, recS, recM);
    };
    // pass 2's form: 1 = register accumulators fed through per-texel lists (round 6), 0 = LDS read-modify-writes behind claim rounds
    static const int sg_form = [] { const char* e = getenv("FS_CV_SG_FORM"); return e ? atoi(e) : 1; }() && kSgTW * kSgTH == 64 ? 1 : 0;
    auto tile_sweep = [&](auto kernel) {
        const unsigned grid = 8u * (unsigned)B * (unsigned)((tiles + 7) >> 3) * (unsigned)K * (unsigned)chunks;
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(64), 0, st, B, K, h, w, D, chunks, tiles_x, tiles_y, curT,
                           (const float4*)recS, (const float2*)recM, Pmat, Ginv, cur_invK, planes, (long long)plane_stride_b,
                           (long long)plane_stride_d, d_src_feats);
    };
    auto sweep16 = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(cv_grid(B, groups, bslices)), dim3(256), 0, st, B, K, h, w, D, bslices, curT, srcT,
                           Pmat, cur_invK, planes, (long long)plane_stride_b, (long long)plane_stride_d, w1, b1, w2, b2, w3,
                           grad_out, d_curT, d_w1, d_b1, d_w2, d_b2, d_w3, d_b3, recS, recM, xs, xm, xhdr);
    };
    if (bwd16) {
        if (C == 48) {
            if (saved) sweep16(cost_volume16_bwd_kernel<48, true>); else sweep16(cost_volume16_bwd_kernel<48, false>);
            if constexpr (kSgTW * kSgTH == 64) { if (sg_form) tile_sweep(cv_src_grad_kernel<48, true, 1>); else tile_sweep(cv_src_grad_kernel<48, true, 0>); }
            else tile_sweep(cv_src_grad_kernel<48, true, 0>);
        } else {
            if (saved) sweep16(cost_volume16_bwd_kernel<16, true>); else sweep16(cost_volume16_bwd_kernel<16, false>);
            if constexpr (kSgTW * kSgTH == 64) { if (sg_form) tile_sweep(cv_src_grad_kernel<16, true, 1>); else tile_sweep(cv_src_grad_kernel<16, true, 0>); }
            else tile_sweep(cv_src_grad_kernel<16, true, 0>);
        }
    } else if (two_pass) {
        if (C == 48) {
            sweep(cost_volume_bwd_kernel<24, true>);
            tile_sweep(cv_src_grad_kernel<48, false, 0>);
        } else {
            sweep(cost_volume_bwd_kernel<8, true>);
            tile_sweep(cv_src_grad_kernel<16, false, 0>);
        }
    } else {
        if (C == 48) sweep(cost_volume_bwd_kernel<24, false>); else sweep(cost_volume_bwd_kernel<8, false>);
        cv_relayout(true, true, d_srcT, d_src_feats, C, hw, B * K, st);
    }
    cv_relayout(!bwd16, true, d_curT, d_cur_feats, C, hw, B, st);
    FS_CHECK_LAUNCH("cost_volume_backward");
    return FS_OK;
}

FS_API int fs_cost_volume_backward(int32_t B, int32_t K, int32_t C, int32_t h, int32_t w, int32_t D,
                                   const float* cur_feats, const float* src_feats,
                                   const float* src_extrinsics, const float* src_Ks,
                                   const float* cur_invK, const float* planes, int64_t plane_stride_b,
                                   int64_t plane_stride_d, int64_t plane_stride_pix, const float* w1,
                                   const float* b1, const float* w2, const float* b2, const float* w3,
                                   const float* grad_out, void* workspace, float* d_cur_feats,
                                   float* d_src_feats, float* d_w1, float* d_b1, float* d_w2, float* d_b2,
                                   float* d_w3, float* d_b3, void* stream_)
{
    return cv_backward_impl(B, K, C, h, w, D, cur_feats, src_feats, src_extrinsics, src_Ks, cur_invK, planes, plane_stride_b,
                            plane_stride_d, plane_stride_pix, w1, b1, w2, b2, w3, grad_out, workspace, d_cur_feats, d_src_feats,
                            d_w1, d_b1, d_w2, d_b2, d_w3, d_b3, nullptr, stream_);
}

FS_API int fs_cost_volume_backward_train(int32_t B, int32_t K, int32_t C, int32_t h, int32_t w, int32_t D,
                                         const float* cur_feats, const float* src_feats,
                                         const float* src_extrinsics, const float* src_Ks,
                                         const float* cur_invK, const float* planes, int64_t plane_stride_b,
                                         int64_t plane_stride_d, int64_t plane_stride_pix, const float* w1,
                                         const float* b1, const float* w2, const float* b2, const float* w3,
                                         const float* grad_out, void* workspace, const void* saved, float* d_cur_feats,
                                         float* d_src_feats, float* d_w1, float* d_b1, float* d_w2, float* d_b2,
                                         float* d_w3, float* d_b3, void* stream_)
{
    if (!saved) return FS_ERR_INVALID_ARG;
    return cv_backward_impl(B, K, C, h, w, D, cur_feats, src_feats, src_extrinsics, src_Ks, cur_invK, planes, plane_stride_b,
                            plane_stride_d, plane_stride_pix, w1, b1, w2, b2, w3, grad_out, workspace, d_cur_feats, d_src_feats,
                            d_w1, d_b1, d_w2, d_b2, d_w3, d_b3, saved, stream_);
}

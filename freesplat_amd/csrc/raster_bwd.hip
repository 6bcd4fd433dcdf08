// raster_bwd.hip -- backward of the tile rasterizer for MI355X (gfx950).
//
// Replaces the autograd backward of the CUDA extension FreeSplat uses
// (src/model/decoder/cuda_splatting.py:114-127; semantics: SURVEY.md Appendix A.5).
//
//   render_bwd      1 single-wavefront workgroup per 8x8 quadrant, back to front over the same sorted list,
//                   survivors compacted in pairs (packed fp32); the per-pixel partials of a pair are summed over
//                   the wavefront in registers (row-swap halving + in-row DPP adds) and leave as ONE global
//                   atomic instruction per pair -- one atomic per (quadrant, Gaussian, component) instead of one
//                   per (pixel, Gaussian, component).
//   preprocess_bwd  1 thread / Gaussian: conic -> cov2D -> Sigma & view position, mean2D ->
//                   mean through the perspective divide, RGB -> SH & view direction.
//                   SH gradients leave through LDS so the [N, M, 3] rows are written coalesced.
#include <cstdlib>

#include "fs_common.h"

namespace fs {

constexpr int kGradStride = 12;  // floats per Gaussian in the accumulation scratch
// layout: 0,1 moments S_x, S_y | 2,3,4 moments S_xx, S_xy, S_yy (render_bwd; preprocess_bwd turns them into the mean2D and
// conic gradients) | 5 opacity | 6,7,8 rgb | 9 view z | 10,11 pad

// Wavefront sums on the gfx950 row-swap instructions: v_permlane32_swap(X, Y) leaves X+Y = (value X summed over lanes
// l, l+32) in the lower half and (value Y ...) in the upper half -- one swap + one add per surviving value, a HALVING
// level -- and v_permlane16_swap does the same between the two 16-lane rows of each half.  What is left is summed
// inside a row with 4 adds that have a DPP row rotation folded in (row_ror 8, 4, 2, 1).
__device__ __forceinline__ float swap_add32(float x, float y)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap_add16(float x, float y)
{
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <int CTRL>
__device__ __forceinline__ float row_add(float v)
{
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row_sum(float v)
{
    v = row_add<0x128>(v);  // row_ror:8
    v = row_add<0x124>(v);  // row_ror:4
    v = row_add<0x122>(v);  // row_ror:2
    return row_add<0x121>(v);  // row_ror:1
}
// Sum 20 per-lane values -- 10 of survivor a, 10 of survivor b -- over the wavefront in 45 instructions: the first
// halving level puts a's sums into the lower half and b's into the upper half (swap_add32), the second leaves
// values 0..4 in the even row and 5..9 in the odd row of each half (swap_add16: five registers, each lane of row r holding
// a partial of value 5*(r&1)+k of survivor r>>1 in register k); inside the rows the five registers are merged PAIRWISE on
// the way (round 3): a rotate-by-2^b level sends lane i the value of lane i -+ 2^b, which differs from i in bit b and in
// no LOWER bit -- so going from the smallest rotation to the largest, the lanes with bit b clear keep register X and the
// lanes with bit b set register Y (own = bit ? Y : X, and what is rotated in is the other choice, bit ? X : Y: each lane
// receives its own value's partial from a lane that kept the other one), and every earlier choice (bits < b) agrees
// between the two lanes.  Rotate by 1, 2, 4: 5 -> 3 -> 2 -> 1 registers; rotate by 8: a plain add.  15 instructions instead
// of 4 rotate-adds on each of 5 registers, and the result is ONE register: lane (row r, column c < 5) holds value
// 5*(r&1) + c of survivor r>>1 (column bits 0, 1 pick among values 0..3, bit 2 picks value 4).
template <int CTRL>
__device__ __forceinline__ float dpp_row(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ float row_merge(float x, float y, unsigned long long mask)
{
    const bool hi = __builtin_amdgcn_inverse_ballot_w64(mask);     // (wave-uniform constant: a v_cndmask on an SGPR pair)
    return (hi ? y : x) + dpp_row<CTRL>(hi ? x : y);
}
template <int NV>   // values per survivor that can be non-zero (the tenth, the depth partial, only when depth gradients flow)
__device__ __forceinline__ float wave_sum_pair(const float (&va)[10], const float (&vb)[10])
{
    float h[10], t[5];
#pragma unroll
    for (int k = 0; k < NV; ++k) h[k] = swap_add32(va[k], vb[k]);
#pragma unroll
    for (int k = NV; k < 10; ++k) h[k] = 0.0f;
#pragma unroll
    for (int k = 0; k < 5; ++k) t[k] = swap_add16(h[k], h[5 + k]);
    constexpr unsigned long long kBit0 = 0xAAAAAAAAAAAAAAAAull, kBit1 = 0xCCCCCCCCCCCCCCCCull, kBit2 = 0xF0F0F0F0F0F0F0F0ull;
    const float u0 = row_merge<0x121>(t[0], t[1], kBit0), u1 = row_merge<0x121>(t[2], t[3], kBit0), u2 = row_add<0x121>(t[4]);   // row_ror:1
    const float w0 = row_merge<0x122>(u0, u1, kBit1), w1 = row_add<0x122>(u2);                                                    // row_ror:2
    return row_add<0x128>(row_merge<0x124>(w0, w1, kBit2));                                                                       // row_ror:4, then 8
}

constexpr int kBwdQuads = 7;  // float4 per slot of two survivors
// Backward blend loop, same organisation as the forward (raster_fwd.hip:blend_quadrant): one single-wavefront
// workgroup per 8x8 quadrant, no barriers; the tile list is walked BACK TO FRONT 64 entries at a time with the
// records of the next batch in flight; the survivors of a batch are compacted (highest list position first) two per
// LDS slot with interleaved fields, so exponent, exp, alpha and all per-Gaussian derivative terms of both run on
// packed fp32, and only the transmittance / behind-colour recurrences stay sequential.  The 2 x 10 per-pixel
// partials are summed over the wavefront by wave_sum_pair and leave as one global atomic instruction per pair
// (20 lanes, one per component): nothing is staged or re-read.
//
// Recurrences, for a pixel with final transmittance Tf, walking j = last .. first:
//   T_j   = T_{j+1} / (1 - a_j)                       transmittance in front of j     (T_{last+1} := Tf)
//   R_j   = colour composited behind j;  R_{j-1} = R_j + a_j (c_j - R_j)              (R_last := 0)
//   dL/da_j = T_j * sum_ch (c_j - R_j)_ch * dL/dC_ch  -  Tf / (1 - a_j) * (bg . dL/dC)
// A survivor that does not touch the pixel enters with a_j = 0 and G_j = 0, which makes every update the identity
// and every partial zero -- no per-value masking.
// DEPTH = false (dL_ddepth == NULL -- FreeSplat's losses never back-propagate the depth map, src/loss/loss_mse.py:32): the
// fourth channel of the behind-colour recurrence, of the score s and of the colour partials does not exist; what is left of
// the (blue, depth) pair runs as scalar operations (packed fp32 has no throughput advantage on this chip, fs_common.h), and
// the wavefront reduction carries 9 instead of 10 values per survivor.
template <bool FAST_EXP, bool DEPTH>
#ifdef FS_BWD_WAVES         // (A/B builds: make VARIANT=b8 EXTRA=-DFS_BWD_WAVES=8 forces <= 64 registers)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(FS_BWD_WAVES, FS_BWD_WAVES))) void render_bwd_kernel(
#else
__global__ __launch_bounds__(64) void render_bwd_kernel(
#endif
    int H, int W, int T, const uint32_t* __restrict__ offsets,
    const uint32_t* __restrict__ point_list, const float4* __restrict__ rec,
    const float* __restrict__ bg, const uint32_t* __restrict__ counters, const float* __restrict__ final_T,
    const int32_t* __restrict__ n_contrib, const float* __restrict__ dL_dcolor,
    const float* __restrict__ dL_ddepth, float* __restrict__ grad)
{
    __shared__ float4 s_pair[33 * kBwdQuads];   // up to 1 carried + 64 new survivors, two per slot
    if (counters && counters[1]) return;  // the forward overflowed its capacity: no image, no lists -> zero gradients
    const int gx = (W + kTile - 1) / kTile, gy = (H + kTile - 1) / kTile;
    const int wave = (blockIdx.x >> 3) & 3;  // quadrant; same workgroup order as the forward
    const int tile = tile_for_block((int)(blockIdx.x >> 5) * 8 + (int)(blockIdx.x & 7), gx, gy);
    if (tile < 0) return;
    const int tx = tile % gx, ty = tile / gx;
    const int lane = threadIdx.x;
    const int px = tx * kTile + (wave & 1) * 8 + (lane & 7);
    const int py = ty * kTile + (wave >> 1) * 8 + (lane >> 3);
    const uint32_t qbit = 1u << wave;
    const bool inside = px < W && py < H;
    const f32x2 pfx = splat2((float)px), pfy = splat2((float)py);
    const uint32_t a = offsets[tile];
    const int n = (int)(offsets[tile + 1] - a);
    if (n == 0) return;
    const uint32_t* const pl = point_list + a;

    const size_t pix = (size_t)py * W + px, HW = (size_t)H * W;
    const float Tf = inside ? final_T[pix] : 0.0f;
    const int last = inside ? n_contrib[pix] : 0;
    const f32x2 g01 = {inside ? dL_dcolor[pix] : 0.0f, inside ? dL_dcolor[HW + pix] : 0.0f};
    const f32x2 g23 = {inside ? dL_dcolor[2 * HW + pix] : 0.0f, (DEPTH && inside && dL_ddepth) ? dL_ddepth[pix] : 0.0f};
    const float Tb = Tf * (bg[0] * g01.x + bg[1] * g01.y + bg[2] * g23.x);

    int maxlast = last;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) maxlast = max(maxlast, __shfl_xor(maxlast, m, 64));
    if (maxlast == 0) return;
    const int m_end = min(n, maxlast);  // entries at or beyond this position contributed to no pixel of the quadrant

    float T_ = Tf;
    f32x2 R01 = splat2(0.0f), R23 = splat2(0.0f);
    const int row = lane >> 4, col = lane & 15;
    float4* const cp = s_pair;

    // One pair of survivors (a further back, b in front of it): recompute, recurrences, partials, wavefront sums, flush.
    auto do_pair = [&](const float4* q) __attribute__((always_inline)) {
        const float4 c0 = q[0], c1 = q[1], c2 = q[2], c3 = q[3];
        const f32x2 dx = (f32x2){c0.x, c0.y} - pfx, dy = (f32x2){c0.z, c0.w} - pfy;
        const f32x2 pw = fma2((f32x2){c1.x, c1.y} * dx, dx, dy * fma2((f32x2){c1.z, c1.w}, dy, (f32x2){c2.x, c2.y} * dx));
        // pw = -power; +0 <= pw <= -threshold as ONE unsigned compare of the bit patterns
        bool on_a = __float_as_int(c3.z) <= last, on_b = __float_as_int(c3.w) <= last;
        on_a = on_a & (__float_as_uint(pw.x) <= __float_as_uint(c2.z));
        on_b = on_b & (__float_as_uint(pw.y) <= __float_as_uint(c2.w));
        if (__builtin_amdgcn_ballot_w64(on_a | on_b) == 0) return;  // wave-uniform
        f32x2 Gr = blend_exp_of_neg<FAST_EXP>(pw);
        f32x2 oe = (f32x2){c3.x, c3.y} * Gr;
        f32x2 al_raw = {fminf(0.99f, oe.x), fminf(0.99f, oe.y)};
        if constexpr (FAST_EXP) {
            // an alpha inside the guard band of the 1/255 threshold (fs_common.h): the pair is re-evaluated with the
            // contract exp, so the accept / reject decisions are those of the exact mode (and of the forward)
            const float glo = alpha_guard_lo(), ghi = alpha_guard_hi();
            const bool nb = (on_a & (al_raw.x >= glo) & !(al_raw.x >= ghi)) | (on_b & (al_raw.y >= glo) & !(al_raw.y >= ghi));
            if (__builtin_amdgcn_ballot_w64(nb) != 0) {  // wave-uniform, rare
                Gr = blend_exp_of_neg<false>(pw);
                oe = (f32x2){c3.x, c3.y} * Gr;
                al_raw = (f32x2){fminf(0.99f, oe.x), fminf(0.99f, oe.y)};
            }
        }
        on_a = on_a & (al_raw.x >= 1.0f / 255.0f);
        on_b = on_b & (al_raw.y >= 1.0f / 255.0f);
        if (__builtin_amdgcn_ballot_w64(on_a | on_b) == 0) return;
        const float4 ka = q[4], kb = q[5];
        const f32x2 al = {on_a ? al_raw.x : 0.0f, on_b ? al_raw.y : 0.0f};
        const f32x2 G = {on_a ? Gr.x : 0.0f, on_b ? Gr.y : 0.0f};
        const f32x2 om = splat2(1.0f) - al;
        // 1 - alpha >= 0.01: one reciprocal (1 ulp) serves both divisions of the reference formula
        const f32x2 rinv = {__builtin_amdgcn_rcpf(om.x), __builtin_amdgcn_rcpf(om.y)};
        float va[10], vb[10];
        f32x2 dLa;
        auto one = [&](const float4& kc, float al1, float rinv1, float& dL1, float (&vv)[10]) __attribute__((always_inline)) {
            T_ = T_ * rinv1;
            const f32x2 d01 = (f32x2){kc.x, kc.y} - R01;
            const float w1 = al1 * T_;
            const f32x2 c01 = splat2(w1) * g01;
            if constexpr (DEPTH) {
                const f32x2 d23 = (f32x2){kc.z, kc.w} - R23;
                const f32x2 s = fma2(d23, g23, d01 * g01);
                R23 = fma2(splat2(al1), d23, R23);
                dL1 = fmaf(s.x + s.y, T_, -(Tb * rinv1));
                const f32x2 c23 = splat2(w1) * g23;
                vv[8] = c23.x; vv[9] = c23.y;
            } else {
                const float d2 = kc.z - R23.x;
                const f32x2 s = d01 * g01;
                R23.x = fmaf(al1, d2, R23.x);
                dL1 = fmaf(fmaf(d2, g23.x, s.x + s.y), T_, -(Tb * rinv1));
                vv[8] = w1 * g23.x; vv[9] = 0.0f;
            }
            R01 = fma2(splat2(al1), d01, R01);
            vv[6] = c01.x; vv[7] = c01.y;
        };
        float dLa_a, dLa_b;
        one(ka, al.x, rinv.x, dLa_a, va);   // survivor a (the one further back)
        one(kb, al.y, rinv.y, dLa_b, vb);   // survivor b (in front of a)
        dLa = (f32x2){dLa_a, dLa_b};
        // per-Gaussian derivative terms of both survivors, packed [a, b].  The five geometric partials leave as MOMENT
        // sums of w = dL/dG * G over the pixels -- S_x = sum w dx, S_y = sum w dy, S_xx = sum w dx^2, S_xy = sum w dx dy,
        // S_yy = sum w dy^2 -- and preprocess_bwd combines them per Gaussian with the conic:
        //   dL/dmean2D = -(W/2, H/2) * (a S_x + b S_y, c S_y + b S_x),   dL/dconic = -1/2 (S_xx, S_xy, S_yy)
        // (linear in the partials, so summing first is the same sum; 9 packed operations per pair instead of 20).
        const f32x2 v_op = G * dLa;
        // w = dL/dG * G = opacity * (dL/dalpha * G) = opacity * v_op: the opacity is per Gaussian, so the moments are summed of
        // v_op alone and preprocess_bwd multiplies the five sums by it -- 6 instead of 9 packed operations per pair (round 6:
        // render_bwd 0.515 -> 0.503 ms per view, profiles/r6_bwd_opacity_ab.txt)
        const f32x2 wdx = v_op * dx, wdy = v_op * dy;
        const f32x2 sxx = wdx * dx, sxy = wdx * dy, syy = wdy * dy;
        va[0] = wdx.x; va[1] = wdy.x; va[2] = sxx.x; va[3] = sxy.x; va[4] = syy.x; va[5] = v_op.x;
        vb[0] = wdx.y; vb[1] = wdy.y; vb[2] = sxx.y; vb[3] = sxy.y; vb[4] = syy.y; vb[5] = v_op.y;
        const float v = wave_sum_pair<DEPTH ? 10 : 9>(va, vb);
        // row r of the wavefront: components 5*(r&1) .. +4 of survivor (r >> 1); lane (r, c < 5) flushes component c
        const float4 ids = q[6];
        if (col < 5) {
            const uint32_t id = __float_as_uint(row < 2 ? ids.x : ids.y);
            if (v != 0.0f) atomicAdd(&grad[(size_t)id * kGradStride + 5 * (row & 1) + col], v);
        }
    };

    // software pipeline over batches of 64 entries, highest batch first: list words two batches ahead, records one
    const int top = (m_end - 1) >> 6;
    auto word = [&](int bi) -> uint32_t {
        const int j = (bi << 6) + lane;
        return (bi >= 0 && j < m_end) ? pl[j] : 0u;
    };
    uint32_t w_cur = word(top), w_nxt = word(top - 1);
    float4 r0 = {}, r1 = {}, r2 = {};
    if (w_cur & qbit) {
        const float4* q = rec + 3 * (size_t)(w_cur >> 4);
        r0 = q[0]; r1 = q[1]; r2 = q[2];
    }
    int rem = 0;
    for (int bi = top; bi >= 0; --bi) {
        const bool hit = (w_cur & qbit) != 0;
        const unsigned long long hits = __ballot(hit);
        const float4 a0 = r0, a1 = r1, a2 = r2;
        const uint32_t w_this = w_cur;
        w_cur = w_nxt;
        if (w_cur & qbit) {
            const float4* q = rec + 3 * (size_t)(w_cur >> 4);
            r0 = q[0]; r1 = q[1]; r2 = q[2];
        }
        w_nxt = word(bi - 2);
        if (!hits) continue;
        const int cnt = __popcll(hits);
        if (hit) {
            const int k = rem + __popcll(hits & ~((2ull << lane) - 1ull));  // carried survivor + survivors above this lane: back to front
            float* d = (float*)(cp + (k >> 1) * kBwdQuads) + (k & 1);
            // (coefficients negated, threshold as unsigned-compare bits: see the forward's compaction)
            d[0] = a0.x; d[2] = a0.y;                    // [x_a x_b y_a y_b]
            d[4] = -a0.z; d[6] = -a0.w;                  // [A_a A_b C_a C_b]   (A = a/2, C = c/2)
            d[8] = -a1.x; d[10] = skip_bits(a1.z);       // [B_a B_b thr_a thr_b] (B = b; thr = bits of -threshold)
            d[12] = a1.y; d[14] = __int_as_float((bi << 6) + lane + 1);  // [op_a op_b pos_a pos_b]
            cp[(k >> 1) * kBwdQuads + 4 + (k & 1)] = make_float4(a2.x, a2.y, a2.z, a1.w);  // [r g b depth]
            d[24] = __uint_as_float(w_this >> 4);        // [id_a id_b . .]
        }
        wave_lds_sync();
        // An odd survivor is CARRIED into the next batch (it stays in the front slot as the `a` half; the next batch's
        // first survivor -- in front of it in the list -- becomes its `b`): one half-empty pair per quadrant, not per batch.
        const int total = rem + cnt, npairs = total >> 1;
        for (int p = 0; p < npairs; ++p) do_pair(cp + p * kBwdQuads);
        rem = total & 1;
        if (rem != 0 && npairs != 0) {
            float4 t = {};
            if (lane < kBwdQuads) t = cp[npairs * kBwdQuads + lane];
            if (lane < kBwdQuads) cp[lane] = t;        // (DS operations of one wavefront execute in order)
        }
        wave_lds_sync();  // the next batch's compaction overwrites the slots
    }
    if (rem != 0) {
        // the quadrant's last survivor has no partner: the unused `b` half must hold finite numbers (it enters with weight 0)
        if (lane == 0) {
            float* d = (float*)cp;
            d[1] = 0.0f; d[3] = 0.0f; d[5] = 0.0f; d[7] = 0.0f; d[9] = 0.0f; d[11] = 0.0f; d[13] = 0.0f;
            d[15] = __int_as_float(0x7fffffff);
            cp[5] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            d[25] = 0.0f;
        }
        wave_lds_sync();
        do_pair(cp);
    }
}

// ------------------------------------------------------------------------------------------
template <int DEG>
__device__ __forceinline__ void sh_backward(const float* __restrict__ sh, float x, float y, float z,
                                            const float* gr, float* gsh, float* gdir)
{
    constexpr int NB = (DEG + 1) * (DEG + 1);
    float b[16];
    sh_basis<DEG>(x, y, z, b);
#pragma unroll
    for (int k = 0; k < NB; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) gsh[3 * k + c] += b[k] * gr[c];
    float dbx[16], dby[16], dbz[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) dbx[k] = dby[k] = dbz[k] = 0.0f;
    if constexpr (DEG > 0) {
        dby[1] = -kSH1; dbz[2] = kSH1; dbx[3] = -kSH1;
    }
    if constexpr (DEG > 1) {
        dbx[4] = kSH2[0] * y; dby[4] = kSH2[0] * x;
        dby[5] = kSH2[1] * z; dbz[5] = kSH2[1] * y;
        dbx[6] = kSH2[2] * (-2.0f * x); dby[6] = kSH2[2] * (-2.0f * y); dbz[6] = kSH2[2] * (4.0f * z);
        dbx[7] = kSH2[3] * z; dbz[7] = kSH2[3] * x;
        dbx[8] = kSH2[4] * (2.0f * x); dby[8] = kSH2[4] * (-2.0f * y);
    }
    if constexpr (DEG > 2) {
        const float xx = x * x, yy = y * y, zz = z * z;
        dbx[9] = kSH3[0] * 6.0f * x * y; dby[9] = kSH3[0] * (3.0f * xx - 3.0f * yy);
        dbx[10] = kSH3[1] * y * z; dby[10] = kSH3[1] * x * z; dbz[10] = kSH3[1] * x * y;
        dbx[11] = kSH3[2] * (-2.0f * x * y); dby[11] = kSH3[2] * (4.0f * zz - xx - 3.0f * yy); dbz[11] = kSH3[2] * 8.0f * y * z;
        dbx[12] = kSH3[3] * (-6.0f * x * z); dby[12] = kSH3[3] * (-6.0f * y * z); dbz[12] = kSH3[3] * (6.0f * zz - 3.0f * xx - 3.0f * yy);
        dbx[13] = kSH3[4] * (4.0f * zz - 3.0f * xx - yy); dby[13] = kSH3[4] * (-2.0f * x * y); dbz[13] = kSH3[4] * 8.0f * x * z;
        dbx[14] = kSH3[5] * 2.0f * x * z; dby[14] = kSH3[5] * (-2.0f * y * z); dbz[14] = kSH3[5] * (xx - yy);
        dbx[15] = kSH3[6] * (3.0f * xx - 3.0f * yy); dby[15] = kSH3[6] * (-6.0f * x * y);
    }
    gdir[0] = gdir[1] = gdir[2] = 0.0f;
#pragma unroll
    for (int k = 1; k < NB; ++k) {
        const float s = sh[3 * k] * gr[0] + sh[3 * k + 1] * gr[1] + sh[3 * k + 2] * gr[2];
        gdir[0] += dbx[k] * s; gdir[1] += dby[k] * s; gdir[2] += dbz[k] * s;
    }
}

// V views of one Gaussian set per launch: the inputs are loaded once and the gradients of all views are summed in
// registers before the single write (per view only the 48-byte record, the tile rect and the 48-byte screen-space
// gradient row are read).  view / proj [V,16], campos [V,3], tanfov [V,2] | NULL, scale [V] | NULL; geom and grad
// hold V buffers geom_stride / grad_stride bytes apart.  `accumulate` adds to what the outputs already hold.
// (two wavefronts per SIMD beat one wavefront at 310 registers by 13 %)
#ifndef FS_PBWD_WAVES
#define FS_PBWD_WAVES 2
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FS_PBWD_WAVES, FS_PBWD_WAVES))) void preprocess_bwd_kernel(
    fs_raster_dims d, int V, const float* __restrict__ means3D, const float* __restrict__ cov3D,
    const float* __restrict__ shs, const float* __restrict__ opacities, const float* __restrict__ view_all, const float* __restrict__ proj_all,
    const float* __restrict__ campos_all, const float* __restrict__ tanfov_dev,
    const float* __restrict__ scale_dev, const char* __restrict__ geom_base, size_t geom_stride,
    const char* __restrict__ grad_base, size_t grad_stride,
    float* __restrict__ dL_dmeans3D, float* __restrict__ dL_dmeans2D, float* __restrict__ dL_dcov3D,
    float* __restrict__ dL_dshs, float* __restrict__ dL_dcolors, float* __restrict__ dL_dopac,
    int accumulate, int row0, int row_end)
{
    // rows [row0, row_end) of the Gaussian set (the whole set, or one chunk of a chunked gradient exchange)
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [256 * M*3] SH grads
    const int base = row0 + blockIdx.x * 256;
    const int cnt = min(256, row_end - base);
    const int t = threadIdx.x;
    const int i = base + t;
    const int per_sh = d.M * 3;
    const bool have_sh = shs != nullptr;
    const bool sh_cm = (d.flags & FS_RASTER_SH_CHANNEL_MAJOR) != 0, cov_full = (d.flags & FS_RASTER_COV_FULL) != 0;
    constexpr int kTriu[6] = {0, 1, 2, 4, 5, 8};

    if (t < cnt) {
        // sums over the views
        float am[3] = {0, 0, 0}, acov[6] = {0, 0, 0, 0, 0, 0}, aop = 0.0f, am2x = 0.0f, am2y = 0.0f;
        float gsh[48];
        float acol[3] = {0, 0, 0};
#pragma unroll
        for (int k = 0; k < 48; ++k) gsh[k] = 0.0f;
        // inputs, once
        const float3 p0 = make_float3(means3D[3 * (size_t)i], means3D[3 * (size_t)i + 1], means3D[3 * (size_t)i + 2]);
        float c0[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) c0[k] = cov_full ? cov3D[9 * (size_t)i + kTriu[k]] : cov3D[6 * (size_t)i + k];
        // (the blend summed the moments of dL/dalpha * G: the Gaussian's opacity completes w = dL/dG * G -- see render_bwd)
        const float opac = opacities[i];
        // the Gaussian's SH coefficients wait in the thread's own slots of the LDS staging area (idle until the gradients are staged
        // there after the view loop; row stride M*3 floats is odd for every degree: conflict-free) instead of 48 registers -- with the
        // 48 gradient sums they spilled 59 dwords at two wavefronts per SIMD (round 6)
        float* const shl = lds + t * per_sh;
        if (have_sh) {
            if (d.flags & FS_RASTER_SH_FP16) {
                const _Float16* hsrc = (const _Float16*)shs + (size_t)i * per_sh;
#pragma unroll
                for (int k = 0; k < 48; ++k) if (k < per_sh) shl[k] = (float)hsrc[sh_cm ? (k % 3) * d.M + k / 3 : k];
            } else {
                const float* fsrc = shs + (size_t)i * per_sh;
#pragma unroll
                for (int k = 0; k < 48; ++k) if (k < per_sh) shl[k] = fsrc[sh_cm ? (k % 3) * d.M + k / 3 : k];
            }
        }
        // the NEXT view's rect, gradient row and clamp bits are loaded while this view's gradients are formed (round 6: with two
        // wavefronts per SIMD every view's dependent loads were an exposed round trip -- V of them per thread)
        auto view_rows = [&](int vi, ushort4& rc, float4& ga0, float4& ga1, float4& ga2, uint8_t& cb) __attribute__((always_inline)) {
            const GeomView g = geom_view((void*)(geom_base + geom_stride * vi), d.N > 0 ? d.N : 1);
            const float4* gp = (const float4*)((const float*)(grad_base + grad_stride * vi) + (size_t)i * kGradStride);
            rc = g.rect[i];
            ga0 = gp[0]; ga1 = gp[1]; ga2 = gp[2];
            cb = have_sh ? g.clamp[i] : (uint8_t)0;
        };
        ushort4 rc_n; float4 gn0, gn1, gn2; uint8_t cb_n;
        view_rows(0, rc_n, gn0, gn1, gn2, cb_n);
      for (int vi = 0; vi < V; ++vi) {
        const float* view = view_all + 16 * vi;
        const float* proj = proj_all + 16 * vi;
        const float* campos = campos_all + 3 * vi;
        const float tanfovx = tanfov_dev ? tanfov_dev[2 * vi] : d.tanfovx;
        const float tanfovy = tanfov_dev ? tanfov_dev[2 * vi + 1] : d.tanfovy;
        const float wscale = scale_dev ? scale_dev[vi] : 1.0f;
        float gm[3] = {0, 0, 0}, gcov[6] = {0, 0, 0, 0, 0, 0};
        const ushort4 rc = rc_n;
        const float ga[12] = {gn0.x, gn0.y, gn0.z, gn0.w, gn1.x, gn1.y, gn1.z, gn1.w, gn2.x, gn2.y, gn2.z, gn2.w};
        const uint8_t cb = cb_n;
        if (vi + 1 < V) view_rows(vi + 1, rc_n, gn0, gn1, gn2, cb_n);
        if (rc.z > rc.x && rc.w > rc.y) {
            // the five moment sums of the blend (S_x, S_y, S_xx, S_xy, S_yy)
            const float S0 = ga[0] * opac, S1 = ga[1] * opac, S2 = ga[2] * opac, S3 = ga[3] * opac, S4 = ga[4] * opac;
            float3 p = p0;
            if (scale_dev) { p.x = p.x * wscale; p.y = p.y * wscale; p.z = p.z * wscale; }
            aop += ga[5];
            const float fx = (float)d.W / (2.0f * tanfovx), fy = (float)d.H / (2.0f * tanfovy);
            float c3[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) c3[k] = c0[k];
            if (scale_dev) {
                const float s2 = wscale * wscale;
#pragma unroll
                for (int k = 0; k < 6; ++k) c3[k] = c3[k] * s2;
            }
            const Cov2D cv = project_cov(view, p, c3, fx, fy, tanfovx, tanfovy);
            const float a = cv.a, b = cv.b, c = cv.c;
            const float denom = a * c - b * b;
            const float d2inv = 1.0f / (denom * denom + 0.0000001f);
            // the blend's moment sums -> gradients of the 2-D mean (pixels) and of the conic (A, B, C) = (c, -b, a) / det
            float gm2x = 0.0f, gm2y = 0.0f;
            if (denom != 0.0f) {
                const float det_inv = 1.0f / denom;
                const float cA = c * det_inv, cB = -b * det_inv, cC = a * det_inv;
                gm2x = -0.5f * (float)d.W * (cA * S0 + cB * S1);
                gm2y = -0.5f * (float)d.H * (cC * S1 + cB * S0);
            }
            am2x += gm2x; am2y += gm2y;
            const float gcx = -0.5f * S2, gcy = -0.5f * S3, gcz = -0.5f * S4;
            float gt[3] = {0, 0, 0};
            if (d2inv != 0.0f) {
                const float dL_da = d2inv * (-c * c * gcx + 2.0f * b * c * gcy + (denom - a * c) * gcz);
                const float dL_dc = d2inv * (-a * a * gcz + 2.0f * a * b * gcy + (denom - a * c) * gcx);
                const float dL_db = d2inv * 2.0f * (b * c * gcx - (denom + 2.0f * b * b) * gcy + a * b * gcz);
                const float* m0 = cv.m0;
                const float* m1 = cv.m1;
                gcov[0] = m0[0] * m0[0] * dL_da + m0[0] * m1[0] * dL_db + m1[0] * m1[0] * dL_dc;
                gcov[3] = m0[1] * m0[1] * dL_da + m0[1] * m1[1] * dL_db + m1[1] * m1[1] * dL_dc;
                gcov[5] = m0[2] * m0[2] * dL_da + m0[2] * m1[2] * dL_db + m1[2] * m1[2] * dL_dc;
                gcov[1] = 2.0f * m0[0] * m0[1] * dL_da + (m0[0] * m1[1] + m0[1] * m1[0]) * dL_db + 2.0f * m1[0] * m1[1] * dL_dc;
                gcov[2] = 2.0f * m0[0] * m0[2] * dL_da + (m0[0] * m1[2] + m0[2] * m1[0]) * dL_db + 2.0f * m1[0] * m1[2] * dL_dc;
                gcov[4] = 2.0f * m0[2] * m0[1] * dL_da + (m0[1] * m1[2] + m0[2] * m1[1]) * dL_db + 2.0f * m1[1] * m1[2] * dL_dc;
                const float s00 = c3[0], s01 = c3[1], s02 = c3[2], s11 = c3[3], s12 = c3[4], s22 = c3[5];
                float u0[3], u1[3];
                u0[0] = m0[0] * s00 + m0[1] * s01 + m0[2] * s02;
                u0[1] = m0[0] * s01 + m0[1] * s11 + m0[2] * s12;
                u0[2] = m0[0] * s02 + m0[1] * s12 + m0[2] * s22;
                u1[0] = m1[0] * s00 + m1[1] * s01 + m1[2] * s02;
                u1[1] = m1[0] * s01 + m1[1] * s11 + m1[2] * s12;
                u1[2] = m1[0] * s02 + m1[1] * s12 + m1[2] * s22;
                float gJ00 = 0, gJ02 = 0, gJ11 = 0, gJ12 = 0;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float gM0 = 2.0f * dL_da * u0[k] + dL_db * u1[k];
                    const float gM1 = 2.0f * dL_dc * u1[k] + dL_db * u0[k];
                    gJ00 += gM0 * view[0 + 4 * k];
                    gJ02 += gM0 * view[2 + 4 * k];
                    gJ11 += gM1 * view[1 + 4 * k];
                    gJ12 += gM1 * view[2 + 4 * k];
                }
                const float tz = 1.0f / cv.tz, tz2 = tz * tz, tz3 = tz2 * tz;
                gt[0] = cv.gmx * (-fx * tz2) * gJ02;
                gt[1] = cv.gmy * (-fy * tz2) * gJ12;
                gt[2] = -fx * tz2 * gJ00 - fy * tz2 * gJ11 + (2.0f * fx * cv.tx) * tz3 * gJ02 + (2.0f * fy * cv.ty) * tz3 * gJ12;
            }
            gt[2] += ga[9];
#pragma unroll
            for (int k = 0; k < 3; ++k)
                gm[k] += view[0 + 4 * k] * gt[0] + view[1 + 4 * k] * gt[1] + view[2 + 4 * k] * gt[2];
            {
                const float4 mh = xform44(proj, p);
                const float mw = 1.0f / (mh.w + 0.0000001f);
                const float mul1 = mh.x * mw * mw, mul2 = mh.y * mw * mw;
                gm[0] += (proj[0] * mw - proj[3] * mul1) * gm2x + (proj[1] * mw - proj[3] * mul2) * gm2y;
                gm[1] += (proj[4] * mw - proj[7] * mul1) * gm2x + (proj[5] * mw - proj[7] * mul2) * gm2y;
                gm[2] += (proj[8] * mw - proj[11] * mul1) * gm2x + (proj[9] * mw - proj[11] * mul2) * gm2y;
            }
            if (!have_sh) {
                acol[0] += ga[6]; acol[1] += ga[7]; acol[2] += ga[8];
            } else {
                float gr[3];
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) gr[ch] = ((cb >> ch) & 1) ? 0.0f : ga[6 + ch];
                const float dox = p.x - campos[0], doy = p.y - campos[1], doz = p.z - campos[2];
                const float len = sqrtf(dox * dox + doy * doy + doz * doz);
                const float x = dox / len, y = doy / len, z = doz / len;
                const float* sh = shl;
                float gdv[3];
                switch (d.sh_degree) {
                    case 0: sh_backward<0>(sh, x, y, z, gr, gsh, gdv); break;
                    case 1: sh_backward<1>(sh, x, y, z, gr, gsh, gdv); break;
                    case 2: sh_backward<2>(sh, x, y, z, gr, gsh, gdv); break;
                    default: sh_backward<3>(sh, x, y, z, gr, gsh, gdv); break;
                }
                const float s2 = dox * dox + doy * doy + doz * doz;
                const float inv32 = 1.0f / sqrtf(s2 * s2 * s2);
                gm[0] += ((s2 - dox * dox) * gdv[0] - doy * dox * gdv[1] - doz * dox * gdv[2]) * inv32;
                gm[1] += (-dox * doy * gdv[0] + (s2 - doy * doy) * gdv[1] - doz * doy * gdv[2]) * inv32;
                gm[2] += (-dox * doz * gdv[0] - doy * doz * gdv[1] + (s2 - doz * doz) * gdv[2]) * inv32;
            }
        }
        {   // (the 1/near rescale of this view: d mean' / d mean = s, d cov' / d cov = s^2)
            const float s2 = wscale * wscale;
#pragma unroll
            for (int k = 0; k < 3; ++k) am[k] += scale_dev ? gm[k] * wscale : gm[k];
#pragma unroll
            for (int k = 0; k < 6; ++k) acov[k] += scale_dev ? gcov[k] * s2 : gcov[k];
        }
      }  // views
        if (accumulate) {
#pragma unroll
            for (int k = 0; k < 3; ++k) am[k] += dL_dmeans3D[3 * (size_t)i + k];
            am2x += dL_dmeans2D[3 * (size_t)i];
            am2y += dL_dmeans2D[3 * (size_t)i + 1];
#pragma unroll
            for (int k = 0; k < 6; ++k) acov[k] += cov_full ? dL_dcov3D[9 * (size_t)i + kTriu[k]] : dL_dcov3D[6 * (size_t)i + k];
            aop += dL_dopac[i];
            if (!have_sh) {
#pragma unroll
                for (int k = 0; k < 3; ++k) acol[k] += dL_dcolors[3 * (size_t)i + k];
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) dL_dmeans3D[3 * (size_t)i + k] = am[k];
        dL_dmeans2D[3 * (size_t)i] = am2x;
        dL_dmeans2D[3 * (size_t)i + 1] = am2y;
        dL_dmeans2D[3 * (size_t)i + 2] = 0.0f;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            if (cov_full) dL_dcov3D[9 * (size_t)i + kTriu[k]] = acov[k];
            else dL_dcov3D[6 * (size_t)i + k] = acov[k];
        }
        if (cov_full) {  // the reference reads the upper triangle only (cuda_splatting.py:126): no gradient below it
            dL_dcov3D[9 * (size_t)i + 3] = 0.0f; dL_dcov3D[9 * (size_t)i + 6] = 0.0f; dL_dcov3D[9 * (size_t)i + 7] = 0.0f;
        }
        dL_dopac[i] = aop;
        if (have_sh) {
            // gsh[] beyond the active degree is still zero; static indices keep it in registers
#pragma unroll
            for (int k = 0; k < 48; ++k)
                if (k < per_sh) lds[t * per_sh + (sh_cm ? (k % 3) * d.M + k / 3 : k)] = gsh[k];
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) dL_dcolors[3 * (size_t)i + k] = acol[k];
        }
    }
    if (have_sh) {
        __syncthreads();
        float* dst = dL_dshs + (size_t)base * per_sh;
        const int total = cnt * per_sh;
        if (accumulate)
            for (int k = t; k < total; k += 256) dst[k] += lds[k];
        else
            for (int k = t; k < total; k += 256) dst[k] = lds[k];
    }
}

__global__ __launch_bounds__(256) void zero_float4_kernel(float4* __restrict__ p, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

}  // namespace fs

using namespace fs;

namespace {

// blend backward of ONE view into its screen-space gradient rows (grad [N, 12], zeroed here)
int launch_render_bwd(const fs_raster_dims& d, const float* bg, const void* geom, const void* binning, const void* image,
                      const uint32_t* counters, const float* dL_dcolor, const float* dL_ddepth, float* grad, hipStream_t st)
{
    const int T = num_tiles(d.H, d.W);
    const size_t P = (size_t)d.H * d.W;
    const GeomView g = geom_view(const_cast<void*>(geom), d.N);
    const uint32_t* offsets = (const uint32_t*)binning;
    const uint32_t* point_list = (const uint32_t*)((const char*)binning + binning_offsets_bytes(d.H, d.W));
    const float* final_T = (const float*)image;
    const int32_t* n_contrib = (const int32_t*)((const char*)image + align_up(P * 4, 256));
    // zero the view's 48 N bytes of screen-space gradient rows with a kernel of our own: hipMemsetAsync's fill kernel ran this at
    // 0.18 TB/s (264 us per view at 1.0 M Gaussians, 19 % of the training step's GPU time: profiles/r6_bwd_fill_ab.txt)
    {
        const size_t n4 = (size_t)d.N * kGradStride / 4;      // (kGradStride = 12 floats: three float4 per Gaussian)
        const unsigned blocks = (unsigned)std::min<size_t>((n4 + 255) / 256, 4096);
        if (blocks) hipLaunchKernelGGL(zero_float4_kernel, dim3(blocks), dim3(256), 0, st, (float4*)grad, n4);
    }
    const int nblk = tile_grid_blocks((d.W + kTile - 1) / kTile, (d.H + kTile - 1) / kTile);
    {
        ScopedStage prof_(kStRenderBwd, st);
        auto go = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3(4 * nblk), dim3(64), 0, st, d.H, d.W, T, offsets, point_list, g.rec, bg, counters,
                               final_T, n_contrib, dL_dcolor, dL_ddepth, grad);
        };
        // FS_RASTER_BWD_DEPTH=1 forces the depth-channel instantiation without a depth gradient (A/B runs)
        static const bool force_depth = getenv("FS_RASTER_BWD_DEPTH") && atoi(getenv("FS_RASTER_BWD_DEPTH")) != 0;
        const bool depth = dL_ddepth != nullptr || force_depth;
        if (d.flags & FS_RASTER_FAST_EXP) { if (depth) go(render_bwd_kernel<true, true>); else go(render_bwd_kernel<true, false>); }
        else { if (depth) go(render_bwd_kernel<false, true>); else go(render_bwd_kernel<false, false>); }
    }
    FS_CHECK_LAUNCH("render_bwd");
    return FS_OK;
}

// screen space -> Gaussian parameters for V views at once (gradients summed over the views)
int launch_preprocess_bwd(const fs_raster_dims& d, int V, const float* means3D, const float* cov3D, const float* shs,
                          const float* opacities, const float* view, const float* proj, const float* campos, const float* tanfov,
                          const float* scale, const void* geom, size_t geom_stride, const void* grad, size_t grad_stride,
                          float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dcov3D, float* dL_dshs, float* dL_dcolors,
                          float* dL_dopacities, int accumulate, hipStream_t st, int row0 = 0, int nrows = -1)
{
    const size_t lds = shs ? (size_t)256 * d.M * 3 * sizeof(float) : 0;
    if (nrows < 0) nrows = d.N - row0;
    if (nrows <= 0) return FS_OK;
    {
        ScopedStage prof_(kStPreprocessBwd, st, V);
        hipLaunchKernelGGL(preprocess_bwd_kernel, dim3((nrows + 255) / 256), dim3(256), lds, st, d, V, means3D, cov3D, shs,
                           opacities, view, proj, campos, tanfov, scale, (const char*)geom, geom_stride, (const char*)grad,
                           grad_stride, dL_dmeans3D, dL_dmeans2D, dL_dcov3D, dL_dshs, dL_dcolors, dL_dopacities,
                           accumulate, row0, row0 + nrows);
    }
    FS_CHECK_LAUNCH("preprocess_bwd");
    return FS_OK;
}

}  // namespace

FS_API int fs_raster_backward(const fs_raster_dims* dims, const float* means3D, const float* cov3D,
                              const float* shs, const float* colors_precomp, const float* opacities, const float* bg,
                              const float* viewmatrix, const float* projmatrix, const float* campos,
                              const float* tanfov_dev, const float* scale_dev,
                              const void* geom, const void* binning, const void* image, const uint32_t* counters,
                              const float* dL_dcolor, const float* dL_ddepth, void* grad_scratch,
                              float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dcov3D,
                              float* dL_dshs, float* dL_dcolors, float* dL_dopacities, int accumulate,
                              void* stream_)
{
    if (!dims) return FS_ERR_INVALID_ARG;
    const fs_raster_dims d = *dims;
    if (d.N < 0 || (d.flags & FS_RASTER_NO_BACKWARD_STATE)) return FS_ERR_INVALID_ARG;   // (that forward kept no n_contrib)
    if (d.N == 0) return FS_OK;  // an empty Gaussian set has empty gradients; its arrays may be NULL (as in the forward)
    if (!means3D || !cov3D || !opacities || !bg || !viewmatrix || !projmatrix || !campos || !geom ||
        !binning || !image || !dL_dcolor || !grad_scratch || !dL_dmeans3D || !dL_dmeans2D ||
        !dL_dcov3D || !dL_dopacities)
        return FS_ERR_INVALID_ARG;
    if ((shs == nullptr) == (colors_precomp == nullptr)) return FS_ERR_INVALID_ARG;
    if (shs ? !dL_dshs : !dL_dcolors) return FS_ERR_INVALID_ARG;
    if (shs && d.M * 3 > 48) return FS_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream_;
    int rc = launch_render_bwd(d, bg, geom, binning, image, counters, dL_dcolor, dL_ddepth, (float*)grad_scratch, st);
    if (rc != FS_OK) return rc;
    return launch_preprocess_bwd(d, 1, means3D, cov3D, shs, opacities, viewmatrix, projmatrix, campos, tanfov_dev, scale_dev, geom, 0,
                                 grad_scratch, 0, dL_dmeans3D, dL_dmeans2D, dL_dcov3D, dL_dshs, dL_dcolors,
                                 dL_dopacities, accumulate, st);
}

namespace {
constexpr int kMaxStreamsBwd = 8;
struct ForkJoinBwd {
    hipEvent_t ready = nullptr, done[kMaxStreamsBwd] = {};
    bool ok = false;
    ForkJoinBwd()
    {
        ok = hipEventCreateWithFlags(&ready, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; i < kMaxStreamsBwd && ok; ++i)
            ok = hipEventCreateWithFlags(&done[i], hipEventDisableTiming) == hipSuccess;
    }
};
}  // namespace

static int raster_backward_views_impl(const fs_raster_dims* dims, int32_t v, const float* means3D, const float* cov3D,
                                    const float* shs, const float* colors_precomp, const float* opacities, const float* bg,
                                    const float* viewmatrix, const float* projmatrix, const float* campos,
                                    const float* tanfov, const float* scale, const void* geom, const void* binning,
                                    const void* image, const uint32_t* counters, const size_t strides[3], const float* dL_dcolor,
                                    const float* dL_ddepth, void* grad_scratch, float* dL_dmeans3D, float* dL_dmeans2D,
                                    float* dL_dcov3D, float* dL_dshs, float* dL_dcolors, float* dL_dopacities,
                                    int32_t accumulate, int32_t n_streams, void* const* streams, void* main_stream,
                                    int32_t row0, int32_t nrows, int32_t with_blend)
{
    if (!dims || v < 0 || !strides || n_streams < 0 || n_streams > kMaxStreamsBwd || (n_streams > 0 && !streams))
        return FS_ERR_INVALID_ARG;
    if (row0 < 0 || (nrows >= 0 && (long long)row0 + nrows > dims->N)) return FS_ERR_INVALID_ARG;
    if (v == 0) return FS_OK;
    const fs_raster_dims d = *dims;
    if (d.N < 0 || (d.flags & FS_RASTER_NO_BACKWARD_STATE)) return FS_ERR_INVALID_ARG;
    if (d.N == 0) return FS_OK;
    if (!means3D || !cov3D || !opacities || !bg || !viewmatrix || !projmatrix || !campos || !geom || !binning || !image ||
        !dL_dcolor || !grad_scratch || !dL_dmeans3D || !dL_dmeans2D || !dL_dcov3D || !dL_dopacities)
        return FS_ERR_INVALID_ARG;
    if ((shs == nullptr) == (colors_precomp == nullptr)) return FS_ERR_INVALID_ARG;
    if (shs ? !dL_dshs : !dL_dcolors) return FS_ERR_INVALID_ARG;
    if (shs && d.M * 3 > 48) return FS_ERR_UNSUPPORTED;
    // events belong to the device that was current when they were created: one cached set per (thread, device)
    static thread_local ForkJoinBwd* fj_dev[64] = {};
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) { set_last_error("hipGetDevice", hipGetLastError()); return FS_ERR_LAUNCH; }
    if (!fj_dev[dev_]) fj_dev[dev_] = new ForkJoinBwd();
    ForkJoinBwd& fj = *fj_dev[dev_];
    const int ns = n_streams <= 1 ? 0 : (n_streams < v ? n_streams : v);
    hipStream_t main = (hipStream_t)main_stream;
    const size_t P = (size_t)d.H * d.W, grad_stride = align_up((size_t)d.N * kGradStride * 4, 256);
    if (!with_blend)   // a later chunk of a chunked gradient exchange: the blend backward of every view already ran
        return launch_preprocess_bwd(d, v, means3D, cov3D, shs, opacities, viewmatrix, projmatrix, campos, tanfov, scale, geom,
                                     strides[0], grad_scratch, grad_stride, dL_dmeans3D, dL_dmeans2D, dL_dcov3D, dL_dshs,
                                     dL_dcolors, dL_dopacities, accumulate, main, row0, nrows);
    if (ns > 0) {
        if (!fj.ok || hipEventRecord(fj.ready, main) != hipSuccess) {
            set_last_error("event record", hipGetLastError());
            return FS_ERR_LAUNCH;
        }
        for (int s = 0; s < ns; ++s)
            if (hipStreamWaitEvent((hipStream_t)streams[s], fj.ready, 0) != hipSuccess) {
                set_last_error("stream wait", hipGetLastError());
                return FS_ERR_LAUNCH;
            }
    }
    int rc = FS_OK;
    for (int i = 0; i < v && rc == FS_OK; ++i) {  // the blend backward of every view, alternating over the streams
        hipStream_t st = ns > 0 ? (hipStream_t)streams[i % ns] : main;
        rc = launch_render_bwd(d, bg + 3 * (size_t)i, (const char*)geom + strides[0] * i,
                               (const char*)binning + strides[1] * i, (const char*)image + strides[2] * i,
                               counters ? counters + 2 * (size_t)i : nullptr, dL_dcolor + 3 * P * i, dL_ddepth ? dL_ddepth + P * i : nullptr,
                               (float*)((char*)grad_scratch + grad_stride * i), st);
    }
    for (int s = 0; s < ns; ++s)
        if (hipEventRecord(fj.done[s], (hipStream_t)streams[s]) != hipSuccess ||
            hipStreamWaitEvent(main, fj.done[s], 0) != hipSuccess) {
            set_last_error("stream join", hipGetLastError());
            return FS_ERR_LAUNCH;
        }
    if (rc != FS_OK) return rc;
    // one pass over the Gaussians for all views
    return launch_preprocess_bwd(d, v, means3D, cov3D, shs, opacities, viewmatrix, projmatrix, campos, tanfov, scale, geom,
                                 strides[0], grad_scratch, grad_stride, dL_dmeans3D, dL_dmeans2D, dL_dcov3D, dL_dshs,
                                 dL_dcolors, dL_dopacities, accumulate, main, row0, nrows);
}

FS_API int fs_raster_backward_views(const fs_raster_dims* dims, int32_t v, const float* means3D, const float* cov3D,
                                    const float* shs, const float* colors_precomp, const float* opacities, const float* bg,
                                    const float* viewmatrix, const float* projmatrix, const float* campos,
                                    const float* tanfov, const float* scale, const void* geom, const void* binning,
                                    const void* image, const uint32_t* counters, const size_t strides[3], const float* dL_dcolor,
                                    const float* dL_ddepth, void* grad_scratch, float* dL_dmeans3D, float* dL_dmeans2D,
                                    float* dL_dcov3D, float* dL_dshs, float* dL_dcolors, float* dL_dopacities,
                                    int32_t accumulate, int32_t n_streams, void* const* streams, void* main_stream)
{
    return raster_backward_views_impl(dims, v, means3D, cov3D, shs, colors_precomp, opacities, bg, viewmatrix, projmatrix, campos,
                                      tanfov, scale, geom, binning, image, counters, strides, dL_dcolor, dL_ddepth, grad_scratch,
                                      dL_dmeans3D, dL_dmeans2D, dL_dcov3D, dL_dshs, dL_dcolors, dL_dopacities, accumulate,
                                      n_streams, streams, main_stream, 0, -1, 1);
}

// The same backward with its per-Gaussian pass restricted to rows [row0, row0 + nrows) of the Gaussian set: a caller that
// exchanges the gradients between GPUs chunk by chunk (view_sharding.GradExchange("chunked")) runs the blend backward of all views
// with the first chunk (with_blend = 1) and only the per-Gaussian pass for the others (with_blend = 0), so that the
// reduce-scatter of chunk c overlaps the pass over chunk c + 1.
FS_API int fs_raster_backward_views_rows(const fs_raster_dims* dims, int32_t v, const float* means3D, const float* cov3D,
                                         const float* shs, const float* colors_precomp, const float* opacities, const float* bg,
                                         const float* viewmatrix, const float* projmatrix, const float* campos,
                                         const float* tanfov, const float* scale, const void* geom, const void* binning,
                                         const void* image, const uint32_t* counters, const size_t strides[3],
                                         const float* dL_dcolor, const float* dL_ddepth, void* grad_scratch,
                                         float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dcov3D, float* dL_dshs,
                                         float* dL_dcolors, float* dL_dopacities, int32_t accumulate, int32_t n_streams,
                                         void* const* streams, void* main_stream, int32_t row0, int32_t nrows, int32_t with_blend)
{
    if (nrows < 0) return FS_ERR_INVALID_ARG;
    return raster_backward_views_impl(dims, v, means3D, cov3D, shs, colors_precomp, opacities, bg, viewmatrix, projmatrix, campos,
                                      tanfov, scale, geom, binning, image, counters, strides, dL_dcolor, dL_ddepth, grad_scratch,
                                      dL_dmeans3D, dL_dmeans2D, dL_dcov3D, dL_dshs, dL_dcolors, dL_dopacities, accumulate,
                                      n_streams, streams, main_stream, row0, nrows, with_blend);
}

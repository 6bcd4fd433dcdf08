// raster_fwd.hip -- forward of the tile-based 3D-Gaussian rasterizer for MI355X (gfx950).
//
// Replaces the CUDA extension FreeSplat calls at src/model/decoder/cuda_splatting.py:114-127
// (semantics: SURVEY.md Appendix A.1-A.4).  Pipeline, all on one stream, no host sync:
//
//   project_bin  1 thread / Gaussian    project, cull, EWA conic, SH->RGB, tile rect, 8x8-quadrant masks of the
//                                       alpha >= 1/255 ellipse, AND the binning in the same launch: per-tile instance
//                                       counts in LDS (privatised per workgroup), one returning global atomic per touched
//                                       tile reserves the workgroup's slots in the tile's FIXED-capacity key area, keys
//                                       (depth_bits<<32 | id<<4 | quadrant mask) written there
//   tile_scan    1 workgroup            exclusive scan of the T tile counts -> compact ranges of the saved lists,
//                                       instance total, overflow flag
//   sort_blend   1 workgroup / tile     LDS bucket sort of the tile's keys by depth (O(n); bitonic network for lists
//                                       > 2048 keys and degenerate tiles) -> list words (id<<4 | mask) in (depth, id)
//                                       order, kept IN LDS (written to the saved list only when a backward follows);
//                                       then each of the 4 wavefronts blends its own 8x8 QUADRANT front to back with no
//                                       further barrier: survivors of 64 list entries compacted pairwise into
//                                       wavefront-private LDS, packed-fp32 exponent / exp / alpha over pairs of Gaussians
// (round 2 ran five kernels: preprocess -> tile_scan -> emit -> tile_sort -> render; the second pass over the Gaussians,
//  the global list round trip between sort and blend and two launches per view are gone)
//
// Design notes (MI355X-first, not the CUDA layout):
//   * no global 64-bit radix sort over all instances: instances are binned per tile with
//     atomics (order irrelevant) and each tile is sorted independently in LDS by the unique
//     key (depth bits, gaussian id), which yields exactly the order a stable global sort of
//     (tile, depth) keys emitted in Gaussian order would give -- "identical tile/depth ordering";
//   * nothing on the path needs the instance count on the host;
//   * per-Gaussian screen-space state is one 48-byte record (3 x dwordx4 gather per instance);
//   * tile -> workgroup mapping is XCD-aware and balanced: workgroup b runs on XCD b % 8; tiles are grouped in 4x4
//     super-tiles (neighbours share most of their Gaussians -> one L2) and super-tile s goes to XCD s % 8
//     (fs_common.h:tile_for_block).
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "fs_common.h"

namespace fs {

// value of lane (l ^ M): DPP / ds_swizzle below 32 (no address register), ds_bpermute for 32
template <int M>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v)
{
    if constexpr (M == 1) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);
    else if constexpr (M == 2) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);
    else if constexpr (M == 3) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x1B, 0xF, 0xF, true);
    else if constexpr (M < 32) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, (M << 10) | 0x1F);
    else return (uint32_t)__shfl_xor((int)v, M, 64);
}
template <int M>
__device__ __forceinline__ unsigned long long lane_xor64(unsigned long long v)
{
    const uint32_t lo = lane_xor<M>((uint32_t)v), hi = lane_xor<M>((uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// ------------------------------------------------------------------------------------------
// preprocess
// ------------------------------------------------------------------------------------------
template <int DEG>
__device__ __forceinline__ void eval_sh(const float* __restrict__ sh, float3 dir, float* rgb,
                                        uint8_t& clampbits, int cs, int ks)
{   // coefficient k of channel c sits at sh[k * ks + c * cs]: (ks, cs) = (3, 1) for [M][3] rows, (1, M) for [3][M]
    float b[16];
    sh_basis<DEG>(dir.x, dir.y, dir.z, b);
    constexpr int NB = (DEG + 1) * (DEG + 1);
    clampbits = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* s = sh + c * cs;
        float acc = b[0] * s[0];
#pragma unroll
        for (int k = 1; k < NB; ++k) acc = acc + b[k] * s[k * ks];
        acc = acc + 0.5f;
        if (acc < 0.0f) clampbits |= (uint8_t)(1u << c);
        rgb[c] = fmaxf(acc, 0.0f);
    }
}

// Stage `per` floats for each of the workgroup's `cnt` Gaussians through LDS with coalesced
// loads (the [N, per] rows are contiguous, so the block's slab is one contiguous range).
__device__ __forceinline__ void stage_rows(float* lds, const float* __restrict__ src, size_t base,
                                           int cnt, int per)
{
    const float* s = src + base * per;
    const int total = cnt * per;
    if ((((uintptr_t)s) & 15) == 0) {
        const int nv = total >> 2;
        // four 16-byte requests in flight per thread (a plain load -> LDS store loop keeps one: the wait for it was the
        // longest phase of the kernel, profiles/phase_trace.py)
        int k = threadIdx.x;
        for (; k + 768 < nv; k += 1024) {
            const float4 v0 = ((const float4*)s)[k], v1 = ((const float4*)s)[k + 256];
            const float4 v2 = ((const float4*)s)[k + 512], v3 = ((const float4*)s)[k + 768];
            ((float4*)lds)[k] = v0; ((float4*)lds)[k + 256] = v1;
            ((float4*)lds)[k + 512] = v2; ((float4*)lds)[k + 768] = v3;
        }
        for (; k < nv; k += 256) ((float4*)lds)[k] = ((const float4*)s)[k];
        for (int k = (nv << 2) + threadIdx.x; k < total; k += blockDim.x) lds[k] = s[k];
    } else {
        for (int k = threadIdx.x; k < total; k += blockDim.x) lds[k] = s[k];
    }
}


// ------------------------------------------------------------------------------------------
// Tile binning helpers of the projection kernel's count and key-writing passes.
// ------------------------------------------------------------------------------------------
constexpr int kBinLds = 4096;  // tiles of a workgroup's bounding box whose counters live in LDS

// Which 8x8-pixel quadrants of tile (tx,ty) can receive alpha >= 1/255 from this Gaussian?
// bit q = (qy<<1)|qx.  Conservative (never drops a contributing quadrant): the maximum over the
// quadrant's pixel-centre rectangle of the exponent `power` (a concave quadratic) is compared with
// the record's skip threshold r1.z, which already carries a safety margin.  A (gaussian, tile)
// instance whose mask is 0 could only ever be skipped by the blend loop, so dropping it leaves the
// image unchanged bit for bit; retained instances keep the reference's (tile, depth, index) order.
struct QuadForm {  // q(u,v) = a u^2 + c v^2 + b u v = -power, per Gaussian, and its level set q <= Q
    float a, b, c;
    float Q;      // -(skip threshold with slack): a pixel can contribute only where q <= Q
    float D;      // 4ac - b^2 (> 0)
    float vplus;  // v of the ellipse's rightmost point (leftmost: -vplus)
    float Vmax;   // v-extent of the ellipse
    float inv2a;
};
__device__ __forceinline__ QuadForm quad_form(const float4 r0, const float4 r1)
{
    // approximate rcp/sqrt are fine here: every interval below is widened by kEps pixels and the
    // threshold itself carries slack, so rounding can only keep (never drop) a borderline quadrant
    QuadForm f;
    f.a = -r0.z; f.c = -r0.w; f.b = -r1.x;
    f.Q = -(r1.z - 1e-3f * (1.0f + fabsf(r1.z)));
    f.D = 4.0f * f.a * f.c - f.b * f.b;
    const float invD = __builtin_amdgcn_rcpf(f.D);
    const float U = __builtin_amdgcn_sqrtf(fmaxf(4.0f * f.c * f.Q * invD, 0.0f));
    f.vplus = -f.b * __builtin_amdgcn_rcpf(2.0f * f.c) * U;
    f.Vmax = __builtin_amdgcn_sqrtf(fmaxf(4.0f * f.a * f.Q * invD, 0.0f));
    f.inv2a = __builtin_amdgcn_rcpf(2.0f * f.a);
    return f;
}
// u-interval [uL, uR] of the level set inside the horizontal band v in [vlo, vhi]: the right boundary
// u_R(v) = (-b v + sqrt(4aQ - D v^2)) / 2a is concave with its maximum at v = vplus, so over the band it
// peaks at clamp(vplus, vlo, vhi); mirrored for the left boundary.  Empty -> uL > uR.
struct Band { float uL, uR; };
constexpr float kEps = 2e-5f;
__device__ __forceinline__ Band band_interval(const QuadForm& f, float vlo, float vhi)
{
    Band o;
    if (f.Q < 0.0f || vlo > f.Vmax + kEps || vhi < -f.Vmax - kEps) { o.uL = 3.0e38f; o.uR = -3.0e38f; return o; }
    const float vr = fminf(vhi, fmaxf(vlo, f.vplus)), vl = fminf(vhi, fmaxf(vlo, -f.vplus));
    const float sr = __builtin_amdgcn_sqrtf(fmaxf(4.0f * f.a * f.Q - f.D * vr * vr, 0.0f));
    const float sl = __builtin_amdgcn_sqrtf(fmaxf(4.0f * f.a * f.Q - f.D * vl * vl, 0.0f));
    o.uR = (-f.b * vr + sr) * f.inv2a;
    o.uL = (-f.b * vl - sl) * f.inv2a;
    const float w = kEps * (1.0f + fabsf(o.uR) + fabsf(o.uL));
    o.uR += w; o.uL -= w;
    return o;
}
// Which 8x8-pixel quadrants of a tile can receive alpha >= 1/255 from this Gaussian?  bit q = (qy<<1)|qx.
// Conservative (never drops a contributing quadrant).  A (gaussian, tile) instance whose mask is 0
// could only ever be skipped by the blend loop, so dropping it leaves the image unchanged bit for bit;
// retained instances keep the reference's (tile, depth, index) order.  NaNs keep everything.
struct RowBands { Band top, bot; };
__device__ __forceinline__ RowBands row_bands(const QuadForm& f, const float4 r0, int ty)
{
    const float v0 = (float)(ty * kTile) - r0.y;
    RowBands r;
    r.top = band_interval(f, v0, v0 + 7.0f);
    r.bot = band_interval(f, v0 + 8.0f, v0 + 15.0f);
    return r;
}
__device__ __forceinline__ uint32_t quad_mask_row(const RowBands& rb, const float4 r0, int tx)
{
    const float u0 = (float)(tx * kTile) - r0.x, u1 = u0 + 8.0f;
    uint32_t m = 0;
    if (!(u0 > rb.top.uR) && !(u0 + 7.0f < rb.top.uL)) m |= 1u;
    if (!(u1 > rb.top.uR) && !(u1 + 7.0f < rb.top.uL)) m |= 2u;
    if (!(u0 > rb.bot.uR) && !(u0 + 7.0f < rb.bot.uL)) m |= 4u;
    if (!(u1 > rb.bot.uR) && !(u1 + 7.0f < rb.bot.uL)) m |= 8u;
    return m;
}

// Quadrant masks of every tile of a (small) rect packed 4 bits per tile, row-major: computed once in
// the projection kernel, used by both of its binning passes.  Per tile row the two 8-pixel bands give two u-intervals; instead of
// testing them against every quadrant column with float compares (28 VALU, 8 of them v_cmp, per tile), each interval is
// turned ONCE into the range of 8-pixel cell columns it can touch -- a bit mask over the rect's <= 32 cells -- and a
// tile's 4 bits are two 2-bit fields of those masks (7 integer ops).  Conservative like quad_mask_row (intervals
// widened by 1e-3 cell on top of the band's own slack; NaNs keep everything); masks may differ from it only by keeping
// a borderline quadrant.
__device__ __forceinline__ uint32_t band_cells(const Band& b, float px, int cell0, int ncell)
{
    const uint32_t full = ncell >= 32 ? 0xFFFFFFFFu : ((1u << ncell) - 1u);
    if (b.uL > b.uR) return 0u;          // empty band
    if (!(b.uL <= b.uR)) return full;    // NaN: keep everything
    // cell c = pixels [8c, 8c+7]; it meets [px+uL, px+uR] iff (px+uL-7)/8 <= c <= (px+uR)/8
    const float lo_f = ceilf((b.uL + px - 7.0f) * 0.125f - 1e-3f), hi_f = floorf((b.uR + px) * 0.125f + 1e-3f);
    const int lo = max((int)lo_f, cell0) - cell0, hi = min((int)hi_f, cell0 + ncell - 1) - cell0;   // (float -> int saturates)
    if (lo > hi) return 0u;
    return ((2u << hi) - (1u << lo)) & full;
}
__device__ __forceinline__ unsigned long long pack_quad_masks(const QuadForm& f, const float4 r0, ushort4 rc)
{
    unsigned long long m = 0;
    const int rw = rc.z - rc.x, cell0 = 2 * rc.x, ncell = 2 * rw;   // small rects: rw <= 16
    int k = 0;
    for (int y = rc.y; y < rc.w; ++y) {
        const RowBands rb = row_bands(f, r0, y);
        const uint32_t top = band_cells(rb.top, r0.x, cell0, ncell), bot = band_cells(rb.bot, r0.x, cell0, ncell);
        for (int x = 0; x < rw; ++x, ++k) {
            const uint32_t nib = ((top >> (2 * x)) & 3u) | (((bot >> (2 * x)) & 3u) << 2);
            m |= (unsigned long long)nib << (4 * k);
        }
    }
    return m;
}

// The same masks for a rect of at most 4 x 4 tiles (99.8 % of the rects at config 3) in a FIXED layout: nibble 4 * row + col,
// so that a nibble index maps to its tile with a shift and a mask, plus `nz`: bit (4 * row + col) set iff that tile is an
// instance (any quadrant bit set; every tile of the rect when culling is off).  The binning passes of bin_wave then walk the SET
// BITS of nz -- 5.7 instances per Gaussian at config 3 -- instead of every tile of the rect (9.2, and 15.7 for the largest rect
// of a wavefront, which sets the trip count of all its lanes).
__device__ __forceinline__ uint32_t spread2(uint32_t x)   // 2-bit groups at 2c -> 4c (c < 4)
{
    x = (x | (x << 4)) & 0x0F0Fu;
    return (x | (x << 2)) & 0x3333u;
}
__device__ __forceinline__ unsigned long long pack_quad_masks4(const QuadForm& f, const float4 r0, ushort4 rc, bool cull, uint32_t& nz)
{
    unsigned long long m = 0;
    const int rw = rc.z - rc.x, rh = rc.w - rc.y, cell0 = 2 * rc.x, ncell = 2 * rw;
    const uint32_t rowfull = (1u << rw) - 1u;
    nz = 0u;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (r < rh) {
            const RowBands rb = row_bands(f, r0, rc.y + r);
            const uint32_t top = band_cells(rb.top, r0.x, cell0, ncell), bot = band_cells(rb.bot, r0.x, cell0, ncell);
            m |= (unsigned long long)(spread2(top) | (spread2(bot) << 2)) << (16 * r);
            uint32_t any = top | bot;                 // cell pairs -> one bit per tile column
            any = (any | (any >> 1)) & 0x55u;
            any = (any | (any >> 1)) & 0x33u;
            any = (any | (any >> 2)) & 0x0Fu;
            nz |= (cull ? any : rowfull) << (4 * r);
        }
    }
    return m;
}

// Workgroup-private tile counters: the 256 Gaussians of a workgroup are neighbours on screen
// (the encoder emits them in pixel order), so their tile rectangles span a small bounding box.
// Counting happens with LDS atomics inside that box and only one global atomic per touched
// tile and workgroup leaves the CU.  Falls back to direct global atomics when the box is larger
// than kBinLds tiles.  Returns the box in bx0,by0,bw,bh (bw*bh == 0: nothing to bin).
struct BinBox { int x0, y0, w, h; bool lds; };

__device__ __forceinline__ BinBox block_bin_box(int* s_box, bool valid, ushort4 rc)
{
    // Bounds first inside each wavefront (shuffles; two 16-bit fields per exchange), then across the four
    // wavefronts through 8 LDS words.  (256 threads doing atomicMin/atomicMax on the same four LDS words
    // serialise completely: that version spent 9 of the workgroup's 25 us here.)
    int x0 = valid ? (int)rc.x : 0x7fff, y0 = valid ? (int)rc.y : 0x7fff;
    int x1 = valid ? (int)rc.z : 0, y1 = valid ? (int)rc.w : 0;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const int lo = __shfl_xor((x0 << 16) | y0, m, 64), hi = __shfl_xor((x1 << 16) | y1, m, 64);
        x0 = min(x0, lo >> 16); y0 = min(y0, lo & 0xffff);
        x1 = max(x1, hi >> 16); y1 = max(y1, hi & 0xffff);
    }
    __syncthreads();  // (s_box may alias memory other threads are still reading)
    if ((threadIdx.x & 63) == 0) {
        s_box[2 * (threadIdx.x >> 6)] = (x0 << 16) | y0;
        s_box[2 * (threadIdx.x >> 6) + 1] = (x1 << 16) | y1;
    }
    __syncthreads();
    BinBox b;
    b.x0 = 0x7fff; b.y0 = 0x7fff;
    int bx1 = 0, by1 = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int lo = s_box[2 * w], hi = s_box[2 * w + 1];
        b.x0 = min(b.x0, lo >> 16); b.y0 = min(b.y0, lo & 0xffff);
        bx1 = max(bx1, hi >> 16); by1 = max(by1, hi & 0xffff);
    }
    b.w = max(0, bx1 - b.x0); b.h = max(0, by1 - b.y0);
    b.lds = (b.w * b.h) <= kBinLds;
    return b;
}

// same for fp16-stored rows (FS_RASTER_SH_FP16: storage-only half precision, fp32 math)
__device__ __forceinline__ void stage_rows_half(float* lds, const _Float16* __restrict__ src, size_t base, int cnt,
                                                int per)
{
    const _Float16* s = src + base * per;
    const int total = cnt * per;
    // 16-byte loads of 8 coefficients, two in flight per thread (round 3 loaded ONE 2-byte value per thread and iteration:
    // 27 dependent trips per workgroup -- the fp16 forward was 7 % SLOWER than the fp32 one, profiles/r4a_bench.json)
    typedef _Float16 half8 __attribute__((ext_vector_type(8)));
    int done = 0;
    if ((((uintptr_t)s) & 15) == 0) {
        const int nv = total >> 3;
        auto put = [&](int k, const half8 v) {
            ((float4*)lds)[2 * k] = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
            ((float4*)lds)[2 * k + 1] = make_float4((float)v[4], (float)v[5], (float)v[6], (float)v[7]);
        };
        int k = threadIdx.x;
        for (; k + 256 < nv; k += 512) {
            const half8 v0 = ((const half8*)s)[k], v1 = ((const half8*)s)[k + 256];
            put(k, v0); put(k + 256, v1);
        }
        for (; k < nv; k += 256) put(k, ((const half8*)s)[k]);
        done = nv << 3;
    }
    for (int k = done + threadIdx.x; k < total; k += blockDim.x) lds[k] = (float)s[k];
}

// One Gaussian into one camera: cull (z <= 0.2), EWA conic + 0.3 px^2, 3-sigma radius, tile rect, SH -> RGB, skip threshold.
// Shared by both projection kernels, so that a Gaussian's record has the same bits whichever launch shape produced it.
struct Projected {
    float4 r0, r1, r2;   // the 48-byte screen-space record (fs_common.h)
    ushort4 rect;
    uint8_t cb;          // SH clamp bits (backward)
    int rad;             // 0 = culled
};
__device__ __forceinline__ Projected project_gaussian(
    const fs_raster_dims& d, bool live, float3 p_in, const float (&c_in)[6], float op, size_t i,
    const float* __restrict__ colors, const float* sh /* this Gaussian's staged SH row (LDS) */, int sh_cs, int sh_ks,
    const float* __restrict__ view, const float* __restrict__ proj, const float* __restrict__ campos,
    float tanfovx, float tanfovy, bool rescale, float wscale)
{
    Projected o;
    o.r0 = make_float4(0, 0, 0, 0); o.r1 = o.r0; o.r2 = o.r0;
    o.rect = make_ushort4(0, 0, 0, 0);
    o.cb = 0;
    o.rad = 0;
    float3 p = p_in;
    if (rescale) { p.x = p.x * wscale; p.y = p.y * wscale; p.z = p.z * wscale; }
    const float3 pv = xform43(view, p);
    if (live && pv.z > 0.2f) {
        const float4 ph = xform44(proj, p);
        const float pw = 1.0f / (ph.w + 0.0000001f);
        const float ndcx = ph.x * pw, ndcy = ph.y * pw;
        const float fx = (float)d.W / (2.0f * tanfovx), fy = (float)d.H / (2.0f * tanfovy);
        float c3[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) c3[k] = c_in[k];
        if (rescale) {
            const float s2 = wscale * wscale;
#pragma unroll
            for (int k = 0; k < 6; ++k) c3[k] = c3[k] * s2;
        }
        const Cov2D cv = project_cov(view, p, c3, fx, fy, tanfovx, tanfovy);
        const float det = cv.a * cv.c - cv.b * cv.b;
        if (det != 0.0f) {
            const float det_inv = 1.0f / det;
            const float cA = cv.c * det_inv, cB = -cv.b * det_inv, cC = cv.a * det_inv;
            const float mid = 0.5f * (cv.a + cv.c);
            const float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
            const float l1 = mid + sq, l2 = mid - sq;
            const float my_radius = ceilf(3.0f * sqrtf(fmaxf(l1, l2)));
            const float px = ((ndcx + 1.0f) * (float)d.W - 1.0f) * 0.5f;
            const float py = ((ndcy + 1.0f) * (float)d.H - 1.0f) * 0.5f;
            const int r = (int)my_radius;
            const int gx = (d.W + kTile - 1) / kTile, gy = (d.H + kTile - 1) / kTile;
            const int x0 = min(gx, max(0, (int)((px - (float)r) / (float)kTile)));
            const int y0 = min(gy, max(0, (int)((py - (float)r) / (float)kTile)));
            const int x1 = min(gx, max(0, (int)((px + (float)r + (float)(kTile - 1)) / (float)kTile)));
            const int y1 = min(gy, max(0, (int)((py + (float)r + (float)(kTile - 1)) / (float)kTile)));
            if ((x1 - x0) * (y1 - y0) > 0) {
                float rgb[3];
                if (colors) {
                    rgb[0] = colors[3 * i];
                    rgb[1] = colors[3 * i + 1];
                    rgb[2] = colors[3 * i + 2];
                } else {
                    float3 dir = make_float3(p.x - campos[0], p.y - campos[1], p.z - campos[2]);
                    const float len = sqrtf(dir.x * dir.x + dir.y * dir.y + dir.z * dir.z);
                    dir.x = dir.x / len; dir.y = dir.y / len; dir.z = dir.z / len;
                    switch (d.sh_degree) {
                        case 0: eval_sh<0>(sh, dir, rgb, o.cb, sh_cs, sh_ks); break;
                        case 1: eval_sh<1>(sh, dir, rgb, o.cb, sh_cs, sh_ks); break;
                        case 2: eval_sh<2>(sh, dir, rgb, o.cb, sh_cs, sh_ks); break;
                        default: eval_sh<3>(sh, dir, rgb, o.cb, sh_cs, sh_ks); break;
                    }
                }
                // Conservative skip threshold for the blend loop: alpha = op*exp(power) < 1/255
                // whenever power < log(1/(255 op)) - margin.  Pure optimisation: pairs inside the
                // margin are evaluated exactly, so the result never depends on this value.
                float thr;
                if (op > 0.0f) {
                    const float l = -logf(255.0f * op);
                    thr = l - (0.001f + 1e-5f * fabsf(l));
                } else {
                    thr = 1.0f;  // power <= 0 < thr: never contributes
                }
                o.r0 = make_float4(px, py, -0.5f * cA, -0.5f * cC);
                o.r1 = make_float4(-cB, op, thr, pv.z);
                o.r2 = make_float4(rgb[0], rgb[1], rgb[2], 0.0f);
                o.rect = make_ushort4((unsigned short)x0, (unsigned short)y0, (unsigned short)x1,
                                      (unsigned short)y1);
                o.rad = r;
            }
        }
    }
    return o;
}

// Mean (12 B) and covariance (24 / 36 B) rows straight into registers: a wavefront's rows are one contiguous
// 0.8 - 2.3 KB range, every fetched line is fully used.
__device__ __forceinline__ void load_mean_cov(const fs_raster_dims& d, const float* __restrict__ means3D,
                                              const float* __restrict__ cov3D, size_t i, float3& p_in, float (&c_in)[6])
{
    p_in = make_float3(means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]);
    if (d.flags & FS_RASTER_COV_FULL) {  // row-major 3x3, upper triangle
        const float* cr = cov3D + 9 * i;
        c_in[0] = cr[0]; c_in[1] = cr[1]; c_in[2] = cr[2]; c_in[3] = cr[4]; c_in[4] = cr[5]; c_in[5] = cr[8];
    } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) c_in[k] = cov3D[6 * i + k];
    }
}

// ------------------------------------------------------------------------------------------
// preprocess_views: projection + binning of ALL views of a call in one launch ("one launch set for
// all v target views").  A workgroup owns CH chunks of 64 consecutive Gaussians; their SH rows are staged in LDS once and
// mean / covariance / opacity sit in registers for the whole launch, so the 148-160 B per Gaussian are read ONCE for the
// v views (the per-view kernel of rounds 1-5 re-read them v times: 279 MB of its traffic per view).  Wavefront w works
// on chunk w % CH and the views w / CH, w / CH + 4 / CH, ...: the view index is wave-uniform (camera matrices arrive by
// scalar loads) and the BINNING IS WAVEFRONT-PRIVATE -- the 64 Gaussians of a chunk are neighbours on screen, their tile
// box (median 44 tiles at config 3) gets its counters in the wavefront's own 2 KB of LDS, and the count / reserve / write
// phases are separated by wave-level fences only.  After the staging barrier the four wavefronts never meet again: the
// returning global atomics and record stores of one view overlap the arithmetic of another.
//   CH = 1: v >= 4 (4 views in flight per workgroup)   CH = 2: v == 2   CH = 4: v == 1 or 3 (every wavefront its own chunk)
// ------------------------------------------------------------------------------------------
constexpr int kWaveBins = 512;   // tiles of a wavefront's bounding box whose counters live in LDS (config 3: 99.9 % of the boxes)

__device__ __forceinline__ void bin_wave(const fs_raster_dims& d, int lane, uint32_t* s_cnt, uint32_t i, const Projected& pr,
                                         uint32_t* __restrict__ tile_counts, unsigned long long* __restrict__ keys,
                                         uint32_t tile_cap)
{
    const float4 r0 = pr.r0, r1 = pr.r1;
    const ushort4 rect = pr.rect;
    const bool valid = pr.rad > 0;
    const bool cull = (d.flags & FS_RASTER_TILE_CULL) != 0;
    const int gx = (d.W + kTile - 1) / kTile;
    const QuadForm qf = quad_form(r0, r1);
    const int rw = rect.z - rect.x, rh = rect.w - rect.y;
    const int area = rw * rh;
    const bool small = area <= 16;
    // every rect of the wavefront at most 4 x 4 tiles (wave-uniform; nearly always): fixed-layout masks, set-bit walks
    const bool all4 = __builtin_amdgcn_ballot_w64(valid && (rw > 4 || rh > 4)) == 0ull;
    unsigned long long qm = 0;
    uint32_t nz = 0u;
    if (all4) { if (valid) qm = pack_quad_masks4(qf, r0, rect, cull, nz); }
    else if (valid && small) qm = pack_quad_masks(qf, r0, rect);
    // key = depth bits : (gaussian id << 4 | quadrant mask); the mask is a function of (id, tile), so
    // ordering by the key is ordering by (depth, id)
    const unsigned long long key_hi = ((unsigned long long)__float_as_uint(r1.w) << 32) | (i << 4);
    // the wavefront's tile box (shuffles; two 16-bit fields per exchange)
    int x0 = valid ? (int)rect.x : 0x7fff, y0 = valid ? (int)rect.y : 0x7fff;
    int x1 = valid ? (int)rect.z : 0, y1 = valid ? (int)rect.w : 0;
#define FS_BOX_STEP(M)                                                                                           \
    {                                                                                                            \
        const int lo = (int)lane_xor<M>((uint32_t)((x0 << 16) | y0)), hi = (int)lane_xor<M>((uint32_t)((x1 << 16) | y1)); \
        x0 = min(x0, lo >> 16); y0 = min(y0, lo & 0xffff);                                                       \
        x1 = max(x1, hi >> 16); y1 = max(y1, hi & 0xffff);                                                       \
    }
    FS_BOX_STEP(32) FS_BOX_STEP(16) FS_BOX_STEP(8) FS_BOX_STEP(4) FS_BOX_STEP(2) FS_BOX_STEP(1)
#undef FS_BOX_STEP
    const int bx0 = __builtin_amdgcn_readfirstlane(x0), by0 = __builtin_amdgcn_readfirstlane(y0);
    const int bw = max(0, __builtin_amdgcn_readfirstlane(x1) - bx0), bh = max(0, __builtin_amdgcn_readfirstlane(y1) - by0);
    const int nb = bw * bh;
    if (nb == 0) return;   // (wave-uniform) nothing of this chunk is on screen
    if (nb <= kWaveBins && all4) {
        for (int k = lane; k < nb; k += 64) s_cnt[k] = 0u;
        wave_lds_sync();
        const int cbase = (rect.y - by0) * bw + (rect.x - bx0);   // the rect's first tile among the wavefront's counters
        for (uint32_t z = nz; z != 0u; z &= z - 1u) {
            const int k = __builtin_ctz(z);
            atomicAdd(&s_cnt[cbase + (k >> 2) * bw + (k & 3)], 1u);
        }
        wave_lds_sync();
        for (int k0 = 0; k0 < nb; k0 += 64) {
            const int k = k0 + lane;
            if (k < nb) {
                const uint32_t c = s_cnt[k];
                const int ky = k / bw, kx = k - ky * bw;
                s_cnt[k] = c ? atomicAdd(&tile_counts[(by0 + ky) * gx + bx0 + kx], c) : 0u;
            }
        }
        wave_lds_sync();
        const uint32_t tbase = (uint32_t)(rect.y * gx + rect.x);
        for (uint32_t z = nz; z != 0u; z &= z - 1u) {
            const int k = __builtin_ctz(z);
            const uint32_t m = (uint32_t)(qm >> (4 * k)) & 15u;
            const uint32_t slot = atomicAdd(&s_cnt[cbase + (k >> 2) * bw + (k & 3)], 1u);
            // (tile * tile_cap + slot < 2^32: tile_capacity() bounds the product)
            if (slot < tile_cap) keys[(tbase + (uint32_t)((k >> 2) * gx + (k & 3))) * tile_cap + slot] = key_hi | m;
        }
        wave_lds_sync();   // (the wavefront's next view reuses the counters)
    } else if (nb <= kWaveBins) {
        for (int k = lane; k < nb; k += 64) s_cnt[k] = 0u;
        wave_lds_sync();
        if (valid) {
            int k = 0;
            for (int y = rect.y; y < rect.w; ++y) {
                RowBands rb = {};
                if (!small) rb = row_bands(qf, r0, y);
                for (int x = rect.x; x < rect.z; ++x, ++k) {
                    const uint32_t m = small ? (uint32_t)(qm >> (4 * k)) & 15u : quad_mask_row(rb, r0, x);
                    if (!cull || m) atomicAdd(&s_cnt[(y - by0) * bw + (x - bx0)], 1u);
                }
            }
        }
        wave_lds_sync();
        // first slot of this wavefront in each touched tile's key area (one returning global atomic per touched tile)
        for (int k0 = 0; k0 < nb; k0 += 64) {
            const int k = k0 + lane;
            if (k < nb) {
                const uint32_t c = s_cnt[k];
                const int ky = k / bw, kx = k - ky * bw;
                s_cnt[k] = c ? atomicAdd(&tile_counts[(by0 + ky) * gx + bx0 + kx], c) : 0u;
            }
        }
        wave_lds_sync();
        if (valid) {
            int k = 0;
            for (int y = rect.y; y < rect.w; ++y) {
                RowBands rb = {};
                if (!small) rb = row_bands(qf, r0, y);
                for (int x = rect.x; x < rect.z; ++x, ++k) {
                    const uint32_t m = small ? (uint32_t)(qm >> (4 * k)) & 15u : quad_mask_row(rb, r0, x);
                    if (!cull || m) {
                        const uint32_t slot = atomicAdd(&s_cnt[(y - by0) * bw + (x - bx0)], 1u);
                        // (tile * tile_cap + slot < 2^32: tile_capacity() bounds the product)
                        if (slot < tile_cap) keys[(uint32_t)(y * gx + x) * tile_cap + slot] = key_hi | m;
                    }
                }
            }
        }
        wave_lds_sync();   // (the wavefront's next view reuses the counters)
    } else if (all4) {     // a chunk that wraps around an image row of the context view: straight to the global counters
        const uint32_t tbase = (uint32_t)(rect.y * gx + rect.x);
        for (uint32_t z = nz; z != 0u; z &= z - 1u) {
            const int k = __builtin_ctz(z);
            const uint32_t m = (uint32_t)(qm >> (4 * k)) & 15u;
            const uint32_t tile = tbase + (uint32_t)((k >> 2) * gx + (k & 3));
            const uint32_t slot = atomicAdd(&tile_counts[tile], 1u);
            if (slot < tile_cap) keys[tile * tile_cap + slot] = key_hi | m;
        }
    } else if (valid) {
        int k = 0;
        for (int y = rect.y; y < rect.w; ++y) {
            RowBands rb = {};
            if (!small) rb = row_bands(qf, r0, y);
            for (int x = rect.x; x < rect.z; ++x, ++k) {
                const uint32_t m = small ? (uint32_t)(qm >> (4 * k)) & 15u : quad_mask_row(rb, r0, x);
                if (!cull || m) {
                    const uint32_t slot = atomicAdd(&tile_counts[y * gx + x], 1u);
                    if (slot < tile_cap) keys[(uint32_t)(y * gx + x) * tile_cap + slot] = key_hi | m;
                }
            }
        }
    }
}

// the tile counters of nv views (view k's at base + k * stride bytes) in one launch
__global__ __launch_bounds__(256) void zero_counts_kernel(char* base, size_t stride, int T)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < T) ((uint32_t*)(base + stride * blockIdx.y))[k] = 0u;
}

// per-view arrays of a call: view i's buffer = base + i * stride (bytes)
struct ViewBuffers {
    char* geom; size_t geom_stride;
    char* scratch; size_t scratch_stride;     // [T] tile counts | [T x tile_cap] keys
    int32_t* radii;                           // [v, N] | nullptr
};

// Six wavefronts per SIMD (80 registers, ~23 of the loop's cold values spilled to scratch): the kernel is bound by the latency
// of its record stores, LDS atomics and returning global atomics, and occupancy buys more than the spills cost -- same-session
// A/B at config 3, 16 views per call (profiles/r6_preprocess_ab.txt): unconstrained (109 registers, 4 wavefronts) 4 270 views/s,
// 5: 4 322, 6: 4 370, 7: 4 346.  (A/B builds: make VARIANT=pw5 EXTRA=-DFS_PRE_WAVES=5)
#ifndef FS_PRE_WAVES
#define FS_PRE_WAVES 6
#endif
#define FS_PRE_OCC __attribute__((amdgpu_waves_per_eu(FS_PRE_WAVES, FS_PRE_WAVES)))
template <int CH>
__global__ __launch_bounds__(256) FS_PRE_OCC void preprocess_views_kernel(
    fs_raster_dims d, int v, const float* __restrict__ means3D, const float* __restrict__ cov3D,
    const float* __restrict__ shs, const float* __restrict__ colors, const float* __restrict__ opacities,
    const float* __restrict__ view, const float* __restrict__ proj, const float* __restrict__ campos,
    const float* __restrict__ tanfov_dev, const float* __restrict__ scale_dev, ViewBuffers vb, uint32_t tile_cap)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int kPerBlock = 64 * CH;
    const int base = blockIdx.x * kPerBlock;
    const int cnt = min(kPerBlock, d.N - base);
    const int per_sh = shs ? d.M * 3 : 0;
    const bool sh_cm = (d.flags & FS_RASTER_SH_CHANNEL_MAJOR) != 0;
    const int sh_cs = sh_cm ? d.M : 1, sh_ks = sh_cm ? 1 : 3;
    float* const l_sh = lds;   // [64 * CH * per_sh]: the only rows wide enough to need staging
    if (shs) {
        if (d.flags & FS_RASTER_SH_FP16) stage_rows_half(l_sh, (const _Float16*)shs, base, cnt, per_sh);
        else stage_rows(l_sh, shs, base, cnt, per_sh);
    }
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int chunk = wave % CH;
    uint32_t* const s_cnt_base = (uint32_t*)(lds + (size_t)kPerBlock * per_sh);
    uint32_t* const s_cnt = s_cnt_base + wave * kWaveBins;
    const int li = chunk * 64 + lane;            // this thread's Gaussian inside the workgroup's slab
    const bool live = li < cnt;
    const uint32_t i = (uint32_t)(base + li);
    // mean / covariance / opacity: loaded once, parked in LDS (SoA: conflict-free) and re-read per view -- ten registers that
    // would otherwise stay live across the whole view loop
    float* const l_g = (float*)(s_cnt_base + 4 * kWaveBins) + li;   // [10][64 * CH]
    {
        float3 p_in = make_float3(0.0f, 0.0f, 0.0f);
        float c_in[6] = {0, 0, 0, 0, 0, 0};
        float op = 0.0f;
        if (live) {
            load_mean_cov(d, means3D, cov3D, i, p_in, c_in);
            op = opacities[i];
        }
        if (wave < CH) {   // (one wavefront per chunk writes; the others loaded the same rows -- L1 hits -- for nothing: 40 B)
            l_g[0 * kPerBlock] = p_in.x; l_g[1 * kPerBlock] = p_in.y; l_g[2 * kPerBlock] = p_in.z;
#pragma unroll
            for (int k = 0; k < 6; ++k) l_g[(3 + k) * kPerBlock] = c_in[k];
            l_g[9 * kPerBlock] = op;
        }
    }
    __syncthreads();   // SH rows staged; the wavefronts go their own ways from here
    const int T = num_tiles(d.H, d.W);
    const size_t keys_off = align_up((size_t)T * 4, 256);
#pragma nounroll
    for (int j0 = wave / CH; j0 < v; j0 += 4 / CH) {
        const int j = __builtin_amdgcn_readfirstlane(j0);
        // (opaque per iteration: otherwise the LDS address of every SH coefficient and the lane-derived shuffle addresses are
        //  hoisted out of the view loop and stay live across it -- 158 registers instead of ~80)
        int li_ = li, lane_ = lane;
        asm volatile("" : "+v"(li_), "+v"(lane_));
        const float* const lg = (const float*)(s_cnt_base + 4 * kWaveBins) + li_;
        const float3 p_in = make_float3(lg[0 * kPerBlock], lg[1 * kPerBlock], lg[2 * kPerBlock]);
        float c_in[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) c_in[k] = lg[(3 + k) * kPerBlock];
        const float op = lg[9 * kPerBlock];
        const float tanfovx = tanfov_dev ? tanfov_dev[2 * j] : d.tanfovx;
        const float tanfovy = tanfov_dev ? tanfov_dev[2 * j + 1] : d.tanfovy;
        const float wscale = scale_dev ? scale_dev[j] : 1.0f;  // scale-invariant rescale (1/near)
        const Projected pr = project_gaussian(d, live, p_in, c_in, op, i, colors, l_sh + (size_t)li_ * per_sh, sh_cs, sh_ks,
                                              view + 16 * j, proj + 16 * j, campos + 3 * j, tanfovx, tanfovy,
                                              scale_dev != nullptr, wscale);
        if (live) {
            const GeomView g = geom_view(vb.geom + vb.geom_stride * j, d.N);
            g.rec[3 * (size_t)i + 0] = pr.r0;
            g.rec[3 * (size_t)i + 1] = pr.r1;
            g.rec[3 * (size_t)i + 2] = pr.r2;
            if (!(d.flags & FS_RASTER_NO_BACKWARD_STATE)) {   // (only the backward reads these)
                g.rect[i] = pr.rect;
                g.clamp[i] = pr.cb;
            }
            if (vb.radii) vb.radii[(size_t)j * d.N + i] = pr.rad;
        }
        char* const sc = vb.scratch + vb.scratch_stride * j;
        bin_wave(d, lane_, s_cnt, i, pr, (uint32_t*)sc, (unsigned long long*)(sc + keys_off), tile_cap);
    }
}

// The per-view projection + binning kernel of rounds 3 - 5 (one workgroup = 256 Gaussians, workgroup-wide binning with
// four barriers): kept for same-session A/Bs against preprocess_views_kernel (FREESPLAT_PREPROCESS=legacy).
__global__ __launch_bounds__(256) void preprocess_kernel(
    fs_raster_dims d, const float* __restrict__ means3D, const float* __restrict__ cov3D,
    const float* __restrict__ shs, const float* __restrict__ colors,
    const float* __restrict__ opacities, const float* __restrict__ view,
    const float* __restrict__ proj, const float* __restrict__ campos,
    const float* __restrict__ tanfov_dev, const float* __restrict__ scale_dev, GeomView g,
    int32_t* __restrict__ radii, uint32_t* __restrict__ tile_counts, unsigned long long* __restrict__ keys,
    uint32_t tile_cap)
{
    const float tanfovx = tanfov_dev ? tanfov_dev[0] : d.tanfovx;
    const float tanfovy = tanfov_dev ? tanfov_dev[1] : d.tanfovy;
    const float wscale = scale_dev ? scale_dev[0] : 1.0f;  // scale-invariant rescale (1/near)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    FS_PT(0, 0);
    const int base = blockIdx.x * 256;
    const int cnt = min(256, d.N - base);
    const int per_sh = d.M * 3;
    const bool sh_cm = (d.flags & FS_RASTER_SH_CHANNEL_MAJOR) != 0;
    const int sh_cs = sh_cm ? d.M : 1, sh_ks = sh_cm ? 1 : 3;
    float* l_sh = lds;                                   // [256 * per_sh]: the only rows wide enough to need staging
    if (shs) {
        if (d.flags & FS_RASTER_SH_FP16) stage_rows_half(l_sh, (const _Float16*)shs, base, cnt, per_sh);
        else stage_rows(l_sh, shs, base, cnt, per_sh);
    }
    const int t = threadIdx.x;
    const bool live = t < cnt;
    const int i = base + t;
    float3 p_in = make_float3(0.0f, 0.0f, 0.0f);
    float c_in[6] = {0, 0, 0, 0, 0, 0};
    float op = 0.0f;
    if (live) {
        load_mean_cov(d, means3D, cov3D, (size_t)i, p_in, c_in);
        op = opacities[i];
    }
    __syncthreads();
    FS_PT(0, 1);  // inputs staged
    const Projected pr = project_gaussian(d, live, p_in, c_in, op, (size_t)i, colors, l_sh + (size_t)t * per_sh, sh_cs, sh_ks,
                                          view, proj, campos, tanfovx, tanfovy, scale_dev != nullptr, wscale);
    const float4 r0 = pr.r0, r1 = pr.r1, r2 = pr.r2;
    const ushort4 rect = pr.rect;
    const uint8_t cb = pr.cb;
    const int rad = pr.rad;
    FS_PT(0, 2);  // projected
    if (live) {
        g.rec[3 * (size_t)i + 0] = r0;
        g.rec[3 * (size_t)i + 1] = r1;
        g.rec[3 * (size_t)i + 2] = r2;
        g.rect[i] = rect;
        g.clamp[i] = cb;
        radii[i] = rad;
    }

    // ---- binning (the staging LDS is dead from here on): count per tile in LDS, reserve, write the keys ----
    __syncthreads();
    FS_PT(0, 3);  // records written
    int* s_box = (int*)lds;
    uint32_t* s_cnt = (uint32_t*)lds + 16;   // per tile of the workgroup's box: instance count, then the next free slot
    const bool valid = rad > 0;
    const bool cull = (d.flags & FS_RASTER_TILE_CULL) != 0;
    const int gx = (d.W + kTile - 1) / kTile;
    const QuadForm qf = quad_form(r0, r1);
    const int area = (rect.z - rect.x) * (rect.w - rect.y);
    const bool small = area <= 16;
    unsigned long long qm = 0;
    if (valid && small) qm = pack_quad_masks(qf, r0, rect);
    // key = depth bits : (gaussian id << 4 | quadrant mask); the mask is a function of (id, tile), so
    // ordering by the key is ordering by (depth, id)
    const unsigned long long key_hi = ((unsigned long long)__float_as_uint(r1.w) << 32) | ((uint32_t)i << 4);
    FS_PT(0, 4);  // quadrant masks
    const BinBox bb = block_bin_box(s_box, valid, rect);
    FS_PT(0, 5);  // workgroup box
    if (bb.w * bb.h == 0) return;
    if (bb.lds) {
        for (int k = t; k < bb.w * bb.h; k += 256) s_cnt[k] = 0;
        __syncthreads();
        if (valid) {
            int k = 0;
            for (int y = rect.y; y < rect.w; ++y) {
                RowBands rb = {};
                if (!small) rb = row_bands(qf, r0, y);
                for (int x = rect.x; x < rect.z; ++x, ++k) {
                    const uint32_t m = small ? (uint32_t)(qm >> (4 * k)) & 15u : quad_mask_row(rb, r0, x);
                    if (!cull || m) atomicAdd(&s_cnt[(y - bb.y0) * bb.w + (x - bb.x0)], 1u);
                }
            }
        }
        __syncthreads();
        FS_PT(0, 6);  // counted
        for (int k = t; k < bb.w * bb.h; k += 256) {
            const uint32_t c = s_cnt[k];
            // first slot of this workgroup in the tile's key area (one returning global atomic per touched tile)
            s_cnt[k] = c ? atomicAdd(&tile_counts[(bb.y0 + k / bb.w) * gx + bb.x0 + k % bb.w], c) : 0u;
        }
        __syncthreads();
        FS_PT(0, 7);  // slots reserved
        if (valid) {
            int k = 0;
            for (int y = rect.y; y < rect.w; ++y) {
                RowBands rb = {};
                if (!small) rb = row_bands(qf, r0, y);
                for (int x = rect.x; x < rect.z; ++x, ++k) {
                    const uint32_t m = small ? (uint32_t)(qm >> (4 * k)) & 15u : quad_mask_row(rb, r0, x);
                    if (!cull || m) {
                        const uint32_t slot = atomicAdd(&s_cnt[(y - bb.y0) * bb.w + (x - bb.x0)], 1u);
                        // (tile * tile_cap + slot < 2^32: tile_capacity() bounds the product)
                        if (slot < tile_cap) keys[(uint32_t)(y * gx + x) * tile_cap + slot] = key_hi | m;
                    }
                }
            }
        }
        FS_PT(0, 8);  // keys written
    } else if (valid) {
        int k = 0;
        for (int y = rect.y; y < rect.w; ++y) {
            RowBands rb = {};
            if (!small) rb = row_bands(qf, r0, y);
            for (int x = rect.x; x < rect.z; ++x, ++k) {
                const uint32_t m = small ? (uint32_t)(qm >> (4 * k)) & 15u : quad_mask_row(rb, r0, x);
                if (!cull || m) {
                    const uint32_t slot = atomicAdd(&tile_counts[y * gx + x], 1u);
                    if (slot < tile_cap) keys[(uint32_t)(y * gx + x) * tile_cap + slot] = key_hi | m;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// tile_scan: exclusive scan over T tile counts (T ~ 5e3; one workgroup of 1024 per view: workgroup b scans view b's
// counters at counts0 + b * counts_stride bytes into offsets0 + b * offsets_stride bytes and counters0 + 2 b)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void tile_scan_kernel(const char* __restrict__ counts0, size_t counts_stride,
                                                         char* __restrict__ offsets0, size_t offsets_stride, int T,
                                                         uint32_t* __restrict__ counters0,
                                                         unsigned long long cap, uint32_t tile_cap)
{
    const uint32_t* __restrict__ counts = (const uint32_t*)(counts0 + counts_stride * blockIdx.x);
    uint32_t* __restrict__ offsets = (uint32_t*)(offsets0 + offsets_stride * blockIdx.x);
    uint32_t* __restrict__ counters = counters0 + 2 * blockIdx.x;
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_max[16];
    __shared__ unsigned long long s_wide[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int per = (T + 1023) / 1024;
    const int lo = min(T, t * per), hi = min(T, lo + per);
    uint32_t s = 0, mx = 0;
    unsigned long long wide = 0;  // the instance total in 64 bits: offsets are 32-bit, a wrapped total must not pass as small
    for (int k = lo; k < hi; ++k) { const uint32_t c = counts[k]; s += c; wide += c; mx = max(mx, c); }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        wide += __shfl_xor(wide, d, 64);
        mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
    }
    if (lane == 0) { s_wide[wave] = wide; s_max[wave] = mx; }
    // inclusive scan of the 1024 partial sums: shuffles inside a wavefront, the 16 wavefront totals through LDS
    uint32_t inc = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const uint32_t v = s_wave[k];
        before += k < wave ? v : 0u;
        total += v;
    }
    inc += before;
    uint32_t run = inc - s;
    for (int k = lo; k < hi; ++k) {
        offsets[k] = run;
        run += counts[k];
    }
    if (t == 1023) {
        unsigned long long total64 = 0;
        uint32_t maxc = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { total64 += s_wide[k]; maxc = max(maxc, s_max[k]); }
        offsets[T] = total;
        counters[0] = total64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)total64;  // saturates: "more than 2^32 - 1"
        // overflow: the saved lists need `total` entries, every tile's key area `count` slots.  Non-zero = the largest
        // tile count (>= 1), from which the caller sizes its retry.
        counters[1] = (total64 > cap || maxc > tile_cap) ? max(maxc, 1u) : 0u;
    }
}

// ------------------------------------------------------------------------------------------
// Per-tile sort by (depth bits, id) -- the networks behind the bucket sort: bitonic network in its "flip" form -- every
// compare-exchange is ascending, so an arbitrary length n works in place: partners >= n are
// virtual +inf and never move.
// ------------------------------------------------------------------------------------------
#ifndef FS_SORT_LDS
#define FS_SORT_LDS 1792     // a multiple of 256 (A/B builds: -DFS_SORT_LDS=2048 -DFS_BLEND_WAVES=6 = rounds 3 - 5)
#endif
constexpr int kSortLds = FS_SORT_LDS;  // keys one LDS sort takes = 64-bit staging slots per workgroup (14 KiB)

template <typename Ptr>
__device__ __forceinline__ void bitonic_sort_any(Ptr a, uint32_t n)
{
    uint32_t P = 1;
    while (P < n) P <<= 1;
    const uint32_t half = P >> 1;
    for (uint32_t k = 2; k <= P; k <<= 1) {
        const uint32_t hk = k >> 1;
        for (uint32_t t = threadIdx.x; t < half; t += blockDim.x) {
            const uint32_t blk = t / hk, off = t % hk;
            const uint32_t i = blk * k + off, j = blk * k + (k - 1 - off);
            if (j < n) {
                const unsigned long long x = a[i], y = a[j];
                if (x > y) { a[i] = y; a[j] = x; }
            }
        }
        __syncthreads();
        for (uint32_t s = k >> 2; s >= 1; s >>= 1) {
            for (uint32_t t = threadIdx.x; t < half; t += blockDim.x) {
                const uint32_t i = (t / s) * 2 * s + (t % s), j = i + s;
                if (j < n) {
                    const unsigned long long x = a[i], y = a[j];
                    if (x > y) { a[i] = y; a[j] = x; }
                }
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------
// Register-resident bitonic sort for tile lists of <= 4096 keys: thread t of the 256 holds the EPT
// consecutive keys [t*EPT, (t+1)*EPT).  Strides below EPT are compare-exchanges between a thread's
// own registers; strides that stay inside a wavefront (partner thread = t ^ m, m < 64) move through
// DPP / ds_swizzle / ds_bpermute, i.e. without LDS round trips or barriers; only the last stages'
// cross-wavefront strides (m >= 64) go through LDS.  For 2048 keys that is 3 LDS exchanges instead
// of the 66 barrier-separated LDS passes of the plain network.  Same "flip" network as
// bitonic_sort_any (all compare-exchanges ascending), so padding keys of ~0 stay at the top.
// ------------------------------------------------------------------------------------------
// Exchange with thread t ^ TM; REV: partner's keys are taken in reversed order (flip step).
template <int EPT, int TM, bool REV>
__device__ __forceinline__ void thread_exchange(unsigned long long (&k)[EPT], int t, unsigned long long* lds)
{
    constexpr int HB = TM >= 128 ? 128 : TM >= 64 ? 64 : TM >= 32 ? 32 : TM >= 16 ? 16 : TM >= 8 ? 8 : TM >= 4 ? 4 : TM >= 2 ? 2 : 1;
    const bool low = (t & HB) == 0;  // this thread holds the lower global indices of every pair
    unsigned long long y[EPT];
    if constexpr (TM < 64) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) y[e] = lane_xor64<TM>(k[REV ? EPT - 1 - e : e]);
    } else {
        constexpr int CH = EPT * 256 > kSortLds ? kSortLds / 256 : EPT;  // keys per thread that fit the exchange buffer
#pragma unroll
        for (int e0 = 0; e0 < EPT; e0 += CH) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < CH; ++e) lds[e * 256 + t] = k[REV ? EPT - 1 - (e0 + e) : e0 + e];
            __syncthreads();
#pragma unroll
            for (int e = 0; e < CH; ++e) y[e0 + e] = lds[e * 256 + (t ^ TM)];
        }
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const unsigned long long x = k[e];
        k[e] = low ? (y[e] < x ? y[e] : x) : (y[e] > x ? y[e] : x);
    }
}

template <int EPT, int J>  // half-cleaner strides J, J/2, ..., 1
struct Clean {
    static __device__ __forceinline__ void run(unsigned long long (&k)[EPT], int t, unsigned long long* lds)
    {
        if constexpr (J >= EPT) {
            thread_exchange<EPT, J / EPT, false>(k, t, lds);
        } else {
#pragma unroll
            for (int e = 0; e < EPT; ++e)
                if ((e & J) == 0) {
                    const unsigned long long a = k[e], b = k[e | J];
                    if (a > b) { k[e] = b; k[e | J] = a; }
                }
        }
        if constexpr (J > 1) Clean<EPT, J / 2>::run(k, t, lds);
    }
};
template <int EPT, int K>  // stages 2, 4, ..., K
struct Stages {
    static __device__ __forceinline__ void run(unsigned long long (&k)[EPT], int t, unsigned long long* lds)
    {
        if constexpr (K > 2) Stages<EPT, K / 2>::run(k, t, lds);
        // flip step: i <-> i ^ (K-1)
        if constexpr (K <= EPT) {
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const int p = e ^ (K - 1);
                if (e < p) {
                    const unsigned long long a = k[e], b = k[p];
                    if (a > b) { k[e] = b; k[p] = a; }
                }
            }
        } else {
            thread_exchange<EPT, K / EPT - 1, true>(k, t, lds);
        }
        if constexpr (K >= 4) Clean<EPT, K / 4>::run(k, t, lds);
    }
};

template <int EPT>
__device__ __forceinline__ void sort_tile_in_registers(const unsigned long long* __restrict__ keys,
                                                       uint32_t* __restrict__ out, uint32_t n,
                                                       unsigned long long* lds)
{
    const int t = threadIdx.x;
    unsigned long long k[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const uint32_t i = (uint32_t)t * EPT + e;
        k[e] = i < n ? keys[i] : ~0ull;
    }
    Stages<EPT, 256 * EPT>::run(k, t, lds);
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const uint32_t i = (uint32_t)t * EPT + e;
        if (i < n) out[i] = (uint32_t)k[e];
    }
}

// Lists a little longer than a power of two (the common case: C3 averages 1156 keys per tile) would pay the
// whole next network for mostly padding.  Two-run form: the first 256*EA keys and the remaining <= 256*EB keys
// (EB < EA) are sorted by their own, smaller networks, laid out back to back through LDS, and merged by the last
// stage of the 512*EA network alone (one flip + its half-cleaners).  In steps x keys-per-thread this is
// 55*4 + 36*1 + 11*8 = 344 instead of 66*8 = 528 for 1024 < n <= 1280.
template <int EA, int EB>
__device__ __forceinline__ void sort_tile_two_runs(const unsigned long long* __restrict__ keys,
                                                   uint32_t* __restrict__ out, uint32_t n, unsigned long long* lds)
{
    static_assert(EB < EA, "second run must be the shorter one");
    constexpr int E = 2 * EA, NA = 256 * EA, NB = 256 * EB;
    const int t = threadIdx.x;
    unsigned long long ka[EA], kb[EB];
#pragma unroll
    for (int e = 0; e < EA; ++e) ka[e] = keys[t * EA + e];  // n > NA
#pragma unroll
    for (int e = 0; e < EB; ++e) {
        const uint32_t i = (uint32_t)(NA + t * EB + e);
        kb[e] = i < n ? keys[i] : ~0ull;
    }
    Stages<EA, NA>::run(ka, t, lds);
    Stages<EB, NB>::run(kb, t, lds);
    unsigned long long k[E];
#pragma unroll
    for (int c0 = 0; c0 < NA + NB; c0 += kSortLds) {  // through the exchange buffer, kSortLds keys at a time
        __syncthreads();
#pragma unroll
        for (int e = 0; e < EA; ++e) {
            const int i = t * EA + e - c0;
            if (i >= 0 && i < kSortLds) lds[i] = ka[e];
        }
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            const int i = NA + t * EB + e - c0;
            if (i >= 0 && i < kSortLds) lds[i] = kb[e];
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = t * E + e - c0;
            if (i >= 0 && i < kSortLds && i + c0 < NA + NB) k[e] = lds[i];
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e)
        if (t * E + e >= NA + NB) k[e] = ~0ull;
    thread_exchange<E, 255, true>(k, t, lds);  // flip step of the last stage: i <-> i ^ (512*EA - 1)
    Clean<E, 256 * E / 4>::run(k, t, lds);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const uint32_t i = (uint32_t)t * E + e;
        if (i < n) out[i] = (uint32_t)k[e];
    }
}


// ------------------------------------------------------------------------------------------
// Bucket sort of one tile's keys in LDS (lists of <= 2048 keys: every tile of the BASELINE configs).
// A tile sees a narrow depth band, and inside it the depth bits of its Gaussians are spread fairly evenly
// (measured on config 3: 2048 buckets over [zmin, zmax] hold <= 10 keys each, 2.4 key comparisons per key), so an
// O(n) distribution sort replaces the O(n log^2 n) network: ~60 VALU per key instead of ~550
// (profiles/r1j_sq_counters.json: tile_sort was 49 M of the pipeline's 201 M VALU wave-instructions per view).
//   1. zmin / zmax of the tile (wave shuffles + 8 LDS words);
//   2. bucket = floor((z - zmin) * CAP / (zmax - zmin + 1)) -- a monotone function of the depth bits, so buckets are
//      ordered front to back --, slot in the bucket by an LDS atomic (arrival order, arbitrary);
//   3. exclusive scan of the CAP bucket counts (each thread owns EPT consecutive buckets);
//   4. keys staged bucket by bucket in LDS; every key counts the keys of ITS bucket that are smaller (full 64-bit
//      compare: equal depths order by Gaussian index, as the reference's stable sort does) -> final position.
// The result is the same total order by (depth bits, id) as before, bit for bit.  Degenerate tiles (a bucket of more
// than kMaxBucket keys: many identical depths) fall back to the register bitonic network on the keys already loaded.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kMaxBucket = 24;
// The sorted list words are left in LDS (`list`, which ALIASES `cnt`: positions are formed in registers, written after a
// barrier) for the blend that follows in the same workgroup; `gout` != nullptr also writes them to the saved list
// (training: the backward walks it).
template <int EPT>
__device__ __forceinline__ void sort_tile_buckets(const unsigned long long* keys, uint32_t* __restrict__ gout,
                                                  uint32_t n, unsigned long long* sk, uint32_t* cnt, uint32_t* s_red,
                                                  int tid = -1)
{
    constexpr int CAP = 256 * EPT;  // capacity == number of buckets
    uint32_t* const list = cnt;
    // (`tid`: the caller's opaque copy of threadIdx.x -- sort_tile_partitioned calls this in a loop and must not have the
    //  lane-derived shuffle addresses of this body hoisted out of it and kept live in registers across the whole loop)
    const int t = tid >= 0 ? tid : (int)threadIdx.x, lane = t & 63, wave = t >> 6;
    unsigned long long k[EPT];
    uint32_t zmin = 0xFFFFFFFFu, zmax = 0u;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const uint32_t i = (uint32_t)(e * 256 + t);
        k[e] = i < n ? keys[i] : ~0ull;
        cnt[i] = 0u;
        if (i < n) {
            const uint32_t z = (uint32_t)(k[e] >> 32);
            zmin = min(zmin, z);
            zmax = max(zmax, z);
        }
    }
    if (t == 0) cnt[CAP] = 0u;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        zmin = min(zmin, (uint32_t)__shfl_xor((int)zmin, m, 64));
        zmax = max(zmax, (uint32_t)__shfl_xor((int)zmax, m, 64));
    }
    if (lane == 0) { s_red[wave] = zmin; s_red[4 + wave] = zmax; }
    __syncthreads();
    zmin = min(min(s_red[0], s_red[1]), min(s_red[2], s_red[3]));
    zmax = max(max(s_red[4], s_red[5]), max(s_red[6], s_red[7]));
    const float scale = (float)CAP / (float)(zmax - zmin + 1u);
    uint32_t bs[EPT];  // bucket << 8 | slot   (slot <= kMaxBucket matters only when not degenerate)
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const uint32_t i = (uint32_t)(e * 256 + t);
        bs[e] = 0u;
        if (i < n) {
            const uint32_t z = (uint32_t)(k[e] >> 32);
            const uint32_t b = min((uint32_t)(CAP - 1), (uint32_t)((float)(z - zmin) * scale));
            const uint32_t slot = atomicAdd(&cnt[b], 1u);
            bs[e] = (b << 8) | min(slot, 255u);
        }
    }
    __syncthreads();
    // exclusive scan of the bucket counts: thread t owns buckets [t*EPT, (t+1)*EPT)
    uint32_t c[EPT], sum = 0u, mx = 0u;
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        c[j] = cnt[t * EPT + j];
        sum += c[j];
        mx = max(mx, c[j]);
    }
    uint32_t inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) s_red[8 + wave] = inc;
    const bool degenerate = __syncthreads_or(mx > kMaxBucket) != 0;
    uint32_t run = inc - sum;
#pragma unroll
    for (int w = 0; w < 3; ++w) run += w < wave ? s_red[8 + w] : 0u;
    if (degenerate) {
        if constexpr ((EPT & (EPT - 1)) == 0) {
            Stages<EPT, CAP>::run(k, t, sk);  // (its exchanges synchronise the workgroup themselves)
            __syncthreads();
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const uint32_t i = (uint32_t)t * EPT + e;
                if (i < n) {
                    list[i] = (uint32_t)k[e];
                    if (gout) gout[i] = (uint32_t)k[e];
                }
            }
        } else {
            // (the register network needs a power-of-two key count per thread: CAP = 1792 sorts in the LDS staging instead)
            __syncthreads();   // (`keys` may BE the staging -- sort_tile_partitioned: every thread holds its keys in registers now)
#pragma unroll
            for (int e = 0; e < EPT; ++e)
                if ((uint32_t)(e * 256 + t) < n) sk[e * 256 + t] = k[e];
            __syncthreads();
            bitonic_sort_any(sk, n);
            for (uint32_t i = (uint32_t)t; i < n; i += 256u) {
                const uint32_t wd = (uint32_t)sk[i];
                list[i] = wd;
                if (gout) gout[i] = wd;
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        cnt[t * EPT + j] = run;
        run += c[j];
    }
    if (t == 255) cnt[CAP] = run;  // == n
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPT; ++e)
        if ((uint32_t)(e * 256 + t) < n) sk[cnt[bs[e] >> 8] + (bs[e] & 255u)] = k[e];
    __syncthreads();
    uint32_t pos[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        pos[e] = 0u;
        if ((uint32_t)(e * 256 + t) < n) {
            const uint32_t b = bs[e] >> 8;
            const uint32_t lo = cnt[b], hi = cnt[b + 1];
            uint32_t r = 0u;
            for (uint32_t j = lo; j < hi; ++j) r += sk[j] < k[e] ? 1u : 0u;
            pos[e] = lo + r;
        }
    }
    __syncthreads();  // every read of the bucket offsets is done: the list may overwrite them
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        if ((uint32_t)(e * 256 + t) < n) {
            list[pos[e]] = (uint32_t)k[e];
            if (gout) gout[pos[e]] = (uint32_t)k[e];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Tile lists LONGER than the LDS sort's capacity (close-up / dense views: round 5's c3_closeup workload has every tile there):
// a two-level distribution sort, O(n) like the bucket sort it builds on, instead of the O(n log^2 n) networks (register bitonic
// up to 4096 keys, ~550 VALU per key; beyond that the in-place global-memory network: ~100 barrier-separated passes over the keys).
//   1. zmin / zmax and a 2048-bin depth histogram of the tile's keys (bins monotone in the depth bits) in LDS;
//   2. exclusive scan; consecutive bins are grouped greedily into GROUPS of <= kSortLds keys (thread 0, binary searches on the
//      prefix sums: <= 64 groups);
//   3. per group: the keys of its depth range are compacted from the tile's key area into the LDS staging (ballot + one LDS
//      atomic per wavefront; arrival order is irrelevant), sorted by sort_tile_buckets -- the full (depth bits, id) order -- and
//      written to the group's range of the tile's saved list.  Groups are depth-ordered and equal depths share a bin, so the
//      concatenation is the same total order as before, bit for bit.
// Cost: 2 + G passes over the keys (L2-resident, coalesced) + G LDS sorts, G ~ n / 1500.  A single bin of more than kSortLds keys
// (thousands of identical depths) or more than 64 groups returns false: the caller falls back to the global network.
// ------------------------------------------------------------------------------------------
constexpr int kMaxGroups = 64;
// (forceinline: as a separate function -- tried to keep its register pressure out of the kernel -- the second group's key count
//  came out wrong on the GPU (gfx950, ROCm 7.2: its output range held 2048 entries instead of m; profiles/r5_long_sort_debug.txt);
//  inlined, the group loop spills ~15 registers to scratch, in this cold path only.)
__device__ __forceinline__ bool sort_tile_partitioned(
    const unsigned long long* __restrict__ kt, uint32_t* __restrict__ gl, uint32_t n,
                                                      unsigned long long* sk, uint32_t* cnt, uint32_t* s_red, uint32_t* s_grp)
{
    constexpr int NB = kSortLds;   // fine bins (== cnt's size - 1)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    uint32_t zmin = 0xFFFFFFFFu, zmax = 0u;
    for (uint32_t i = t; i < n; i += 256) {
        const uint32_t z = (uint32_t)(kt[i] >> 32);
        zmin = min(zmin, z);
        zmax = max(zmax, z);
    }
    for (int i = t; i <= NB; i += 256) cnt[i] = 0u;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        zmin = min(zmin, (uint32_t)__shfl_xor((int)zmin, m, 64));
        zmax = max(zmax, (uint32_t)__shfl_xor((int)zmax, m, 64));
    }
    if (lane == 0) { s_red[wave] = zmin; s_red[4 + wave] = zmax; }
    __syncthreads();
    zmin = min(min(s_red[0], s_red[1]), min(s_red[2], s_red[3]));
    zmax = max(max(s_red[4], s_red[5]), max(s_red[6], s_red[7]));
    const float scale = (float)NB / (float)(zmax - zmin + 1u);
    auto bin_of = [&](unsigned long long key) { return min((uint32_t)(NB - 1), (uint32_t)((float)((uint32_t)(key >> 32) - zmin) * scale)); };
    for (uint32_t i = t; i < n; i += 256) atomicAdd(&cnt[bin_of(kt[i])], 1u);
    __syncthreads();
    // exclusive scan of the NB bin counts: thread t owns bins [t * 8, t * 8 + 8)
    constexpr int BPT = NB / 256;
    uint32_t c[BPT], sum = 0u;
#pragma unroll
    for (int j = 0; j < BPT; ++j) { c[j] = cnt[t * BPT + j]; sum += c[j]; }
    uint32_t inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) s_red[8 + wave] = inc;
    __syncthreads();
    uint32_t run = inc - sum;
#pragma unroll
    for (int w = 0; w < 3; ++w) run += w < wave ? s_red[8 + w] : 0u;
#pragma unroll
    for (int j = 0; j < BPT; ++j) { cnt[t * BPT + j] = run; run += c[j]; }
    if (t == 255) cnt[NB] = run;   // == n
    __syncthreads();
    // groups of consecutive bins holding <= kSortLds keys: s_grp[g] = first bin, s_grp[kMaxGroups + 1 + g] = keys before it
    if (t == 0) {
        uint32_t g = 0, b = 0;
        bool ok = true;
        while (b < (uint32_t)NB && ok) {
            const uint32_t base = cnt[b];
            uint32_t lo = b, hi = NB;              // largest e in (b, NB] with cnt[e] - base <= kSortLds
            while (lo < hi) {
                const uint32_t mid = (lo + hi + 1) >> 1;
                if (cnt[mid] - base <= (uint32_t)kSortLds) lo = mid; else hi = mid - 1;
            }
            if (lo == b || g >= (uint32_t)kMaxGroups) { ok = false; break; }   // one bin alone is too long / too many groups
            s_grp[g] = b;
            s_grp[kMaxGroups + 1 + g] = base;
            ++g;
            b = lo;
        }
        s_grp[g] = NB;
        s_grp[kMaxGroups + 1 + g] = n;
        s_grp[2 * kMaxGroups + 2] = ok ? g : 0xFFFFFFFFu;
    }
    __syncthreads();
    const uint32_t G = s_grp[2 * kMaxGroups + 2];
    if (G == 0xFFFFFFFFu) return false;
    for (uint32_t g = 0; g < G; ++g) {
        const uint32_t b0 = s_grp[g], b1 = s_grp[g + 1], off = s_grp[kMaxGroups + 1 + g], m = s_grp[kMaxGroups + 2 + g] - off;
        if (m == 0u) continue;   // (workgroup-uniform)
        if (t == 0) s_red[12] = 0u;
        __syncthreads();
        for (uint32_t i0 = 0; i0 < n; i0 += 256) {
            const uint32_t i = i0 + t;
            unsigned long long key = 0ull;
            bool in = false;
            if (i < n) {
                key = kt[i];
                const uint32_t b = bin_of(key);
                in = b >= b0 && b < b1;
            }
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(in);
            if (mask != 0ull) {
                uint32_t base = 0u;
                if (lane == 0) base = atomicAdd(&s_red[12], (uint32_t)__popcll(mask));
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (in) sk[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u))] = key;
            }
        }
        __syncthreads();
        // The group must have received exactly the keys the histogram counted for its bins.  It always has when this function
        // is inlined; as a separate (noinline) function the second group's count came out wrong on gfx950 / ROCm 7.2
        // (profiles/r5_long_sort_debug.txt, root cause not found: ADVICE r5).  Should a toolchain change bring that back, the
        // tile declines here and takes the global network -- slower, never wrong.  (workgroup-uniform: read after the barrier)
        if (s_red[12] != m) return false;
        int tt = t;
        asm volatile("" : "+v"(tt));   // opaque per iteration (see sort_tile_buckets)
        sort_tile_buckets<kSortLds / 256>(sk, gl + off, m, sk, cnt, s_red, tt);
        __syncthreads();   // the sort's last LDS writes (its list copy) precede the next group's zeroing of the counters
    }
    return true;
}

// ------------------------------------------------------------------------------------------
// Blend loop of one wavefront = one 8x8 quadrant of the tile, fully independent of the other three (no workgroup
// barrier): front-to-back alpha compositing over the tile's sorted list.
//   * the list words carry a 4-bit quadrant mask (computed once at binning time): the wavefront ballots 64 entries at a
//     time against its own bit and gathers the 48-byte records of the entries that have it one batch ahead, so the global
//     latency of batch i+1 is covered by the blending of batch i; entries that cannot touch the quadrant cost 1/64 of a
//     VALU test instead of a full per-pixel evaluation;
//   * the survivors of a batch are COMPACTED into a wavefront-private LDS area, two per slot with their fields
//     interleaved ([x_a x_b y_a y_b] ...), so that the walk reads register PAIRS straight from LDS
//     (6 ds_read_b128 per two survivors) and the exponent, the exp and alpha of both run on packed fp32
//     (v_pk_{add,mul,fma}_f32): ~50 VALU per two survivors instead of ~90 with scalar fp32 and broadcast reads;
//   * four survivors per step (the two packed exponent / exp chains are independent, the scheduler interleaves them); the
//     full steps of a batch carry no "does this half exist" logic, only the batch's last, partial step does.
// Arithmetic per pixel is the same sequence of IEEE operations as the oracle => same bits.
// LDS_LIST: `pl` points into LDS (the list the workgroup has just sorted) or to the saved list in global memory (lists
// longer than the LDS sort's capacity): two instantiations, so that each uses its own address space's instructions.
// TRACK: keep the per-pixel contributor count (n_contrib, the backward's starting point); off with
// FS_RASTER_NO_BACKWARD_STATE (inference: 4 fewer selects per 4 survivors)
// ------------------------------------------------------------------------------------------
constexpr int kPairQuads = 6;  // float4 per slot of two survivors
constexpr int kPairArea = 34 * kPairQuads;  // float4 per wavefront: up to 3 carried + 64 new survivors = 67 entries, two per slot
template <bool FAST_EXP, bool TRACK, bool LDS_LIST>
__device__ __forceinline__ void blend_quadrant(const uint32_t* pl, int n, const float4* __restrict__ rec, float4* const cp,
                                               int H, int W, int tx, int ty, int wave, const float* __restrict__ bg,
                                               float* __restrict__ out_color, float* __restrict__ out_depth,
                                               float* __restrict__ out_alpha, float* __restrict__ final_T,
                                               int32_t* __restrict__ n_contrib)
{
    const int lane = threadIdx.x & 63;
    const int px = tx * kTile + (wave & 1) * 8 + (lane & 7);
    const int py = ty * kTile + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const f32x2 pfx = splat2((float)px), pfy = splat2((float)py);
    const uint32_t qbit = 1u << wave;

    float T_ = 1.0f;
    f32x2 C01 = splat2(0.0f), C2D = splat2(0.0f);
    int last = 0;
    // Per-pixel predicates live as explicit 64-bit WAVE MASKS in SGPRs (ballot in, inverse_ballot out): the compiler
    // keeps a loop-carried `bool done` as a 0/1 byte in a VGPR and re-materialises the mask for every use (v_and +
    // v_cmp + s_xor per survivor, v_cndmask + v_or to update it: ~4 of ~30 VALU per survivor in round 2's loop).
    unsigned long long done_m = __builtin_amdgcn_ballot_w64(!inside);

    // VM: mask of the pixels this survivor can reach (power in range, alpha >= 1/255, pixel not yet saturated)
#define FS_BLEND_ONE(VM, AL, OM, KR, POS)                                                           \
    {                                                                                               \
        const float test_T = T_ * (OM);                                                             \
        const unsigned long long vis = (VM) & ~done_m;                                              \
        const unsigned long long okc = __builtin_amdgcn_ballot_w64(test_T >= 0.0001f);              \
        done_m |= vis & ~okc;                     /* T would fall below 1e-4: the pixel is done, this one not applied */ \
        const bool ok = __builtin_amdgcn_inverse_ballot_w64(vis & okc);                             \
        const f32x2 wgt = splat2(ok ? (AL) * T_ : 0.0f); /* weight 0: sums unchanged (finite colours) */ \
        C01 = fma2((f32x2){(KR).x, (KR).y}, wgt, C01);                                              \
        C2D = fma2((f32x2){(KR).z, (KR).w}, wgt, C2D);                                              \
        T_ = ok ? test_T : T_;                                                                      \
        if constexpr (TRACK) last = ok ? __float_as_int(POS) : last;                                \
    }
    // One step = two slots = four survivors a, b | c, d, blended strictly in list order.  PARTIAL: the quadrant's very
    // last step, `left` (1..3) entries; full steps carry no "does this entry exist" logic.
    auto step = [&](const float4* q, auto partial, int left) __attribute__((always_inline)) {
        constexpr bool PARTIAL = decltype(partial)::value;
        const float4 c0 = q[0], c1 = q[1], c2 = q[2], c3 = q[3];
        const float4 e0 = q[6], e1 = q[7], e2 = q[8], e3 = q[9];
        const f32x2 dx = (f32x2){c0.x, c0.y} - pfx, dy = (f32x2){c0.z, c0.w} - pfy;
        const f32x2 ex = (f32x2){e0.x, e0.y} - pfx, ey = (f32x2){e0.z, e0.w} - pfy;
        // -power = A dx^2 + dy (C dy + B dx): 3 multiplications + 2 fmas (the oracle's own association)
        const f32x2 pw = fma2((f32x2){c1.x, c1.y} * dx, dx, dy * fma2((f32x2){c1.z, c1.w}, dy, (f32x2){c2.x, c2.y} * dx));
        const f32x2 pv = fma2((f32x2){e1.x, e1.y} * ex, ex, ey * fma2((f32x2){e1.z, e1.w}, ey, (f32x2){e2.x, e2.y} * ex));
        // +0 <= q <= -threshold  <=>  bits(q) <= bits(-threshold) as unsigned (negative q and NaN compare above)
        unsigned long long ma = __builtin_amdgcn_ballot_w64(__float_as_uint(pw.x) <= __float_as_uint(c2.z));
        unsigned long long mb = __builtin_amdgcn_ballot_w64(__float_as_uint(pw.y) <= __float_as_uint(c2.w));
        unsigned long long mc = __builtin_amdgcn_ballot_w64(__float_as_uint(pv.x) <= __float_as_uint(e2.z));
        unsigned long long md = __builtin_amdgcn_ballot_w64(__float_as_uint(pv.y) <= __float_as_uint(e2.w));
        if constexpr (PARTIAL) { mb = left > 1 ? mb : 0ull; mc = left > 2 ? mc : 0ull; md = 0ull; }
        if ((((ma | mb) | (mc | md)) & ~done_m) == 0ull) return;  // wave-uniform: nobody in range
        const float4 ka = q[4], kb = q[5], kc = q[10], kd = q[11];
        f32x2 ew = blend_exp_of_neg<FAST_EXP>(pw), ev = blend_exp_of_neg<FAST_EXP>(pv);
        f32x2 ow = (f32x2){c3.x, c3.y} * ew, ov = (f32x2){e3.x, e3.y} * ev;
        f32x2 aw = (f32x2){fminf(0.99f, ow.x), fminf(0.99f, ow.y)};
        f32x2 av = (f32x2){fminf(0.99f, ov.x), fminf(0.99f, ov.y)};
        unsigned long long ga, gb, gc, gd;   // alpha >= 1/255
        if constexpr (FAST_EXP) {
            // hardware exp; a step with an alpha inside the guard band of the 1/255 threshold (fs_common.h) is
            // re-evaluated with the contract exp, so the accept / reject decisions are those of the exact mode
            const float glo = alpha_guard_lo(), ghi = alpha_guard_hi();
            ga = __builtin_amdgcn_ballot_w64(aw.x >= ghi); gb = __builtin_amdgcn_ballot_w64(aw.y >= ghi);
            gc = __builtin_amdgcn_ballot_w64(av.x >= ghi); gd = __builtin_amdgcn_ballot_w64(av.y >= ghi);   // nothing in [glo, ghi): same as >= 1/255
            const unsigned long long nb = (__builtin_amdgcn_ballot_w64(aw.x >= glo) & ~ga) | (__builtin_amdgcn_ballot_w64(aw.y >= glo) & ~gb)
                                        | (__builtin_amdgcn_ballot_w64(av.x >= glo) & ~gc) | (__builtin_amdgcn_ballot_w64(av.y >= glo) & ~gd);
            if (__builtin_expect(nb != 0ull, 0)) {  // wave-uniform, rare
                ew = blend_exp_of_neg<false>(pw); ev = blend_exp_of_neg<false>(pv);
                ow = (f32x2){c3.x, c3.y} * ew; ov = (f32x2){e3.x, e3.y} * ev;
                aw = (f32x2){fminf(0.99f, ow.x), fminf(0.99f, ow.y)};
                av = (f32x2){fminf(0.99f, ov.x), fminf(0.99f, ov.y)};
                ga = __builtin_amdgcn_ballot_w64(aw.x >= 1.0f / 255.0f); gb = __builtin_amdgcn_ballot_w64(aw.y >= 1.0f / 255.0f);
                gc = __builtin_amdgcn_ballot_w64(av.x >= 1.0f / 255.0f); gd = __builtin_amdgcn_ballot_w64(av.y >= 1.0f / 255.0f);
            }
        } else {
            ga = __builtin_amdgcn_ballot_w64(aw.x >= 1.0f / 255.0f); gb = __builtin_amdgcn_ballot_w64(aw.y >= 1.0f / 255.0f);
            gc = __builtin_amdgcn_ballot_w64(av.x >= 1.0f / 255.0f); gd = __builtin_amdgcn_ballot_w64(av.y >= 1.0f / 255.0f);
        }
        const f32x2 mw = splat2(1.0f) - aw, mv = splat2(1.0f) - av;
        FS_BLEND_ONE(ma & ga, aw.x, mw.x, ka, c3.z)
        FS_BLEND_ONE(mb & gb, aw.y, mw.y, kb, c3.w)
        FS_BLEND_ONE(mc & gc, av.x, mv.x, kc, e3.z)
        if constexpr (!PARTIAL) FS_BLEND_ONE(md & gd, av.y, mv.y, kd, e3.w)
    };

    // software pipeline: list words two batches ahead, records one batch ahead
    uint32_t w_cur = lane < n ? pl[lane] : 0u;
    uint32_t w_nxt = 64 + lane < n ? pl[64 + lane] : 0u;
    float4 r0 = {}, r1 = {}, r2 = {};
    if (w_cur & qbit) {
        const float4* q = rec + 3 * (size_t)(w_cur >> 4);
        r0 = q[0]; r1 = q[1]; r2 = q[2];
    }
    // Survivors that do not fill a step of four are CARRIED into the next batch (they stay at the front of the
    // compaction area, the next batch's survivors are appended behind them): one partial step per quadrant instead of
    // one per batch (~0.4 of a step per batch of ~9).
    int rem = 0;
    for (int c = 0; c < n; c += 64) {
        if (done_m == ~0ull) break;  // every pixel of the quadrant is saturated (or outside the image)
        const bool hit = (w_cur & qbit) != 0;
        const unsigned long long hits = __ballot(hit);
        const float4 a0 = r0, a1 = r1, a2 = r2;
        w_cur = w_nxt;
        w_nxt = c + 128 + lane < n ? pl[c + 128 + lane] : 0u;
        if (w_cur & qbit) {
            const float4* q = rec + 3 * (size_t)(w_cur >> 4);
            r0 = q[0]; r1 = q[1]; r2 = q[2];
        }
        if (!hits) continue;
        if (hit) {
            const int k = rem + __builtin_amdgcn_mbcnt_hi((uint32_t)(hits >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hits, 0));
            float* d = (float*)(cp + (k >> 1) * kPairQuads) + (k & 1);
            // (coefficients NEGATED: the walk evaluates q = -power >= 0, bit for bit the negation of the oracle's power,
            //  so that "0 >= power >= threshold" is one unsigned compare of q's bits against skip_bits)
            d[0] = a0.x; d[2] = a0.y;                   // [x_a x_b y_a y_b]
            d[4] = -a0.z; d[6] = -a0.w;                 // [A_a A_b C_a C_b]   (A = a/2, C = c/2)
            d[8] = -a1.x; d[10] = skip_bits(a1.z);      // [B_a B_b thr_a thr_b] (B = b; thr = bits of -threshold)
            d[12] = a1.y; d[14] = __int_as_float(c + lane + 1);  // [op_a op_b pos_a pos_b]
            cp[(k >> 1) * kPairQuads + 4 + (k & 1)] = make_float4(a2.x, a2.y, a2.z, a1.w);  // [r g b depth]
        }
        wave_lds_sync();
        const int total = rem + __popcll(hits), nfull = total >> 2;
        for (int p = 0; p < nfull; ++p) step(cp + 2 * p * kPairQuads, std::false_type{}, 4);
        rem = total & 3;
        if (rem != 0 && nfull != 0) {
            // the leftover entries sit in the slot pair behind the last full step: move that pair to the front
            // (DS operations of one wavefront execute in order: the steps' reads are done, the copy's read precedes its write)
            float4 t = {};
            if (lane < 2 * kPairQuads) t = cp[2 * nfull * kPairQuads + lane];
            if (lane < 2 * kPairQuads) cp[lane] = t;
        }
        wave_lds_sync();   // (the next batch's compaction appends behind the carried entries)
    }
    if (rem != 0 && done_m != ~0ull) {
        // the quadrant's last, partial step: its unused entries enter with weight 0, their colours must still be finite
        if (lane >= rem && lane < 4) cp[(lane >> 1) * kPairQuads + 4 + (lane & 1)] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        wave_lds_sync();
        step(cp, std::true_type{}, rem);
    }
#undef FS_BLEND_ONE
    if (inside) {
        const size_t pix = (size_t)py * W + px, HW = (size_t)H * W;
        final_T[pix] = T_;
        if constexpr (TRACK) n_contrib[pix] = last;
        out_color[pix] = fmaf(T_, bg[0], C01.x);
        out_color[HW + pix] = fmaf(T_, bg[1], C01.y);
        out_color[2 * HW + pix] = fmaf(T_, bg[2], C2D.x);
        out_depth[pix] = C2D.y;
        out_alpha[pix] = 1.0f - T_;
    }
}

// ------------------------------------------------------------------------------------------
// sort_blend: one workgroup per tile.  All four wavefronts sort the tile's keys in LDS (bucket sort above), the sorted
// list words stay in LDS, then each wavefront blends its own quadrant from them without another workgroup barrier.
// LDS: 14 KiB key staging for 1 792 keys (reused as the four wavefronts' compaction areas once the sort is done) + 7 KiB bucket
// offsets (reused as the sorted list) = 21.6 KiB -> 7 workgroups per CU, 7 wavefronts per SIMD at 68 - 72 registers (round 6;
// rounds 3 - 5: 2 048 keys, 24.6 KiB, 6 wavefronts at 80 registers -- the longest tile list of config 3 is 1 338 entries.
// Same-session A/B, profiles/r6_blend_waves_ab.txt: +1.3 % views/s at config 3, +7 % on the close-up workload).
// Lists longer than the LDS sort's 2048 keys (none in the BASELINE configs; every tile of the c3_closeup workload) are
// sorted into the saved list in global memory by the two-level distribution sort (sort_tile_partitioned: histogram ->
// groups of <= 2048 keys -> one LDS bucket sort per group) and blended from there; a tile it declines (thousands of equal
// depths, > 64 groups, a group whose compacted count disagrees with the histogram) falls back to the in-place global network.
// ------------------------------------------------------------------------------------------
static_assert(4 * kPairArea * sizeof(float4) <= kSortLds * sizeof(unsigned long long), "compaction areas must fit the key staging");
template <bool FAST_EXP, bool TRACK>
#ifndef FS_BLEND_WAVES
#define FS_BLEND_WAVES 7      // (A/B builds: make VARIANT=w6 EXTRA="-DFS_BLEND_WAVES=6 -DFS_SORT_LDS=2048")
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FS_BLEND_WAVES, FS_BLEND_WAVES))) void sort_blend_kernel(
    int H, int W, const uint32_t* __restrict__ counts, const uint32_t* __restrict__ offsets, uint32_t tile_cap,
    unsigned long long* __restrict__ keys, uint32_t* __restrict__ point_list, const float4* __restrict__ rec,
    const float* __restrict__ bg, const uint32_t* __restrict__ counters,
    float* __restrict__ out_color, float* __restrict__ out_depth, float* __restrict__ out_alpha,
    float* __restrict__ final_T, int32_t* __restrict__ n_contrib)
{
    __shared__ __attribute__((aligned(16))) unsigned long long sk[kSortLds];
    __shared__ uint32_t s_cnt[kSortLds + 1];
    __shared__ uint32_t s_red[16];
    __shared__ uint32_t s_grp[2 * kMaxGroups + 3];   // sort_tile_partitioned: group bins | keys before each group | group count
    if (counters[1]) return;  // overflowed capacity: key areas / saved lists are not backed by memory
    const int gx = (W + kTile - 1) / kTile, gy = (H + kTile - 1) / kTile;
    const int tile = tile_for_block(blockIdx.x, gx, gy);  // XCD-aware, balanced (fs_common.h)
    if (tile < 0) return;
    FS_PT(2, 0);
    const uint32_t n = counts[tile];
    unsigned long long* const kt = keys + (size_t)tile * tile_cap;
    uint32_t* const gl = point_list + offsets[tile];   // this tile's range of the saved (compact) list
    const bool in_lds = n <= (uint32_t)kSortLds;
    if (n == 0) {
    } else if (n <= 256u) {
        sort_tile_buckets<1>(kt, TRACK ? gl : nullptr, n, sk, s_cnt, s_red);
    } else if (n <= 512u) {
        sort_tile_buckets<2>(kt, TRACK ? gl : nullptr, n, sk, s_cnt, s_red);
    } else if (n <= 1024u) {
        sort_tile_buckets<4>(kt, TRACK ? gl : nullptr, n, sk, s_cnt, s_red);
    } else if (n <= (uint32_t)kSortLds) {
        sort_tile_buckets<kSortLds / 256>(kt, TRACK ? gl : nullptr, n, sk, s_cnt, s_red);
    }
#ifdef FS_LONG_SORT_NETWORKS   // (rounds 1 - 4: register bitonic networks up to 4096 keys, the global-memory network beyond; A/B builds)
    else if (n <= 2560u) {
        sort_tile_two_runs<8, 2>(kt, gl, n, sk);
    } else if (n <= 3072u) {
        sort_tile_two_runs<8, 4>(kt, gl, n, sk);
    } else if (n <= 4096u) {
        sort_tile_in_registers<16>(kt, gl, n, sk);
    }
#endif
    else if (!sort_tile_partitioned(kt, gl, n, sk, s_cnt, s_red, s_grp)) {
        // degenerate (thousands of identical depths in one tile): in place in global memory, the plain network
        __syncthreads();
        bitonic_sort_any(kt, n);
        for (uint32_t k = threadIdx.x; k < n; k += 256) gl[k] = (uint32_t)kt[k];
    }
    if (!in_lds) __threadfence_block();
    FS_PT(2, 1);  // this wavefront's part of the sort done
    __syncthreads();  // list complete (LDS or global); the key staging is free for the compaction areas
    FS_PT(2, 2);  // list complete
    const int wave = threadIdx.x >> 6;
    float4* const cp = (float4*)sk + wave * kPairArea;
    const int tx = tile % gx, ty = tile / gx;
    if (in_lds)
        blend_quadrant<FAST_EXP, TRACK, true>(s_cnt, (int)n, rec, cp, H, W, tx, ty, wave, bg, out_color, out_depth, out_alpha,
                                              final_T, n_contrib);
    else
        blend_quadrant<FAST_EXP, TRACK, false>(gl, (int)n, rec, cp, H, W, tx, ty, wave, bg, out_color, out_depth, out_alpha,
                                               final_T, n_contrib);
    FS_PT(2, 3);  // quadrant 0 blended
}

}  // namespace fs

// ============================================================================================
// C ABI
// ============================================================================================
using namespace fs;

FS_API int fs_raster_buffer_sizes(int32_t N, int32_t H, int32_t W, int64_t cap, size_t out[4])
{
    if (N < 0 || H <= 0 || W <= 0 || cap < 0 || !out) return FS_ERR_INVALID_ARG;
    const size_t T = (size_t)num_tiles(H, W), P = (size_t)H * W;
    out[0] = geom_bytes(N > 0 ? N : 1);
    out[1] = binning_offsets_bytes(H, W) + align_up((size_t)(cap > 0 ? cap : 1) * 4, 256);
    out[2] = align_up(P * 4, 256) * 2;
    out[3] = align_up(T * 4, 256) + align_up(T * (size_t)tile_capacity(cap, (int)T) * 8, 256);
    return FS_OK;
}

FS_API const uint32_t* fs_raster_tile_ranges(const void* binning, int32_t, int32_t)
{
    return (const uint32_t*)binning;
}
FS_API const uint32_t* fs_raster_point_list(const void* binning, int32_t H, int32_t W)
{
    return (const uint32_t*)((const char*)binning + binning_offsets_bytes(H, W));
}
FS_API const float* fs_raster_geom_records(const void* geom) { return (const float*)geom; }
FS_API const float* fs_raster_final_T(const void* image) { return (const float*)image; }
FS_API const int32_t* fs_raster_n_contrib(const void* image, int32_t H, int32_t W)
{
    return (const int32_t*)((const char*)image + align_up((size_t)H * W * 4, 256));
}

namespace fs {
// FREESPLAT_PREPROCESS=legacy: the per-view projection kernel of rounds 3 - 5 (same-session A/Bs)
static bool legacy_preprocess()
{
    static const bool on = [] { const char* e = getenv("FREESPLAT_PREPROCESS"); return e && strcmp(e, "legacy") == 0; }();
    return on;
}
constexpr int kMaxViewsInFlight = 16;   // key areas (scratch slots) a multi-view call may use at once
// FREESPLAT_RASTER_BATCH: views per projection launch of a multi-view call (default 16 = all views of a call in flight).  The blends of one batch run on the
// side streams while the projection of the next batch runs on the main stream.
static int raster_batch()
{
    static const int n = [] {
        const char* e = getenv("FREESPLAT_RASTER_BATCH");
        const int x = e ? atoi(e) : 16;
        return x < 1 ? 1 : (x > kMaxViewsInFlight ? kMaxViewsInFlight : x);
    }();
    return n;
}

static int check_forward_args(const fs_raster_dims& d, const float* means3D, const float* cov3D, const float* shs,
                              const float* colors_precomp, const float* opacities, const int32_t* radii, int64_t cap)
{
    if (d.N < 0 || d.H <= 0 || d.W <= 0 || cap < 1) return FS_ERR_INVALID_ARG;
    if (d.N > 0) {  // an empty Gaussian set renders the background; its arrays may be NULL
        if (!means3D || !cov3D || !opacities || !radii) return FS_ERR_INVALID_ARG;
        if ((shs == nullptr) == (colors_precomp == nullptr)) return FS_ERR_INVALID_ARG;
    }
    if (shs && (d.sh_degree < 0 || d.sh_degree > 3 || (d.sh_degree + 1) * (d.sh_degree + 1) > d.M))
        return FS_ERR_UNSUPPORTED;
    const int gx = (d.W + kTile - 1) / kTile, gy = (d.H + kTile - 1) / kTile;
    if (gx > 32767 || gy > 32767) return FS_ERR_UNSUPPORTED;  // tile coordinates travel as 15-bit fields (ushort4 rects, packed shuffles)
    if (d.N > (1 << 28)) return FS_ERR_UNSUPPORTED;  // list entries are (id << 4 | quadrant mask)
    if (cap > 0xFFFFFFFFll) return FS_ERR_UNSUPPORTED;  // tile ranges are 32-bit
    return FS_OK;
}

// Projection + binning + tile ranges of `nv` views in ONE launch set on `st`: a 2-D fill of the views' tile counters, the
// projection of every Gaussian into the nv cameras (preprocess_views_kernel) and one scan workgroup per view.  All per-view
// pointers address the FIRST of the nv views; view k's buffers lie k strides further.
static int launch_binning(const fs_raster_dims& d, int nv, const float* means3D, const float* cov3D, const float* shs,
                          const float* colors_precomp, const float* opacities, const float* viewmatrix,
                          const float* projmatrix, const float* campos, const float* tanfov, const float* scale,
                          void* geom, size_t geom_stride, void* binning, size_t binning_stride, void* scratch,
                          size_t scratch_stride, int64_t cap, int32_t* radii, uint32_t* counters, hipStream_t st)
{
    const int T = num_tiles(d.H, d.W);
    const uint32_t tile_cap = tile_capacity(cap, T);
    hipLaunchKernelGGL(zero_counts_kernel, dim3((T + 255) / 256, nv), dim3(256), 0, st, (char*)scratch, scratch_stride, T);
    FS_CHECK_LAUNCH("zero tile counts");
    if (d.N > 0) {
        const int M = shs ? d.M : 0;
        ViewBuffers vb;
        vb.geom = (char*)geom; vb.geom_stride = geom_stride;
        vb.scratch = (char*)scratch; vb.scratch_stride = scratch_stride;
        vb.radii = radii;
        const int ch = nv >= 4 ? 1 : (nv == 2 ? 2 : 4);
        const size_t lds = (size_t)(64 * ch * M * 3) * sizeof(float) + (size_t)4 * kWaveBins * sizeof(uint32_t)
                           + (size_t)(10 * 64 * ch) * sizeof(float);
        const dim3 grid((d.N + 64 * ch - 1) / (64 * ch));
        ScopedStage prof_(kStPreprocess, st, nv);
#define FS_LAUNCH_PRE(CH)                                                                                            \
        hipLaunchKernelGGL((preprocess_views_kernel<CH>), grid, dim3(256), lds, st, d, nv, means3D, cov3D, shs,      \
                           colors_precomp, opacities, viewmatrix, projmatrix, campos, tanfov, scale, vb, tile_cap)
        if (ch == 1) FS_LAUNCH_PRE(1); else if (ch == 2) FS_LAUNCH_PRE(2); else FS_LAUNCH_PRE(4);
#undef FS_LAUNCH_PRE
    }
    FS_CHECK_LAUNCH("preprocess_views");
    {
        ScopedStage prof_(kStTileScan, st, nv);
        hipLaunchKernelGGL(tile_scan_kernel, dim3(nv), dim3(1024), 0, st, (const char*)scratch, scratch_stride,
                           (char*)binning, binning_stride, T, counters, (unsigned long long)cap, tile_cap);
    }
    FS_CHECK_LAUNCH("tile_scan");
    return FS_OK;
}

// Sort + blend of ONE view whose keys are binned (launch_binning, or the legacy projection kernel).
static int launch_blend(const fs_raster_dims& d, const float* bg, void* geom, void* binning, void* image, void* scratch,
                        int64_t cap, float* out_color, float* out_depth, float* out_alpha, const uint32_t* counters,
                        hipStream_t st)
{
    const int gx = (d.W + kTile - 1) / kTile, gy = (d.H + kTile - 1) / kTile;
    const int T = gx * gy;
    const size_t P = (size_t)d.H * d.W;
    const GeomView g = geom_view(geom, d.N > 0 ? d.N : 1);
    uint32_t* offsets = (uint32_t*)binning;
    uint32_t* point_list = (uint32_t*)((char*)binning + binning_offsets_bytes(d.H, d.W));
    float* final_T = (float*)image;
    int32_t* n_contrib = (int32_t*)((char*)image + align_up(P * 4, 256));
    uint32_t* counts = (uint32_t*)scratch;
    unsigned long long* keys = (unsigned long long*)((char*)scratch + align_up((size_t)T * 4, 256));
    const uint32_t tile_cap = tile_capacity(cap, T);
    const int nblk = tile_grid_blocks(gx, gy);
    {
        ScopedStage prof_(kStRender, st);
#define FS_LAUNCH_RENDER(F, TR)                                                                                     \
        hipLaunchKernelGGL((sort_blend_kernel<F, TR>), dim3(nblk), dim3(256), 0, st, d.H, d.W, counts, offsets, tile_cap, \
                           keys, point_list, g.rec, bg, counters, out_color, out_depth, out_alpha, final_T, n_contrib)
        const bool track = !(d.flags & FS_RASTER_NO_BACKWARD_STATE);
        if (d.flags & FS_RASTER_FAST_EXP) { if (track) FS_LAUNCH_RENDER(true, true); else FS_LAUNCH_RENDER(true, false); }
        else { if (track) FS_LAUNCH_RENDER(false, true); else FS_LAUNCH_RENDER(false, false); }
#undef FS_LAUNCH_RENDER
    }
    FS_CHECK_LAUNCH("sort_blend");
    return FS_OK;
}

// rounds 3 - 5: projection + binning of one view by the per-view kernel, its own memset and single-workgroup scan
static int launch_binning_legacy(const fs_raster_dims& d, const float* means3D, const float* cov3D, const float* shs,
                                 const float* colors_precomp, const float* opacities, const float* viewmatrix,
                                 const float* projmatrix, const float* campos, const float* tanfov_dev, const float* scale_dev,
                                 void* geom, void* binning, void* scratch, int64_t cap, int32_t* radii, uint32_t* counters,
                                 hipStream_t st)
{
    const int T = num_tiles(d.H, d.W);
    GeomView g = geom_view(geom, d.N > 0 ? d.N : 1);
    uint32_t* counts = (uint32_t*)scratch;
    unsigned long long* keys = (unsigned long long*)((char*)scratch + align_up((size_t)T * 4, 256));
    const uint32_t tile_cap = tile_capacity(cap, T);
    if (hipMemsetAsync(counts, 0, (size_t)T * 4, st) != hipSuccess) {
        set_last_error("memset tile counts", hipGetLastError());
        return FS_ERR_LAUNCH;
    }
    const int M = shs ? d.M : 0;
    if (d.N > 0) {
        size_t lds = (size_t)(256 * M * 3) * sizeof(float);
        if (lds < (size_t)(16 + kBinLds) * 4) lds = (size_t)(16 + kBinLds) * 4;
        {
            ScopedStage prof_(kStPreprocess, st);
            hipLaunchKernelGGL(preprocess_kernel, dim3((d.N + 255) / 256), dim3(256), lds, st, d, means3D,
                               cov3D, shs, colors_precomp, opacities, viewmatrix, projmatrix, campos,
                               tanfov_dev, scale_dev, g, radii, counts, keys, tile_cap);
        }
        FS_CHECK_LAUNCH("preprocess");
    }
    {
        ScopedStage prof_(kStTileScan, st);
        hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, st, (const char*)counts, (size_t)0, (char*)binning,
                           (size_t)0, T, counters, (unsigned long long)cap, tile_cap);
    }
    FS_CHECK_LAUNCH("tile_scan");
    return FS_OK;
}
}  // namespace fs

FS_API int fs_raster_forward(const fs_raster_dims* dims, const float* means3D, const float* cov3D,
                             const float* shs, const float* colors_precomp, const float* opacities,
                             const float* bg, const float* viewmatrix, const float* projmatrix,
                             const float* campos, const float* tanfov_dev, const float* scale_dev,
                             void* geom, void* binning, void* image, void* scratch, int64_t cap, float* out_color, float* out_depth,
                             float* out_alpha, int32_t* radii, uint32_t* counters, void* stream_)
{
    if (!dims || !bg || !viewmatrix || !projmatrix || !campos || !geom || !binning || !image ||
        !scratch || !out_color || !out_depth || !out_alpha || !counters)
        return FS_ERR_INVALID_ARG;
    const fs_raster_dims d = *dims;
    int rc = check_forward_args(d, means3D, cov3D, shs, colors_precomp, opacities, radii, cap);
    if (rc != FS_OK) return rc;
    hipStream_t st = (hipStream_t)stream_;
    if (legacy_preprocess())
        rc = launch_binning_legacy(d, means3D, cov3D, shs, colors_precomp, opacities, viewmatrix, projmatrix, campos,
                                   tanfov_dev, scale_dev, geom, binning, scratch, cap, radii, counters, st);
    else
        rc = launch_binning(d, 1, means3D, cov3D, shs, colors_precomp, opacities, viewmatrix, projmatrix, campos, tanfov_dev,
                            scale_dev, geom, 0, binning, 0, scratch, 0, cap, radii, counters, st);
    if (rc != FS_OK) return rc;
    return launch_blend(d, bg, geom, binning, image, scratch, cap, out_color, out_depth, out_alpha, counters, st);
}

// ---- all views of one call in ONE host call (decoder path) ---------------------------------------------------
namespace fs {
constexpr int kMaxStreams = 8;
struct ForkJoin {  // cached events: fork `main` into the side streams, join them back
    hipEvent_t ready = nullptr, done[kMaxStreams] = {}, batch[kMaxViewsInFlight] = {};
    bool ok = false;
    ForkJoin()
    {
        ok = hipEventCreateWithFlags(&ready, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; i < kMaxStreams && ok; ++i) ok = hipEventCreateWithFlags(&done[i], hipEventDisableTiming) == hipSuccess;
        for (int i = 0; i < kMaxViewsInFlight && ok; ++i) ok = hipEventCreateWithFlags(&batch[i], hipEventDisableTiming) == hipSuccess;
    }
};
}  // namespace fs

FS_API int fs_raster_scratch_slots(int32_t v, int32_t n_streams)
{
    if (v < 0 || n_streams < 0) return FS_ERR_INVALID_ARG;
    if (fs::legacy_preprocess()) return n_streams > 1 ? (n_streams < v ? n_streams : (v > 0 ? v : 1)) : 1;
    return v < 1 ? 1 : (v < fs::kMaxViewsInFlight ? v : fs::kMaxViewsInFlight);
}

FS_API int fs_raster_forward_views(const fs_raster_dims* dims, int32_t v, const float* means3D, const float* cov3D,
                                   const float* shs, const float* colors_precomp, const float* opacities,
                                   const float* bg, const float* viewmatrix, const float* projmatrix,
                                   const float* campos, const float* tanfov, const float* scale, void* geom,
                                   void* binning, void* image, void* scratch, const size_t strides[4], int64_t cap,
                                   float* out_color, float* out_depth, float* out_alpha, int32_t* radii,
                                   uint32_t* counters, int32_t n_streams, void* const* streams, void* main_stream)
{
    if (!dims || v < 0 || !strides || n_streams < 0 || n_streams > fs::kMaxStreams || (n_streams > 0 && !streams))
        return FS_ERR_INVALID_ARG;
    if (v == 0) return FS_OK;
    if (!bg || !viewmatrix || !projmatrix || !campos || !geom || !binning || !image || !scratch || !out_color ||
        !out_depth || !out_alpha || !counters)
        return FS_ERR_INVALID_ARG;
    const fs_raster_dims d = *dims;
    int rc = check_forward_args(d, means3D, cov3D, shs, colors_precomp, opacities, radii, cap);
    if (rc != FS_OK) return rc;
    // events belong to the device that was current when they were created: one cached set per (thread, device)
    static thread_local fs::ForkJoin* fj_dev[64] = {};
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) { set_last_error("hipGetDevice", hipGetLastError()); return FS_ERR_LAUNCH; }
    if (!fj_dev[dev_]) fj_dev[dev_] = new fs::ForkJoin();
    fs::ForkJoin& fj = *fj_dev[dev_];
    const int ns = n_streams <= 1 ? 0 : (n_streams < v ? n_streams : v);
    hipStream_t main = (hipStream_t)main_stream;
    if (ns > 0 && !fj.ok) { set_last_error("event create", hipGetLastError()); return FS_ERR_LAUNCH; }
    const size_t P = (size_t)d.H * d.W;
    const bool legacy = fs::legacy_preprocess();
    auto join = [&]() -> int {   // main waits for everything queued on the side streams
        for (int s = 0; s < ns; ++s)
            if (hipEventRecord(fj.done[s], (hipStream_t)streams[s]) != hipSuccess ||
                hipStreamWaitEvent(main, fj.done[s], 0) != hipSuccess) {
                set_last_error("stream join", hipGetLastError());
                return FS_ERR_LAUNCH;
            }
        return FS_OK;
    };
    auto blend = [&](int i, void* scratch_i, hipStream_t st) {
        return fs::launch_blend(d, bg + 3 * (size_t)i, (char*)geom + strides[0] * i, (char*)binning + strides[1] * i,
                                (char*)image + strides[2] * i, scratch_i, cap, out_color + 3 * P * i, out_depth + P * i,
                                out_alpha + P * i, counters + 2 * (size_t)i, st);
    };
    if (legacy) {   // rounds 3 - 5: whole views alternate over the streams, one key area per stream
        if (ns > 0) {
            if (hipEventRecord(fj.ready, main) != hipSuccess) { set_last_error("event record", hipGetLastError()); return FS_ERR_LAUNCH; }
            for (int s = 0; s < ns; ++s)
                if (hipStreamWaitEvent((hipStream_t)streams[s], fj.ready, 0) != hipSuccess) {
                    set_last_error("stream wait", hipGetLastError());
                    return FS_ERR_LAUNCH;
                }
        }
        for (int i = 0; i < v && rc == FS_OK; ++i) {
            const int s = ns > 0 ? i % ns : 0;
            hipStream_t st = ns > 0 ? (hipStream_t)streams[s] : main;
            void* scratch_i = (char*)scratch + strides[3] * s;
            rc = fs::launch_binning_legacy(d, means3D, cov3D, shs, colors_precomp, opacities, viewmatrix + 16 * (size_t)i,
                                           projmatrix + 16 * (size_t)i, campos + 3 * (size_t)i,
                                           tanfov ? tanfov + 2 * (size_t)i : nullptr, scale ? scale + i : nullptr,
                                           (char*)geom + strides[0] * i, (char*)binning + strides[1] * i, scratch_i, cap,
                                           radii ? radii + (size_t)d.N * i : nullptr, counters + 2 * (size_t)i, st);
            if (rc == FS_OK) rc = blend(i, scratch_i, st);
        }
        const int jr = join();   // join even after a failed launch: never leave `main` unordered
        return rc != FS_OK ? rc : jr;
    }
    // One projection launch set per BATCH of views on `main`, the blends of the batch fan out over the side streams; `main`
    // goes straight on to the next batch's projection, which overlaps those blends.  View i's key area is scratch slot
    // i % kMaxViewsInFlight: after kMaxViewsInFlight views the streams are joined before the slots are reused.
    const int VB = fs::raster_batch();
    for (int c0 = 0; c0 < v && rc == FS_OK; c0 += fs::kMaxViewsInFlight) {
        const int c1 = c0 + fs::kMaxViewsInFlight < v ? c0 + fs::kMaxViewsInFlight : v;
        for (int b0 = c0, k = 0; b0 < c1 && rc == FS_OK; b0 += VB, ++k) {
            const int nb = b0 + VB < c1 ? VB : c1 - b0;
            char* const scratch_b = (char*)scratch + strides[3] * (size_t)(b0 - c0);
            rc = fs::launch_binning(d, nb, means3D, cov3D, shs, colors_precomp, opacities, viewmatrix + 16 * (size_t)b0,
                                    projmatrix + 16 * (size_t)b0, campos + 3 * (size_t)b0,
                                    tanfov ? tanfov + 2 * (size_t)b0 : nullptr, scale ? scale + b0 : nullptr,
                                    (char*)geom + strides[0] * b0, strides[0], (char*)binning + strides[1] * b0, strides[1],
                                    scratch_b, strides[3], cap, radii ? radii + (size_t)d.N * b0 : nullptr,
                                    counters + 2 * (size_t)b0, main);
            if (rc != FS_OK) break;
            if (ns > 0 && hipEventRecord(fj.batch[k], main) != hipSuccess) { set_last_error("event record", hipGetLastError()); rc = FS_ERR_LAUNCH; break; }
            for (int i = b0; i < b0 + nb && rc == FS_OK; ++i) {
                hipStream_t st = ns > 0 ? (hipStream_t)streams[i % ns] : main;
                if (ns > 0 && hipStreamWaitEvent(st, fj.batch[k], 0) != hipSuccess) { set_last_error("stream wait", hipGetLastError()); rc = FS_ERR_LAUNCH; break; }
                rc = blend(i, scratch_b + strides[3] * (size_t)(i - b0), st);
            }
        }
        const int jr = join();
        if (rc == FS_OK) rc = jr;
    }
    return rc;
}

#ifdef FS_PHASE_TRACE
// debug builds only: copy out (and clear) the phase timestamps of this translation unit's kernels
FS_API int fs_debug_phase_trace(unsigned long long* dst)
{
    static unsigned long long z[fs::kPtKernels * fs::kPtBlocks * fs::kPtSlots];
    if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(fs::g_phase_trace), sizeof(z)) != hipSuccess) return FS_ERR_LAUNCH;
    return hipMemcpyToSymbol(HIP_SYMBOL(fs::g_phase_trace), z, sizeof(z)) == hipSuccess ? FS_OK : FS_ERR_LAUNCH;
}
#endif


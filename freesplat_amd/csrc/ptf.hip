// ptf.hip -- the data-dependent core of one Pixel-wise Triplet Fusion fold step, MI355X (gfx950).
//
// Replaces the index-producing part of EncoderFreeSplat.fuse_gaussians
// (src/model/encoder/encoder_freesplat.py:443-482 and the ~fusion_mask selections of :508-519;
// exact semantics: SURVEY.md Appendix C): project the M global Gaussians into view i, z-buffer
// them per pixel (scatter amin), test depth consistency, pick the winners, and emit the three
// ordered index lists the fold needs:
//     keep_idx   ascending m that are NOT fused           (global[~mask])
//     fuse_idx   ascending m that ARE fused, fuse_pix their pixels of view i
//     append_pix ascending pixels of view i that start a new Gaussian (~fusion_mask)
// The reference does this with scatter_reduce_, two torch.isin passes over M-sized index tensors,
// boolean-mask indexing and four host syncs per view; here it is 6 short launches and no sync:
// atomicMin on the float bits (z > 0, so uint order == float order), byte flags, and a
// two-level ballot/prefix-sum compaction that preserves ascending order.
//
// Integer/index work => bit-exact against the oracle: compiled with -ffp-contract=off, the
// projection uses plain IEEE mul/add/div in a fixed order, rounding is round-half-to-even.
#include "fs_common.h"

namespace fs {

constexpr uint32_t kZInit = 0x461C4000u;  // bits of 10000.0f (encoder_freesplat.py:464)
constexpr int kScanBlock = 1024;          // elements per compaction workgroup (256 threads x 4)

__global__ __launch_bounds__(256) void ptf_fill_kernel(uint32_t* __restrict__ zbuf, int P)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p < P) zbuf[p] = kZInit;
}

// w2c = inverse(extrinsics_i) row-major [16]; kpix = {fx, fy, cx, cy} in pixels
// (every kernel below takes its element count either from the host (`M`) or, when `Mp` is non-null, from device memory:
//  fs_ptf_fold_step chains the fold steps of all views without a host sync, sizing grids for an upper bound)
__global__ __launch_bounds__(256) void ptf_project_kernel(int M, const int32_t* __restrict__ Mp, int h, int w,
                                                          const float* __restrict__ xyz,
                                                          const float* __restrict__ w2c,
                                                          const float* __restrict__ kpix,
                                                          int32_t* __restrict__ pix_of,
                                                          uint32_t* __restrict__ zbits_of,
                                                          uint32_t* __restrict__ zbuf)
{
    if (Mp) M = *Mp;
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const float x = xyz[3 * (size_t)m], y = xyz[3 * (size_t)m + 1], z = xyz[3 * (size_t)m + 2];
    const float cx = ((w2c[0] * x + w2c[1] * y) + w2c[2] * z) + w2c[3];
    const float cy = ((w2c[4] * x + w2c[5] * y) + w2c[6] * z) + w2c[7];
    const float cz = ((w2c[8] * x + w2c[9] * y) + w2c[10] * z) + w2c[11];
    const float px = (cx / cz) * kpix[0] + kpix[2];
    const float py = (cy / cz) * kpix[1] + kpix[3];
    const float col = __builtin_rintf(px), row = __builtin_rintf(py);  // torch.round: half to even
    const bool valid = row >= 0.0f && row < (float)h && col >= 0.0f && col < (float)w && cz > 0.0f;
    int32_t pix = -1;
    const uint32_t zb = __float_as_uint(cz);
    if (valid) {
        pix = (int)row * w + (int)col;
        atomicMin(&zbuf[pix], zb);
    }
    pix_of[m] = pix;
    zbits_of[m] = zb;
}

__device__ __forceinline__ bool fusion_mask(uint32_t zb, float d, float depth_thres)
{
    // |zbuf - d_i| < max(0.05 d_i, depth_thres)   (encoder_freesplat.py:468)
    return fabsf(__uint_as_float(zb) - d) < fmaxf(d * 0.05f, depth_thres);
}

// Flags AND their per-block counts in one launch (rounds 1 - 3: ptf_flags_kernel, one thread per element, then
// ptf_count_kernel re-reading the bytes): workgroup b < nbM decides `win` for the 1024 Gaussians [1024 b, +1024), workgroup
// nbM + b decides `app` for 1024 pixels; a thread owns 4 consecutive elements (one 4-byte store of its flags).
__global__ __launch_bounds__(256) void ptf_flags_count_kernel(int M, const int32_t* __restrict__ Mp, int P, int nbM,
                                                              const int32_t* __restrict__ pix_of,
                                                              const uint32_t* __restrict__ zbits_of,
                                                              const uint32_t* __restrict__ zbuf,
                                                              const float* __restrict__ depth_i, float depth_thres,
                                                              uint8_t* __restrict__ win, uint8_t* __restrict__ app,
                                                              uint32_t* __restrict__ block_counts)
{
    __shared__ uint32_t s_w[4];
    if (Mp) M = *Mp;
    const bool second = (int)blockIdx.x >= nbM;
    const int n = second ? P : M;
    const int blk = second ? blockIdx.x - nbM : blockIdx.x;
    const int base = blk * kScanBlock + threadIdx.x * 4;
    uint32_t v = 0;
    // the 16-byte loads below need 16-byte aligned arrays: depth_i = depths + i * P is only 4-byte aligned when h * w is not a
    // multiple of 4 -- such a call takes the scalar path (workgroup-uniform)
    const bool al16 = second ? ((((uintptr_t)zbuf | (uintptr_t)depth_i) & 15) == 0)
                             : ((((uintptr_t)pix_of | (uintptr_t)zbits_of) & 15) == 0);
    if (base + 3 < n && al16) {
        // a full quad: the four elements' loads are issued together (a loop that may leave early serialises the dependent
        // gathers pix_of -> zbuf / depth_i: 192 workgroups of latency instead of 768)
        if (second) {
            const uint4 zb = *(const uint4*)(zbuf + base);
            const float4 dd = *(const float4*)(depth_i + base);
            v = (fusion_mask(zb.x, dd.x, depth_thres) ? 0u : 1u) | (fusion_mask(zb.y, dd.y, depth_thres) ? 0u : 1u << 8) |
                (fusion_mask(zb.z, dd.z, depth_thres) ? 0u : 1u << 16) | (fusion_mask(zb.w, dd.w, depth_thres) ? 0u : 1u << 24);
        } else {
            const int4 px = *(const int4*)(pix_of + base);
            const uint4 zo = *(const uint4*)(zbits_of + base);
            const int pix[4] = {px.x, px.y, px.z, px.w};
            const uint32_t own[4] = {zo.x, zo.y, zo.z, zo.w};
            uint32_t zb[4];
            float dd[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int q = pix[k] >= 0 ? pix[k] : 0;
                zb[k] = zbuf[q];
                dd[k] = depth_i[q];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                v |= ((pix[k] >= 0 && zb[k] == own[k] && fusion_mask(zb[k], dd[k], depth_thres)) ? 1u : 0u) << (8 * k);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = base + k;
            if (e >= n) break;
            bool on;
            if (second) {
                on = !fusion_mask(zbuf[e], depth_i[e], depth_thres);
            } else {
                const int pix = pix_of[e];
                on = false;
                if (pix >= 0) {
                    const uint32_t zb = zbuf[pix];
                    on = zb == zbits_of[e] && fusion_mask(zb, depth_i[pix], depth_thres);
                }
            }
            v |= (on ? 1u : 0u) << (8 * k);
        }
    }
    uint8_t* f = second ? app : win;
    if (base + 3 < n && ((((uintptr_t)(f + base)) & 3) == 0)) {
        *(uint32_t*)(f + base) = v;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (base + k < n) f[base + k] = (uint8_t)((v >> (8 * k)) & 1u);
    }
    uint32_t c = __popc(v);
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) c += __shfl_xor(c, s, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// ---- order-preserving compaction -----------------------------------------------------------------
__device__ __forceinline__ uint32_t load4(const uint8_t* __restrict__ f, int base, int n)
{
    // 4 consecutive byte flags (0/1) starting at `base`, zero past n
    uint32_t v = 0;
    if (base + 3 < n && ((((uintptr_t)(f + base)) & 3) == 0)) {
        v = *(const uint32_t*)(f + base);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (base + k < n) v |= (uint32_t)f[base + k] << (8 * k);
    }
    return v;
}

// exclusive scan of the two block-count ranges (one workgroup); counts = {n_keep, n_fuse, n_append}
__global__ __launch_bounds__(1024) void ptf_scan_blocks_kernel(int M, const int32_t* __restrict__ Mp, int nbM, int nbP,
                                                               uint32_t* __restrict__ block_counts,
                                                               int32_t* __restrict__ counts)
{
    __shared__ uint32_t part[1024];
    if (Mp) M = *Mp;
    for (int range = 0; range < 2; ++range) {
        uint32_t* a = block_counts + (range ? nbM : 0);
        const int n = range ? nbP : nbM;
        const int t = threadIdx.x;
        const int per = (n + 1023) / 1024;
        const int lo = min(n, t * per), hi = min(n, lo + per);
        uint32_t s = 0;
        for (int k = lo; k < hi; ++k) s += a[k];
        part[t] = s;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const uint32_t v = (t >= off) ? part[t - off] : 0u;
            __syncthreads();
            part[t] += v;
            __syncthreads();
        }
        uint32_t run = part[t] - s;
        for (int k = lo; k < hi; ++k) {
            const uint32_t c = a[k];
            a[k] = run;
            run += c;
        }
        if (t == 1023) {
            if (range == 0) { counts[1] = (int32_t)part[1023]; counts[0] = M - (int32_t)part[1023]; }
            else { counts[2] = (int32_t)part[1023]; counts[3] = counts[0] + counts[1] + counts[2]; }  // [3]: size of the next state
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void ptf_emit_kernel(int M, const int32_t* __restrict__ Mp, int P, int nbM,
                                                       const uint8_t* __restrict__ win,
                                                       const uint8_t* __restrict__ app,
                                                       const int32_t* __restrict__ pix_of,
                                                       const uint32_t* __restrict__ block_offsets,
                                                       long long* __restrict__ keep_idx,
                                                       long long* __restrict__ fuse_idx,
                                                       long long* __restrict__ fuse_pix,
                                                       long long* __restrict__ append_pix, uint32_t* __restrict__ zbuf)
{
    __shared__ uint32_t s_w[4];
    if (Mp) M = *Mp;
    const bool second = (int)blockIdx.x >= nbM;
    const uint8_t* f = second ? app : win;
    const int n = second ? P : M;
    const int blk = second ? blockIdx.x - nbM : blockIdx.x;
    const int base = blk * kScanBlock + threadIdx.x * 4;
    const uint32_t v = load4(f, base, n);
    if (second) {
        // the z-buffer's last reader (the flags) is done: leave it cleared for the next fold step of this scratch, which
        // then needs no fill launch of its own (fs_ptf_fold)
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (base + k < n) zbuf[base + k] = 0xFFFFFFFFu;
    }
    const uint32_t c = __popc(v);
    // exclusive scan of c over the 256 threads: wave inclusive scan, then 4 wave totals via LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = c;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const uint32_t o = __shfl_up(inc, s, 64);
        if (lane >= s) inc += o;
    }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint32_t wave_off = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < wave) wave_off += s_w[k];
    uint32_t rank = block_offsets[blockIdx.x] + wave_off + inc - c;  // set flags before `base`
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int e = base + k;
        if (e >= n) break;
        const bool on = (v >> (8 * k)) & 1u;
        if (second) {
            if (on) append_pix[rank] = e;
        } else if (on) {
            fuse_idx[rank] = e;
            fuse_pix[rank] = pix_of[e];
        } else {
            keep_idx[(uint32_t)e - rank] = e;  // rank = number of fused entries before e
        }
        rank += on ? 1u : 0u;
    }
}

// ------------------------------------------------------------------------------------------
// Inference-path data movement of one fold step (encoder_freesplat.py:485-519), fused:
//   ptf_gru_inputs  builds the GRU's concatenated input rows [hid | he | x | xe] (networks.py:201-206,
//                   positional encodings of encoder_freesplat.py:62-77, 485-486) straight from the
//                   state / view arrays through the index lists -- replaces ~12 index_select / cat /
//                   sin / cos launches;
//   ptf_write_state writes the next global state in its final order
//                   [kept (copied) | fused (GRU output + density-weighted blends) | appended pixels]
//                   -- replaces ~40 boolean-mask / cat launches and O(M) temporaries per field.
// ------------------------------------------------------------------------------------------
// one 16-lane group per fused pair: lanes 0..15 move the two 64-float latents as float4 and share the 24 encodings
__global__ __launch_bounds__(256) void ptf_gru_inputs_kernel(int n_fuse, const int32_t* __restrict__ counts,
                                                            const long long* __restrict__ fuse_idx,
                                                            const long long* __restrict__ fuse_pix,
                                                            const float* __restrict__ G, const float* __restrict__ R,
                                                            const float* __restrict__ O, const float* __restrict__ g_i,
                                                            const float* __restrict__ rho_i,
                                                            const float* __restrict__ om_i, float* __restrict__ cat)
{
    if (counts) n_fuse = counts[1];
    const int t = blockIdx.x * 16 + (threadIdx.x >> 4), c = threadIdx.x & 15;
    if (t >= n_fuse) return;
    const long long m = fuse_idx[t], p = fuse_pix[t];
    float* row = cat + (size_t)t * 176;
    ((float4*)row)[c] = ((const float4*)(G + m * 64))[c];                 // hid      [0,64)
    ((float4*)(row + 88))[c] = ((const float4*)(g_i + p * 64))[c];        // x        [88,152)
    // he = PE(rho_i[p], O[m]) [64,88) :486, xe = PE(R[m], om_i[p]) [152,176) :485 -- four scalars x six octaves, spread
    // over the 16 lanes (lane c: scalar c >> 2, octaves c & 3 and (c & 3) + 4), the same arithmetic per term as
    // fs_common.h:pos_enc2 (which the fused GRU kernel uses), one 8-byte store per (sin, cos) pair
    const int sc = c >> 2, kq = c & 3;
    const float v = sc == 0 ? rho_i[p] : sc == 1 ? O[m] : sc == 2 ? R[m] : om_i[p];
    float* out = row + (sc < 2 ? 64 : 152) + 12 * (sc & 1);
    float rv, ev;
    rev2pi(v, rv, ev);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = kq + 4 * j;
        if (k < 6) {
            const float f = (float)(1 << k);
            const float tt = rv * f;                                       // (exact)
            const float ph = (tt - rintf(tt)) + ev * f;
            *(float2*)(out + 2 * k) = make_float2(__builtin_amdgcn_sinf(ph), __builtin_amdgcn_cosf(ph));
        }
    }
}

struct PtfState { float *G, *X, *R, *O, *E, *D; };

// SPLIT: the fused rows whose latent row the GRU wrote itself (`fused` == NULL: the fold) are handled by ptf_write_state_fused_kernel,
// four lanes per row -- they move ~300 bytes, not a 256-byte latent row; this launch then walks the kept and the appended rows only.
template <bool SPLIT>
__global__ __launch_bounds__(256) void ptf_write_state_kernel(
    int n_keep, int n_fuse, int n_app, const int32_t* __restrict__ counts, const long long* __restrict__ keep_idx,
    const long long* __restrict__ fuse_idx,
    const long long* __restrict__ fuse_pix, const long long* __restrict__ app_pix, PtfState s, const float* __restrict__ g_i,
    const float* __restrict__ x_i, const float* __restrict__ rho_i, const float* __restrict__ om_i,
    const float* __restrict__ d_i, const float* __restrict__ E_i, const float* __restrict__ fused, PtfState o)
{
    if (counts) { n_keep = counts[0]; n_fuse = counts[1]; n_app = counts[2]; }
    const int c = threadIdx.x & 15;
    const int n_out = n_keep + n_fuse + n_app;
    // (grid-stride: with the counts on the device the fold sizes its grid for the worst case -- 235 k workgroups at 968x1296 of
    //  which 34 k have rows -- so the grid is capped and the workgroups walk)
    if constexpr (SPLIT) {
        // kept + appended rows only, FOUR rows per 16-lane group in flight: index loads, then the rows' loads, then the stores -- a
        // group that moved one row per trip exposed the index -> row -> store chain once per 688 bytes
        constexpr int U = 4;
        const long long n_ka = (long long)n_keep + n_app, G = (long long)gridDim.x * 16;
        for (long long r0 = (long long)blockIdx.x * 16 + (threadIdx.x >> 4); r0 < n_ka; r0 += U * G) {
            long long src[U];
            bool kept[U], live[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long r = r0 + u * G;
                live[u] = r < n_ka; kept[u] = r < n_keep;
                src[u] = 0;
                if (live[u]) src[u] = kept[u] ? keep_idx[r] : app_pix[r - n_keep];
            }
            float4 g[U], e[U];
            float x[U], q[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long m = src[u];
                g[u] = e[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); x[u] = q[u] = 0.0f;
                if (!live[u]) continue;       // (no loads for rows that do not exist: thousands of groups reading row 0 make one L2 channel the bottleneck)
                g[u] = ((const float4*)((kept[u] ? s.G : g_i) + m * 64))[c];
                if (c < 4) e[u] = ((const float4*)(kept[u] ? s.E + m * 16 : E_i))[c];
                if (c >= 4 && c < 7) x[u] = (kept[u] ? s.X : x_i)[3 * m + (c - 4)];
                if (c >= 7 && c < 10) q[u] = (c == 7 ? (kept[u] ? s.R : rho_i) : (c == 8 ? (kept[u] ? s.O : om_i) : (kept[u] ? s.D : d_i)))[m];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!live[u]) continue;
                const long long r = r0 + u * G;
                const size_t row = (size_t)(r < n_keep ? r : r + n_fuse);
                ((float4*)(o.G + row * 64))[c] = g[u];
                if (c < 4) ((float4*)(o.E + row * 16))[c] = e[u];
                if (c >= 4 && c < 7) o.X[3 * row + (c - 4)] = x[u];
                if (c == 7) o.R[row] = q[u];
                if (c == 8) o.O[row] = q[u];
                if (c == 9) o.D[row] = q[u];
            }
        }
        return;
    }
    for (long long r0 = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);; r0 += (long long)gridDim.x * 16) {
    const int row = (int)r0;
    if (r0 >= n_out) return;
    float4* oG = (float4*)(o.G + (size_t)row * 64);
    if (row < n_keep) {                                                     // global[~mask]          :492
        const long long m = keep_idx[row];
        oG[c] = ((const float4*)(s.G + m * 64))[c];
        if (c < 4) ((float4*)(o.E + (size_t)row * 16))[c] = ((const float4*)(s.E + m * 16))[c];
        if (c == 4) { o.X[3 * (size_t)row] = s.X[3 * m]; o.X[3 * (size_t)row + 1] = s.X[3 * m + 1]; o.X[3 * (size_t)row + 2] = s.X[3 * m + 2]; }
        if (c == 5) { o.R[row] = s.R[m]; o.O[row] = s.O[m]; o.D[row] = s.D[m]; }
    } else if (row < n_keep + n_fuse) {                                     // fused entries          :493-506
        const int t = row - n_keep;
        const long long m = fuse_idx[t], p = fuse_pix[t];
        if (fused) oG[c] = ((const float4*)(fused + (size_t)t * 64))[c];      // (NULL: the GRU kernel wrote the row itself)
        const float w0 = s.R[m], w1 = rho_i[p], ws = w0 + w1;
        if (c < 4) {
            const float4 a = ((const float4*)(s.E + m * 16))[c], b = ((const float4*)E_i)[c];
            ((float4*)(o.E + (size_t)row * 16))[c] = make_float4((a.x * w0 + b.x * w1) / ws, (a.y * w0 + b.y * w1) / ws,
                                                                  (a.z * w0 + b.z * w1) / ws, (a.w * w0 + b.w * w1) / ws);
        }
        if (c == 4) {
#pragma unroll
            for (int k = 0; k < 3; ++k) o.X[3 * (size_t)row + k] = (s.X[3 * m + k] * w0 + x_i[3 * p + k] * w1) / ws;
        }
        if (c == 5) { o.R[row] = ws; o.O[row] = s.O[m] + om_i[p]; o.D[row] = (s.D[m] * w0 + d_i[p] * w1) / ws; }
    } else {                                                                // ~fusion_mask pixels    :508-519
        const long long p = app_pix[row - n_keep - n_fuse];
        oG[c] = ((const float4*)(g_i + p * 64))[c];
        if (c < 4) ((float4*)(o.E + (size_t)row * 16))[c] = ((const float4*)E_i)[c];
        if (c == 4) { o.X[3 * (size_t)row] = x_i[3 * p]; o.X[3 * (size_t)row + 1] = x_i[3 * p + 1]; o.X[3 * (size_t)row + 2] = x_i[3 * p + 2]; }
        if (c == 5) { o.R[row] = rho_i[p]; o.O[row] = om_i[p]; o.D[row] = d_i[p]; }
    }
    }
}

// The fused rows of ptf_write_state_kernel when the GRU has already written their latent rows (the fold): four lanes per row, lane q =
// float4 q of the blended extrinsics; q = 0 also the position, q = 1 the scalars.  Same expressions as the 16-lane form.
__global__ __launch_bounds__(256) void ptf_write_state_fused_kernel(
    int n_keep, int n_fuse, const int32_t* __restrict__ counts, const long long* __restrict__ fuse_idx,
    const long long* __restrict__ fuse_pix, PtfState s, const float* __restrict__ x_i, const float* __restrict__ rho_i,
    const float* __restrict__ om_i, const float* __restrict__ d_i, const float* __restrict__ E_i, PtfState o)
{
    if (counts) { n_keep = counts[0]; n_fuse = counts[1]; }
    const int c = threadIdx.x & 3;
    for (int t = blockIdx.x * 64 + (threadIdx.x >> 2); t < n_fuse; t += gridDim.x * 64) {       // (grid-stride: capped grid)
        const size_t row = (size_t)n_keep + t;
        const long long m = fuse_idx[t], p = fuse_pix[t];
        const float w0 = s.R[m], w1 = rho_i[p], ws = w0 + w1;
        {
            const float4 a = ((const float4*)(s.E + m * 16))[c], b = ((const float4*)E_i)[c];
            ((float4*)(o.E + row * 16))[c] = make_float4((a.x * w0 + b.x * w1) / ws, (a.y * w0 + b.y * w1) / ws,
                                                          (a.z * w0 + b.z * w1) / ws, (a.w * w0 + b.w * w1) / ws);
        }
        if (c == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) o.X[3 * row + k] = (s.X[3 * m + k] * w0 + x_i[3 * p + k] * w1) / ws;
        }
        if (c == 1) { o.R[row] = ws; o.O[row] = s.O[m] + om_i[p]; o.D[row] = (s.D[m] * w0 + d_i[p] * w1) / ws; }
    }
}

// ------------------------------------------------------------------------------------------
// Training path: backward of one fold step's data movement (the differentiable part of
// encoder_freesplat.py:485-519 around the GRU).  Forward of a step:
//   out = [ in[keep] | fused rows | view rows[app] ],   fused row t (m = fuse[t], p = fpix[t], w0 = R[m], w1 = rho_i[p]):
//     G = GRU(cat[t]);  X, E, D = (in * w0 + view * w1) / (w0 + w1);  R = w0 + w1;  O = O[m] + om_i[p]
// ptf_write_state_bwd turns the gradient of `out` into the gradient of `in` (every in-row is kept or fused exactly
// once: plain stores) and of the view's arrays (a pixel may be fused by several tied Gaussians: float atomics into
// zero-initialised arrays); the GRU rows' gradient is the contiguous block d_out.G[n_keep : n_keep+n_fuse].
// ptf_gru_inputs_bwd scatters the gradient of the GRU's concatenated input rows [hid | he | x | xe] back through
// the gather and the positional encodings (he = PE(rho_i[p], O[m]), xe = PE(R[m], om_i[p]); :62-77, 485-486).
// ------------------------------------------------------------------------------------------
struct PtfGrad { float *G, *X, *R, *O, *E, *D; };   // any member may be NULL (no gradient for that field)

// SPLIT: the fused rows are handled by ptf_write_state_bwd_fused_kernel (4 lanes per row instead of 16: a fused row moves ~300 bytes,
// not a 256-byte latent row, and 10 of its 16 lanes had nothing to do) -- this launch then walks the kept and the appended rows only.
template <bool SPLIT>
__global__ __launch_bounds__(256) void ptf_write_state_bwd_kernel(
    int n_keep, int n_fuse, int n_app, const long long* __restrict__ keep_idx, const long long* __restrict__ fuse_idx,
    const long long* __restrict__ fuse_pix, const long long* __restrict__ app_pix, PtfState s,
    const float* __restrict__ x_i, const float* __restrict__ rho_i, const float* __restrict__ d_i,
    const float* __restrict__ E_i, PtfGrad go, PtfGrad gs, float* __restrict__ g_lat_i, float* __restrict__ g_x_i,
    float* __restrict__ g_rho_i, float* __restrict__ g_om_i, float* __restrict__ g_d_i)
{
    const int c = threadIdx.x & 15;
    const int n_out = n_keep + n_fuse + n_app;
    const float4 z4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if constexpr (SPLIT) {
        // kept + appended rows only (the grid covers a quarter of them): FOUR rows per 16-lane group in flight -- indices, then every
        // load (the out row's gradient; an appended pixel's accumulated values), then the stores.  Lane c: float4 c of the latent row;
        // lanes 0 - 3 the extrinsics, 4 - 6 the position, 7 - 9 the scalars R, O, D.
        constexpr int U = 4;
        const long long n_ka = (long long)n_keep + n_app, G = (long long)gridDim.x * 16;
        const long long r0 = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
        long long dst[U];
        bool kept[U], live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long r = r0 + u * G;
            live[u] = r < n_ka; kept[u] = r < n_keep;
            dst[u] = 0;
            if (live[u]) dst[u] = kept[u] ? keep_idx[r] : app_pix[r - n_keep];
        }
        float4 g[U], e[U], acc[U];
        float x[U], q[U], ax[U], aq[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            g[u] = e[u] = acc[u] = z4; x[u] = q[u] = ax[u] = aq[u] = 0.0f;
            if (!live[u]) continue;            // (no loads for rows that do not exist)
            const long long r = r0 + u * G;
            const size_t row = (size_t)(kept[u] ? r : r + n_fuse);
            const long long m = dst[u];
            if (go.G) g[u] = ((const float4*)(go.G + row * 64))[c];
            if (c < 4 && go.E && kept[u]) e[u] = ((const float4*)(go.E + row * 16))[c];
            if (c >= 4 && c < 7 && go.X) x[u] = go.X[3 * row + (c - 4)];
            if (c >= 7 && c < 10) { const float* gp = c == 7 ? go.R : (c == 8 ? go.O : go.D); if (gp) q[u] = gp[row]; }
            if (!kept[u]) {                    // an appended pixel is appended once and never fused: plain read-modify-write
                acc[u] = ((const float4*)(g_lat_i + m * 64))[c];
                if (c >= 4 && c < 7) ax[u] = g_x_i[3 * m + (c - 4)];
                if (c >= 7 && c < 10) aq[u] = (c == 7 ? g_rho_i : (c == 8 ? g_om_i : g_d_i))[m];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!live[u]) continue;
            const long long m = dst[u];
            if (kept[u]) {
                ((float4*)(gs.G + m * 64))[c] = g[u];
                if (c < 4) ((float4*)(gs.E + m * 16))[c] = e[u];
                if (c >= 4 && c < 7) gs.X[3 * m + (c - 4)] = x[u];
                if (c == 7) gs.R[m] = q[u];
                if (c == 8) gs.O[m] = q[u];
                if (c == 9) gs.D[m] = q[u];
            } else {
                ((float4*)(g_lat_i + m * 64))[c] = make_float4(acc[u].x + g[u].x, acc[u].y + g[u].y, acc[u].z + g[u].z, acc[u].w + g[u].w);
                if (c >= 4 && c < 7 && go.X) g_x_i[3 * m + (c - 4)] = ax[u] + x[u];
                if (c == 7 && go.R) g_rho_i[m] = aq[u] + q[u];
                if (c == 8 && go.O) g_om_i[m] = aq[u] + q[u];
                if (c == 9 && go.D) g_d_i[m] = aq[u] + q[u];
            }
        }
        return;
    }
    const int row = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (row >= n_out) return;
    const float4 dG = go.G ? ((const float4*)(go.G + (size_t)row * 64))[c] : z4;
    if (row < n_keep) {
        const long long m = keep_idx[row];
        ((float4*)(gs.G + m * 64))[c] = dG;
        if (c < 4) ((float4*)(gs.E + m * 16))[c] = go.E ? ((const float4*)(go.E + (size_t)row * 16))[c] : z4;
        if (c == 4) {
#pragma unroll
            for (int k = 0; k < 3; ++k) gs.X[3 * m + k] = go.X ? go.X[3 * (size_t)row + k] : 0.0f;
        }
        if (c == 5) {
            gs.R[m] = go.R ? go.R[row] : 0.0f;
            gs.O[m] = go.O ? go.O[row] : 0.0f;
            gs.D[m] = go.D ? go.D[row] : 0.0f;
        }
    } else if (row < n_keep + n_fuse) {
        const int t = row - n_keep;
        const long long m = fuse_idx[t], p = fuse_pix[t];
        const float w0 = s.R[m], w1 = rho_i[p], ws = w0 + w1, inv = 1.0f / ws;
        // (the latent row's gradient goes through the GRU: ptf_gru_inputs_bwd writes gs.G[m])
        float dw0 = 0.0f, dw1 = 0.0f;  // partial sums of this lane; reduced over the 16-lane group below
        if (c < 4) {
            const float4 a = ((const float4*)(s.E + m * 16))[c], b = ((const float4*)E_i)[c];
            const float4 g = go.E ? ((const float4*)(go.E + (size_t)row * 16))[c] : z4;
            const float ox = (a.x * w0 + b.x * w1) * inv, oy = (a.y * w0 + b.y * w1) * inv;
            const float oz = (a.z * w0 + b.z * w1) * inv, ow = (a.w * w0 + b.w * w1) * inv;
            ((float4*)(gs.E + m * 16))[c] = make_float4(g.x * w0 * inv, g.y * w0 * inv, g.z * w0 * inv, g.w * w0 * inv);
            dw0 += (g.x * (a.x - ox) + g.y * (a.y - oy) + g.z * (a.z - oz) + g.w * (a.w - ow)) * inv;
            dw1 += (g.x * (b.x - ox) + g.y * (b.y - oy) + g.z * (b.z - oz) + g.w * (b.w - ow)) * inv;
        }
        if (c == 4) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float a = s.X[3 * m + k], b = x_i[3 * p + k], g = go.X ? go.X[3 * (size_t)row + k] : 0.0f;
                const float o = (a * w0 + b * w1) * inv;
                gs.X[3 * m + k] = g * w0 * inv;
                atomicAdd(&g_x_i[3 * p + k], g * w1 * inv);
                dw0 += g * (a - o) * inv;
                dw1 += g * (b - o) * inv;
            }
        }
        if (c == 5) {
            const float a = s.D[m], b = d_i[p], g = go.D ? go.D[row] : 0.0f;
            const float o = (a * w0 + b * w1) * inv;
            gs.D[m] = g * w0 * inv;
            atomicAdd(&g_d_i[p], g * w1 * inv);
            dw0 += g * (a - o) * inv;
            dw1 += g * (b - o) * inv;
            const float gR = go.R ? go.R[row] : 0.0f, gO = go.O ? go.O[row] : 0.0f;
            dw0 += gR; dw1 += gR;          // R_out = w0 + w1
            gs.O[m] = gO;                  // O_out = O[m] + om_i[p]
            atomicAdd(&g_om_i[p], gO);
        }
        // sum dw0 / dw1 over lanes 0..5 of the 16-lane group (xor shuffles stay inside the group)
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) {
            dw0 += __shfl_xor(dw0, d, 64);
            dw1 += __shfl_xor(dw1, d, 64);
        }
        if (c == 0) {
            gs.R[m] = dw0;                 // (+ the positional-encoding term: ptf_gru_inputs_bwd adds it)
            atomicAdd(&g_rho_i[p], dw1);
        }
    } else {
        const long long p = app_pix[row - n_keep - n_fuse];
        float4* q = (float4*)(g_lat_i + p * 64) + c;   // an appended pixel is appended once and never fused
        const float4 v = *q;
        *q = make_float4(v.x + dG.x, v.y + dG.y, v.z + dG.z, v.w + dG.w);
        if (c == 4 && go.X) {
#pragma unroll
            for (int k = 0; k < 3; ++k) g_x_i[3 * p + k] += go.X[3 * (size_t)row + k];
        }
        if (c == 5) {
            if (go.R) g_rho_i[p] += go.R[row];
            if (go.O) g_om_i[p] += go.O[row];
            if (go.D) g_d_i[p] += go.D[row];
        }
    }
}

// The fused rows of ptf_write_state_bwd_kernel, four lanes per row (lane q: float4 q of the 16 extrinsics floats; q = 0 also the
// position, q = 1 the depth / density / weight scalars; dw0 / dw1 summed over the four lanes) -- same arithmetic, same order per term.
__global__ __launch_bounds__(256) void ptf_write_state_bwd_fused_kernel(
    int n_keep, int n_fuse, const long long* __restrict__ fuse_idx, const long long* __restrict__ fuse_pix, PtfState s,
    const float* __restrict__ x_i, const float* __restrict__ rho_i, const float* __restrict__ d_i,
    const float* __restrict__ E_i, PtfGrad go, PtfGrad gs, float* __restrict__ g_x_i,
    float* __restrict__ g_rho_i, float* __restrict__ g_om_i, float* __restrict__ g_d_i)
{
    const int t = blockIdx.x * 64 + (threadIdx.x >> 2), c = threadIdx.x & 3;
    if (t >= n_fuse) return;                  // (whole quads leave together: the shuffles below stay inside a quad)
    const int row = n_keep + t;
    const float4 z4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const long long m = fuse_idx[t], p = fuse_pix[t];
    const float w0 = s.R[m], w1 = rho_i[p], ws = w0 + w1, inv = 1.0f / ws;
    float dw0 = 0.0f, dw1 = 0.0f;
    {
        const float4 a = ((const float4*)(s.E + m * 16))[c], b = ((const float4*)E_i)[c];
        const float4 g = go.E ? ((const float4*)(go.E + (size_t)row * 16))[c] : z4;
        const float ox = (a.x * w0 + b.x * w1) * inv, oy = (a.y * w0 + b.y * w1) * inv;
        const float oz = (a.z * w0 + b.z * w1) * inv, ow = (a.w * w0 + b.w * w1) * inv;
        ((float4*)(gs.E + m * 16))[c] = make_float4(g.x * w0 * inv, g.y * w0 * inv, g.z * w0 * inv, g.w * w0 * inv);
        dw0 += (g.x * (a.x - ox) + g.y * (a.y - oy) + g.z * (a.z - oz) + g.w * (a.w - ow)) * inv;
        dw1 += (g.x * (b.x - ox) + g.y * (b.y - oy) + g.z * (b.z - oz) + g.w * (b.w - ow)) * inv;
    }
    if (c == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float a = s.X[3 * m + k], b = x_i[3 * p + k], g = go.X ? go.X[3 * (size_t)row + k] : 0.0f;
            const float o = (a * w0 + b * w1) * inv;
            gs.X[3 * m + k] = g * w0 * inv;
            atomicAdd(&g_x_i[3 * p + k], g * w1 * inv);
            dw0 += g * (a - o) * inv;
            dw1 += g * (b - o) * inv;
        }
    }
    if (c == 1) {
        const float a = s.D[m], b = d_i[p], g = go.D ? go.D[row] : 0.0f;
        const float o = (a * w0 + b * w1) * inv;
        gs.D[m] = g * w0 * inv;
        atomicAdd(&g_d_i[p], g * w1 * inv);
        dw0 += g * (a - o) * inv;
        dw1 += g * (b - o) * inv;
        const float gR = go.R ? go.R[row] : 0.0f, gO = go.O ? go.O[row] : 0.0f;
        dw0 += gR; dw1 += gR;          // R_out = w0 + w1
        gs.O[m] = gO;                  // O_out = O[m] + om_i[p]
        atomicAdd(&g_om_i[p], gO);
    }
#pragma unroll
    for (int d = 2; d >= 1; d >>= 1) {
        dw0 += __shfl_xor(dw0, d, 64);
        dw1 += __shfl_xor(dw1, d, 64);
    }
    if (c == 0) {
        gs.R[m] = dw0;                 // (+ the positional-encoding term: ptf_gru_inputs_bwd adds it)
        atomicAdd(&g_rho_i[p], dw1);
    }
}

// Backward of the positional encodings of a fused pair, spread over its 16 lanes: four scalars (rho_i[p], O[m] of he; R[m],
// om_i[p] of xe) x six octaves = 24 (sin, cos) pairs.  Lane c works for scalar c >> 2 on the octaves c & 3 and (c & 3) + 4;
// the quad sums its lanes.  d/dv of out[2k] = sin(v f), out[2k+1] = cos(v f), f = 2^k: f (g[2k] cos(v f) - g[2k+1] sin(v f)),
// with the forward's phase reduction (fs_common.h:rev2pi) and the hardware v_sin_f32 / v_cos_f32.
// (Rounds 2 - 3 called cosf / sinf 24 times on each of lanes 0 and 1 -- libm's full-range forms, ~1350 VALU instructions
// per wavefront of 4 pairs, half of this kernel's 116 us.)
__device__ __forceinline__ float pos_enc_bwd_lane(float v, const float* __restrict__ g, int kq)
{
    float rv, ev;
    rev2pi(v, rv, ev);
    float d = 0.0f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = kq + 4 * j;
        if (k < 6) {
            const float f = (float)(1 << k);
            const float t = rv * f;                                    // (exact)
            const float ph = (t - rintf(t)) + ev * f;
            d += f * (g[2 * k] * __builtin_amdgcn_cosf(ph) - g[2 * k + 1] * __builtin_amdgcn_sinf(ph));
        }
    }
    d += __shfl_xor(d, 1);
    return d + __shfl_xor(d, 2);
}

__global__ __launch_bounds__(256) void ptf_gru_inputs_bwd_kernel(int n_fuse, const long long* __restrict__ fuse_idx,
                                                                const long long* __restrict__ fuse_pix,
                                                                const float* __restrict__ R, const float* __restrict__ O,
                                                                const float* __restrict__ rho_i,
                                                                const float* __restrict__ om_i,
                                                                const float* __restrict__ dcat, float* __restrict__ gG,
                                                                float* __restrict__ gR, float* __restrict__ gO,
                                                                float* __restrict__ g_lat_i, float* __restrict__ g_rho_i,
                                                                float* __restrict__ g_om_i)
{
    const int t = blockIdx.x * 16 + (threadIdx.x >> 4), c = threadIdx.x & 15;
    if (t >= n_fuse) return;                       // (whole 16-lane groups leave: the quad shuffles below stay inside one)
    const long long m = fuse_idx[t], p = fuse_pix[t];
    const float* row = dcat + (size_t)t * 176;
    ((float4*)(gG + m * 64))[c] = ((const float4*)row)[c];                    // hid: in-row m is fused exactly once
    // x: several tied rows may share pixel p -> atomics; lane c takes floats c, c + 16, c + 32, c + 48, so that one atomic
    // instruction covers 64 CONTIGUOUS bytes of the pixel's row (with 4 c + e it touched every fourth float of all 256)
    float* q = g_lat_i + p * 64 + c;
#pragma unroll
    for (int j = 0; j < 4; ++j) atomicAdd(q + 16 * j, row[88 + 16 * j + c]);
    // he = PE(rho_i[p], O[m]) at [64,88), xe = PE(R[m], om_i[p]) at [152,176)
    const int sc = c >> 2;
    const float v = sc == 0 ? rho_i[p] : sc == 1 ? O[m] : sc == 2 ? R[m] : om_i[p];
    const float d = pos_enc_bwd_lane(v, row + (sc < 2 ? 64 : 152) + 12 * (sc & 1), c & 3);
    if ((c & 3) == 0) {
        if (sc == 0) atomicAdd(&g_rho_i[p], d);
        else if (sc == 1) gO[m] += d;
        else if (sc == 2) gR[m] += d;
        else atomicAdd(&g_om_i[p], d);
    }
}

__host__ __device__ inline size_t ptf_scratch_layout(int M, int P, size_t off[7])
{
    const int nbM = (M + kScanBlock - 1) / kScanBlock, nbP = (P + kScanBlock - 1) / kScanBlock;
    size_t o = 0;
    off[0] = o; o += align_up((size_t)P * 4, 256);            // zbuf
    off[1] = o; o += align_up((size_t)M * 4, 256);            // pix_of
    off[2] = o; o += align_up((size_t)M * 4, 256);            // zbits_of
    off[3] = o; o += align_up((size_t)M, 256);                // win
    off[4] = o; o += align_up((size_t)P, 256);                // app
    off[5] = o; o += align_up((size_t)(nbM + nbP) * 4, 256);  // block counts / offsets
    off[6] = o;
    return o;
}

}  // namespace fs

using namespace fs;

FS_API size_t fs_ptf_scratch_bytes(int32_t M, int32_t h, int32_t w)
{
    if (M < 0 || h <= 0 || w <= 0) return 0;
    size_t off[7];
    return ptf_scratch_layout(M > 0 ? M : 1, h * w, off);
}

// M: number of state rows, or (Mp != NULL) an upper bound of *Mp used for sizing grids and the scratch layout
// zbuf_clean: the z-buffer (the first P words of the scratch) already holds 0xFFFFFFFF -- the previous step's emit kernel
// leaves it so -- and the fill launch is skipped
static int ptf_match_impl(int32_t M, const int32_t* Mp, int32_t h, int32_t w, const float* xyz, const float* w2c,
                          const float* kpix, const float* depth_i, float depth_thres, void* scratch,
                          int64_t* keep_idx, int64_t* fuse_idx, int64_t* fuse_pix, int64_t* append_pix,
                          int32_t* counts, void* stream_, bool zbuf_clean = false)
{
    if (M < 0 || h <= 0 || w <= 0 || !w2c || !kpix || !depth_i || !scratch || !append_pix || !counts)
        return FS_ERR_INVALID_ARG;
    if (M > 0 && (!xyz || !keep_idx || !fuse_idx || !fuse_pix)) return FS_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream_;
    const int P = h * w;
    size_t off[7];
    ptf_scratch_layout(M > 0 ? M : 1, P, off);
    char* s = (char*)scratch;
    uint32_t* zbuf = (uint32_t*)(s + off[0]);
    int32_t* pix_of = (int32_t*)(s + off[1]);
    uint32_t* zbits_of = (uint32_t*)(s + off[2]);
    uint8_t* win = (uint8_t*)(s + off[3]);
    uint8_t* app = (uint8_t*)(s + off[4]);
    uint32_t* blocks = (uint32_t*)(s + off[5]);
    const int nbM = (M + kScanBlock - 1) / kScanBlock, nbP = (P + kScanBlock - 1) / kScanBlock;
    ScopedStage prof_(kStPtf, st);
    if (!zbuf_clean) hipLaunchKernelGGL(ptf_fill_kernel, dim3((P + 255) / 256), dim3(256), 0, st, zbuf, P);
    if (M > 0)
        hipLaunchKernelGGL(ptf_project_kernel, dim3((M + 255) / 256), dim3(256), 0, st, M, Mp, h, w, xyz, w2c, kpix,
                           pix_of, zbits_of, zbuf);
    hipLaunchKernelGGL(ptf_flags_count_kernel, dim3(nbM + nbP), dim3(256), 0, st, M, Mp, P, nbM, pix_of, zbits_of, zbuf,
                       depth_i, depth_thres, win, app, blocks);
    hipLaunchKernelGGL(ptf_scan_blocks_kernel, dim3(1), dim3(1024), 0, st, M, Mp, nbM, nbP, blocks, counts);
    hipLaunchKernelGGL(ptf_emit_kernel, dim3(nbM + nbP), dim3(256), 0, st, M, Mp, P, nbM, win, app, pix_of, blocks,
                       (long long*)keep_idx, (long long*)fuse_idx, (long long*)fuse_pix, (long long*)append_pix, zbuf);
    FS_CHECK_LAUNCH("ptf_match");
    return FS_OK;
}

FS_API int fs_ptf_match(int32_t M, int32_t h, int32_t w, const float* xyz, const float* w2c,
                        const float* kpix, const float* depth_i, float depth_thres, void* scratch,
                        int64_t* keep_idx, int64_t* fuse_idx, int64_t* fuse_pix, int64_t* append_pix,
                        int32_t* counts, void* stream_)
{
    return ptf_match_impl(M, nullptr, h, w, xyz, w2c, kpix, depth_i, depth_thres, scratch, keep_idx, fuse_idx, fuse_pix,
                          append_pix, counts, stream_);
}

FS_API int fs_ptf_gru_inputs(int32_t n_fuse, const int64_t* fuse_idx, const int64_t* fuse_pix, const float* G,
                             const float* R, const float* O, const float* g_i, const float* rho_i,
                             const float* om_i, float* cat, void* stream_)
{
    if (n_fuse < 0) return FS_ERR_INVALID_ARG;
    if (n_fuse == 0) return FS_OK;
    if (!fuse_idx || !fuse_pix || !G || !R || !O || !g_i || !rho_i || !om_i || !cat) return FS_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream_;
    ScopedStage prof_(kStPtf, st);
    hipLaunchKernelGGL(ptf_gru_inputs_kernel, dim3((n_fuse + 15) / 16), dim3(256), 0, st, n_fuse, (const int32_t*)nullptr,
                       (const long long*)fuse_idx, (const long long*)fuse_pix, G, R, O, g_i, rho_i, om_i, cat);
    FS_CHECK_LAUNCH("ptf_gru_inputs");
    return FS_OK;
}

FS_API int fs_ptf_write_state(int32_t n_keep, int32_t n_fuse, int32_t n_app, const int64_t* keep_idx,
                              const int64_t* fuse_idx, const int64_t* fuse_pix, const int64_t* append_pix,
                              const float* G, const float* X, const float* R, const float* O, const float* E,
                              const float* D, const float* g_i, const float* x_i, const float* rho_i,
                              const float* om_i, const float* d_i, const float* E_i, const float* fused,
                              float* oG, float* oX, float* oR, float* oO, float* oE, float* oD, void* stream_)
{
    if (n_keep < 0 || n_fuse < 0 || n_app < 0) return FS_ERR_INVALID_ARG;
    const long long n_out = (long long)n_keep + n_fuse + n_app;
    if (n_out == 0) return FS_OK;
    if (!g_i || !x_i || !rho_i || !om_i || !d_i || !E_i || !oG || !oX || !oR || !oO || !oE || !oD)
        return FS_ERR_INVALID_ARG;
    if ((n_keep || n_fuse) && (!G || !X || !R || !O || !E || !D)) return FS_ERR_INVALID_ARG;
    if (n_fuse && (!fused || !fuse_idx || !fuse_pix)) return FS_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream_;
    ScopedStage prof_(kStPtf, st);
    PtfState s{const_cast<float*>(G), const_cast<float*>(X), const_cast<float*>(R), const_cast<float*>(O),
               const_cast<float*>(E), const_cast<float*>(D)};
    PtfState o{oG, oX, oR, oO, oE, oD};
    hipLaunchKernelGGL(ptf_write_state_kernel<false>, dim3((unsigned)((n_out + 15) / 16)), dim3(256), 0, st, n_keep, n_fuse,
                       n_app, (const int32_t*)nullptr, (const long long*)keep_idx, (const long long*)fuse_idx, (const long long*)fuse_pix,
                       (const long long*)append_pix, s, g_i, x_i, rho_i, om_i, d_i, E_i, fused, o);
    FS_CHECK_LAUNCH("ptf_write_state");
    return FS_OK;
}

// ---- one whole fold step without host involvement -----------------------------------------------------------
namespace {
struct FoldLayout { size_t match, keep, fuse, fpix, app, cat, fused, total; };
FoldLayout fold_layout(int M_max, int P)
{
    FoldLayout L;
    size_t off[7];
    const size_t nf = (size_t)(M_max < P ? M_max : P);
    size_t o = 0;
    L.match = o; o += align_up(ptf_scratch_layout(M_max > 0 ? M_max : 1, P, off), 256);
    L.keep = o;  o += align_up((size_t)(M_max > 0 ? M_max : 1) * 8, 256);
    L.fuse = o;  o += align_up((nf > 0 ? nf : 1) * 8, 256);
    L.fpix = o;  o += align_up((nf > 0 ? nf : 1) * 8, 256);
    L.app = o;   o += align_up((size_t)P * 8, 256);
    L.cat = o;   o += align_up((nf > 0 ? nf : 1) * 176 * 4, 256);
    L.fused = o; o += align_up((nf > 0 ? nf : 1) * 64 * 4, 256);
    L.total = o;
    return L;
}
}  // namespace

FS_API size_t fs_ptf_fold_scratch_bytes(int32_t M_max, int32_t h, int32_t w)
{
    if (M_max < 0 || h <= 0 || w <= 0) return 0;
    return fold_layout(M_max, h * w).total;
}

static int fold_step_impl(int32_t M_max, const int32_t* M_dev, int32_t h, int32_t w, const float* G, const float* X,
                          const float* R, const float* O, const float* E, const float* D, const float* g_i,
                          const float* x_i, const float* rho_i, const float* om_i, const float* d_i, const float* E_i,
                          const float* w2c, const float* kpix, float depth_thres, const float* gru_tables,
                          void* scratch, float* oG, float* oX, float* oR, float* oO, float* oE, float* oD,
                          int32_t* counts, void* stream_, bool zbuf_clean, float* save_side = nullptr, float* save_act = nullptr,
                          float* save_cat = nullptr)
{
    if (M_max <= 0 || h <= 0 || w <= 0 || !G || !X || !R || !O || !E || !D || !g_i || !x_i || !rho_i || !om_i || !d_i ||
        !E_i || !w2c || !kpix || !gru_tables || !scratch || !oG || !oX || !oR || !oO || !oE || !oD || !counts)
        return FS_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream_;
    const int P = h * w;
    const FoldLayout L = fold_layout(M_max, P);
    char* s = (char*)scratch;
    long long *keep = (long long*)(s + L.keep), *fuse = (long long*)(s + L.fuse), *fpix = (long long*)(s + L.fpix),
              *app = (long long*)(s + L.app);
    float *cat = (float*)(s + L.cat), *fused = (float*)(s + L.fused);
    int rc = ptf_match_impl(M_max, M_dev, h, w, X, w2c, kpix, d_i, depth_thres, s + L.match, (int64_t*)keep,
                            (int64_t*)fuse, (int64_t*)fpix, (int64_t*)app, counts, stream_, zbuf_clean);
    if (rc != FS_OK) return rc;
    const int nf_max = M_max < P ? M_max : P;
    ScopedStage prof_(kStPtf, st);
    (void)cat; (void)fused;  // (the GRU gathers and encodes its input rows itself and writes its rows into the out state)
    rc = launch_ptf_gru_gather(nf_max, counts, (const long long*)fuse, (const long long*)fpix, G, R, O, g_i, rho_i, om_i,
                               gru_tables, oG, true, st, save_side, save_act, save_cat);
    if (rc != FS_OK) return rc;
    PtfState si{const_cast<float*>(G), const_cast<float*>(X), const_cast<float*>(R), const_cast<float*>(O),
                const_cast<float*>(E), const_cast<float*>(D)};
    PtfState so{oG, oX, oR, oO, oE, oD};
    const long long n_out_max = (long long)M_max + P;
    // (the counts are on the device: both grids cover their worst case, workgroups past the rows that exist leave at once)
    static const bool split = [] { const char* e = getenv("FS_PTF_WS_SPLIT"); return !(e && atoi(e) == 0); }();
    if (split) {
        const unsigned cap = 256 * 8 * 4;        // 8 workgroups per CU resident, four rounds of them: enough to balance, few enough to be cheap when empty
        hipLaunchKernelGGL(ptf_write_state_kernel<true>, dim3(std::min<unsigned>((unsigned)((n_out_max + 15) / 16), cap)), dim3(256), 0, st, 0, 0, 0,
                           (const int32_t*)counts, (const long long*)keep, (const long long*)fuse, (const long long*)fpix,
                           (const long long*)app, si, g_i, x_i, rho_i, om_i, d_i, E_i, (const float*)nullptr, so);
        hipLaunchKernelGGL(ptf_write_state_fused_kernel, dim3(std::min<unsigned>((unsigned)((nf_max + 63) / 64), cap)), dim3(256), 0, st, 0, 0,
                           (const int32_t*)counts, (const long long*)fuse, (const long long*)fpix, si, x_i, rho_i, om_i, d_i, E_i, so);
    } else {
        hipLaunchKernelGGL(ptf_write_state_kernel<false>, dim3((unsigned)((n_out_max + 15) / 16)), dim3(256), 0, st, 0, 0, 0,
                           (const int32_t*)counts, (const long long*)keep, (const long long*)fuse, (const long long*)fpix,
                           (const long long*)app, si, g_i, x_i, rho_i, om_i, d_i, E_i, (const float*)nullptr, so);
    }
    FS_CHECK_LAUNCH("ptf_write_state");
    return FS_OK;
}

FS_API int fs_ptf_fold_step(int32_t M_max, const int32_t* M_dev, int32_t h, int32_t w, const float* G, const float* X,
                            const float* R, const float* O, const float* E, const float* D, const float* g_i,
                            const float* x_i, const float* rho_i, const float* om_i, const float* d_i, const float* E_i,
                            const float* w2c, const float* kpix, float depth_thres, const float* gru_tables,
                            void* scratch, float* oG, float* oX, float* oR, float* oO, float* oE, float* oD,
                            int32_t* counts, void* stream_)
{
    return fold_step_impl(M_max, M_dev, h, w, G, X, R, O, E, D, g_i, x_i, rho_i, om_i, d_i, E_i, w2c, kpix, depth_thres,
                          gru_tables, scratch, oG, oX, oR, oO, oE, oD, counts, stream_, false);
}

// fs_ptf_fold_step for a TRAINING fold (round 6): the GRU additionally leaves, for fused pair t (the order of the step's fuse list),
// row t of `side` [min(M_max, h w), fs_ptf_gru_side_cols()] (columns 6 .. 9 only), `act` [.. rounded up to 16, fs_ptf_gru_act_cols()] (gates, lane
// order) and `cat` [.., 176] (the pair's gathered + encoded input row: what fs_ptf_gru_inputs would re-gather), which
// fs_ptf_gru_backward_saved / fs_ptf_gru_weight_grads consume instead of re-running the forward.  Requires fs_ptf_gru_stream_t_rows() > 0.
FS_API int fs_ptf_fold_step_save(int32_t M_max, const int32_t* M_dev, int32_t h, int32_t w, const float* G, const float* X,
                                 const float* R, const float* O, const float* E, const float* D, const float* g_i,
                                 const float* x_i, const float* rho_i, const float* om_i, const float* d_i, const float* E_i,
                                 const float* w2c, const float* kpix, float depth_thres, const float* gru_tables,
                                 void* scratch, float* oG, float* oX, float* oR, float* oO, float* oE, float* oD,
                                 int32_t* counts, float* side, float* act, float* cat, void* stream_)
{
    if (!side || !act || !cat) return FS_ERR_INVALID_ARG;
    return fold_step_impl(M_max, M_dev, h, w, G, X, R, O, E, D, g_i, x_i, rho_i, om_i, d_i, E_i, w2c, kpix, depth_thres,
                          gru_tables, scratch, oG, oX, oR, oO, oE, oD, counts, stream_, false, side, act, cat);
}

// Camera constants of the fold in one launch: thread i scales the normalised intrinsics of view i to pixels
// (encoder_freesplat.py:445-448, one multiply each: same bits as torch); all threads replicate view 0's extrinsics per
// pixel (the initial per-Gaussian extrinsics, :441).  The world-to-camera matrices are NOT formed here: a pixel's
// round-half-even decision can hinge on their last bit; they come from fs_invert_4x4 (framing.hip: double precision inside,
// rounded once -- tests/test_ptf_hip.py::test_invert_4x4_vs_float64_inverse), in the fold and in the tests' oracle alike.
__global__ __launch_bounds__(256) void ptf_cameras_kernel(int V, int P, int h, int w, const float* __restrict__ Es,
                                                          const float* __restrict__ Kn, float* __restrict__ kpix,
                                                          float* __restrict__ E0, uint32_t* __restrict__ zbuf)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t < (long long)P * 4) ((float4*)E0)[t] = ((const float4*)Es)[t & 3];
    if (zbuf && t < P) zbuf[t] = 0xFFFFFFFFu;      // (fs_ptf_fold: the first step's z-buffer, cleared here)
    if (t >= V) return;
    const float* K = Kn + 9 * t;
    kpix[4 * t] = K[0] * (float)w; kpix[4 * t + 1] = K[4] * (float)h;
    kpix[4 * t + 2] = K[2] * (float)w; kpix[4 * t + 3] = K[5] * (float)h;
}

namespace {
size_t fold_camera_bytes(int V, int P) { return align_up((size_t)V * 4 * 4, 256) + align_up((size_t)P * 64, 256) + align_up((size_t)V * 64, 256); }
}

// The per-view pixel intrinsics kpix [V,4] = (fx w, fy h, cx w, cy h) and the initial per-Gaussian extrinsics E0 [P,16]
// (view 0's matrix on every row) in one launch: what fs_ptf_fold prepares internally, for callers that drive
// fs_ptf_fold_step themselves (the training path).  P = h * w.
FS_API int fs_ptf_cameras(int32_t V, int32_t h, int32_t w, const float* Es, const float* Kn, float* kpix, float* E0,
                          void* stream_)
{
    if (V < 1 || h <= 0 || w <= 0 || !Es || !Kn || !kpix || !E0) return FS_ERR_INVALID_ARG;
    const long long P = (long long)h * w;
    const long long nt = P * 4 > V ? P * 4 : V;
    hipLaunchKernelGGL(ptf_cameras_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, V, (int)P,
                       h, w, Es, Kn, kpix, E0, (uint32_t*)nullptr);
    FS_CHECK_LAUNCH("ptf_cameras");
    return FS_OK;
}

FS_API size_t fs_ptf_fold_bytes(int32_t V, int32_t h, int32_t w)
{
    if (V < 2 || h <= 0 || w <= 0) return 0;
    return fold_layout((V - 1) * h * w, h * w).total + fold_camera_bytes(V, h * w);
}

// All fold steps of one scene in one host call: views 1 .. V-1 are folded into the state that starts as view 0.
// lat [V,P,64], xs [V,P,3], rho / om / dep [V,P], Es [V,16] (camera-to-world), w2c [V,16] (its inverse; NULL: computed here), Kn [V,9]
// (normalised intrinsics);
// bufA / bufB: two sets of 6 state arrays (G, X, R, O, E, D) with V*P rows each (2*P for V == 2, bufB unused), written
// alternately; counts [V,4].  The final state is in set A if (V - 1) is odd, else B, with counts[V-1][3] rows.
// scratch: fs_ptf_fold_bytes(V, h, w).
FS_API int fs_ptf_fold(int32_t V, int32_t h, int32_t w, const float* lat, const float* xs, const float* rho,
                       const float* om, const float* dep, const float* Es, const float* w2c, const float* Kn,
                       float depth_thres, const float* gru_tables, void* scratch, float* const* bufA, float* const* bufB,
                       int32_t* counts, void* stream_)
{
    if (V < 2 || h <= 0 || w <= 0 || !lat || !xs || !rho || !om || !dep || !Es || !Kn || !gru_tables || !scratch ||
        !bufA || !bufB || !counts)
        return FS_ERR_INVALID_ARG;
    const size_t P = (size_t)h * w;
    hipStream_t st = (hipStream_t)stream_;
    char* cam = (char*)scratch + fold_layout((int)((V - 1) * P), (int)P).total;
    float* kpix = (float*)cam;
    float* E0 = (float*)(cam + align_up((size_t)V * 4 * 4, 256));
    if (!w2c) {     // the world-to-camera matrices are formed here (fs_invert_4x4, into the scratch): one launch, no extra call
        float* inv = (float*)(cam + align_up((size_t)V * 4 * 4, 256) + align_up(P * 64, 256));
        const int rc = fs_invert_4x4(V, Es, inv, stream_);
        if (rc != FS_OK) return rc;
        w2c = inv;
    }
    {
        const long long nt = (long long)P * 4 > V ? (long long)P * 4 : V;
        // (the z-buffer is the first P words of the scratch for every step's layout; cleared here for step 1, by each
        //  step's emit kernel for the next)
        hipLaunchKernelGGL(ptf_cameras_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, V, (int)P, h, w, Es,
                           Kn, kpix, E0, (uint32_t*)scratch);
        FS_CHECK_LAUNCH("ptf_cameras");
    }
    const float* cur[6] = {lat, xs, rho, om, E0, dep};
    for (int i = 1; i < V; ++i) {
        float* const* out = (i & 1) ? bufA : bufB;
        const int rc = fold_step_impl((int32_t)(i * P), i == 1 ? nullptr : counts + 4 * (i - 1) + 3, h, w, cur[0], cur[1],
                                        cur[2], cur[3], cur[4], cur[5], lat + i * P * 64, xs + i * P * 3, rho + i * P,
                                        om + i * P, dep + i * P, Es + 16 * (size_t)i, w2c + 16 * (size_t)i,
                                        kpix + 4 * (size_t)i, depth_thres, gru_tables, scratch, out[0], out[1], out[2],
                                        out[3], out[4], out[5], counts + 4 * i, stream_, true);
        if (rc != FS_OK) return rc;
        for (int k = 0; k < 6; ++k) cur[k] = out[k];
    }
    return FS_OK;
}


// ---- training path ------------------------------------------------------------------------------------------
// The four ordered index lists of the fold step that last used `scratch` (fs_ptf_fold_step keeps them there):
// lists[0..3] = keep_idx, fuse_idx, fuse_pix, append_pix (device pointers into scratch; lengths = counts[0..2]).
FS_API int fs_ptf_fold_step_lists(int32_t M_max, int32_t h, int32_t w, void* scratch, int64_t** lists)
{
    if (M_max <= 0 || h <= 0 || w <= 0 || !scratch || !lists) return FS_ERR_INVALID_ARG;
    const FoldLayout L = fold_layout(M_max, h * w);
    char* s = (char*)scratch;
    lists[0] = (int64_t*)(s + L.keep); lists[1] = (int64_t*)(s + L.fuse);
    lists[2] = (int64_t*)(s + L.fpix); lists[3] = (int64_t*)(s + L.app);
    return FS_OK;
}

// Backward of fs_ptf_write_state.  g_out[6] / g_in[6]: gradients of the out / in state arrays in the order
// G, X, R, O, E, D (entries of g_out may be NULL = zero gradient; g_in all required, every row is written except
// g_in G of the fused rows, which fs_ptf_gru_inputs_backward writes).  The gradients of the view's arrays
// (g_lat_i [P,64], g_x_i [P,3], g_rho_i / g_om_i / g_d_i [P]) are ACCUMULATED: zero them before the first step.
// The gradient of the GRU output rows is g_out[0] + n_keep*64 (n_fuse contiguous rows).
FS_API int fs_ptf_write_state_backward(int32_t n_keep, int32_t n_fuse, int32_t n_app, const int64_t* keep_idx,
                                       const int64_t* fuse_idx, const int64_t* fuse_pix, const int64_t* append_pix,
                                       const float* X, const float* R, const float* E, const float* D,
                                       const float* x_i, const float* rho_i, const float* d_i, const float* E_i,
                                       float* const* g_out, float* const* g_in, float* g_lat_i, float* g_x_i,
                                       float* g_rho_i, float* g_om_i, float* g_d_i, void* stream_)
{
    if (n_keep < 0 || n_fuse < 0 || n_app < 0 || !g_out || !g_in) return FS_ERR_INVALID_ARG;
    const long long n_out = (long long)n_keep + n_fuse + n_app;
    if (n_out == 0) return FS_OK;
    if (!g_lat_i || !g_x_i || !g_rho_i || !g_om_i || !g_d_i || !x_i || !rho_i || !d_i || !E_i) return FS_ERR_INVALID_ARG;
    if (n_keep || n_fuse) {
        if (!X || !R || !E || !D) return FS_ERR_INVALID_ARG;
        for (int k = 0; k < 6; ++k)
            if (!g_in[k]) return FS_ERR_INVALID_ARG;
    }
    hipStream_t st = (hipStream_t)stream_;
    ScopedStage prof_(kStPtf, st);
    PtfState s{nullptr, const_cast<float*>(X), const_cast<float*>(R), nullptr, const_cast<float*>(E), const_cast<float*>(D)};
    PtfGrad go{g_out[0], g_out[1], g_out[2], g_out[3], g_out[4], g_out[5]};
    PtfGrad gs{g_in[0], g_in[1], g_in[2], g_in[3], g_in[4], g_in[5]};
    // FS_PTF_WS_BWD_SPLIT=0: one launch, 16 lanes per row for every row (rounds 2 - 5; A/B)
    static const bool split = [] { const char* e = getenv("FS_PTF_WS_BWD_SPLIT"); return !(e && atoi(e) == 0); }();
    if (split) {
        const long long n_ka = (long long)n_keep + n_app;
        if (n_ka > 0)
            hipLaunchKernelGGL(ptf_write_state_bwd_kernel<true>, dim3((unsigned)((n_ka + 63) / 64)), dim3(256), 0, st, n_keep, n_fuse,
                               n_app, (const long long*)keep_idx, (const long long*)fuse_idx, (const long long*)fuse_pix,
                               (const long long*)append_pix, s, x_i, rho_i, d_i, E_i, go, gs, g_lat_i, g_x_i, g_rho_i, g_om_i, g_d_i);
        if (n_fuse > 0)
            hipLaunchKernelGGL(ptf_write_state_bwd_fused_kernel, dim3((unsigned)((n_fuse + 63) / 64)), dim3(256), 0, st, n_keep, n_fuse,
                               (const long long*)fuse_idx, (const long long*)fuse_pix, s, x_i, rho_i, d_i, E_i, go, gs, g_x_i, g_rho_i,
                               g_om_i, g_d_i);
    } else {
        hipLaunchKernelGGL(ptf_write_state_bwd_kernel<false>, dim3((unsigned)((n_out + 15) / 16)), dim3(256), 0, st, n_keep, n_fuse,
                           n_app, (const long long*)keep_idx, (const long long*)fuse_idx, (const long long*)fuse_pix,
                           (const long long*)append_pix, s, x_i, rho_i, d_i, E_i, go, gs, g_lat_i, g_x_i, g_rho_i, g_om_i, g_d_i);
    }
    FS_CHECK_LAUNCH("ptf_write_state_bwd");
    return FS_OK;
}

// Backward of fs_ptf_gru_inputs: dcat [n_fuse,176] -> g_G [M,64] rows fuse_idx (stored), g_R / g_O [M] (added to what
// fs_ptf_write_state_backward stored), g_lat_i / g_rho_i / g_om_i of the view (accumulated, atomics).
FS_API int fs_ptf_gru_inputs_backward(int32_t n_fuse, const int64_t* fuse_idx, const int64_t* fuse_pix, const float* R,
                                      const float* O, const float* rho_i, const float* om_i, const float* dcat,
                                      float* g_G, float* g_R, float* g_O, float* g_lat_i, float* g_rho_i, float* g_om_i,
                                      void* stream_)
{
    if (n_fuse < 0) return FS_ERR_INVALID_ARG;
    if (n_fuse == 0) return FS_OK;
    if (!fuse_idx || !fuse_pix || !R || !O || !rho_i || !om_i || !dcat || !g_G || !g_R || !g_O || !g_lat_i || !g_rho_i ||
        !g_om_i)
        return FS_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream_;
    ScopedStage prof_(kStPtf, st);
    hipLaunchKernelGGL(ptf_gru_inputs_bwd_kernel, dim3((n_fuse + 15) / 16), dim3(256), 0, st, n_fuse,
                       (const long long*)fuse_idx, (const long long*)fuse_pix, R, O, rho_i, om_i, dcat, g_G, g_R, g_O,
                       g_lat_i, g_rho_i, g_om_i);
    FS_CHECK_LAUNCH("ptf_gru_inputs_bwd");
    return FS_OK;
}
